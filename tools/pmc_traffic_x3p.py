#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes of the pipelined fp32-class attention (snf_sparse_attn_fwd_x3_hl):
calibration read / write of known size, then REPS calls on operand sets that rotate through more than the Infinity Cache.
usage: python tools/pmc_traffic_x3p.py cfgA|cfgB|cfgC     (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, see pmc_traffic_x3p.sh)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import ops  # noqa: E402

WL = {"cfgB": (32768, 768, 6, 200), "cfgC": (100000, 768, 6, 512), "cfgA": (8192, 384, 6, 200)}
REPS = 8
dev = torch.device("cuda")
N, D, h, K = WL[sys.argv[1] if len(sys.argv) > 1 else "cfgB"]
g = torch.Generator().manual_seed(5)
nset = 3 if N > 50000 else 4
xs = [torch.randn(32768, 768, generator=g).to(dev) for _ in range(2)]
imgs = [ops.split_hl_rows(torch.randn(N, 2 * D, generator=g).to(dev)) for _ in range(nset)]
kpf = torch.randn(K, D, generator=g).to(dev)
if ops.x3_hl_kpfrag_supported(K, h, D // h):   # as the model dispatches it: the key projection writes the fragment image
    kpf = ops.linear_rows_x3_kpfrag(kpf, (torch.randn(D, D, generator=g) / D ** 0.5).to(dev), None, h)
w = torch.randn(1, 768, generator=g).to(dev)
b = torch.zeros(1, device=dev)
outs = [torch.empty(32768, 768, device=dev) for _ in range(2)]
torch.cuda.synchronize()
for it in range(REPS):
    i = it % nset
    ops.critic(xs[it % 2], w, b)                              # calibration read : 32768 * 768 * 4 bytes
    outs[it % 2].fill_(1.0)                                   # calibration write: 32768 * 768 * 4 bytes
    ops.sparse_attn_fwd_x3_hl(imgs[i][:, :2 * D], imgs[i][:, 2 * D:], kpf, h)
torch.cuda.synchronize()
print("calls %d, algorithmic bytes per call %d" % (REPS, 8 * N * D + 8 * K * D))
