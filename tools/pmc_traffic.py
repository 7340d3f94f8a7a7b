#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, one counter per rocprofv3 pass).

Runs, on operand sets that rotate through more than the 256 MiB Infinity Cache:
  * a calibration READ of known size  (critic kernel: N*D*4 bytes read, N*4 written),
  * a calibration WRITE of known size (torch fill of N*D*4 bytes),
  * the config-B attention (sparse_attn_mfma_kernel + reduce_partials_kernel).
tools/pmc_traffic_summary.py turns the counter CSVs into bytes per launch, with the unit/gfx950 corrections of
MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request on wide streaming reads -> x2; WRITE_SIZE calibrated here)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import ops  # noqa: E402

dev = torch.device("cuda")
N, D, h, K = 32768, 768, 6, 200
g = torch.Generator().manual_seed(5)
nset = 4
xs = [torch.randn(N, D, generator=g).to(dev) for _ in range(nset)]
qvs = [torch.randn(N, 2 * D, generator=g).to(dev).to(torch.bfloat16) for _ in range(nset)]
qvf = [torch.randn(N, 2 * D, generator=g).to(dev) for _ in range(2)]       # fp32 operands of the fp32-class (x3) kernel
kpf = torch.randn(K, D, generator=g).to(dev)
kp = torch.randn(K, D, generator=g).to(dev).to(torch.bfloat16)   # as the model's bf16 path passes it
w = torch.randn(1, D, generator=g).to(dev)
b = torch.zeros(1, device=dev)
outs = [torch.empty(N, D, device=dev) for _ in range(nset)]
torch.cuda.synchronize()
for it in range(12):
    i = it % nset
    ops.critic(xs[i], w, b)                                   # calibration read : N*D*4 bytes
    outs[i].fill_(1.0)                                        # calibration write: N*D*4 bytes
    ops.sparse_attn_fwd_mfma(qvs[i][:, :D], qvs[i][:, D:], kp, N, h)
    ops.sparse_attn_fwd_x3(qvf[it % 2][:, :D], qvf[it % 2][:, D:], kpf, h)
torch.cuda.synchronize()
print("known bytes: critic read %d, fill write %d, attention algorithmic %d (Q,V bf16 + Kp + O)"
      % (N * D * 4, N * D * 4, 2 * N * D * 2 + K * D * (2 + 4)))
