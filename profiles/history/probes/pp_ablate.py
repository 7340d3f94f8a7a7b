#!/usr/bin/env python3
"""Dev: timing ablations of the ping-pong attention kernel (bit 0: no V re-staging, bit 1: no Q reloads)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import timed
from snuffy_amd import _ffi, ops
dev = torch.device("cuda")
N, D, h, K = 32768, 768, 6, 200
g = torch.Generator().manual_seed(1)
kp = torch.randn(K, D, generator=g).to(dev).to(torch.bfloat16)
qvs = [torch.randn(N, 2 * D, generator=g).to(dev).to(torch.bfloat16) for _ in range(4)]
st = {"i": 0}
def f():
    st["i"] = (st["i"] + 1) % 4
    ops.sparse_attn_fwd_mfma(qvs[st["i"]][:, :D], qvs[st["i"]][:, D:], kp, N, h)
lib = _ffi.load()
for flag in (0, 1, 2, 3):
    lib.snf_debug_attn_trace_wg(flag)
    t = timed(f, 20, warmup=3)
    print(f"ablation flag {flag}: {t*1e3:.1f} us")
lib.snf_debug_attn_trace_wg(0)
