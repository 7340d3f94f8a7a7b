"""Timing of the exact-fp32 attention (snf_sparse_attn_fwd_f32): matrix-core forms (v_mfma_f32_32x32x2_f32) vs the vector-ALU kernels.
python tools/exact_attn_bench.py [n k h dk]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import _ffi, ops  # noqa: E402

shapes = [tuple(int(x) for x in sys.argv[1:5])] if len(sys.argv) >= 5 else [(30000, 500, 4, 192), (30000, 900, 4, 96), (32768, 200, 6, 128),
                                                                              (8192, 200, 6, 64)]
for n, k, h, dk in shapes:
    d = h * dk
    q, kp, v = torch.randn(n, d, device="cuda"), torch.randn(k, d, device="cuda"), torch.randn(n, d, device="cuda")
    for mode in (1, 0):
        _ffi.load().snf_debug_exact_attn_mfma(mode)
        for _ in range(2):
            ops.sparse_attn_fwd(q, kp, v, h)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.sparse_attn_fwd(q, kp, v, h)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        print(f"n={n} k={k} h={h} dk={dk} {'mfma f32' if mode else 'vector ALU'}: {us:.0f} us  {4 * n * k * d / us / 1e6:.1f} TFLOP/s", flush=True)
    if ops.x3u_attn_supported(k, dk):
        for _ in range(2):
            ops.sparse_attn_fwd_x3u(q, kp, v, h)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.sparse_attn_fwd_x3u(q, kp, v, h)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        print(f"n={n} k={k} h={h} dk={dk} split-bf16 x3 (unfused): {us:.0f} us  {4 * n * k * d / us / 1e6:.1f} TFLOP/s", flush=True)
_ffi.load().snf_debug_exact_attn_mfma(1)
