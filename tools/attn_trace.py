#!/usr/bin/env python3
"""Per-phase cycle anatomy of the MFMA attention kernel (workgroup 0) from in-kernel s_memtime stamps."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import _ffi, ops  # noqa: E402

dev = torch.device("cuda")
lib = _ffi.load()
lib.snf_debug_attn_trace.argtypes = [ctypes.c_void_p]
lib.snf_debug_attn_trace.restype = None
N, D, h, K = 32768, 768, 6, int(sys.argv[1]) if len(sys.argv) > 1 else 200
WG = int(sys.argv[2]) if len(sys.argv) > 2 else 0     # traced workgroup (42 = first one that straddles two heads)
lib.snf_debug_attn_trace_wg.argtypes = [ctypes.c_int]
lib.snf_debug_attn_trace_wg.restype = None
lib.snf_debug_attn_trace_wg(WG)
g = torch.Generator().manual_seed(0)
qv = torch.randn(N, 2 * D, generator=g).to(dev).to(torch.bfloat16)
q, vt = qv[:, :D], qv[:, D:]
kp = torch.randn(K, D, generator=g).to(dev).to(torch.bfloat16)
for _ in range(3):
    ops.sparse_attn_fwd_mfma(q, vt, kp, N, h)
buf = torch.zeros(64 * 8 * 4, dtype=torch.int64, device=dev)
lib.snf_debug_attn_trace(ctypes.c_void_p(buf.data_ptr()))
ops.sparse_attn_fwd_mfma(q, vt, kp, N, h)
torch.cuda.synchronize()
lib.snf_debug_attn_trace(None)
t = buf.cpu().view(64, 8, 4)
pm = [int(t[sl, 0, 0]) for sl in (60, 56, 58, 61)]
print("prologue detail (softmax wave 0): entry -> Q loads issued %d | Kp loads issued %d | Kp landed, converted, stored, "
      "barrier %d" % tuple(pm[i + 1] - pm[i] for i in range(3)))
ms = [int(t[60, 0, 0]), int(t[61, 0, 0]), int(t[62, 0, 0]), int(t[63, 0, 0])]
print("kernel milestones (s_memtime ticks): prologue(Kp->LDS) %d | main loop %d | last GEMM2 + flush (pooling wave 0) %d | "
      "total %d" % (ms[1] - ms[0], ms[2] - ms[1], ms[3] - ms[2], ms[3] - ms[0]))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.sparse_attn_fwd_mfma(q, vt, kp, N, h)
e1.record()
torch.cuda.synchronize()
print("event time per call (main + reduce kernels): %.1f us" % (e0.elapsed_time(e1) * 100))
# softmax waves stamp 0 step start, 1 GEMM1 done, 2 barrier B passed, 3 softmax done, 4 barrier A passed (next 0 = P written);
# pooling waves stamp 5 barrier B passed, 6 GEMM2 done, 7 barrier A passed
for it in range(7):
    if int(t[it, 0, 0]) == 0:
        break
    print(f"step {it}:")
    rows = {"softmax: GEMM1": (0, 1), "softmax: wait B": (1, 2), "softmax: Q loads + softmax": (2, 3), "softmax: wait A": (3, 4),
            "pooling: GEMM2 (from B)": (5, 6), "pooling: wait A": (6, 7)}
    for nm, (p0, p1) in rows.items():
        print(f"   {nm:28s} " + " ".join(f"w{w}:{int(t[it, p1, w]) - int(t[it, p0, w]):7d}" for w in range(4)))
    nxt = [int(t[it + 1, 0, w]) or int(t[62, 0, w]) for w in range(4)]
    print(f"   {'softmax: write P (+loop)':28s} " + " ".join(f"w{w}:{nxt[w] - int(t[it, 4, w]):7d}" for w in range(4)))
    print(f"   {'step total':28s} " + " ".join(f"w{w}:{nxt[w] - int(t[it, 0, w]):7d}" for w in range(4)))
