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
g = torch.Generator().manual_seed(0)
qv = torch.randn(N, 2 * D, generator=g).to(dev).to(torch.bfloat16)
q, vt = qv[:, :D], qv[:, D:]
kp = torch.randn(K, D, generator=g).to(dev)
for _ in range(3):
    ops.sparse_attn_fwd_mfma(q, vt, kp, N, h)
buf = torch.zeros(64 * 8 * 4, dtype=torch.int64, device=dev)
lib.snf_debug_attn_trace(ctypes.c_void_p(buf.data_ptr()))
ops.sparse_attn_fwd_mfma(q, vt, kp, N, h)
torch.cuda.synchronize()
lib.snf_debug_attn_trace(None)
t = buf.cpu().view(64, 8, 4)
pm = [int(t[sl, 0, 0]) for sl in (60, 56, 58, 59, 57, 61)]
print("prologue detail: entry -> Q loads issued %d | Kp loads issued %d | P image zeroed %d | accumulators zeroed %d | "
      "Kp landed, converted, stored, barrier %d" % tuple(pm[i + 1] - pm[i] for i in range(5)))
ms = [int(t[sl, 0, 0]) for sl in (60, 61, 62, 63)]
print("kernel milestones (wave 0, s_memtime ticks): prologue(Kp->LDS) %d | main loop %d | drain+flush %d | total %d"
      % (ms[1] - ms[0], ms[2] - ms[1], ms[3] - ms[2], ms[3] - ms[0]))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.sparse_attn_fwd_mfma(q, vt, kp, N, h)
e1.record()
torch.cuda.synchronize()
print("event time per call (main + reduce kernels): %.1f us" % (e0.elapsed_time(e1) * 100))
names = ["GEMM1", "softmax||GEMM2(+loads)", "wait barrier A", "publish P", "wait barrier B", "loop/top"]
for it in range(3):
    if int(t[it, 0, 0]) == 0:
        break
    row = []
    for w in range(4):
        st = [int(t[it, p, w]) for p in range(6)]
        nxt = int(t[it + 1, 0, w]) if int(t[it + 1, 0, w]) else st[5]
        d = [st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], nxt - st[5]]
        row.append(d)
    print(f"iter {it}:")
    for p, nm in enumerate(names):
        print(f"   {nm:26s} " + " ".join(f"w{w}:{row[w][p]:7d}" for w in range(4)))
    print(f"   {'total':26s} " + " ".join(f"w{w}:{sum(row[w]):7d}" for w in range(4)))
