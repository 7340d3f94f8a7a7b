#!/usr/bin/env python3
"""Phase anatomy of the single-workgroup top-k kernel (s_memtime stamps of thread 0).  Needs a library built with
-DSNF_TOPK_TRACE (add it to EXTRA_FLAGS["topk.hip"] in snuffy_amd/build.py); the shipped build has no stamps."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import _ffi, ops
lib = _ffi.load()
n, k = int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 200
c = torch.randn(n, generator=torch.Generator().manual_seed(2)).cuda()
for _ in range(3):
    ops.topk(c, k)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
lib.snf_debug_topk_trace(buf)
t = list(buf)
names = ["issue loads", "loads landed", "p0 zero+sync", "p0 atomics", "p0 scan -> p1", "p1 atomics", "p1 scan -> p2", "p2 atomics",
         "p2 scan", "collect", "rank sort"]
prev = t[0]
for i in range(1, 11):
    print("%-16s %7d" % (names[i], t[i] - prev)); prev = t[i]
print("total", t[10] - t[0])
