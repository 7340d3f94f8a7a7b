#!/usr/bin/env python3
"""Barrier-to-barrier timeline of workgroup 0 of snf_gemm_bf16 (dev build with -DSNF_GEMM_TRACE: waves 0 and 4 stamp s_memtime before
and after every s_barrier; 100 MHz ticks).  Shows where a tile's time goes: the K steps, and the gap at the tile boundary (epilogue).
    SNUFFY_HIP_LIB=snuffy_amd/build/variants/lib_trace.so python tools/gemm_trace.py m n k [act]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m, n, k = [int(v) for v in sys.argv[1:4]]
act = sys.argv[4] if len(sys.argv) > 4 else "none"
dev = torch.device("cuda")
buf = torch.zeros(320, dtype=torch.int64, device=dev)
os.environ["SNF_GEMM_TRACE_PTR"] = hex(buf.data_ptr())
from snuffy_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(0)
a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev)
w = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
b = torch.randn(n, generator=g).to(dev)
out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    buf.zero_()
    ops.gemm_bf16(a, w, b, act, out=out, tile_n=256)
    torch.cuda.synchronize()
t = buf.cpu().view(2, 160)
ns = k // 32
for grp in range(2):
    st = [int(v) for v in t[grp] if int(v)]
    t0 = st[0]
    # stamps come in pairs (before, after) per barrier; two barriers per step
    print("group %d: %d stamps; tick = 10 ns" % (grp, len(st)))
    line = []
    for i in range(0, len(st) - 1, 2):
        arrive, leave = st[i] - t0, st[i + 1] - t0
        line.append("%d(+%d)" % (arrive, leave - arrive))
    per = 2 * ns
    for j in range(0, len(line), per):
        print("  ", " ".join(line[j:j + per]))
