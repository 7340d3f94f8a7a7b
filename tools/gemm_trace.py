#!/usr/bin/env python3
"""In-kernel s_memtime anatomy of the GEMM main loop (needs the SNF_GEMM_TRACE build: tools/gemm_variants.sh trace).
   SNUFFY_HIP_LIB=snuffy_amd/build/variants/lib_trace.so python tools/gemm_trace.py m n k [tile_n]
Per phase and wave group: load part (barrier release -> arrival), wait at the barrier, MFMA part, wait at the barrier."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import ops  # noqa: E402

m, n, k = [int(v) for v in sys.argv[1:4]]
tile_n = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev)
w = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
buf = torch.zeros(max(n, 1024), dtype=torch.float32, device=dev)          # the trace build takes it as the stamp buffer
for _ in range(3):
    buf.zero_()
    ops.gemm_bf16(a, w, buf[:n], "none", tile_n=tile_n)
torch.cuda.synchronize()
st = buf.view(torch.int64).cpu().numpy()[:320].reshape(2, 160)
t0 = st[0][0]
for grp in range(2):
    s = st[grp]
    s = s[s > 0]
    print(f"group {grp}: {len(s)} stamps; first at +{s[0]-t0}")
    # stamps come in fours per phase: L-end(arrive), L-barrier-release, M-end(arrive), M-barrier-release
    rows = []
    for i in range(0, len(s) - 4, 4):
        load = s[i] - (s[i - 1] if i else s[i])
        rows.append((load, s[i + 1] - s[i], s[i + 2] - s[i + 1], s[i + 3] - s[i + 2]))
    for j, r in enumerate(rows[:24]):
        print(f"   phase {j:2d}: load {r[0]:5d}  wait {r[1]:5d}  mfma {r[2]:5d}  wait {r[3]:5d}   (sum {sum(r)})")
    tot = s[-1] - s[0]
    print(f"   {len(rows)} phases in {tot} ticks = {tot/max(1,len(rows)):.0f} per phase")
