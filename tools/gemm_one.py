#!/usr/bin/env python3
"""One GEMM shape, a few launches (target of the rocprofv3 passes in tools/pmc_gemm.sh).  python tools/gemm_one.py m n k [tile_n] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import ops  # noqa: E402

m, n, k = [int(v) for v in sys.argv[1:4]]
tile_n = int(sys.argv[4]) if len(sys.argv) > 4 else 0
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 6
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
As = [torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev) for _ in range(3)]
w = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
b = torch.randn(n, generator=g).to(dev)
out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
for i in range(iters):
    ops.gemm_bf16(As[i % 3], w, b, "none", out=out, tile_n=tile_n)
torch.cuda.synchronize()
