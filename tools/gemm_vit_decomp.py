#!/usr/bin/env python3
"""ViT-S/16 batch-512 GEMM shapes on snf_gemm_bf16, for the timing ablations of tools/gemm_variants.sh (round 6):
   for v in base nomfma nostore; do SNUFFY_HIP_LIB=snuffy_amd/build/variants/lib_$v.so python tools/gemm_vit_decomp.py; done"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import timed  # noqa: E402
from snuffy_amd import ops  # noqa: E402

dev = torch.device("cuda")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100864
SHAPES = [(M, 1152, 384, "none", "qkv"), (M, 384, 384, "none", "proj"), (M, 1536, 384, "gelu", "fc1"), (M, 1536, 384, "none", "fc1-noact"),
          (M, 1536, 384, "relu", "fc1-relu"), (M, 384, 1536, "none", "fc2"), (M, 64, 384, "relu", "adapter-down"), (M, 384, 64, "none", "adapter-up")]
g = torch.Generator().manual_seed(0)
tag = os.path.basename(os.environ.get("SNUFFY_HIP_LIB", "shipped"))
for m, n, k, act, label in SHAPES:
    nset = max(2, int(600e6 // (m * (k + n) * 2)))
    As = [torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev) for _ in range(nset)]
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(n, generator=g).to(dev)
    outs = [torch.empty(m, n, dtype=torch.bfloat16, device=dev) for _ in range(nset)]
    st = {"i": 0}
    line = f"{tag:16s} {label:12s} m={m} n={n} k={k} {act:5s}:"
    for tn in (256, 128):
        def ours():
            i = st["i"] = (st["i"] + 1) % nset
            ops.gemm_bf16(As[i], w, b, act, out=outs[i], tile_n=tn)
        t = timed(ours, 20, warmup=3)
        line += f"  tile_n={tn} {t*1e3:7.1f} us {2.0*m*n*k/t/1e9:7.1f} TF/s"
    print(line, flush=True)
    del As, outs
