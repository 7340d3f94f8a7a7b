#!/usr/bin/env python3
"""fp32-class projections: the one-pass kernel on interleaved images (snf_gemm_hl_bf16) vs the same contraction as a bf16 GEMM over 3 k concatenated
columns vs the fp32 library GEMM, on the aggregator's and the extractor's shapes.   python tools/gemm_x3_bench.py [cfgB cfgA vit]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import timed  # noqa: E402
from snuffy_amd import ops  # noqa: E402
from tools.gemm_bench import SHAPES  # noqa: E402

dev = torch.device("cuda")


def main():
    g = torch.Generator().manual_seed(0)
    for name in (sys.argv[1:] or ["cfgB"]):
        for m, n, k, act, label in SHAPES[name]:
            a = torch.randn(m, k, generator=g).to(dev)
            w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
            b = torch.randn(n, generator=g).to(dev)
            a3, w3 = ops.split3_rows(a), ops.split3_weight(w)
            out = torch.empty(m, n, dtype=torch.float32, device=dev)
            flops = 2.0 * m * n * k
            t_lib = timed(lambda: torch.addmm(b, a, w.t(), out=out), 5, warmup=2)
            t_cat = timed(lambda: ops.gemm_x3(a3, w3, b, "none", out=out), 10, warmup=2)
            a_hl, w_hl = ops.split_hl_rows(a), ops.split_hl_weight(w)
            t_one = timed(lambda: ops.gemm_hl(a_hl, w_hl, b, "none", out=out), 10, warmup=2)
            print(f"{name:5s} {label:11s} m={m} n={n} k={k}: fp32 library {t_lib*1e3:8.1f} us {flops/t_lib/1e9:6.1f} TF/s | x3 concatenated "
                  f"{t_cat*1e3:7.1f} us {flops/t_cat/1e9:6.1f} TF/s ({3*flops/t_cat/1e9:6.1f} issued) | x3 one pass {t_one*1e3:7.1f} us "
                  f"{flops/t_one/1e9:6.1f} TF/s ({3*flops/t_one/1e9:6.1f} issued)", flush=True)


if __name__ == "__main__":
    main()
