#!/usr/bin/env python3
"""snf_gemm_bf16 vs the library GEMM (hipBLASLt through torch) on the hot path's shapes.  python tools/gemm_bench.py [vit]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import timed  # noqa: E402
from snuffy_amd import ops  # noqa: E402

dev = torch.device("cuda")
SHAPES = {
    "cfgB": [(32768, 1536, 768, "none", "Q|V"), (32768, 3072, 768, "relu", "FFN-in"), (32768, 768, 3072, "none", "FFN-out")],
    "cfgA": [(8192, 768, 384, "none", "Q|V"), (8192, 1536, 384, "relu", "FFN-in"), (8192, 384, 1536, "none", "FFN-out")],
    "vit": [(100864, 384, 768, "none", "patch-embed"), (100864, 1152, 384, "none", "qkv"), (100864, 384, 384, "none", "proj"),
            (100864, 1536, 384, "gelu", "fc1"), (100864, 384, 1536, "none", "fc2")],
}


def main():
    which = sys.argv[1:] or ["cfgB"]
    g = torch.Generator().manual_seed(0)
    for name in which:
        for m, n, k, act, label in SHAPES[name]:
            nset = max(2, int(600e6 // (m * (k + n) * 2)))
            As = [torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev) for _ in range(nset)]
            w = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
            b = torch.randn(n, generator=g).to(dev)
            bh = b.to(torch.bfloat16)
            outs = [torch.empty(m, n, dtype=torch.bfloat16, device=dev) for _ in range(nset)]
            st = {"i": 0}

            def lib():
                i = st["i"] = (st["i"] + 1) % nset
                if act == "relu":
                    torch._addmm_activation(bh, As[i], w.t(), out=outs[i])
                elif act == "gelu":
                    torch._addmm_activation(bh, As[i], w.t(), use_gelu=True, out=outs[i])
                else:
                    torch.addmm(bh, As[i], w.t(), out=outs[i])
            flops = 2.0 * m * n * k
            t_lib = timed(lib, 20, warmup=3)
            line = f"{name:5s} {label:11s} m={m} n={n} k={k} {act:5s}: library {t_lib*1e3:7.1f} us {flops/t_lib/1e9:7.1f} TF/s"
            for tn in (256, 128):
                def ours():
                    i = st["i"] = (st["i"] + 1) % nset
                    ops.gemm_bf16(As[i], w, b, act, out=outs[i], tile_n=tn)
                t = timed(ours, 20, warmup=3)
                line += f" | ours tile_n={tn} {t*1e3:7.1f} us {flops/t/1e9:7.1f} TF/s"
            print(line, flush=True)
            del As, outs


if __name__ == "__main__":
    main()
