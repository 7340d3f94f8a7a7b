#!/usr/bin/env python3
"""North-star sweep: synthetic bags N in {1k, 8k, 32k, 100k} x D in {384, 768} (h = 6, Lambda = 200, bf16 path, eval forward)
on one MI355X: slides/s of the whole aggregator, the sparse-attention and top-Lambda kernels alone, and their share of the
8 TB/s HBM roof (at the kernels' own operand width and at the fp32 byte figure of SURVEY section 8(d)).

    python tools/sweep.py [--steps 30] > gpurun_out/sweep.md
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_net, kernel_rooflines  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--lam", type=int, default=200)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    print("| N | D | K | slides/s (graph replay) | slides/s (eager issue) | ms/bag | attention µs | top-Λ µs | attn frac (bf16 bytes) "
          "| attn+top-Λ frac (§8(d) fp32 bytes) |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for D in (384, 768):
        for N in (1000, 8192, 32768, 100000):
            wl = dict(N=N, D=D, h=6, lam=args.lam)
            net = build_net(D, 6, args.lam, "bf16", dev).eval()
            nb = max(2, min(8, int(2.0e9 // (N * D * 4))))
            g = torch.Generator().manual_seed(1234)
            bags = [torch.randn(1, N, D, generator=g).to(dev) for _ in range(nb)]
            res = {}
            with torch.no_grad():
                for mode in ("eager", "graph"):
                    net.configure(graph_max_patches=(1 << 20) if mode == "graph" else 0)
                    for i in range(2 * nb + 4):
                        net(bags[i % nb])
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for i in range(args.steps):
                        net(bags[i % nb])
                    e1.record()
                    torch.cuda.synchronize()
                    res[mode] = e0.elapsed_time(e1) / args.steps
            ms = res["graph"]
            r = kernel_rooflines(wl, "bf16", dev, "sweep")
            ra, ru = r["roofline"], r["roofline_topk_attn"]
            print("| %d | %d | %d | %.0f | %.0f | %.3f | %.1f | %.1f | %.3f | %.3f |"
                  % (N, D, min(args.lam, N), 1e3 / ms, 1e3 / res["eager"], ms, ra["us_per_launch"], ru["us_topk"], ra["frac"],
                     ru["survey_8d_frac"]), flush=True)
            del bags, net
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
