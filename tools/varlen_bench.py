#!/usr/bin/env python3
"""Small bags: one bag per forward (graph replay, the best the per-bag path does) against MILNet.forward_bags (B bags per launch,
eager issue and graph replay).  slides/s of the whole aggregator on one MI355X, eval forward, h = 6, Lambda = 200.

    python tools/varlen_bench.py [--steps 20] > gpurun_out/varlen.md
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_net  # noqa: E402


def rate(fn, bags_per_call, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return bags_per_call * steps * 1e3 / e0.elapsed_time(e1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--lam", type=int, default=200)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    print("| N | D | arithmetic | per bag, graph replay (slides/s) | B | packed, eager issue | packed, graph replay | x per-bag |")
    print("|" + "---|" * 8)
    g = torch.Generator().manual_seed(1234)
    for D in (384, 768):
        for N, Bs in ((1000, (8, 16, 64)), (4096, (8, 16)), (8192, (4, 8, 16))):
            for prec in ("fp32", "bf16"):
                net = build_net(D, 6, args.lam, prec, dev).eval()
                pool = [torch.randn(1, N, D, generator=g).to(dev) for _ in range(max(Bs))]
                with torch.no_grad():
                    net.configure(graph_max_patches=1 << 20)
                    it = iter(range(10 ** 9))
                    base = rate(lambda: net(pool[next(it) % len(pool)]), 1, 4 * args.steps)
                    for B in Bs:
                        bags = pool[:B]
                        net.configure(graph_max_patches=0)
                        eager = rate(lambda: net.forward_bags(bags), B, args.steps)
                        net.configure(graph_max_patches=1 << 20)
                        graph = rate(lambda: net.forward_bags(bags), B, args.steps)
                        print("| %d | %d | %s | %.0f | %d | %.0f | %.0f | %.1f |" % (N, D, prec, base, B, eager, graph, graph / base),
                              flush=True)
                del net, pool
                torch.cuda.empty_cache()
    musk_like(dev, args.steps)


def musk_like(dev, steps):
    """The MIL benchmark shape (train.py:993-995): 92 bags of 2-40 instances, D = 166, --num_heads 2, Lambda = 200 (every row of a
    bag is selected): one bag per forward against all 92 in one set of launches."""
    import numpy as np
    rs = np.random.RandomState(0)
    sizes = [int(v) for v in rs.randint(2, 41, 92)]
    g = torch.Generator().manual_seed(5)
    print("\n| set | arithmetic | per bag, eager (slides/s) | per bag, graph replay | packed, eager | packed, graph replay |")
    print("|" + "---|" * 6)
    for prec in ("fp32", "bf16"):
        net = build_net(166, 2, 200, prec, dev).eval()
        bags = [torch.randn(1, n, 166, generator=g).to(dev) for n in sizes]
        with torch.no_grad():
            net.configure(graph_max_patches=0)
            e_bag = rate(lambda: [net(x) for x in bags], len(bags), steps)
            e_pk = rate(lambda: net.forward_bags(bags), len(bags), steps)
            net.configure(graph_max_patches=1 << 16)
            g_bag = rate(lambda: [net(x) for x in bags], len(bags), steps)
            g_pk = rate(lambda: net.forward_bags(bags), len(bags), steps)
        print("| 92 bags x 2-40 instances, D=166, h=2 | %s | %.0f | %.0f | %.0f | %.0f |" % (prec, e_bag, g_bag, e_pk, g_pk), flush=True)


if __name__ == "__main__":
    main()
