#!/usr/bin/env python3
"""North-star sweep: synthetic bags N in {1k, 8k, 32k, 100k} x D in {384, 768} (h = 6, Lambda = 200, eval forward) on one MI355X,
in BOTH arithmetics (fp32-class = the reference's, bf16), next to the reference's CPU path (the parity-checked torch-CPU port of
oracle/, timed on this host's cores; core count stated): slides/s of the whole aggregator, the sparse-attention and top-Lambda
kernels alone and their share of the 8 TB/s HBM roof (at the kernels' own operand width and at the fp32 byte figure of SURVEY 8(d)).

    python tools/sweep.py [--steps 30] [--cpu-threads 32] [--no-cpu] > gpurun_out/sweep.md
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_net, kernel_rooflines  # noqa: E402


def cpu_rate(N, D, lam, threads, budget_s=6.0, h=6, r=0.0):
    """slides/s of the CPU port (oracle.milnet_forward, A materialised as the reference does) at `threads` threads."""
    from oracle import snuffy_oracle as orc          # CPU baseline column: the checker, timed -- never the product path
    net = build_net(D, h, lam, "fp32", "cpu", r)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    x = torch.randn(N, D, generator=torch.Generator().manual_seed(1234))
    torch.set_num_threads(threads)
    with torch.no_grad():
        orc.milnet_forward(x, sd, h, "relu", lam, r, 1)
        t0, n = time.perf_counter(), 0
        while True:
            orc.milnet_forward(x, sd, h, "relu", lam, r, 1)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 50:
                break
    return n / el


def gpu_ms(net, bags, steps, graph, preroll_s=0.3):
    """ms per forward: warm-up calls (captures included), an untimed time-based pre-roll of the same step (as bench.py: the region
    starts at the sustained clock, not on the ramp or behind one-time costs), then the best of two timed regions.  Round 5's committed
    sweep held one cold first row (N = 1000, D = 384, fp32-class: 651 slides/s against 4.5 k in every other run of that shape --
    VERDICT r5 weak #6); a fresh process does not reproduce it (4502 / 4524 / 4507 slides/s on three first rows), the guard stays."""
    nb = len(bags)
    net.configure(graph_max_patches=(1 << 20) if graph else 0)
    with torch.no_grad():
        for i in range(2 * nb + 4):
            net(bags[i % nb])
        torch.cuda.synchronize()
        t0, i = time.perf_counter(), 0
        while time.perf_counter() - t0 < preroll_s:
            net(bags[i % nb])
            i += 1
            if i % 16 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        best = float("inf")
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(steps):
                net(bags[i % nb])
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / steps)
    return best


def packed_rate(net, N, D, dev, steps, nbags=16):
    """slides/s of MILNet.forward_bags over `nbags` bags of N patches (graph replay)."""
    g = torch.Generator().manual_seed(4321)
    bags = [torch.randn(1, N, D, generator=g).to(dev) for _ in range(nbags)]
    net.configure(graph_max_patches=1 << 20)
    with torch.no_grad():
        for _ in range(3):
            net.forward_bags(bags)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            net.forward_bags(bags)
        e1.record()
        torch.cuda.synchronize()
    return nbags * steps * 1e3 / e0.elapsed_time(e1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--lam", type=int, default=200)
    ap.add_argument("--cpu-threads", type=int, default=32, help="threads of the CPU column (bench.py's sweep finds 16-32 best on a 128-core host)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cpu_model = "?"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = [ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    print("one MI355X; CPU column: oracle/snuffy_oracle.py (torch-CPU fp32 port of the reference's op sequence) on %s, %d threads of %d cores\n"
          % (cpu_model, args.cpu_threads, os.cpu_count() or 0))
    print("| N | D | K | CPU slides/s | fp32-class slides/s | x CPU | bf16 slides/s | x CPU | fp32 attention µs | frac of 8 TB/s | bf16 attention µs "
          "| frac (bf16 bytes) | top-Λ µs (in pipeline) | fp32 select + gather / key projection + attention frac (§8(d) bytes) | bf16 select + gather / key projection + attention frac (§8(d) bytes) | bf16 slides/s eager issue "
          "| fp32-class packed (64 bags per launch at N = 1000, 16 at 8192) | bf16 packed |")
    print("|" + "---|" * 18)
    for D in (384, 768):
        for N in (1000, 8192, 32768, 100000):
            wl = dict(N=N, D=D, h=6, lam=args.lam)
            nb = max(2, min(8, int(2.0e9 // (N * D * 4))))
            g = torch.Generator().manual_seed(1234)
            bags = [torch.randn(1, N, D, generator=g).to(dev) for _ in range(nb)]
            ms, roof, packed = {}, {}, {}
            for prec in ("fp32", "bf16"):
                net = build_net(D, 6, args.lam, prec, dev).eval()
                ms[prec] = gpu_ms(net, bags, args.steps, True)
                if prec == "bf16":
                    ms["bf16_eager"] = gpu_ms(net, bags, args.steps, False)
                if N <= 8192:      # small bags: MILNet.forward_bags, 16 bags per set of launches (graph replay)
                    packed[prec] = packed_rate(net, N, D, dev, args.steps, nbags=64 if N <= 1024 else 16)
                del net
                roof[prec] = kernel_rooflines(wl, prec, dev, "sweep")
            cpu = float("nan") if args.no_cpu else cpu_rate(N, D, args.lam, args.cpu_threads)
            rf, rb = roof["fp32"], roof["bf16"]
            print("| %d | %d | %d | %.2f | %.0f | %.0f | %.0f | %.0f | %.1f | %.3f | %.1f | %.3f | %.1f | %.3f | %.3f | %.0f | %s | %s |"
                  % (N, D, min(args.lam, N), cpu, 1e3 / ms["fp32"], 1e3 / ms["fp32"] / cpu, 1e3 / ms["bf16"], 1e3 / ms["bf16"] / cpu,
                     rf["roofline"]["us_per_launch"], rf["roofline"]["frac"], rb["roofline"]["us_per_launch"], rb["roofline"]["frac"],
                     rb["roofline_topk_attn"]["us_topk"], rf["roofline_topk_attn"]["survey_8d_frac"],
                     rb["roofline_topk_attn"]["survey_8d_frac"], 1e3 / ms["bf16_eager"],
                     "%.0f" % packed["fp32"] if "fp32" in packed else "–", "%.0f" % packed["bf16"] if "bf16" in packed else "–"),
                  flush=True)
            del bags
            torch.cuda.empty_cache()
    # the reference's published recipes (reference README.md:609-669; SURVEY 8(d): "also report h=4"): 4 heads, a random patch share
    # (device sampler, graph replay; "parity" = the reference's numpy draws on the host), bags of the CAMELYON16 mean length
    from bench import WORKLOADS
    print("\n| README recipe | N | D | h | dk | K (top + random) | CPU slides/s | fp32-class slides/s | x CPU | fp32-class, parity sampler | bf16 slides/s "
          "| fp32-class attention µs | frac of 8 TB/s | attention kernel |")
    print("|" + "---|" * 14)
    for name in ("readme_dino_scratch", "readme_dino_adapter", "readme_mae_adapter"):
        wl = WORKLOADS[name]
        N, D, h, lam, r = wl["N"], wl["D"], wl["h"], wl["lam"], wl["r"]
        g = torch.Generator().manual_seed(1234)
        bags = [torch.randn(1, N, D, generator=g).to(dev) for _ in range(4)]
        ms = {}
        for prec, smp in (("fp32", "device"), ("fp32", "reference"), ("bf16", "device")):
            net = build_net(D, h, lam, prec, dev, r).eval().configure(sampler=smp)
            ms[(prec, smp)] = gpu_ms(net, bags, args.steps, smp == "device")
            del net
        rf = kernel_rooflines(wl, "fp32", dev, name)["roofline"]
        cpu = float("nan") if args.no_cpu else cpu_rate(N, D, lam, args.cpu_threads, h=h, r=r)
        import math
        print("| %s | %d | %d | %d | %d | %d (%d + %d) | %.2f | %.0f | %.0f | %.0f | %.0f | %.1f | %.3f | %s |"
              % (name, N, D, h, D // h, lam, math.ceil(lam * (1 - r)), int(lam * r), cpu, 1e3 / ms[("fp32", "device")],
                 1e3 / ms[("fp32", "device")] / cpu, 1e3 / ms[("fp32", "reference")], 1e3 / ms[("bf16", "device")], rf["us_per_launch"],
                 rf["frac"], rf["kernel"]), flush=True)
        del bags
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
