"""Selector timings (one GPU): the one-workgroup top-k, the multi-workgroup form on its own, and the fused pipeline
critic(+histogram) -> select next to critic -> one-workgroup top-k.   python tools/topk_bench.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import ops  # noqa: E402


def timed(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = "cuda"
    print("| N | D | K | top-k 1 WG us | hist+select us | critic us | critic+hist us | critic -> 1 WG us | critic+hist -> select us | critic_ln -> 1 WG | critic_ln+hist -> select |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for n, d, k in ((8192, 384, 200), (12288, 384, 200), (32768, 384, 200), (32768, 768, 200), (100000, 768, 200), (100000, 768, 512), (300000, 768, 512)):
        g = torch.Generator().manual_seed(n)
        x = torch.randn(n, d, generator=g).to(dev)
        w = (torch.randn(1, d, generator=g) / math.sqrt(d)).to(dev)
        b = torch.zeros(1, device=dev)
        s = ops.critic(x, w, b).view(-1)
        keep = ops.SELECT_FUSED_MIN_N
        ops.SELECT_FUSED_MIN_N = 1 << 40
        t1 = timed(lambda: ops.topk(s, k))
        t3 = timed(lambda: ops.critic(x, w, b))
        t5 = timed(lambda: ops.topk(ops.critic(x, w, b).view(-1), k))
        t7 = timed(lambda: ops.topk(ops.critic_ln(x, w, b, 1e-5)[0].view(-1), k))
        ops.SELECT_FUSED_MIN_N = 0
        t2 = timed(lambda: ops.topk_hist_select(s, k))

        def crit_only():
            ops.critic_select(x, w, b)
            ops.selector(x.device).pending = None
            ops.selector(x.device).state.zero_()
        t4 = timed(crit_only) - timed(lambda: ops.selector(x.device).state.zero_())
        t6 = timed(lambda: ops.topk(ops.critic_select(x, w, b)[0].view(-1), k))
        t8 = timed(lambda: ops.topk(ops.critic_select(x, w, b, 1e-5)[0].view(-1), k))
        ops.SELECT_FUSED_MIN_N = keep
        print("| %d | %d | %d | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f |" % (n, d, k, t1, t2, t3, t4, t5, t6, t7, t8))


if __name__ == "__main__":
    main()
