#!/usr/bin/env python3
"""The fp32-class / bf16 attention launch at config B timed three ways: eager issue over rotating cold operand sets (bench.py's
roofline), the same launches captured into ONE HIP graph (no host between them), and one operand set only (warm)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import ops  # noqa: E402

N, D, h, K = 32768, 768, 6, 200
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(7)
kp = torch.randn(K, D, generator=g).to(dev)


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for prec in ("fp32", "bf16"):
    dt = torch.float32 if prec == "fp32" else torch.bfloat16
    nset = max(2, int(math.ceil(600e6 / (2 * N * D * (4 if prec == "fp32" else 2)))))
    qvs = [torch.randn(N, 2 * D, generator=g).to(dev).to(dt) for _ in range(nset)]
    kpi = kp.to(dt)

    def call(i):
        if prec == "fp32":
            ops.sparse_attn_fwd_x3(qvs[i][:, :D], qvs[i][:, D:], kp, h)
        else:
            ops.sparse_attn_fwd_mfma(qvs[i][:, :D], qvs[i][:, D:], kpi, N, h)
    st = {"i": 0}

    def rot():
        st["i"] = (st["i"] + 1) % nset
        call(st["i"])
    eager_cold = timed(rot, 20)
    eager_warm = timed(lambda: call(0), 20)
    for _ in range(3):
        rot()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(2 * nset):
            call(i % nset)
    graph_cold = timed(graph.replay, 5) / (2 * nset)
    graph1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph1):
        for i in range(6):
            call(0)
    graph_warm = timed(graph1.replay, 5) / 6
    print("%s (strided halves of [N, 2D], %d operand sets): eager cold %.1f us | eager one set %.1f | graph cold %.1f | graph one set %.1f"
          % (prec, nset, eager_cold, eager_warm, graph_cold, graph_warm), flush=True)
    del qvs
