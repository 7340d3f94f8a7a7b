"""Does the one-pass GEMM's K step get shorter when its operands are L2-resident?  Small problems launched repeatedly (operands a few
MB: hot in the XCDs' L2s from the previous launch), the K axis varied: the step time is the slope.   python tools/gemm_hl_l2_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import timed  # noqa: E402
from snuffy_amd import ops  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
for m, n in ((1024, 1024), (2048, 2048), (4096, 4096), (32768, 768), (32768, 3072)):
    ts = {}
    for k in (512, 1024, 2048, 4096):
        if m * k * 4 > 1.7e9:
            continue
        a = torch.randn(m, k, generator=g).to(dev)
        w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
        a_hl, w_hl = ops.split_hl_rows(a), ops.split_hl_weight(w)
        out = torch.empty(m, n, dtype=torch.float32, device=dev)
        ts[k] = timed(lambda: ops.gemm_hl(a_hl, w_hl, None, "none", out=out), 30, warmup=5) * 1e3
    ks = sorted(ts)
    tiles = ((m + 255) // 256) * ((n + 255) // 256)
    rounds = max(1.0, tiles / 256.0)
    slope = (ts[ks[-1]] - ts[ks[0]]) / ((ks[-1] - ks[0]) / 32) / rounds
    print("m=%6d n=%5d (%4d tiles, operands %5.1f + %5.1f MB at k=%d): %s  -> %.2f us per 32-column step and round"
          % (m, n, tiles, m * ks[-1] * 4 / 1e6, n * ks[-1] * 4 / 1e6, ks[-1], "  ".join("k=%d %.1f us" % (k, ts[k]) for k in ks), slope), flush=True)
