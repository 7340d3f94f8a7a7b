"""Development: config-B gemm_hl launches as the fp32-class bag issues them (Q|V -> hl image, FFN-in relu -> hl image, FFN-out + residual),
timed over 300 launches each; run under different SNF_GEMM_* environment switches for same-box A/B comparisons."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import ops  # noqa: E402

m = 32768
for name, n, k, kw in (("Q|V", 1536, 768, dict(hl_out=True)), ("FFN-in", 3072, 768, dict(act="relu", hl_out=True)), ("FFN-out", 768, 3072, dict(resid=True))):
    a = ops.split_hl_rows(torch.randn(m, k, device="cuda"))
    w = ops.split_hl_weight(torch.randn(n, k, device="cuda") / k ** 0.5)
    b = torch.randn(n, device="cuda")
    kw = dict(kw)
    if kw.pop("resid", False):
        kw["resid"] = torch.randn(m, n, device="cuda")
    for _ in range(20):
        ops.gemm_hl(a, w, b, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        ops.gemm_hl(a, w, b, **kw)
    e1.record()
    torch.cuda.synchronize()
    print("%-8s %.1f us" % (name, e0.elapsed_time(e1) * 1e3 / 300), flush=True)
