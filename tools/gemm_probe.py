#!/usr/bin/env python3
"""Time the four dense projections of one config-B bag as the bf16 path issues them (library GEMMs), plus alternatives."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import timed  # noqa: E402
from snuffy_amd import ops  # noqa: E402

dev = torch.device("cuda")
N, D, F = 32768, 768, 3072
ldv = N + 64
g = torch.Generator().manual_seed(0)
bf = torch.bfloat16
xhat = torch.randn(ldv, D, generator=g).to(dev).to(bf)
wq = (torch.randn(D, D, generator=g) * 0.03).to(dev).to(bf)
wv = (torch.randn(D, D, generator=g) * 0.03).to(dev).to(bf)
w1 = (torch.randn(F, D, generator=g) * 0.03).to(dev).to(bf)
w2 = (torch.randn(D, F, generator=g) * 0.03).to(dev).to(bf)
bq = torch.randn(D, generator=g).to(dev).to(bf)
b1 = torch.randn(F, generator=g).to(dev).to(bf)
hid = torch.randn(N, F, generator=g).to(dev).to(bf)


def report(name, fn, flop, iters=20):
    t = timed(fn, iters, warmup=5)
    print(f"{name:58s} {t*1e3:8.1f} us  {flop/t/1e9:8.1f} TFLOP/s", flush=True)


fl_p, fl_f = 2 * N * D * D, 2 * N * D * F
print("tunableop:", os.environ.get("PYTORCH_TUNABLEOP_ENABLED", "0"), " ldv =", ldv)
report("q   = addmm(bq, xhat[:n], wq.t())            [N,D]", lambda: torch.addmm(bq, xhat[:N], wq.t()), fl_p)
report("q   = mm(xhat[:n], wq.t())  (no bias)", lambda: torch.mm(xhat[:N], wq.t()), fl_p)
report("vt  = addmm(bv[:,None], wv, xhat.t())        [D,ldv]", lambda: torch.addmm(bq.unsqueeze(1), wv, xhat.t()), 2 * ldv * D * D)
report("vt  = mm(wv, xhat.t())  (no bias)", lambda: torch.mm(wv, xhat.t()), 2 * ldv * D * D)
report("vt  = mm(wv, xhat[:n].t())  (n columns)", lambda: torch.mm(wv, xhat[:N].t()), fl_p)
wqv = torch.cat([wq, wv]).contiguous()
report("qv  = mm(xhat[:n], [wq;wv].t())              [N,2D]", lambda: torch.mm(xhat[:N], wqv.t()), 2 * fl_p)
report("hid = _addmm_activation(b1, xhat[:n], w1.t())  [N,F]", lambda: torch._addmm_activation(b1, xhat[:N], w1.t()), fl_f)
report("hid = mm(xhat[:n], w1.t())", lambda: torch.mm(xhat[:N], w1.t()), fl_f)
report("zb  = mm(hid, w2.t())                        [N,D]", lambda: torch.mm(hid, w2.t()), fl_f)
out = torch.empty(N, D, dtype=bf, device=dev)
report("zb  = mm(hid, w2.t(), out=)", lambda: torch.mm(hid, w2.t(), out=out), fl_f)
