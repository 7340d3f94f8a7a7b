"""Random single-bag models against the CPU oracle (fp64): selection bit-exact, logits / A within the arithmetic's class.  Covers
shapes the fixtures do not (N up to 40 k, Lambda up to 512 incl. key chunks, head widths inside and outside the MFMA kernels,
depth 1-2, every activation, random share).  python tools/fuzz_model.py [n_cases]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import snuffy_oracle as orc  # noqa: E402  (the checker, not the product)
from snuffy_amd.snuffy import build_milnet  # noqa: E402

DEV = "cuda"
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(7)
bad = 0
for case in range(ncases):
    d, h = [(64, 1), (128, 2), (384, 6), (768, 6), (166, 2), (256, 4), (512, 4), (96, 3)][rs.randint(8)]
    lam = int(rs.choice([8, 50, 200, 224, 300, 512]))
    act = ["relu", "gelu", "leakyrelu", "selu"][rs.randint(4)]
    depth = int(rs.choice([1, 1, 2]))
    r = float(rs.choice([0.0, 0.0, 0.25]))
    n = int(np.clip(np.round(np.exp(rs.uniform(0, np.log(40000)))), 1, 40000))
    precision = ["fp32", "bf16"][rs.randint(2)]
    torch.manual_seed(case)
    net = build_milnet(d, h, act, lam, r, depth).to(DEV).eval()
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() > 1:
                torch.nn.init.xavier_uniform_(p)
    net.configure(precision=precision, return_attention=True)
    x = torch.randn(1, n, d, generator=torch.Generator().manual_seed(100 + case)).to(DEV)
    sd = {k: v.detach().cpu().double() for k, v in net.state_dict().items()}
    np.random.seed(case)
    with torch.no_grad():
        classes, logits, attn = net(x)
    sel_gpu = net.b_classifier.encoder.layers[0].last_selection
    np.random.seed(case)
    c64, l64, a64, sels = orc.milnet_forward(x[0].cpu().double(), sd, h, act, lam, r, depth)
    top = sel_gpu[0].cpu().numpy()
    ref_sel = sels[0].numpy()
    ok_sel = np.array_equal(np.concatenate([top, sel_gpu[1].cpu().numpy()]) if sel_gpu[1] is not None else top, ref_sel)
    if not ok_sel:
        # fp32 critic scores vs fp64: near-ties may order differently -- accept only if the score gap at the disagreement is at rounding level
        cs = c64[:, 0].numpy()
        k1 = len(top)
        diff = set(top.tolist()) ^ set(ref_sel[:k1].tolist())
        gap = max(abs(cs[i] - np.sort(cs)[::-1][k1 - 1]) for i in diff) if diff else 0.0
        ok_sel = gap < 1e-5 * max(1.0, np.abs(cs).max())
    el = (logits[0].cpu().double() - l64).abs().max().item() / max(1.0, l64.abs().max().item())
    ea = (attn[0].cpu().double() - a64).abs().max().item() if ok_sel and attn.shape[1:] == a64.shape else float("nan")
    tl, ta = (1e-3, 1e-3) if precision == "fp32" else (2e-2, 5e-2)
    flag = "" if (ok_sel and el <= tl and (not ea == ea or ea <= ta)) else "   <-- CHECK"
    if flag:
        bad += 1
    print("case %2d N=%5d D=%3d h=%d Lambda=%3d r=%.2f depth=%d %-9s %s: selection %s  |dlogit| %.1e  |dA| %.1e%s"
          % (case, n, d, h, lam, r, depth, act, precision, "exact" if ok_sel else "DIFFERS", el, ea, flag), flush=True)
    del net, x
print("cases outside the class:", bad)
