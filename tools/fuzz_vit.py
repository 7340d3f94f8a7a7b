"""ViT-S/16 + adapter extractor at odd batch sizes (row counts that are not multiples of any tile) against the ViT oracle, both
arithmetics and the plain-fp32 setting.  python tools/fuzz_vit.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vit_oracle as vorc  # noqa: E402  (the checker)
from snuffy_amd import vit  # noqa: E402

DEV = "cuda"
torch.manual_seed(0)
model = vit.vit_small(patch_size=16, adapter_ffn_scalar="10", adapter_ffn_num=32, adapter_d_model=384)
with torch.no_grad():
    for n_, p in model.named_parameters():
        if "adaptmlp.up_proj" in n_ or n_.endswith(".bias"):
            p.normal_(0.0, 0.02)
model = model.to(DEV).eval()
sd = {k: v.cpu() for k, v in model.state_dict().items()}
bad = 0
for bsz in (1, 2, 3, 7, 13, 50, 129):
    x = torch.rand(bsz, 3, 224, 224, generator=torch.Generator().manual_seed(bsz)).to(DEV)
    take = list(range(min(bsz, 4)))
    ref = vorc.vit_forward(x[take].cpu(), sd, 16, 12, 6, 10.0, "dino_adapter")
    for precision, gemm, tol in (("fp32", "x3", 1e-3), ("fp32", "library", 1e-4), ("bf16", "x3", 1e-2)):
        vit.FP32_GEMM = gemm
        model.configure(precision)
        with torch.no_grad():
            feats = model(x)
        e = (feats[take].cpu().float() - ref).abs().max().item() / ref.abs().max().item()
        flag = "" if e < tol else "   <-- CHECK"
        bad += bool(flag)
        print("batch %3d %s/%s: rel err %.2e%s" % (bsz, precision, gemm, e, flag), flush=True)
vit.FP32_GEMM = "x3"
print("cases outside the class:", bad)
