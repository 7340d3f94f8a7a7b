#!/usr/bin/env python3
"""ViT extractor throughput (BASELINE.json config 4: DINO ViT-S/16 + adapter, 224x224 tiles, batch 512, 1 MI355X).
   python tools/bench_vit.py [--batch 512] [--steps 5] [--precision bf16|fp32] [--arch vit_small|vit_base]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import vit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--arch", default="vit_small")
ap.add_argument("--gemm-table", action="store_true", help="apply snuffy_amd/tuning/gemm_gfx950.csv (library GEMM selections)")
ap.add_argument("--tune-gemms", default=None, metavar="CSV", help="tune unseen GEMM shapes online and record them in CSV")
a = ap.parse_args()
if a.gemm_table or a.tune_gemms:
    from snuffy_amd.gemm_tuning import use_pretuned_gemms
    use_pretuned_gemms(path=a.tune_gemms, tune_missing=bool(a.tune_gemms))
dev = torch.device("cuda")
torch.manual_seed(0)
width = {"vit_small": 384, "vit_base": 768}[a.arch]
model = getattr(vit, a.arch)(patch_size=16, adapter_ffn_scalar="10", adapter_ffn_num=32, adapter_d_model=width)
model = model.to(dev).eval().configure(a.precision)
x = torch.rand(a.batch, 3, 224, 224, device=dev)
for _ in range(a.warmup):
    model(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    model(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
heads = width // 64
T = 197
flops_img = 12 * (2 * T * width * 3 * width + 4 * T * T * width + 2 * T * width * width + 16 * T * width * width
                  + 4 * T * width * 32) + 2 * 196 * 768 * width
print(json.dumps({"metric": "images/sec", "value": round(a.batch / dt, 1), "unit": "img/s", "arch": a.arch + "/16+adapter",
                  "batch": a.batch, "ms_per_batch": round(dt * 1e3, 2), "dtype": a.precision,
                  "model_tflops_per_s": round(flops_img * a.batch / dt / 1e12, 1),
                  "mfma_frac_of_2.5PF": round(flops_img * a.batch / dt / 2.5e15, 4),
                  "gemm_table": bool(a.gemm_table or a.tune_gemms)}))
