#!/usr/bin/env python3
"""Kernel micro-benchmarks (HIP events on torch's current stream).  python tools/kbench.py [attn|topk|rows|all]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, timed  # noqa: E402
from snuffy_amd import ops  # noqa: E402

dev = torch.device("cuda")


def attn(wlname="cfgB", dts=("bf16", "f32"), iters=30, K=None, nset=4, N=None):
    wl = WORKLOADS[wlname]
    N0, D, h, lam = wl["N"], wl["D"], wl["h"], wl["lam"]
    N = N or N0
    K = K or min(lam, 256)
    g = torch.Generator().manual_seed(1)
    kp = torch.randn(K, D, generator=g).to(dev)
    for dt in dts:
        tdt = torch.bfloat16 if dt == "bf16" else torch.float32
        elt = 2 if dt == "bf16" else 4
        qvs = [torch.randn(N, 2 * D, generator=g).to(dev).to(tdt) for _ in range(nset)]   # fused projection output
        st = {"i": 0}
        kpi = kp.to(tdt)   # bf16 runs hand over a bf16 Kp, as the model's bf16 path does

        def f():
            st["i"] = (st["i"] + 1) % nset
            ops.sparse_attn_fwd_mfma(qvs[st["i"]][:, :D], qvs[st["i"]][:, D:], kpi, N, h)
        t = timed(f, iters, warmup=3)
        b = 2 * N * D * elt + K * D * (elt + 4)
        print(f"attn_mfma {wlname} N={N} K={K} nset={nset} {dt}: {t*1e3:8.1f} us  {b/t/1e6:8.1f} GB/s algorithmic  ({b/t/1e6/8000*100:.1f}% of 8 TB/s)"
              f"  {4*N*K*D/t/1e9:.1f} TFLOP/s")
        del qvs


def attn_x3(wlname="cfgB", iters=20, nset=3, exact=True):
    wl = WORKLOADS[wlname]
    N, D, h, lam = wl["N"], wl["D"], wl["h"], wl["lam"]
    K = min(lam, 224)
    g = torch.Generator().manual_seed(1)
    kp = torch.randn(K, D, generator=g).to(dev)
    qvs = [torch.randn(N, 2 * D, generator=g).to(dev) for _ in range(nset)]
    st = {"i": 0}

    def f():
        st["i"] = (st["i"] + 1) % nset
        ops.sparse_attn_fwd_x3(qvs[st["i"]][:, :D], qvs[st["i"]][:, D:], kp, h)

    def fe():
        st["i"] = (st["i"] + 1) % nset
        ops.sparse_attn_fwd(qvs[st["i"]][:, :D].contiguous(), kp, qvs[st["i"]][:, D:].contiguous(), h)
    t = timed(f, iters, warmup=3)
    b = 2 * N * D * 4 + 2 * K * D * 4
    print(f"attn_x3 {wlname} N={N} K={K} f32 operands: {t*1e3:8.1f} us  {b/t/1e6:8.1f} GB/s algorithmic ({b/t/1e6/8000*100:.1f}% of 8 TB/s)"
          f"  {4*N*K*D/t/1e9:.1f} TFLOP/s useful ({3*4*N*K*D/t/1e9:.1f} issued)")
    if exact:
        t = timed(fe, 5, warmup=1)
        print(f"attn exact (vector ALU) {wlname}: {t*1e3:8.1f} us")


def topk():
    g = torch.Generator().manual_seed(2)
    for n, k in [(8192, 200), (32768, 200), (100000, 512)]:
        c = torch.randn(n, generator=g).to(dev)
        t = timed(lambda: ops.topk(c, k), 30)
        print(f"topk n={n} k={k}: {t*1e3:.1f} us")


def rows():
    g = torch.Generator().manual_seed(3)
    N, D = 32768, 768
    x = torch.randn(N, D, generator=g).to(dev)
    w = torch.randn(1, D, generator=g).to(dev)
    b = torch.zeros(1, device=dev)
    gam = torch.ones(D, device=dev)
    t = timed(lambda: ops.critic(x, w, b), 20)
    print(f"critic: {t*1e3:.1f} us {N*D*4/t/1e6:.0f} GB/s")
    t = timed(lambda: ops.layernorm_rows(x, gam, gam, 1e-5), 20)
    print(f"layernorm f32->f32: {t*1e3:.1f} us {2*N*D*4/t/1e6:.0f} GB/s")
    t = timed(lambda: ops.layernorm_rows(x, None, None, 1e-5, out_dtype=torch.bfloat16), 20)
    print(f"layernorm f32->bf16: {t*1e3:.1f} us {N*D*6/t/1e6:.0f} GB/s")
    zb = torch.randn(N, D, generator=g).to(dev).to(torch.bfloat16)
    t = timed(lambda: ops.ln_mean_head(x, gam, gam, 1e-5, w, b, add_bf16=zb, add_bias=gam), 20)
    print(f"ln_mean_head(+bf16 add): {t*1e3:.1f} us {N*D*6/t/1e6:.0f} GB/s")
    hb = torch.randn(N, 4 * D, generator=g).to(dev).to(torch.bfloat16)
    bb = torch.zeros(4 * D, device=dev)
    t = timed(lambda: ops.bias_act_(hb, bb, "gelu"), 20)
    print(f"bias_act bf16 gelu: {t*1e3:.1f} us {N*4*D*4/t/1e6:.0f} GB/s")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "attnx":   # experiments: where does the time go?
        attn("cfgB", dts=("bf16",))
        attn("cfgB", dts=("bf16",), nset=1)           # operands stay in the 256 MiB Infinity Cache
        attn("cfgB", dts=("bf16",), N=4096, nset=1)   # L2-resident, 1/8 of the work
        attn("cfgB", dts=("bf16",), K=32)             # 1 key block: memory traffic unchanged, 1/7 of the math
        attn("cfgB", dts=("bf16",), K=128)
    if what == "attnB":
        attn("cfgB", dts=("bf16",), iters=10)
    if what in ("attn", "all"):
        attn("cfgB")
        attn("cfgA", dts=("bf16",))
    if what == "x3B":
        attn_x3("cfgB", iters=10, exact=False)
    if what in ("x3", "all"):
        attn_x3("cfgB")
        attn_x3("cfgA")
    if what in ("topk", "all"):
        topk()
    if what in ("rows", "all"):
        rows()
