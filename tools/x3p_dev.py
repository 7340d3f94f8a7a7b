"""Development driver of the pipelined fp32-class attention kernel (snf_sparse_attn_fwd_x3_hl): parity against the round-3 kernel
and the fp64 oracle, then timing on cold rotating operands.  python tools/x3p_dev.py [n k h [dk]] [--time] [--kbw=1|2]
(--kbw: key blocks per wave of the dk = 128 family, snf_debug_x3p_kbw; default 1)"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import ops  # noqa: E402

DEV = "cuda:0"


def ref64(q, kp, v, h):
    n, d = q.shape
    k = kp.shape[0]
    dk = d // h
    qh = q.double().view(n, h, dk).transpose(0, 1)
    kh = kp.double().view(k, h, dk).transpose(0, 1)
    vh = v.double().view(n, h, dk).transpose(0, 1)
    p = torch.softmax(qh @ kh.transpose(1, 2) / math.sqrt(dk), dim=-1)
    o = (p.transpose(1, 2) @ vh).transpose(0, 1).reshape(k, d)
    return o, p


def check(n, k, h, dk=128, seed=0):
    g = torch.Generator().manual_seed(seed)
    d = h * dk
    q, kp, v = torch.randn(n, d, generator=g), torch.randn(k, d, generator=g), torch.randn(n, d, generator=g)
    qv = torch.cat([q, v], dim=1).to(DEV)
    img = ops.split_hl_rows(qv)                     # [n, 4 d] bf16: Q image | V image
    o, attn, lse = ops.sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], kp.to(DEV), h, need_attn=True, need_lse=True)
    o2, a2, _ = ops.sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], kp.to(DEV), h)
    torch.cuda.synchronize()
    o_ref, p_ref = ref64(q, kp, v, h)
    sc = o_ref.abs().max()
    e_o = float((o.cpu().double() - o_ref).abs().max() / sc)
    e_o2 = float((o2.cpu().double() - o_ref).abs().max() / sc)
    e_p = float((attn.cpu().double() - p_ref).abs().max())
    print(f"n={n} k={k} h={h} dk={dk}: O err {e_o:.2e} (no aux {e_o2:.2e}) P err {e_p:.2e} bitwise aux==noaux {bool(torch.equal(o, o2))}", flush=True)
    if k <= 224 and os.environ.get("X3P_VS_R3"):
        o3, a3, _ = ops.sparse_attn_fwd_x3(qv[:, :d], qv[:, d:], kp.to(DEV), h, need_attn=True)
        print(f"   vs round-3 kernel: O {float((o - o3).abs().max() / sc):.2e}  P {float((attn - a3).abs().max()):.2e}")
    return e_o < 2e-5 and e_o2 < 2e-5 and e_p < 6e-6


def timeit(n, k, h, dk=128, reps=120, nbuf=6):
    d = h * dk
    bufs = []
    for i in range(nbuf):
        qv = torch.randn(n, 2 * d, device=DEV)
        bufs.append((qv, ops.split_hl_rows(qv)))
    kp = torch.randn(k, d, device=DEV)
    legs = [("x3_hl", lambda b: ops.sparse_attn_fwd_x3_hl(b[1][:, :2 * d], b[1][:, 2 * d:], kp, h))]
    if os.environ.get("X3P_VS_R3"):
        legs.append(("x3   ", lambda b: ops.sparse_attn_fwd_x3(b[0][:, :d], b[0][:, d:], kp, h)))
    for name, fn in legs:
        if name.startswith("x3 ") and k > 224 * 8:
            continue
        for b in bufs:
            fn(b)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for r in range(reps):
            ev[r][0].record()
            fn(bufs[r % nbuf])
            ev[r][1].record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev[reps // 6:])     # (the first launches run on the clock ramp)
        mid = ts[len(ts) // 4: 3 * len(ts) // 4]
        iqm = sum(mid) / len(mid)                                             # interquartile mean: the figure to compare builds with
        byts = 8 * n * d + 8 * k * d
        print(f"{name} n={n} k={k} h={h} dk={dk}: median {ts[len(ts) // 2]:.1f} us  min {ts[0]:.1f} us  iqm {iqm:.2f} us  -> "
              f"{byts / ts[len(ts) // 2] / 1e6:.2f} TB/s ({byts / ts[len(ts) // 2] / 1e6 / 8:.3f} of 8 TB/s)", flush=True)


if __name__ == "__main__":
    from snuffy_amd import _ffi
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    for a in sys.argv[1:]:
        if a.startswith("--kbw="):
            _ffi.load().snf_debug_x3p_kbw(int(a[6:]))
            print("key blocks per wave (dk = 128):", a[6:])
    ok = True
    if args:
        shapes = [tuple(int(x) for x in args[:4])]
    else:
        shapes = [(64, 200, 1), (1000, 200, 6), (4097, 224, 3), (5000, 100, 2), (2000, 256, 2), (777, 129, 2), (3000, 160, 2), (6401, 200, 6),
                  (3000, 512, 6), (1000, 200, 6, 64), (4097, 256, 3, 64), (777, 129, 2, 64), (5000, 97, 12, 64), (3000, 512, 6, 64)]
    for sh in shapes:
        ok = check(*sh) and ok
    print("PARITY", "OK" if ok else "FAIL")
    if "--time" in sys.argv:
        timeit(*(shapes[0] if args else (32768, 200, 6)))
