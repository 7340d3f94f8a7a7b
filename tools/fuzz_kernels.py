"""Random-shape fuzz of the round-3 kernels (segmented top-k, skinny projection, ragged and varlen attention) against numpy / fp64 / their
per-bag forms; prints every mismatch and a total (0 on the final build).  python tools/fuzz_kernels.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import ops
DEV = "cuda"
rs = np.random.RandomState(0)
g = torch.Generator().manual_seed(0)
bad = 0
# 1. segmented top-k
for it in range(60):
    nb = int(rs.randint(1, 12)); k = int(rs.choice([1, 7, 64, 200, 512, 1000]))
    sizes = [int(v) for v in np.clip(np.round(np.exp(rs.uniform(0, np.log(40000), nb))), 1, 40000)]
    pk = ops.PackedBags(sizes, DEV)
    s = torch.randn(pk.total, generator=g)
    if it % 3 == 0: s = torch.round(s * 4) / 4          # many ties
    s = s.to(DEV)
    got = ops.topk_segmented(s, pk, k).cpu().numpy()
    sc = s.cpu().numpy()
    for b, n in enumerate(sizes):
        lo = int(pk.host[b]); kb = min(k, n)
        ref = np.lexsort((np.arange(n), -sc[lo:lo + n]))[:kb]
        if not np.array_equal(got[b, :kb], ref):
            bad += 1; print("topk mismatch", sizes, k, b)
print("topk fuzz done", bad)
# 2. skinny linear
for it in range(40):
    r = int(rs.randint(1, 700)); c = int(rs.randint(1, 1200)); k = 16 * int(rs.randint(1, 200))
    x = torch.randn(r, k, generator=g).to(DEV); w = (torch.randn(c, k, generator=g) / k ** 0.5).to(DEV); b = torch.randn(c, generator=g).to(DEV)
    y = ops.linear_rows_x3(x, w, b)
    ref = x.double() @ w.double().t() + b.double()
    e = (y.double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    if e > 3e-5: bad += 1; print("skinny", r, c, k, e)
print("skinny fuzz done", bad)
# 3. ragged attention vs exact per-bag
for it in range(25):
    h = int(rs.choice([1, 2, 3, 6])); dk = int(rs.choice([16, 64, 83, 115, 128])); d = h * dk
    nb = int(rs.randint(1, 10)); sizes = [int(v) for v in rs.randint(1, 400, nb)]
    kcap = int(rs.choice([8, 64, 200, 256])); kbs = [min(kcap, n) for n in sizes]
    if not ops.ragged_attn_supported(max(kbs), dk): continue
    pk = ops.PackedBags(sizes, DEV); rag = pk.ragged(kbs)
    q = torch.randn(pk.total, d, generator=g).to(DEV); v = torch.randn(pk.total, d, generator=g).to(DEV)
    kp = (torch.randn(sum(kbs), d, generator=g) * 0.5).to(DEV)
    out, attn, _ = ops.sparse_attn_fwd_ragged(q, v, kp, pk, rag, h, need_attn=True)
    for b, n in enumerate(sizes):
        lo, k0, kb = int(pk.host[b]), int(rag.koff[b]), kbs[b]
        o1, a1, _ = ops.sparse_attn_fwd(q[lo:lo + n], kp[k0:k0 + kb], v[lo:lo + n], h, need_attn=True)
        e = (out[k0:k0 + kb] - o1).abs().max().item() / max(1.0, o1.abs().max().item()); ea = (attn[:, lo:lo + n, :kb] - a1).abs().max().item()
        if e > 3e-5 or ea > 3e-6: bad += 1; print("ragged", h, dk, sizes, kbs, b, e, ea)
print("ragged fuzz done", bad)
# 4. varlen attention: composition independence + vs per-bag
for it in range(20):
    prec = ["bf16", "fp32"][it % 2]
    d, h = [(384, 6), (768, 6), (128, 1), (256, 2)][rs.randint(4)]; dk = d // h
    k = int(rs.choice([5, 32, 100, 200, 224]))
    if prec == "fp32" and k <= 32: k = 33 if dk == 128 else k
    nb = int(rs.randint(2, 8)); sizes = [int(v) for v in np.clip(np.round(np.exp(rs.uniform(np.log(k), np.log(20000), nb))), k, 20000)]
    pk = ops.PackedBags(sizes, DEV)
    dt = torch.bfloat16 if prec == "bf16" else torch.float32
    qv = torch.randn(pk.total, 2 * d, generator=g).to(DEV).to(dt); kp = (torch.randn(nb * k, d, generator=g) * 0.5).to(DEV).to(dt)
    fn = ops.sparse_attn_fwd_mfma_varlen if prec == "bf16" else ops.sparse_attn_fwd_x3_varlen
    try:
        out, attn, lse = fn(qv[:, :d], qv[:, d:], kp, pk, k, h, need_attn=True, need_lse=True)
    except Exception as exc:
        print("varlen unsupported", prec, d, h, k, str(exc)[:80]); continue
    for b, n in enumerate(sizes):
        lo = int(pk.host[b]); qb = qv[lo:lo + n]
        o2, a2, l2 = fn(qb[:, :d], qb[:, d:], kp[b * k:(b + 1) * k], ops.PackedBags([n], DEV), k, h, need_attn=True, need_lse=True)
        if not (torch.equal(out[b * k:(b + 1) * k], o2) and torch.equal(attn[:, lo:lo + n], a2) and torch.equal(lse[:, lo:lo + n], l2)):
            bad += 1; print("composition", prec, sizes, k, b)
        if prec == "bf16":
            o1, a1, l1 = ops.sparse_attn_fwd_mfma(qb[:, :d], qb[:, d:], kp[b * k:(b + 1) * k], n, h, need_attn=True, need_lse=True)
        else:
            o1, a1, l1 = ops.sparse_attn_fwd_x3(qb[:, :d], qb[:, d:], kp[b * k:(b + 1) * k], h, need_attn=True, need_lse=True)
        e = (out[b * k:(b + 1) * k] - o1).abs().max().item() / max(1e-6, o1.abs().max().item())
        if e > 3e-6 or not torch.equal(attn[:, lo:lo + n], a1): bad += 1; print("vs per-bag", prec, sizes, k, b, e)
print("varlen fuzz done", bad)
print("TOTAL BAD", bad)
