"""Parity of every HIP kernel (through the C ABI / ctypes) against the CPU oracle.  Needs a real MI355X."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import snuffy_oracle as orc
from tests.helpers import golden_files, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ops():
    from snuffy_amd import ops as o
    return o


# ---------------------------------------------------------------- K2 top-k: bit-exact
def test_topk_golden_tiefree_and_ties():
    z = np.load(golden_files("f2_")[0])
    for n in (5000, 32768):
        c = torch.from_numpy(z[f"tiefree_c_{n}"]).to(DEV)
        for k in (1, 10, 200, 512, 1024):
            idx = ops().topk(c, k).cpu().numpy()
            assert np.array_equal(idx, z[f"tiefree_order_{n}"][:k]), (n, k)
    c = torch.from_numpy(z["ties_c"]).to(DEV)
    assert np.array_equal(ops().topk(c, 1024).cpu().numpy(), z["ties_stable_order"])
    sp = torch.from_numpy(z["special_c"]).to(DEV)
    assert np.array_equal(ops().topk(sp, sp.numel()).cpu().numpy(), z["special_stable_order"])


@pytest.mark.parametrize("n,k", [(1, 1), (2, 2), (63, 10), (4096, 200), (4097, 200), (8192, 2048), (100000, 512),
                                 (100000, 1), (40, 40), (300000, 900)])
def test_topk_random_vs_oracle(n, k):
    g = torch.Generator().manual_seed(n * 7 + k)
    c = torch.randn(n, generator=g)
    m = min(c[::5].numel(), c[1::5].numel())
    c[::5][:m] = c[1::5][:m]  # inject exact ties
    want = orc.topk_desc_stable(c, k).numpy()
    got = ops().topk(c.to(DEV), k).cpu().numpy()
    assert np.array_equal(got, want)


def test_topk_strided_and_gather_and_nan():
    g = torch.Generator().manual_seed(3)
    c2 = torch.randn(5000, 3, generator=g)
    x = torch.randn(5000, 96, generator=g)
    cd = c2.to(DEV)
    idx, xs = ops().topk(cd[:, 1], 77, x=x.to(DEV))
    want = orc.topk_desc_stable(c2[:, 1], 77)
    assert np.array_equal(idx.cpu().numpy(), want.numpy())
    assert torch.equal(xs.cpu(), x[want])
    c = torch.randn(1000, generator=g)
    c[17] = float("nan")
    c[500] = float("inf")
    got = ops().topk(c.to(DEV), 3).cpu().tolist()
    assert got[:2] == [17, 500]          # NaN first (torch.sort semantics), then +inf


# ---------------------------------------------------------------- K1 + K2 fused selector: bit-exact, every path
def _score_cases(n, seed):
    g = torch.Generator().manual_seed(seed)
    c = torch.randn(n, generator=g)
    yield "randn", c
    t = c.clone()
    m = min(t[::5].numel(), t[1::5].numel())
    t[::5][:m] = t[1::5][:m]
    yield "pair ties", t
    yield "few values", torch.randint(0, 7, (n,), generator=g).float() - 3.0      # crowded threshold bin -> in-launch fallback
    yield "all equal", torch.full((n,), 0.25)
    if n >= 8:
        sp = c.clone()
        sp[n // 3] = float("nan")
        sp[n // 2] = float("inf")
        sp[n // 2 + 1] = float("-inf")
        sp[5] = -0.0
        sp[6] = 0.0
        yield "specials", sp
    yield "tiny spread", 1.0 + 1e-6 * torch.randn(n, generator=g)                 # every key in ONE first-digit bin


@pytest.mark.parametrize("n", [1, 5, 4096, 4097, 12288, 32768, 100000, 300001])
def test_topk_multi_workgroup_form_vs_oracle(n):
    """snf_topk_hist_select_f32 (histogram launch + multi-workgroup select) against the stable descending order."""
    for name, c in _score_cases(n, 11 * n + 1):
        cd = c.to(DEV)
        for k in sorted({1, min(n, 10), min(n, 200), min(n, 512), min(n, 2048)}):
            want = orc.topk_desc_stable(c, k).numpy()
            got = ops().topk_hist_select(cd, k).cpu().numpy()
            assert np.array_equal(got, want), (name, n, k)
    if n >= 8:                                        # strided scores (one class column of a multi-class critic)
        c2 = torch.randn(n, 3, generator=torch.Generator().manual_seed(n))
        k = min(n, 77)
        assert np.array_equal(ops().topk_hist_select(c2.to(DEV)[:, 1], k).cpu().numpy(), orc.topk_desc_stable(c2[:, 1], k).numpy())


@pytest.mark.parametrize("n,d,k", [(16385, 384, 200), (32768, 768, 200), (100000, 768, 512), (50000, 96, 2048)])
def test_fused_selector_critic_plus_select(n, d, k):
    """critic_select (scores + first-digit histogram in one pass over the bag, optionally the normalised bf16 copy) followed by
    topk: the scores equal the plain critic's bit for bit and the selection equals the one-workgroup kernel's and the oracle's."""
    o = ops()
    g = torch.Generator().manual_seed(n + d)
    x = torch.randn(n, d, generator=g).to(DEV)
    w = (torch.randn(1, d, generator=g) / math.sqrt(d)).to(DEV)
    b = torch.randn(1, generator=g).to(DEV)
    s_ref = o.critic(x, w, b)
    for eps in (None, 1e-5):
        s, xhat = o.critic_select(x, w, b, eps)
        assert torch.equal(s, s_ref)
        if eps is not None:
            assert torch.equal(xhat, o.critic_ln(x, w, b, eps)[1])
        sel = o.selector(x.device)
        assert sel.pending is not None
        idx = o.topk(s.view(-1), k)
        assert sel.pending is None                   # consumed by the fused select
        assert np.array_equal(idx.cpu().numpy(), orc.topk_desc_stable(s_ref.view(-1).cpu(), k).numpy())
        assert int(sel.state.view(torch.int32)[: 4 * 2048 + 3].abs().sum()) == 0      # counted part left zeroed
    # a histogram that was never consumed (aborted forward) is noticed: first on the host side ...
    s, _ = o.critic_select(x, w, b)
    s2, _ = o.critic_select(x, w, b)
    assert np.array_equal(o.topk(s2.view(-1), k).cpu().numpy(), orc.topk_desc_stable(s_ref.view(-1).cpu(), k).numpy())
    # ... and, if the host lost track, inside the launch (total != n -> exact one-workgroup selection, state cleaned)
    sel = o.selector(x.device)
    before = int(sel.state.view(torch.int32)[4 * 2048 + 3])
    s, _ = o.critic_select(x, w, b)
    sel.pending = None
    s3, _ = o.critic_select(x, w, b)                # adds a second histogram on top of the first
    assert np.array_equal(o.topk(s3.view(-1), k).cpu().numpy(), orc.topk_desc_stable(s_ref.view(-1).cpu(), k).numpy())
    assert int(sel.state.view(torch.int32)[4 * 2048 + 3]) == before + 1
    s4, _ = o.critic_select(x, w, b)                # and the next bag is back on the fast path
    assert np.array_equal(o.topk(s4.view(-1), k).cpu().numpy(), orc.topk_desc_stable(s_ref.view(-1).cpu(), k).numpy())
    assert int(sel.state.view(torch.int32)[4 * 2048 + 3]) == before + 1


def test_fused_selector_state_is_per_stream():
    """SURVEY 8b (re-entrant): two streams of one device run critic + select pipelines at the same time, each on a selector state of
    its own -- both get the single-stream answer, every time (one shared histogram would mix their counts)."""
    o = ops()
    n, d, k = 40000, 384, 300
    g = torch.Generator().manual_seed(4)
    xs = [torch.randn(n, d, generator=g).to(DEV) for _ in range(2)]
    w = (torch.randn(1, d, generator=g) / math.sqrt(d)).to(DEV)
    b = torch.zeros(1, device=DEV)
    want = [orc.topk_desc_stable(o.critic(x, w, b).view(-1).cpu(), k).numpy() for x in xs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    states = []
    for rep in range(6):
        got = [None, None]
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                for _ in range(3):                                   # several bags in flight per stream
                    s, _ = o.critic_select(xs[i], w, b)
                    got[i] = o.topk(s.view(-1), k)
                if rep == 0:
                    states.append(o.selector(xs[i].device))
        torch.cuda.synchronize()
        for i in (0, 1):
            assert np.array_equal(got[i].cpu().numpy(), want[i]), (rep, i)
    assert states[0] is not states[1] and states[0].state.data_ptr() != states[1].state.data_ptr()


# ---------------------------------------------------------------- K1 critic
@pytest.mark.parametrize("n,d,c", [(1, 64, 1), (1000, 166, 1), (4099, 384, 2), (513, 768, 1), (100, 2048, 3)])
def test_critic(n, d, c):
    g = torch.Generator().manual_seed(n + d)
    x, w, b = torch.randn(n, d, generator=g), torch.randn(c, d, generator=g) / math.sqrt(d), torch.randn(c, generator=g)
    s, mv, mi = ops().critic(x.to(DEV), w.to(DEV), b.to(DEV), want_max=True)
    ref = F.linear(x.double(), w.double(), b.double())
    assert (s.cpu().double() - ref).abs().max() < 2e-5
    s_cpu = s.cpu()
    mvr, mir = s_cpu.max(dim=0)
    assert torch.equal(mv.cpu(), mvr)
    assert torch.equal(mi.cpu(), mir)


# ---------------------------------------------------------------- K5 LayerNorm (+ fused patch rows)
@pytest.mark.parametrize("n,d", [(1, 64), (333, 166), (2050, 384), (1000, 768), (64, 2048)])
def test_layernorm_rows(n, d):
    g = torch.Generator().manual_seed(d)
    x = torch.randn(n, d, generator=g) * 3 + 1
    gam, bet = torch.randn(d, generator=g), torch.randn(d, generator=g)
    ref = F.layer_norm(x.double(), (d,), gam.double(), bet.double(), 1e-5)
    out, mean, rstd = ops().layernorm_rows(x.to(DEV), gam.to(DEV), bet.to(DEV), 1e-5, want_stats=True)
    assert (out.cpu().double() - ref).abs().max() < 2e-5
    assert (mean.cpu() - x.mean(1)).abs().max() < 1e-5
    assert rel_err(rstd.cpu(), 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)) < 1e-5
    # plain normalisation to bf16
    ob = ops().layernorm_rows(x.to(DEV), None, None, 1e-5, out_dtype=torch.bfloat16)
    refn = F.layer_norm(x.double(), (d,), None, None, 1e-5)
    assert (ob.cpu().double() - refn).abs().max() < 2e-2
    # patched rows
    k = min(n, 7)
    sel = torch.randperm(n, generator=g)[:k]
    rows = torch.randn(k, d, generator=g)
    y = x.clone()
    y[sel] = rows
    slot = ops().slot_map(sel.to(DEV), n)
    m = torch.full((n,), -1, dtype=torch.int32)
    m[sel] = torch.arange(k, dtype=torch.int32)
    assert torch.equal(slot.cpu(), m)
    outp = ops().layernorm_rows(x.to(DEV), gam.to(DEV), bet.to(DEV), 1e-5, slot=slot, patch_rows=rows.to(DEV))
    refp = F.layer_norm(y.double(), (d,), gam.double(), bet.double(), 1e-5)
    assert (outp.cpu().double() - refp).abs().max() < 2e-5
    # re-normalise K rows into an existing bf16 buffer
    buf = ob.clone()
    ops().layernorm_rows(rows.to(DEV), None, None, 1e-5, out=buf, out_row_idx=sel.to(DEV))
    refb = F.layer_norm(y.double(), (d,), None, None, 1e-5)
    assert (buf.cpu().double() - refb).abs().max() < 2e-2


def test_gather_scatter():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1000, 166, generator=g)
    sel = torch.randperm(1000, generator=g)[:200]
    rows = torch.randn(200, 166, generator=g)
    xd, sd, rd = x.to(DEV), sel.to(DEV), rows.to(DEV)
    assert torch.equal(ops().gather_rows(xd, sd).cpu(), x[sel])
    y = x.clone()
    y[sel] = rows
    assert torch.equal(ops().scatter_rows(xd, sd, rd).cpu(), y)
    assert torch.equal(xd.cpu(), x)                      # input untouched
    z = xd.clone()
    ops().scatter_add_rows_(z, sd, rd)
    w = x.clone()
    w[sel] += rows
    assert torch.equal(z.cpu(), w)


@pytest.mark.parametrize("act", ["relu", "gelu", "leakyrelu", "selu", "none"])
def test_bias_act(act):
    g = torch.Generator().manual_seed(9)
    for n, f in [(100, 256), (37, 166), (513, 3072)]:
        h = torch.randn(n, f, generator=g) * 2
        b = torch.randn(f, generator=g)
        ref = (h + b).double()
        ref = ref if act == "none" else orc.ACTIVATIONS[act](ref)
        hd = h.to(DEV)
        ops().bias_act_(hd, b.to(DEV), act)
        assert (hd.cpu().double() - ref).abs().max() < 1e-5
        if f % 8 == 0:
            hb = h.to(DEV).to(torch.bfloat16)
            refb = (hb.cpu().float() + b).double()
            refb = refb if act == "none" else orc.ACTIVATIONS[act](refb)
            ops().bias_act_(hb, b.to(DEV), act)
            assert rel_err(hb.cpu().float(), refb.float()) < 1e-2


@pytest.mark.parametrize("n,d,c", [(1, 64, 1), (1000, 166, 1), (5000, 384, 2), (3001, 768, 1)])
def test_ln_mean_head(n, d, c):
    g = torch.Generator().manual_seed(n)
    z = torch.randn(n, d, generator=g) * 2 + 0.5
    gam, bet = torch.randn(d, generator=g), torch.randn(d, generator=g)
    w, b = torch.randn(c, d, generator=g) / math.sqrt(d), torch.randn(c, generator=g)
    pooled_ref = F.layer_norm(z.double(), (d,), gam.double(), bet.double(), 1e-5).mean(0)
    ref = F.linear(pooled_ref, w.double(), b.double())
    logits, pooled, _ = ops().ln_mean_head(z.to(DEV), gam.to(DEV), bet.to(DEV), 1e-5, w.to(DEV), b.to(DEV))
    assert (pooled.cpu().double() - pooled_ref).abs().max() < 1e-5
    assert (logits.cpu().double() - ref).abs().max() < 1e-5
    # fused residual assembly
    k = min(n, 9)
    sel = torch.randperm(n, generator=g)[:k]
    delta = torch.randn(k, d, generator=g)
    addb = (torch.randn(n, d, generator=g)).to(torch.bfloat16)
    bias = torch.randn(d, generator=g)
    zz = z + addb.float() + bias
    zz[sel] += delta
    ref2 = F.linear(F.layer_norm(zz.double(), (d,), gam.double(), bet.double(), 1e-5).mean(0), w.double(), b.double())
    slot = ops().slot_map(sel.to(DEV), n)
    l2, _, zo = ops().ln_mean_head(z.to(DEV), gam.to(DEV), bet.to(DEV), 1e-5, w.to(DEV), b.to(DEV), add_bf16=addb.to(DEV),
                                   add_bias=bias.to(DEV), slot=slot, delta_rows=delta.to(DEV), want_z=True)
    assert (l2.cpu().double() - ref2).abs().max() < 1e-5
    assert (zo.cpu() - zz).abs().max() < 1e-5


# ---------------------------------------------------------------- K7 sparse attention
def attn_ref(q, kp, v, h):
    o, p = orc.sparse_attention(q.double(), kp.double(), v.double(), h)
    return o, p


@pytest.mark.parametrize("n,k,h,dk", [(40, 40, 2, 83), (1000, 200, 6, 16), (3000, 900, 4, 24), (777, 64, 1, 128),
                                      (257, 5, 3, 7), (1, 1, 2, 32), (5000, 512, 6, 128), (17, 17, 2, 256),
                                      # the README recipes' head widths (h = 4) and the limits of the matrix-core forms
                                      (8192, 500, 4, 192), (4001, 900, 4, 96), (333, 1024, 2, 192), (600, 1025, 2, 64), (2000, 130, 3, 8)])
def test_sparse_attn_exact(n, k, h, dk):
    g = torch.Generator().manual_seed(n + k)
    d = h * dk
    q, kp, v = torch.randn(n, d, generator=g), torch.randn(k, d, generator=g), torch.randn(n, d, generator=g)
    o_ref, p_ref = attn_ref(q, kp, v, h)
    o, attn, lse = ops().sparse_attn_fwd(q.to(DEV), kp.to(DEV), v.to(DEV), h, need_attn=True, need_lse=True)
    assert (attn.cpu().double() - p_ref).abs().max() < 1e-6
    assert rel_err(o.cpu(), o_ref) < 1e-5
    s = torch.matmul(q.double().view(n, h, dk).transpose(0, 1), kp.double().view(k, h, dk).transpose(0, 1).transpose(1, 2))
    lse_ref = torch.logsumexp(s / math.sqrt(dk), dim=-1)
    assert (lse.cpu().double() - lse_ref).abs().max() < 1e-4
    o2, _, _ = ops().sparse_attn_fwd(q.to(DEV), kp.to(DEV), v.to(DEV), h)     # P in workspace
    assert torch.equal(o2, o)
    # round 5: the same products on v_mfma_f32_32x32x2_f32 (head widths % 8 == 0, <= 1024 keys; P^T V for every shape) against the
    # vector-ALU kernels -- exact fp32 both ways, equal up to the order of the fmaf chains
    from snuffy_amd import _ffi
    _ffi.load().snf_debug_exact_attn_mfma(0)
    try:
        o3, a3, l3 = ops().sparse_attn_fwd(q.to(DEV), kp.to(DEV), v.to(DEV), h, need_attn=True, need_lse=True)
    finally:
        _ffi.load().snf_debug_exact_attn_mfma(1)
    assert (a3 - attn).abs().max() < 5e-7 and rel_err(o3.cpu(), o.cpu()) < 3e-6 and (l3 - lse).abs().max() < 2e-5
    assert (attn.sum(-1) - 1).abs().max() < 1e-5


@pytest.mark.parametrize("n,k,h,dk", [(8192, 500, 4, 192), (4001, 900, 4, 96), (333, 1024, 2, 192), (2000, 130, 3, 16), (50, 7, 2, 48),
                                      (3000, 200, 2, 256), (1, 1, 1, 32)])
def test_sparse_attn_x3u_fp32_class(n, k, h, dk):
    """snf_sparse_attn_fwd_x3u_f32 (round 5): scores + softmax in the fp32-class arithmetic of the pipelined kernels -- split-bf16 x 3,
    fp32 accumulate -- and P^T V exact, for head widths those kernels do not take (dk = 192: reference README.md:661-669).  Same
    bounds against fp64 as snf_sparse_attn_fwd_x3; deterministic; the same answer with and without the A output."""
    g = torch.Generator().manual_seed(n * 3 + k + dk)
    d = h * dk
    q, kp, v = torch.randn(n, d, generator=g), torch.randn(k, d, generator=g), torch.randn(n, d, generator=g)
    o_ref, p_ref = attn_ref(q, kp, v, h)
    assert ops().x3u_attn_supported(k, dk)
    o, attn, lse = ops().sparse_attn_fwd_x3u(q.to(DEV), kp.to(DEV), v.to(DEV), h, need_attn=True, need_lse=True)
    assert (attn.cpu().double() - p_ref).abs().max() < 1e-5
    assert rel_err(o.cpu(), o_ref) < 2e-5
    s_ref = (q.double().view(n, h, dk).transpose(0, 1) @ kp.double().view(k, h, dk).transpose(0, 1).transpose(1, 2)) / dk ** 0.5
    assert (lse.cpu().double() - torch.logsumexp(s_ref, dim=-1)).abs().max() < 3e-5
    assert (attn.sum(-1) - 1).abs().max() < 1e-5
    o2, a2, _ = ops().sparse_attn_fwd_x3u(q.to(DEV), kp.to(DEV), v.to(DEV), h)
    assert a2 is None and torch.equal(o2, o)
    if d % 4 == 0:                       # the halves of a fused [Q | V] projection output, used in place (row pitch 2 d)
        qv = torch.cat([q, v], dim=1).to(DEV)
        o3, _, _ = ops().sparse_attn_fwd_x3u(qv[:, :d], kp.to(DEV), qv[:, d:], h)
        assert torch.equal(o3, o)
    from snuffy_amd import SnuffyHipError
    with pytest.raises(SnuffyHipError):
        ops().sparse_attn_fwd_x3u(torch.zeros(8, 2 * 83, device=DEV), torch.zeros(4, 2 * 83, device=DEV), torch.zeros(8, 2 * 83, device=DEV), 2)


@pytest.mark.parametrize("n,k,h,dk,p_drop", [(4000, 200, 6, 128, 0.1), (1000, 224, 2, 128, 0.5), (3001, 256, 3, 64, 0.1), (700, 37, 4, 64, 0.25),
                                             (65, 5, 1, 128, 0.1)])
def test_sparse_attn_x3_in_kernel_dropout(n, k, h, dk, p_drop):
    """snf_sparse_attn_fwd_x3_dropout (round 5, the training forward of reference snuffy.py:166-167): O = (P o M)^T V with M regenerated
    in registers from (seed, offset) -- against the mask TENSOR of snf_dropout_mask_f32 for the same state applied to the kernel's own
    undropped P in fp64; the P written out is the undropped one, bit for bit the plain launch's; more keys than one launch: refused."""
    from snuffy_amd import SnuffyHipError
    g = torch.Generator().manual_seed(n + k + dk)
    d = h * dk
    q, kp, v = (torch.randn(n, d, generator=g).to(DEV), torch.randn(k, d, generator=g).to(DEV), torch.randn(n, d, generator=g).to(DEV))
    seed, offset = 1234567 + n, (1 << 40) + 17 * k
    assert ops().x3_attn_dropout_supported(k, dk)
    o_plain, p_plain, _ = ops().sparse_attn_fwd_x3(q, v, kp, h, need_attn=True)
    o, p, _ = ops().sparse_attn_fwd_x3(q, v, kp, h, need_attn=True, dropout=(p_drop, seed, offset))
    assert torch.equal(p, p_plain)
    mask = ops().dropout_mask(h, n, k, p_drop, seed, offset, DEV)
    kept = (mask > 0).float().mean().item()
    assert abs(kept - (1 - p_drop)) < 0.02 + 2.0 / (h * n * k) ** 0.5
    ref = torch.bmm((p.double() * mask.double()).transpose(1, 2), v.double().view(n, h, dk).transpose(0, 1)).transpose(0, 1).reshape(k, d)
    assert rel_err(o.cpu(), ref.cpu()) < 2e-5
    assert rel_err(o.cpu(), o_plain.cpu()) > 1e-3                      # and it is not the undropped product
    o2, _, _ = ops().sparse_attn_fwd_x3(q, v, kp, h, need_attn=True, dropout=(p_drop, seed, offset))
    assert torch.equal(o2, o)
    o0, _, _ = ops().sparse_attn_fwd_x3(q, v, kp, h, need_attn=True, dropout=(0.0, seed, offset))
    assert torch.equal(o0, o_plain)
    if dk == 128:
        kk = 300
        with pytest.raises(SnuffyHipError):
            ops().sparse_attn_fwd_x3(q, v, torch.randn(kk, d, generator=g).to(DEV), h, dropout=(p_drop, seed, offset))


@pytest.mark.parametrize("n,k,h,dk", [(3000, 200, 6, 128), (1001, 77, 3, 64), (500, 300, 2, 96)])
def test_sparse_attn_bwd_reads_the_halves_of_a_fused_projection_in_place(n, k, h, dk):
    """snf_sparse_attn_bwd_ld_f32 (round 5): q / v as row-strided column halves of one [n, 2 d] tensor -- bit for bit the gradients of
    the contiguous call (the training chain hands the Q | V projection's output over without two 100 MB copies per step)."""
    g = torch.Generator().manual_seed(n + k)
    d = h * dk
    qv = torch.randn(n, 2 * d, generator=g).to(DEV)
    kp, dout = torch.randn(k, d, generator=g).to(DEV), torch.randn(k, d, generator=g).to(DEV)
    q, v = qv[:, :d], qv[:, d:]
    _, p, _ = ops().sparse_attn_fwd(q.contiguous(), kp, v.contiguous(), h, need_attn=True)
    mask = ops().dropout_mask(h, n, k, 0.1, 7, 99, DEV)
    for m in (None, mask):
        a = ops().sparse_attn_bwd(q, kp, v, p, dout, h, mask=m)
        b = ops().sparse_attn_bwd(q.contiguous(), kp, v.contiguous(), p, dout, h, mask=m)
        for x, y in zip(a, b):
            assert x.is_contiguous() and torch.equal(x, y)


@pytest.mark.parametrize("n,k,h,dk", [(5000, 200, 6, 128), (30000, 500, 2, 192), (1001, 36, 3, 64), (700, 300, 2, 96), (40, 8, 1, 32)])
def test_exact_attention_lds_staged_kernels_are_the_direct_ones_bit_for_bit(n, k, h, dk):
    """Round 5: P^T V (forward, dKp) and dQ / dV of the exact attention stage their operands through LDS; snf_debug_exact_attn_mfma(2) selects
    the kernels that read every operand straight from L2 -- the same fmaf chains in the same order, so every output is bit-identical."""
    from snuffy_amd import _ffi
    g = torch.Generator().manual_seed(n + k + dk)
    d = h * dk
    q, kp, v = (torch.randn(n, d, generator=g).to(DEV), torch.randn(k, d, generator=g).to(DEV), torch.randn(n, d, generator=g).to(DEV))
    dout = torch.randn(k, d, generator=g).to(DEV)
    mask = ops().dropout_mask(h, n, k, 0.1, 3, 5, DEV)
    res = {}
    try:
        for mode in (1, 2):
            _ffi.load().snf_debug_exact_attn_mfma(mode)
            o, p, _ = ops().sparse_attn_fwd(q, kp, v, h, need_attn=True)
            res[mode] = (o, p) + tuple(ops().sparse_attn_bwd(q, kp, v, p, dout, h, mask=mask)) + tuple(ops().sparse_attn_bwd(q, kp, v, p, dout, h))
    finally:
        _ffi.load().snf_debug_exact_attn_mfma(1)
    for x, y in zip(res[1], res[2]):
        assert torch.equal(x, y)


def bf16r(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("n,k,h,dk", [(1, 1, 1, 64), (100, 31, 2, 64), (128, 32, 6, 64), (129, 33, 3, 128),
                                      (1000, 64, 6, 128), (4099, 100, 6, 64), (2500, 200, 6, 128), (3000, 224, 2, 128),
                                      (777, 256, 6, 64), (1500, 224, 1, 128), (1500, 250, 1, 64), (8192, 200, 6, 64), (640, 129, 4, 128),
                                      (1500, 250, 1, 128), (3000, 512, 6, 128), (2000, 600, 2, 64), (700, 1792, 1, 128),
                                      (300, 7, 12, 64), (897, 65, 1, 128), (127, 1, 4, 128), (2049, 193, 5, 64)])
def test_sparse_attn_mfma(n, k, h, dk, dt):
    g = torch.Generator().manual_seed(n * 3 + k)
    d = h * dk
    q, kp, v = torch.randn(n, d, generator=g), torch.randn(k, d, generator=g), torch.randn(n, d, generator=g)
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    # q and v as the two halves of one fused projection buffer [n, 2d] (row pitch 2d), as the model passes them
    qv = torch.cat([q, v], dim=1).to(DEV).to(tdt)
    qd, vd = qv[:, :d], qv[:, d:]
    o, attn, lse = ops().sparse_attn_fwd_mfma(qd, vd, kp.to(DEV), n, h, need_attn=True, need_lse=True)
    # (a) against the exact oracle: bf16-class tolerance (north star: 1e-2)
    o_ref, p_ref = attn_ref(q, kp, v, h)
    assert (attn.cpu().double() - p_ref).abs().max() < 1e-2
    assert rel_err(o.cpu(), o_ref) < 1e-2
    # (b) against the oracle fed with the same bf16-rounded operands: tight (catches any layout slip)
    o_r, p_r = attn_ref(bf16r(q), bf16r(kp), bf16r(v), h)
    assert (attn.cpu().double() - p_r).abs().max() < 2e-5 + 2e-3 * float(p_r.max())
    assert rel_err(o.cpu(), o_r) < 3e-3
    # rows of P sum to one; sum over keys of O equals the column sums of V (size-independent checksum)
    assert (attn.sum(-1) - 1).abs().max() < 1e-4
    assert rel_err(o.cpu().view(k, h, dk).sum(0), bf16r(v).view(n, h, dk).sum(0)) < 5e-3
    # run-to-run determinism
    o2, attn2, _ = ops().sparse_attn_fwd_mfma(q.to(DEV).to(tdt), v.to(DEV).to(tdt), kp.to(DEV), n, h, need_attn=True)  # contiguous
    assert torch.equal(o2, o) and torch.equal(attn2, attn)
    # without materialising A the output is the same
    o3, a3, _ = ops().sparse_attn_fwd_mfma(qd, vd, kp.to(DEV), n, h)
    assert a3 is None and (torch.equal(o3, o) or rel_err(o3.cpu(), o.cpu()) < 2e-3)
    # a bf16 Kp (what the model's bf16 key projection hands over) is read as it is: same bits as the library's own
    # round-to-nearest-even of the f32 Kp
    o4, a4, _ = ops().sparse_attn_fwd_mfma(qd, vd, kp.to(DEV).to(torch.bfloat16), n, h, need_attn=True)
    assert torch.equal(o4, o) and torch.equal(a4, attn)


@pytest.mark.parametrize("n,k,h,dk,dt", [(40000, 200, 6, 128, "bf16"), (40000, 300, 6, 128, "bf16"), (70000, 256, 6, 64, "f32"),
                                         (33000, 224, 3, 128, "f32")])
def test_sparse_attn_mfma_workgroups_straddle_heads(n, k, h, dk, dt):
    """More row tiles than compute units: every workgroup walks several (head, tile) items and some cross a head boundary
    (Kp refetched by all 8 waves, partial tiles flushed mid-loop) -- with the attention / lse outputs and, for k = 300, the
    key-chunked variant.  Checked against the oracle fed with the same bf16-rounded operands and through the
    size-independent checksums."""
    g = torch.Generator().manual_seed(n + k)
    d = h * dk
    q, kp, v = torch.randn(n, d, generator=g), torch.randn(k, d, generator=g), torch.randn(n, d, generator=g)
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    qv = torch.cat([q, v], dim=1).to(DEV).to(tdt)
    qd, vd = qv[:, :d], qv[:, d:]
    o, attn, lse = ops().sparse_attn_fwd_mfma(qd, vd, kp.to(DEV), n, h, need_attn=True, need_lse=True)
    o_r, p_r = attn_ref(bf16r(q), bf16r(kp), bf16r(v), h)
    assert (attn.cpu().double() - p_r).abs().max() < 2e-5 + 2e-3 * float(p_r.max())
    assert rel_err(o.cpu(), o_r) < 3e-3
    assert (attn.sum(-1) - 1).abs().max() < 1e-4
    assert rel_err(o.cpu().view(k, h, dk).sum(0), bf16r(v).view(n, h, dk).sum(0)) < 5e-3
    o2, a2, _ = ops().sparse_attn_fwd_mfma(qd, vd, kp.to(DEV).to(torch.bfloat16), n, h)      # no attention output, bf16 Kp
    assert a2 is None and (torch.equal(o2, o) or rel_err(o2.cpu(), o.cpu()) < 2e-3)


@pytest.mark.parametrize("n,k,h", [(32768, 200, 6), (5000, 224, 3), (64, 193, 1), (130, 100, 2), (40000, 128, 6), (1, 97, 1), (777, 210, 4)])
def test_sparse_attn_inference_call_pattern(n, k, h):
    """The inference call pattern of the model (bf16 q / v / kp, no A / lse, dk = 128) over tile counts from one to several per
    workgroup: against the oracle on the same bf16-rounded operands, against the same call with lse requested (the training
    variant of the kernel), the column-sum checksum, and an f32 Kp rounded by the library."""
    dk = 128
    g = torch.Generator().manual_seed(n + 3 * k)
    d = h * dk
    q, kp, v = torch.randn(n, d, generator=g), torch.randn(k, d, generator=g), torch.randn(n, d, generator=g)
    qv = torch.cat([q, v], dim=1).to(DEV).to(torch.bfloat16)
    qd, vd = qv[:, :d], qv[:, d:]
    kpd = kp.to(DEV).to(torch.bfloat16)
    o, a_none, _ = ops().sparse_attn_fwd_mfma(qd, vd, kpd, n, h)
    assert a_none is None
    o_r, _ = attn_ref(bf16r(q), bf16r(kp), bf16r(v), h)
    assert rel_err(o.cpu(), o_r) < 3e-3
    o_two_role, _, _ = ops().sparse_attn_fwd_mfma(qd, vd, kpd, n, h, need_lse=True)
    assert rel_err(o.cpu(), o_two_role.cpu()) < 2e-3
    assert rel_err(o.cpu().view(k, h, dk).sum(0), bf16r(v).view(n, h, dk).sum(0)) < 5e-3
    o2, _, _ = ops().sparse_attn_fwd_mfma(qd, vd, kp.to(DEV), n, h)                # f32 Kp: rounded by the library first
    assert torch.equal(o2, o)


def test_sparse_attn_mfma_online_max_spike():
    """A key row that dominates one query (score >> others) must not overflow or lose the other rows."""
    n, k, h, dk = 512, 200, 6, 128
    g = torch.Generator().manual_seed(0)
    d = h * dk
    q, kp, v = torch.randn(n, d, generator=g), torch.randn(k, d, generator=g), torch.randn(n, d, generator=g)
    q[7] *= 40.0
    kp[3] *= 25.0
    o, attn, _ = ops().sparse_attn_fwd_mfma(q.to(DEV), v.to(DEV), kp.to(DEV), n, h, need_attn=True)
    assert torch.isfinite(o).all() and torch.isfinite(attn).all()
    o_r, p_r = attn_ref(bf16r(q), bf16r(kp), bf16r(v), h)
    assert (attn.cpu().double() - p_r).abs().max() < 5e-3
    assert rel_err(o.cpu(), o_r) < 3e-3


@pytest.mark.parametrize("n,k,h,dk", [(1000, 200, 6, 128), (4097, 224, 3, 128), (33, 7, 2, 128), (1, 1, 1, 64), (700, 256, 6, 64),
                                      (2500, 150, 6, 64), (5000, 100, 2, 128), (31, 40, 4, 64), (300, 33, 1, 128),
                                      # more keys than one LDS image: key chunks with exact cross-chunk statistics
                                      (3000, 512, 6, 128), (700, 225, 2, 128), (1500, 601, 3, 64), (257, 1790, 1, 128), (100, 2048, 2, 64)])
def test_sparse_attn_x3_fp32_class(n, k, h, dk):
    """snf_sparse_attn_fwd_x3 (split-bf16 x 3 on the matrix cores) against the fp64 oracle on the UNROUNDED operands: the
    reference's own arithmetic class (fp32), not the bf16 one.  Measured: P <= 3.5e-6, O <= 1e-5 of its scale (the exact
    vector-ALU kernel: 4e-7 / 9e-7; the bf16 kernel: 5e-3) -- two bf16 halves carry 16 mantissa bits, the dropped lo x lo
    term is 2^-17 relative."""
    g = torch.Generator().manual_seed(n * 5 + k)
    d = h * dk
    q, kp, v = torch.randn(n, d, generator=g), torch.randn(k, d, generator=g), torch.randn(n, d, generator=g)
    qv = torch.cat([q, v], dim=1).to(DEV)                       # column halves of one buffer, as the fused projection leaves them
    o, attn, lse = ops().sparse_attn_fwd_x3(qv[:, :d], qv[:, d:], kp.to(DEV), h, need_attn=True, need_lse=True)
    o_ref, p_ref = attn_ref(q, kp, v, h)
    assert (attn.cpu().double() - p_ref).abs().max() < 6e-6
    assert rel_err(o.cpu(), o_ref) < (2e-5 if k <= 256 else 4e-5)      # few rows under many keys: O is a short sum of small P
    s_ref = (q.double().view(n, h, dk).transpose(0, 1) @ kp.double().view(k, h, dk).transpose(0, 1).transpose(1, 2)) / dk ** 0.5
    assert (lse.cpu().double() - torch.logsumexp(s_ref, dim=-1)).abs().max() < 3e-5
    assert (attn.sum(-1) - 1).abs().max() < 1e-5
    # same answer as the exact vector-ALU kernel to fp32 rounding, bit-identical run to run, and without the A / lse outputs
    o_ex, a_ex, _ = ops().sparse_attn_fwd(q.to(DEV), kp.to(DEV), v.to(DEV), h, need_attn=True)
    assert (attn - a_ex).abs().max() < 6e-6 and rel_err(o.cpu(), o_ex.cpu()) < (2e-5 if k <= 256 else 4e-5)
    o2, a2, _ = ops().sparse_attn_fwd_x3(q.to(DEV), v.to(DEV), kp.to(DEV), h)
    assert a2 is None and torch.equal(o2, o)


def test_sparse_attn_x3_config_b_walks_heads_and_spike():
    """Config-B size (every workgroup walks 24 tiles, some across a head change) + a dominating key (no overflow)."""
    n, k, h, dk = 32768, 200, 6, 128
    g = torch.Generator().manual_seed(11)
    d = h * dk
    q, kp, v = torch.randn(n, d, generator=g), torch.randn(k, d, generator=g), torch.randn(n, d, generator=g)
    q[7] *= 40.0
    kp[3] *= 25.0
    o, attn, _ = ops().sparse_attn_fwd_x3(q.to(DEV), v.to(DEV), kp.to(DEV), h, need_attn=True)
    assert torch.isfinite(o).all() and torch.isfinite(attn).all()
    rows = torch.arange(0, n, 61)
    o_ref, p_ref = attn_ref(q, kp, v, h)
    # the split carries 16 mantissa bits: the error of a score is ~2^-17 |q| |k| scale, so it grows with the spiked operands
    # (|s| ~ 1e2 .. 1e3 here, where the un-spiked shapes above sit at 3.5e-6); still inside the fp32 class bound of 1e-3
    assert (attn[:, rows.to(DEV), :].cpu().double() - p_ref[:, rows, :]).abs().max() < 1e-3
    assert rel_err(o.cpu(), o_ref) < 1e-3
    from snuffy_amd import SnuffyHipError
    with pytest.raises(SnuffyHipError):
        ops().sparse_attn_fwd_x3(q.to(DEV), v.to(DEV), torch.zeros(8 * 224 + 1, d, device=DEV), h)   # more than 8 key chunks


@pytest.fixture
def x3p_two_key_blocks_per_wave():
    """Switches the dk = 128 family of snf_sparse_attn_fwd_x3_hl to its round-5 form (one wave per SIMD, two key blocks per wave)
    for one test; the default form is restored afterwards."""
    from snuffy_amd import _ffi
    _ffi.load().snf_debug_x3p_kbw(2)
    yield
    _ffi.load().snf_debug_x3p_kbw(1)


X3P_SHAPES = [(1000, 200, 6, 128), (4097, 224, 3, 128), (33, 128, 2, 128), (1, 97, 1, 128), (5000, 100, 2, 128), (2000, 256, 2, 128),
              (777, 129, 2, 128), (300, 160, 1, 128), (6401, 200, 6, 128),
              # more than 256 keys: key chunks with exact cross-chunk statistics
              (3000, 512, 6, 128), (700, 257, 2, 128), (1500, 601, 3, 128), (257, 1790, 1, 128), (100, 2048, 2, 128),
              # last chunk far below the 97-key minimum of a launch of its own: 5 x 256 + 65, 7 x 256 + 1 -- masked in the merged launch
              (400, 1345, 1, 128), (300, 1793, 2, 128), (300, 1345, 2, 64),
              # dk = 64 (round 5): config A's head width
              (1000, 200, 6, 64), (4097, 256, 3, 64), (33, 128, 2, 64), (1, 97, 1, 64), (5000, 100, 12, 64), (777, 129, 2, 64),
              (300, 160, 1, 64), (8192, 200, 6, 64), (3000, 512, 6, 64), (700, 257, 2, 64), (257, 1790, 1, 64)]


@pytest.mark.parametrize("n,k,h,dk", [sh for sh in X3P_SHAPES if sh[3] == 128 and sh[1] > 128])
def test_sparse_attn_x3_hl_two_key_blocks_per_wave(n, k, h, dk, x3p_two_key_blocks_per_wave):
    """The round-5 form of the dk = 128 family (5 .. 8 key blocks per launch on 3 .. 4 waves of 512 registers, loader / light-wave /
    all-issue DMA configurations, key chunks): same bounds against fp64 as the default form, and equal to it up to the order of the
    fp32 row sums."""
    test_sparse_attn_x3_hl_fp32_class(n, k, h, dk)
    from snuffy_amd import _ffi
    g = torch.Generator().manual_seed(n * 7 + k)
    d = h * dk
    q, kp, v = torch.randn(n, d, generator=g), torch.randn(k, d, generator=g), torch.randn(n, d, generator=g)
    img = ops().split_hl_rows(torch.cat([q, v], dim=1).to(DEV))
    o2, a2, _ = ops().sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], kp.to(DEV), h, need_attn=True)
    _ffi.load().snf_debug_x3p_kbw(1)
    o1, a1, _ = ops().sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], kp.to(DEV), h, need_attn=True)
    assert (a1 - a2).abs().max() < 2e-6 and rel_err(o2.cpu(), o1.cpu()) < 4e-6


@pytest.mark.parametrize("n,k,h,dk", X3P_SHAPES)
def test_sparse_attn_x3_hl_fp32_class(n, k, h, dk):
    """snf_sparse_attn_fwd_x3_hl (the pipelined split-bf16 x 3 kernel on PRE-SPLIT hl operands) against the fp64 oracle on the
    unrounded operands: same arithmetic class as snf_sparse_attn_fwd_x3 (the hl image carries exactly the hi / lo halves that
    kernel derives from the fp32 tensor), same bounds.  Tile counts from one to several per workgroup, ragged last tiles, every
    key-block count the kernel is built for (4 .. 8), rows past the end, head changes inside a workgroup's range; dk = 128 and 64."""
    g = torch.Generator().manual_seed(n * 7 + k)
    d = h * dk
    q, kp, v = torch.randn(n, d, generator=g), torch.randn(k, d, generator=g), torch.randn(n, d, generator=g)
    img = ops().split_hl_rows(torch.cat([q, v], dim=1).to(DEV))            # [n, 4 d]: image of Q | image of V, as the projection leaves it
    qi, vi = img[:, :2 * d], img[:, 2 * d:]
    o, attn, lse = ops().sparse_attn_fwd_x3_hl(qi, vi, kp.to(DEV), h, need_attn=True, need_lse=True)
    o_ref, p_ref = attn_ref(q, kp, v, h)
    assert (attn.cpu().double() - p_ref).abs().max() < (6e-6 if k >= 128 else 1e-5)   # (about 100 keys: probabilities of 1e-2 and more)
    assert rel_err(o.cpu(), o_ref) < (2e-5 if k <= 256 else 4e-5)      # few rows under many keys: O is a short sum of small P
    s_ref = (q.double().view(n, h, dk).transpose(0, 1) @ kp.double().view(k, h, dk).transpose(0, 1).transpose(1, 2)) / dk ** 0.5
    assert (lse.cpu().double() - torch.logsumexp(s_ref, dim=-1)).abs().max() < 3e-5
    assert (attn.sum(-1) - 1).abs().max() < 1e-5
    # without the A / lse outputs (the inference call): the same bits, and bit-identical run to run (fixed-order reduction)
    o2, a2, _ = ops().sparse_attn_fwd_x3_hl(qi, vi, kp.to(DEV), h)
    o3, _, _ = ops().sparse_attn_fwd_x3_hl(qi, vi, kp.to(DEV), h)
    assert a2 is None and torch.equal(o2, o3) and torch.equal(o2, o)
    # and the round-3 kernel on the fp32 tensors the images were made of
    if k <= 224:
        o4, a4, _ = ops().sparse_attn_fwd_x3(q.to(DEV), v.to(DEV), kp.to(DEV), h, need_attn=True)
        assert (attn - a4).abs().max() < (6e-6 if k >= 128 else 1e-5) and rel_err(o.cpu(), o4.cpu()) < 2e-5


def test_sparse_attn_x3_hl_config_b_spike_and_domain():
    """Config-B size (24 tiles per workgroup, head changes inside six workgroups' ranges) + a dominating key (no overflow) +
    the shapes the entry point refuses."""
    n, k, h, dk = 32768, 200, 6, 128
    g = torch.Generator().manual_seed(11)
    d = h * dk
    q, kp, v = torch.randn(n, d, generator=g), torch.randn(k, d, generator=g), torch.randn(n, d, generator=g)
    q[7] *= 40.0
    kp[3] *= 25.0
    img = ops().split_hl_rows(torch.cat([q, v], dim=1).to(DEV))
    o, attn, _ = ops().sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], kp.to(DEV), h, need_attn=True)
    assert torch.isfinite(o).all() and torch.isfinite(attn).all()
    rows = torch.arange(0, n, 61)
    o_ref, p_ref = attn_ref(q, kp, v, h)
    assert (attn[:, rows.to(DEV), :].cpu().double() - p_ref[:, rows, :]).abs().max() < 1e-3   # spiked operands: see the x3 test
    assert rel_err(o.cpu(), o_ref) < 1e-3
    # column-sum checksum at full size: sum_k O[k, :] = sum_n V[n, :] (every row's probabilities sum to one)
    assert rel_err(o.cpu().view(k, h, dk).sum(0), v.view(n, h, dk).sum(0)) < 1e-4
    from snuffy_amd import SnuffyHipError
    with pytest.raises(SnuffyHipError):
        ops().sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], torch.zeros(96, d, device=DEV), h)    # fewer than 4 key blocks
    with pytest.raises(SnuffyHipError):
        ops().sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], torch.zeros(2049, d, device=DEV), h)  # more than 8 chunks
    with pytest.raises(SnuffyHipError):
        ops().sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], torch.zeros(200, d, device=DEV), 24)  # dk = 32


@pytest.mark.parametrize("n,k,h,dk", [(3000, 200, 6, 128), (5000, 97, 2, 128), (4000, 256, 3, 128), (3000, 512, 6, 128), (900, 1024, 2, 128),
                                      (3000, 200, 6, 64), (8192, 200, 6, 64), (2000, 512, 4, 64),
                                      # round 5: chunks are whole key blocks -- any chunked key count has the fused form (a shorter last chunk
                                      # runs with its trailing key blocks masked): the README recipe's 900 keys, 257 = 160 + 97, 1790
                                      (2000, 900, 4, 128), (700, 257, 2, 128), (1500, 601, 3, 64), (300, 1790, 1, 128)])
def test_sparse_attn_x3_hl_fused_key_projection(n, k, h, dk):
    """snf_linear_rows_x3_kpfrag_f32 + snf_sparse_attn_fwd_x3_hl_kpfrag: the key projection writes the attention kernel's fragment
    image itself -- bit-identical to projection -> fp32 Kp -> snf_sparse_attn_fwd_x3_hl (same products, same rounding points)."""
    o_ = ops()
    d = h * dk
    g = torch.Generator().manual_seed(n + k)
    xs = torch.randn(k, d, generator=g).to(DEV)
    w = (torch.randn(d, d, generator=g) / d ** 0.5).to(DEV)
    b = torch.randn(d, generator=g).to(DEV)
    img = o_.split_hl_rows(torch.randn(n, 2 * d, generator=g).to(DEV))
    assert o_.x3_hl_kpfrag_supported(k, h, dk)
    kp = o_.linear_rows_x3(xs, w, b)
    frag = o_.linear_rows_x3_kpfrag(xs, w, b, h)
    o1, a1, l1 = o_.sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], kp, h, need_attn=True, need_lse=True)
    o2, a2, l2 = o_.sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], frag, h, need_attn=True, need_lse=True)
    assert torch.equal(o1, o2) and torch.equal(a1, a2) and torch.equal(l1, l2)
    o3, _, _ = o_.sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], frag, h)
    assert torch.equal(o3, o2)
    nb = o_.linear_rows_x3_kpfrag(xs, w, None, h)                                 # no bias
    o4, _, _ = o_.sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], nb, h)
    o5, _, _ = o_.sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], o_.linear_rows_x3(xs, w, None), h)
    assert torch.equal(o4, o5)


@pytest.mark.parametrize("n,k,h,dk", [(3000, 200, 6, 128), (32768, 200, 6, 128), (8192, 200, 6, 64), (700, 257, 2, 128), (5000, 900, 4, 128),
                                      (100000, 512, 6, 128), (300, 97, 1, 64)])
def test_gather_fused_into_the_key_projection(n, k, h, dk):
    """snf_gather_linear_rows_x3_kpfrag_f32 (round 6): gather + row -> slot map + key projection in one launch -- the same bytes as
    snf_gather_slot_map_f32 followed by snf_linear_rows_x3_kpfrag_f32 (snuffy.py:131,145-147,190)."""
    o_ = ops()
    d = h * dk
    g = torch.Generator().manual_seed(n + 3 * k)
    x = torch.randn(n, d, generator=g).to(DEV)
    idx = torch.randperm(n, generator=g)[:k].to(DEV)
    w = (torch.randn(d, d, generator=g) / d ** 0.5).to(DEV)
    b = torch.randn(d, generator=g).to(DEV)
    xs0, slot0 = o_.gather_slot_map(x, idx)
    frag0 = o_.linear_rows_x3_kpfrag(xs0, w, b, h)
    frag1, xs1, slot1 = o_.gather_linear_rows_x3_kpfrag(x, idx, w, b, h)
    assert torch.equal(xs0, xs1) and torch.equal(slot0, slot1)
    if k <= 256:                                  # one key chunk: every byte of the image is written
        assert torch.equal(frag0.buf, frag1.buf)
    m = min(n, 4096)
    img = o_.split_hl_rows(torch.randn(m, 2 * d, generator=g).to(DEV))
    o0, a0, _ = o_.sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], frag0, h, need_attn=True)
    o1, a1, _ = o_.sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], frag1, h, need_attn=True)
    assert torch.equal(o0, o1) and torch.equal(a0, a1)
    frag2, xs2, slot2 = o_.gather_linear_rows_x3_kpfrag(x, idx, w, None, h, want_map=False)
    o2, _, _ = o_.sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], frag2, h)
    o3, _, _ = o_.sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], o_.linear_rows_x3_kpfrag(xs0, w, None, h), h)
    assert slot2 is None and torch.equal(xs2, xs0) and torch.equal(o2, o3)


def test_sparse_attn_x3_hl_fused_key_projection_domain():
    assert ops().x3_hl_kpfrag_supported(257, 2, 128)          # (round 5: chunks of 160 + 97 keys, on key-block boundaries)
    assert not ops().x3_hl_kpfrag_supported(200, 24, 32)      # dk = 32
    assert not ops().x3_hl_kpfrag_supported(96, 6, 128)
    with pytest.raises(ValueError):
        ops().linear_rows_x3_kpfrag(torch.zeros(96, 256, device=DEV), torch.zeros(256, 256, device=DEV), None, 2)   # fewer than 97 keys


def test_mfma_rejects_unsupported_shapes():
    from snuffy_amd import SnuffyHipError
    q = torch.zeros(64, 96, device=DEV)
    with pytest.raises(SnuffyHipError):
        ops().sparse_attn_fwd_mfma(q, torch.zeros(64, 96, device=DEV), torch.zeros(8, 96, device=DEV), 64, 2)  # dk=48
    q = torch.zeros(64, 256, device=DEV)
    with pytest.raises(SnuffyHipError):   # dk = 128: 8 key chunks of 224 at most
        ops().sparse_attn_fwd_mfma(q, torch.zeros(64, 256, device=DEV), torch.zeros(1793, 256, device=DEV), 64, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,c", [(1, 64, 1), (1000, 166, 1), (4099, 384, 2), (513, 768, 1)])
def test_critic_ln_equals_critic_plus_layernorm(n, d, c):
    """The fused pass (snf_critic_ln_f32) must give bit-identical scores and bf16 xhat to the two separate kernels."""
    g = torch.Generator().manual_seed(n + d)
    x = (torch.randn(n, d, generator=g) * 2 + 0.3).to(DEV)
    w = torch.randn(c, d, generator=g).to(DEV)
    b = torch.randn(c, generator=g).to(DEV)
    s_f, xhat_f = ops().critic_ln(x, w, b, 1e-5)
    s_ref = ops().critic(x, w, b)
    xhat_ref = ops().layernorm_rows(x, None, None, 1e-5, out_dtype=torch.bfloat16)
    assert torch.equal(s_f, s_ref)
    assert torch.equal(xhat_f.view(torch.int16), xhat_ref.view(torch.int16))


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,k", [(1, 64, 1), (1000, 166, 40), (32768, 768, 200), (5000, 384, 2048)])
def test_gather_slot_map_equals_separate_kernels(n, d, k):
    g = torch.Generator().manual_seed(n + k)
    x = torch.randn(n, d, generator=g).to(DEV)
    idx = torch.randperm(n, generator=g)[:k].to(DEV)
    xs, m = ops().gather_slot_map(x, idx)
    assert torch.equal(xs, ops().gather_rows(x, idx))
    assert torch.equal(m, ops().slot_map(idx, n))
    assert torch.equal(xs, x[idx])
    xs2, m2, xs16 = ops().gather_slot_map(x, idx, bf16_copy=True)       # + the bf16 copy for the bf16 key projection
    assert torch.equal(xs2, xs) and torch.equal(m2, m)
    assert torch.equal(xs16.view(torch.int16), xs.to(torch.bfloat16).view(torch.int16))


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,h,dk,drop", [(1, 1, 1, 8, 0.0), (50, 7, 2, 16, 0.0), (1000, 200, 6, 64, 0.0), (777, 130, 3, 83, 0.0),
                                           (2048, 200, 6, 128, 0.0), (300, 64, 2, 32, 0.3), (5000, 512, 2, 64, 0.0),
                                           # round 5 (f32 matrix-core forms): the README head widths, a mask at an odd k, k > 1024
                                           (4100, 500, 4, 192, 0.0), (3001, 900, 4, 96, 0.1), (999, 131, 2, 24, 0.2), (400, 1100, 1, 64, 0.0)])
def test_sparse_attn_bwd_matches_autograd_reference(n, k, h, dk, drop):
    """snf_sparse_attn_bwd_f32 against torch.autograd through the plain fp64 formulation of snuffy.py:160-168."""
    g = torch.Generator().manual_seed(n + k + dk)
    d = h * dk
    q, kp, v = (torch.randn(s, d, generator=g) for s in (n, k, n))
    dout = torch.randn(k, d, generator=g)
    mask = None
    if drop > 0:
        mask = (torch.rand(h, n, k, generator=g) >= drop).float() / (1.0 - drop)
    _, p, _ = ops().sparse_attn_fwd(q.to(DEV), kp.to(DEV), v.to(DEV), h, need_attn=True)
    dq, dkp, dv = ops().sparse_attn_bwd(q.to(DEV), kp.to(DEV), v.to(DEV), p, dout.to(DEV), h,
                                        mask=None if mask is None else mask.to(DEV))
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, kp, v))
    qh, kh, vh = (t.view(-1, h, dk).transpose(0, 1) for t in (qd, kd, vd))
    pr = torch.softmax(qh @ kh.transpose(1, 2) / dk ** 0.5, dim=-1)
    if mask is not None:
        pr = pr * mask.double()
    o = (pr.transpose(1, 2) @ vh).transpose(0, 1).reshape(k, d)
    o.backward(dout.double())
    for got, ref, name in ((dq, qd.grad, "dq"), (dkp, kd.grad, "dkp"), (dv, vd.grad, "dv")):
        assert rel_err(got.cpu(), ref) < 2e-5, name
    # deterministic
    dq2, dkp2, dv2 = ops().sparse_attn_bwd(q.to(DEV), kp.to(DEV), v.to(DEV), p, dout.to(DEV), h,
                                           mask=None if mask is None else mask.to(DEV))
    assert torch.equal(dq, dq2) and torch.equal(dkp, dkp2) and torch.equal(dv, dv2)
    # the vector-ALU kernels (round 1) compute the same fp32 products in another order
    from snuffy_amd import _ffi
    _ffi.load().snf_debug_exact_attn_mfma(0)
    try:
        dq3, dkp3, dv3 = ops().sparse_attn_bwd(q.to(DEV), kp.to(DEV), v.to(DEV), p, dout.to(DEV), h,
                                               mask=None if mask is None else mask.to(DEV))
    finally:
        _ffi.load().snf_debug_exact_attn_mfma(1)
    for a_, b_ in ((dq, dq3), (dkp, dkp3), (dv, dv3)):
        assert rel_err(a_.cpu(), b_.cpu()) < 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("dk", [128, 64])
@pytest.mark.parametrize("n,k,h,drop", [(1, 1, 1, 0.0), (100, 31, 2, 0.0), (129, 33, 3, 0.0), (1000, 64, 6, 0.0),
                                        (2500, 200, 6, 0.0), (3000, 224, 2, 0.0), (640, 129, 4, 0.25), (4099, 100, 1, 0.0),
                                        (8192, 200, 2, 0.1)])     # the last one: dKp through the row-chunked batched GEMM
def test_sparse_attn_bwd_mfma(n, k, h, drop, dt, dk):
    """MFMA backward (dk = 128) against fp64 autograd on the SAME bf16-rounded operands (layout slips show as O(1) errors)
    and against the exact operands at bf16-class tolerance."""
    g = torch.Generator().manual_seed(7 * n + k)
    d = h * dk
    q, kp, v = (torch.randn(s, d, generator=g) for s in (n, k, n))
    dout = torch.randn(k, d, generator=g)
    mask = None
    if drop > 0:
        mask = (torch.rand(h, n, k, generator=g) >= drop).float() / (1.0 - drop)
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    qv = torch.cat([q, v], dim=1).to(DEV).to(tdt)                      # row-strided halves, as the model passes them
    qd_, vd_ = qv[:, :d], qv[:, d:]
    _, _, lse = ops().sparse_attn_fwd_mfma(qd_, vd_, kp.to(DEV), n, h, need_lse=True)
    dq, dkp, dv = ops().sparse_attn_bwd_mfma(qd_, vd_, kp.to(DEV), dout.to(DEV), lse, h,
                                             mask=None if mask is None else mask.to(DEV))

    def ref(qq, kk, vv, dd):
        qd, kd, vd = (t.double().requires_grad_(True) for t in (qq, kk, vv))
        qh, kh, vh = (t.view(-1, h, dk).transpose(0, 1) for t in (qd, kd, vd))
        pr = torch.softmax(qh @ kh.transpose(1, 2) / dk ** 0.5, dim=-1)
        if mask is not None:
            pr = pr * mask.double()
        o = (pr.transpose(1, 2) @ vh).transpose(0, 1).reshape(k, d)
        o.backward(dd.double())
        return qd.grad, kd.grad, vd.grad
    # the kernel rounds q, v, kp, dout to bf16 and keeps P / dP / dS as bf16 MFMA operands: bf16-class tolerance
    rq, rk, rv = ref(bf16r(q), bf16r(kp), bf16r(v), bf16r(dout))
    def close(got, want, tol, name):   # relative to the gradient's own scale; a single key has dQ = dKp = 0 exactly
        scale_ = max(float(want.abs().max()), 1e-3)
        assert float((got.cpu().double() - want).abs().max()) < tol * 4 * scale_ or rel_err(got.cpu(), want) < tol, name
    for got, want, name in ((dq, rq, "dq"), (dkp, rk, "dkp"), (dv, rv, "dv")):
        close(got, want, 1.5e-2, name)
    eq, ek, ev = ref(q, kp, v, dout)
    for got, want, name in ((dq, eq, "dq"), (dkp, ek, "dkp"), (dv, ev, "dv")):
        close(got, want, 2e-2, name)
    dq2, dkp2, dv2 = ops().sparse_attn_bwd_mfma(qd_, vd_, kp.to(DEV), dout.to(DEV), lse, h,
                                                mask=None if mask is None else mask.to(DEV))
    assert torch.equal(dq, dq2) and torch.equal(dkp, dkp2) and torch.equal(dv, dv2)
    # gradients written as bf16 into the two halves of one [n, 2 d] buffer == the fp32 outputs rounded once
    dq3, dkp3, dv3 = ops().sparse_attn_bwd_mfma(qd_, vd_, kp.to(DEV), dout.to(DEV), lse, h,
                                                mask=None if mask is None else mask.to(DEV), fused_bf16_grads=True)
    assert dq3._base is dv3._base and dq3._base.shape == (n, 2 * h * dk) and dq3.dtype == torch.bfloat16
    assert torch.equal(dq3, dq.to(torch.bfloat16)) and torch.equal(dv3, dv.to(torch.bfloat16)) and torch.equal(dkp3, dkp)


# ---------------------------------------------------------------- attention dropout: mask regenerated in the kernels
@pytest.mark.parametrize("h,n,k,p,seed,offset", [(2, 37, 200, 0.1, 1234, 0), (6, 300, 203, 0.1, 0, 7), (1, 5, 3, 0.5, 2 ** 63 + 11, 2 ** 40 + 3),
                                                 (3, 1000, 64, 0.9, 42, 2 ** 62 - 1), (2, 10, 8, 0.0, 1, 1)])
def test_dropout_mask_kernel_equals_host_philox(h, n, k, p, seed, offset):
    """snf_dropout_mask_f32 (the mask the attention kernels regenerate in registers) == the numpy Philox4x32-10 restatement,
    element for element; the kept fraction is 1 - p."""
    from oracle import philox_ref
    got = ops().dropout_mask(h, n, k, p, seed, offset, DEV).cpu().numpy()
    want = philox_ref.dropout_mask(h, n, k, p, seed, offset)
    assert np.array_equal(got, want)
    if p > 0 and h * n * k > 10000:
        assert abs(float((got > 0).mean()) - (1 - p)) < 0.01


@pytest.mark.parametrize("n,k,h,dk,dt", [(3000, 200, 6, 128, "bf16"), (777, 130, 2, 128, "f32"), (2048, 256, 6, 64, "bf16")])
def test_mfma_attention_dropout_forward_and_backward(n, k, h, dk, dt):
    """Train-mode attention on the MFMA kernels: the forward's O / A and the backward's gradients equal the oracle evaluated
    with the SAME mask, reconstructed on the host from (seed, offset); the backward with the regenerated mask equals the
    backward fed with the mask tensor."""
    from oracle import philox_ref
    p_drop, seed, offset = 0.1, 987654321, 5
    g = torch.Generator().manual_seed(n + k)
    d = h * dk
    q, kp, v, dout = (torch.randn(s, d, generator=g) for s in (n, k, n, k))
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    qv = torch.cat([q, v], dim=1).to(DEV).to(tdt)
    qd_, vd_ = qv[:, :d], qv[:, d:]
    mask = torch.from_numpy(philox_ref.dropout_mask(h, n, k, p_drop, seed, offset))
    o, attn, lse = ops().sparse_attn_fwd_mfma(qd_, vd_, kp.to(DEV), n, h, need_attn=True, need_lse=True,
                                              dropout=(p_drop, seed, offset))
    o0, attn0, lse0 = ops().sparse_attn_fwd_mfma(qd_, vd_, kp.to(DEV), n, h, need_attn=True, need_lse=True)
    assert torch.equal(lse, lse0)                                        # the statistics are those of the un-dropped softmax
    assert torch.equal(attn.cpu(), attn0.cpu() * mask)                   # A = P o M, bit for bit
    qr, kr, vr = bf16r(q), bf16r(kp), bf16r(v)
    _, p_r = attn_ref(qr, kr, vr, h)
    o_r = ((p_r * mask.double()).transpose(1, 2) @ vr.double().view(n, h, dk).transpose(0, 1)).transpose(0, 1).reshape(k, d)
    assert rel_err(o.cpu(), o_r) < 3e-3
    # backward: regenerated mask == explicit mask tensor, and both == fp64 autograd with that mask
    dq, dkp, dv = ops().sparse_attn_bwd_mfma(qd_, vd_, kp.to(DEV), dout.to(DEV), lse, h, dropout=(p_drop, seed, offset))
    dq2, dkp2, dv2 = ops().sparse_attn_bwd_mfma(qd_, vd_, kp.to(DEV), dout.to(DEV), lse, h, mask=mask.to(DEV))
    assert torch.equal(dq, dq2) and torch.equal(dkp, dkp2) and torch.equal(dv, dv2)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (qr, kr, vr))
    qh, kh, vh = (t.view(-1, h, dk).transpose(0, 1) for t in (qd, kd, vd))
    pr = torch.softmax(qh @ kh.transpose(1, 2) / dk ** 0.5, dim=-1) * mask.double()
    (pr.transpose(1, 2) @ vh).transpose(0, 1).reshape(k, d).backward(bf16r(dout).double())
    for got, want, name in ((dq, qd.grad, "dq"), (dkp, kd.grad, "dkp"), (dv, vd.grad, "dv")):
        assert rel_err(got.cpu(), want) < 1.5e-2, name


def test_training_attention_is_reproducible_under_manual_seed():
    """SparseAttnFn draws (seed, offset) from torch's CPU generator: same torch.manual_seed -> same dropped forward and same
    gradients; bf16-autocast keeps no [h, N, K] tensor for the mask."""
    from snuffy_amd import autograd as SA
    g = torch.Generator().manual_seed(3)
    n, k, h, d = 1500, 200, 6, 768
    q, kp, v = (torch.randn(s, d, generator=g).to(DEV).requires_grad_(True) for s in (n, k, n))
    outs = []
    for seed in (5, 5, 6):
        torch.manual_seed(seed)
        for t in (q, kp, v):
            t.grad = None
        o, p = SA.SparseAttnFn.apply(q, kp, v, h, 0.1, True, False)
        assert p is None
        o.square().sum().backward()
        outs.append((o.detach().clone(), q.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert not torch.equal(outs[0][0], outs[2][0])


@pytest.mark.parametrize("n,d", [(1000, 384), (20000, 768), (333, 64), (40000, 1024)])
def test_critic_ln_hl_one_pass_equals_the_two_kernels(n, d):
    """snf_critic_ln_hl_f32 (fp32-class path: critic + LayerNorm_0 with affine -> interleaved hi / lo image, one read of the bag)
    against snf_critic_f32 + snf_layernorm_rows_hl_f32: scores bit for bit; the image to the last fp32 bit of the normalised
    value (the two kernels' row statistics can differ in the last place, which shows in ~1 % of the lo halves and, where a value
    sits on a bf16 rounding boundary, moves hi by one step with lo compensating -- the same 2^-17 class either way); above the
    fused-selector threshold the top-k that follows starts from the histogram the pass counted."""
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(n + d)
    x = (torch.randn(n, d, generator=g) * 2 + 0.3).to(DEV)
    w = (torch.randn(1, d, generator=g) / d ** 0.5).to(DEV)
    b = torch.randn(1, generator=g).to(DEV)
    gamma = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(d, generator=g)).to(DEV)
    s, img = ops.critic_ln_hl(x, w, b, gamma, beta, 1e-5)
    s_ref = ops.critic(x, w, b)
    img_ref = ops.layernorm_rows_hl(x, gamma, beta, 1e-5)
    assert torch.equal(s, s_ref)

    def value(im):
        v = im.view(n, d // 32, 2, 32).float()
        return (v[:, :, 0] + v[:, :, 1]).reshape(n, d)
    v1, v2 = value(img), value(img_ref)
    assert (v1 - v2).abs().max().item() <= 2.0 ** -14 * max(1.0, v2.abs().max().item())
    assert (img.view(torch.int16) != img_ref.view(torch.int16)).float().mean().item() < 0.03
    exact = torch.nn.functional.layer_norm(x.double(), (d,), gamma.double(), beta.double(), 1e-5)
    assert (v1.double() - exact).abs().max().item() <= 2.0 ** -14 * max(1.0, exact.abs().max().item())
    top = ops.topk(s.view(-1), 200)
    ref = torch.sort(s_ref.view(-1), descending=True, stable=True)[1][:200]
    assert torch.equal(top, ref)
