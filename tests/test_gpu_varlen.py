"""Varlen path: many bags per launch.  The segmented kernels against their per-bag forms (top-k and head
bit-identical; attention: same P bit for bit, O up to the fp32 order of the partial sums, and independent of the batch composition
bit for bit), MILNet.forward_bags against the per-bag forwards (same selections, same draws of the random share) and against the
CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import snuffy_oracle as orc
from tests.helpers import build_amd_milnet

pytestmark = pytest.mark.gpu
DEV = "cuda"

SIZES = [300, 1000, 129, 5000, 2048, 777]


def _packed(sizes):
    from snuffy_amd import ops
    return ops.PackedBags(sizes, DEV)


def test_topk_segmented_matches_per_bag_selection():
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(0)
    for sizes, k in ((SIZES, 100), ([200, 9000, 20000, 40000, 201], 200), ([64, 64], 64), ([70000, 3000], 150)):
        pk = _packed(sizes)
        s = torch.randn(pk.total, generator=g)
        s[::7] = s[3]                      # ties: the stable rule (ascending index) must hold inside every bag
        s[5] = float("inf")
        s[pk.total - 1] = float("-inf")
        s = s.to(DEV)
        got = ops.topk_segmented(s, pk, k)
        for b, n in enumerate(sizes):
            lo = int(pk.host[b])
            ref = ops.topk(s[lo:lo + n].clone(), k)
            assert torch.equal(got[b], ref), (sizes, b)
            c = s[lo:lo + n].cpu().numpy()
            order = np.lexsort((np.arange(n), -c))[:k]          # descending score, ascending index
            assert np.array_equal(got[b].cpu().numpy(), order)


@pytest.mark.parametrize("d,h,k", [(768, 6, 200), (768, 6, 224), (384, 6, 200), (384, 6, 37), (256, 2, 64)])
@pytest.mark.parametrize("need_attn", [False, True])
def test_attention_bf16_varlen_vs_per_bag_and_composition_independent(d, h, k, need_attn):
    from snuffy_amd import ops
    sizes = [n for n in SIZES if n >= k] + [k]
    pk = _packed(sizes)
    g = torch.Generator().manual_seed(1)
    qv = torch.randn(pk.total, 2 * d, generator=g).to(DEV).to(torch.bfloat16)
    kp = (torch.randn(pk.bags * k, d, generator=g) * 0.5).to(DEV).to(torch.bfloat16)
    q, v = qv[:, :d], qv[:, d:]
    out, attn, lse = ops.sparse_attn_fwd_mfma_varlen(q, v, kp, pk, k, h, need_attn=need_attn, need_lse=need_attn)
    assert out.shape == (pk.bags * k, d)
    for b, n in enumerate(sizes):
        lo = int(pk.host[b])
        qb = qv[lo:lo + n]
        o1, a1, l1 = ops.sparse_attn_fwd_mfma(qb[:, :d], qb[:, d:], kp[b * k:(b + 1) * k], n, h, need_attn=need_attn,
                                              need_lse=need_attn)
        # the single-bag entry point spreads a small bag over more workgroups: same P, O up to the fp32 order of the partial sums
        assert (out[b * k:(b + 1) * k] - o1).abs().max().item() <= 2e-6 * o1.abs().max().item(), (b, n)
        if need_attn:
            assert torch.equal(attn[:, lo:lo + n], a1)
            assert torch.equal(lse[:, lo:lo + n], l1)
        # ... and a bag's result does not depend on what it is packed with: alone in a varlen launch, bit for bit
        o2, a2, l2 = ops.sparse_attn_fwd_mfma_varlen(qb[:, :d], qb[:, d:], kp[b * k:(b + 1) * k], _packed([n]), k, h,
                                                     need_attn=need_attn, need_lse=need_attn)
        assert torch.equal(out[b * k:(b + 1) * k], o2), (b, n)
        if need_attn:
            assert torch.equal(attn[:, lo:lo + n], a2) and torch.equal(lse[:, lo:lo + n], l2)
    # and against fp64 on one bag (the kernels agree with each other; this pins them to the definition)
    b = 1
    lo, n = int(pk.host[b]), sizes[b]
    dk = d // h
    qd = qv[lo:lo + n, :d].double().view(n, h, dk).transpose(0, 1)
    vd = qv[lo:lo + n, d:].double().view(n, h, dk).transpose(0, 1)
    kd = kp[b * k:(b + 1) * k].double().view(k, h, dk).transpose(0, 1)
    p = torch.softmax(qd @ kd.transpose(1, 2) / dk ** 0.5, dim=-1)
    ref = (p.transpose(1, 2) @ vd).transpose(0, 1).reshape(k, d)
    err = (out[b * k:(b + 1) * k].double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-2, err


@pytest.mark.parametrize("d,h,k", [(768, 6, 200), (384, 6, 200), (384, 6, 50), (768, 6, 100)])
@pytest.mark.parametrize("need_attn", [False, True])
def test_attention_x3_varlen_vs_per_bag_and_composition_independent(d, h, k, need_attn):
    from snuffy_amd import ops
    sizes = [n for n in SIZES if n >= k] + [k]
    pk = _packed(sizes)
    g = torch.Generator().manual_seed(2)
    qv = torch.randn(pk.total, 2 * d, generator=g).to(DEV)
    kp = (torch.randn(pk.bags * k, d, generator=g) * 0.5).to(DEV)
    q, v = qv[:, :d], qv[:, d:]
    out, attn, lse = ops.sparse_attn_fwd_x3_varlen(q, v, kp, pk, k, h, need_attn=need_attn, need_lse=need_attn)
    for b, n in enumerate(sizes):
        lo = int(pk.host[b])
        qb = qv[lo:lo + n]
        o1, a1, l1 = ops.sparse_attn_fwd_x3(qb[:, :d], qb[:, d:], kp[b * k:(b + 1) * k], h, need_attn=need_attn, need_lse=need_attn)
        assert (out[b * k:(b + 1) * k] - o1).abs().max().item() <= 2e-6 * o1.abs().max().item(), (b, n)
        if need_attn:
            assert torch.equal(attn[:, lo:lo + n], a1)
            assert torch.equal(lse[:, lo:lo + n], l1)
        o2, a2, l2 = ops.sparse_attn_fwd_x3_varlen(qb[:, :d], qb[:, d:], kp[b * k:(b + 1) * k], _packed([n]), k, h,
                                                   need_attn=need_attn, need_lse=need_attn)
        assert torch.equal(out[b * k:(b + 1) * k], o2), (b, n)
        if need_attn:
            assert torch.equal(attn[:, lo:lo + n], a2) and torch.equal(lse[:, lo:lo + n], l2)
    b = 0
    lo, n = int(pk.host[b]), sizes[b]
    dk = d // h
    qd = qv[lo:lo + n, :d].double().view(n, h, dk).transpose(0, 1)
    vd = qv[lo:lo + n, d:].double().view(n, h, dk).transpose(0, 1)
    kd = kp[b * k:(b + 1) * k].double().view(k, h, dk).transpose(0, 1)
    p = torch.softmax(qd @ kd.transpose(1, 2) / dk ** 0.5, dim=-1)
    ref = (p.transpose(1, 2) @ vd).transpose(0, 1).reshape(k, d)
    err = (out[b * k:(b + 1) * k].double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-5, err


@pytest.mark.parametrize("d", [384, 768, 102])
def test_head_varlen_bit_identical_to_per_bag(d):
    from snuffy_amd import ops
    sizes = SIZES + [1, 3]
    pk = _packed(sizes)
    g = torch.Generator().manual_seed(3)
    z = torch.randn(pk.total, d, generator=g).to(DEV)
    add = torch.randn(pk.total, d, generator=g).to(DEV).to(torch.bfloat16)
    bias = torch.randn(d, generator=g).to(DEV)
    gamma, beta = torch.randn(d, generator=g).to(DEV), torch.randn(d, generator=g).to(DEV)
    wh, bh = torch.randn(2, d, generator=g).to(DEV), torch.randn(2, generator=g).to(DEV)
    # two patched rows in every bag (slots in packed coordinates)
    sel = torch.tensor([int(pk.host[b]) + j for b, n in enumerate(sizes) for j in ((0, n - 1) if n > 1 else (0,))], device=DEV)
    delta = torch.randn(sel.shape[0], d, generator=g).to(DEV)
    slot = ops.slot_map(sel, pk.total)
    for kw in (dict(), dict(add_bf16=add, add_bias=bias, slot=slot, delta_rows=delta)):
        logits, pooled = ops.ln_mean_head_varlen(z, pk, gamma, beta, 1e-5, wh, bh, **kw)
        for b, n in enumerate(sizes):
            lo = int(pk.host[b])
            kb = {}
            if kw:
                in_bag = (sel >= lo) & (sel < lo + n)
                kb = dict(add_bf16=add[lo:lo + n], add_bias=bias, slot=ops.slot_map(sel[in_bag] - lo, n), delta_rows=delta[in_bag])
            l1, p1, _ = ops.ln_mean_head(z[lo:lo + n], gamma, beta, 1e-5, wh, bh, **kb)
            assert torch.equal(logits[b], l1), (b, n)
            assert torch.equal(pooled[b], p1)


def _net(d, h, lam, r, depth, precision, seed=0):
    torch.manual_seed(seed)
    net = build_amd_milnet(d, h, "relu", lam, r, depth).to(DEV).eval()
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() > 1:
                torch.nn.init.xavier_uniform_(p)
    net.configure(precision=precision, return_attention=True)
    return net


def _bags(sizes, d, seed=5):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(1, n, d, generator=g).to(DEV) for n in sizes]


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("d,lam,r,depth", [(384, 200, 0.0, 1), (768, 200, 0.0, 1), (384, 64, 0.25, 2), (384, 200, 0.5, 1)])
def test_forward_bags_matches_per_bag_forwards(precision, d, lam, r, depth):
    """Same selections (bit-exact, random share included), same A / logits up to the rounding of projections run over a
    different row count (tile / library choice), for a length mix that includes a bag of exactly Lambda patches."""
    sizes = [lam, 1000, 333, 4100, 2048]
    net = _net(d, 6, lam, r, depth, precision)
    bags = _bags(sizes, d)
    with torch.no_grad():
        np.random.seed(11)
        ref, sel_ref = [], []
        for x in bags:
            ref.append(net(x))
            sel_ref.append([tuple(None if t is None else t.clone() for t in l.last_selection) for l in net.b_classifier.encoder.layers])
        np.random.seed(11)
        assert net._packable(bags)
        got = net.forward_bags(bags)
        after_packed = np.random.rand()
        np.random.seed(11)
        [net(x) for x in bags]
        assert np.random.rand() == after_packed          # the numpy stream is left where the per-bag loop leaves it
    for li, layer in enumerate(net.b_classifier.encoder.layers):       # selections, random share included: bit-exact
        top, rnd = layer.last_selection_bags
        for b in range(len(bags)):
            assert torch.equal(top[b], sel_ref[b][li][0])
            assert (rnd is None and sel_ref[b][li][1] is None) or torch.equal(rnd[b], sel_ref[b][li][1])
    # fp32-class: products to 2^-17 in both runs, but the packed projections may take another kernel form (row count) and the
    # B x K selected rows go through the x3 GEMM instead of the fp32 library -- measured 5e-6 on A at depth 2
    tol_logit, tol_a = (2e-5, 2e-5) if precision == "fp32" else (2e-2, 2e-2)
    for b, ((c0, y0, a0), (c1, y1, a1)) in enumerate(zip(ref, got)):
        assert c1.shape == c0.shape and y1.shape == y0.shape and a1.shape == a0.shape
        assert torch.equal(c0, c1)                       # critic scores: same kernel, row-wise
        assert (y0 - y1).abs().max().item() <= tol_logit * max(1.0, y0.abs().max().item()), (b, y0, y1)
        assert (a0 - a1).abs().max().item() <= tol_a, b


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16", 2e-2)])
def test_forward_bags_vs_oracle(precision, tol):
    d, h, lam = 384, 6, 200
    sizes = [1000, 1500, 600, 250]
    net = _net(d, h, lam, 0.0, 1, precision)
    bags = _bags(sizes, d, seed=9)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    with torch.no_grad():
        got = net.forward_bags(bags)
    top, _ = net.b_classifier.encoder.layers[0].last_selection_bags
    for b, x in enumerate(bags):
        classes, logits, attn, sels = orc.milnet_forward(x[0].cpu(), sd, h, "relu", lam, 0.0, 1)
        assert np.array_equal(sels[0].numpy(), top[b].cpu().numpy())                   # bit-exact top-Lambda indices per bag
        assert (got[b][0][0].cpu() - classes).abs().max().item() <= 2e-5
        assert (got[b][1][0].cpu() - logits).abs().max().item() <= tol * max(1.0, logits.abs().max().item())
        assert (got[b][2][0].cpu() - attn).abs().max().item() <= tol


def test_forward_bags_falls_back_when_not_packable():
    net = _net(384, 6, 200, 0.0, 1, "bf16")
    bags = _bags([1000, 150], 384)          # the second bag is shorter than Lambda: K differs per bag
    with torch.no_grad():
        assert not net._packable(bags)
        got = net.forward_bags(bags)
        ref = [net(x) for x in bags]
    for (c0, y0, a0), (c1, y1, a1) in zip(ref, got):
        assert torch.equal(c0, c1) and torch.equal(y0, y1) and torch.equal(a0, a1)


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-5), ("bf16", 2e-2)])
def test_trainer_valid_packs_small_bags(precision, tol):
    """Trainer.valid with --eval_bags_per_launch 4 against the per-bag loop: same predictions / loss to rounding, for a set that
    mixes packable chunks, a chunk with a bag shorter than Lambda (forward_bags falls back) and a bag above the pack limit."""
    from snuffy_amd.train import Snuffy, get_args_parser
    a = get_args_parser().parse_args([])
    a.feats_size, a.optimizer, a.num_epochs, a.precision, a.random_patch_share = 384, "adamw", 2, precision, 0.25
    torch.manual_seed(0)
    np.random.seed(0)
    tr = Snuffy(a)
    sizes = [400, 900, 250, 1300, 600, 150, 700, 800, 5000, 300, 310]
    g = torch.Generator().manual_seed(4)
    feats = [torch.randn(n, 384, generator=g).numpy() for n in sizes]
    labels = [np.array([i % 2], dtype=np.float32) for i in range(len(sizes))]
    np.random.seed(3)
    ref = tr.valid((labels, feats))
    a.eval_bags_per_launch, a.eval_pack_max_patches = 4, 4096
    np.random.seed(3)
    got = tr.valid((labels, feats))
    assert np.abs(ref["predictions"] - got["predictions"]).max() <= tol
    assert abs(ref["epoch_valid_loss"] - got["epoch_valid_loss"]) <= tol
    assert np.array_equal(ref["labels"], got["labels"])


# ---- ragged groups: bags shorter than Lambda (they select all their rows) and head widths outside the MFMA kernels --------------
def test_topk_segmented_with_bags_shorter_than_k():
    from snuffy_amd import ops
    sizes = [5, 300, 1, 40, 200, 199]
    pk = _packed(sizes)
    g = torch.Generator().manual_seed(6)
    s = torch.randn(pk.total, generator=g)
    s[::5] = s[1]
    s = s.to(DEV)
    got = ops.topk_segmented(s, pk, 200)
    for b, n in enumerate(sizes):
        lo = int(pk.host[b])
        kb = min(200, n)
        c = s[lo:lo + n].cpu().numpy()
        order = np.lexsort((np.arange(n), -c))[:kb]
        assert np.array_equal(got[b, :kb].cpu().numpy(), order), (b, n)


@pytest.mark.parametrize("d,h", [(166, 2), (230, 2), (384, 6), (64, 1)])
def test_ragged_attention_vs_exact_per_bag(d, h):
    from snuffy_amd import ops
    sizes = [5, 40, 1, 17, 300, 2, 33]
    kbs = [min(200, n) for n in sizes]
    pk = _packed(sizes)
    rag = pk.ragged(kbs)
    g = torch.Generator().manual_seed(8)
    q = torch.randn(pk.total, d, generator=g).to(DEV)
    v = torch.randn(pk.total, d, generator=g).to(DEV)
    kp = (torch.randn(sum(kbs), d, generator=g) * 0.5).to(DEV)
    out, attn, lse = ops.sparse_attn_fwd_ragged(q, v, kp, pk, rag, h, need_attn=True, need_lse=True)
    assert out.shape == (sum(kbs), d) and attn.shape == (h, pk.total, max(kbs))
    for b, n in enumerate(sizes):
        lo, k0, kb = int(pk.host[b]), int(rag.koff[b]), kbs[b]
        o1, a1, l1 = ops.sparse_attn_fwd(q[lo:lo + n], kp[k0:k0 + kb], v[lo:lo + n], h, need_attn=True, need_lse=True)
        assert (out[k0:k0 + kb] - o1).abs().max().item() <= 2e-5 * max(1.0, o1.abs().max().item()), (b, n)
        assert (attn[:, lo:lo + n, :kb] - a1).abs().max().item() <= 2e-6
        assert (lse[:, lo:lo + n] - l1).abs().max().item() <= 2e-5
    # and the definition (fp64), one bag
    b = 4
    lo, n, k0, kb, dk = int(pk.host[b]), sizes[b], int(rag.koff[b]), kbs[b], d // h
    qd = q[lo:lo + n].double().view(n, h, dk).transpose(0, 1)
    vd = v[lo:lo + n].double().view(n, h, dk).transpose(0, 1)
    kd = kp[k0:k0 + kb].double().view(kb, h, dk).transpose(0, 1)
    p = torch.softmax(qd @ kd.transpose(1, 2) / dk ** 0.5, dim=-1)
    ref = (p.transpose(1, 2) @ vd).transpose(0, 1).reshape(kb, d)
    assert (out[k0:k0 + kb].double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


def _musk_like(n_bags, d, seed):
    rs = np.random.RandomState(seed)
    sizes = [int(v) for v in rs.randint(2, 41, n_bags)]
    sizes[3] = 1
    return sizes


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("d,h,depth", [(166, 2, 1), (230, 2, 2)])
def test_forward_bags_on_a_musk_shaped_set(precision, d, h, depth):
    """The MIL benchmark shape (train.py:993-995: D = 166 / 230, --num_heads 2, bags of 1-40 instances, all shorter than
    Lambda = 200 so every row is selected): 40 bags in one set of launches against the per-bag forwards and the oracle."""
    sizes = _musk_like(40, d, 3)
    net = _net(d, h, 200, 0.0, depth, precision)
    bags = _bags(sizes, d, seed=12)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    with torch.no_grad():
        groups = net._pack_groups(bags)
        assert groups == [(list(range(len(bags))), True)]
        got = net.forward_bags(bags)
        top, _ = net.b_classifier.encoder.layers[0].last_selection_bags
        ref = [net(x) for x in bags]
    tol = 2e-5 if precision == "fp32" else 2e-2
    for b, ((c0, y0, a0), (c1, y1, a1)) in enumerate(zip(ref, got)):
        n = sizes[b]
        assert c1.shape == c0.shape and y1.shape == y0.shape and a1.shape == a0.shape == (1, h, n, n)
        assert torch.equal(c0, c1)
        assert (y0 - y1).abs().max().item() <= tol * max(1.0, y0.abs().max().item()), b
        assert (a0 - a1).abs().max().item() <= tol
        if precision == "fp32" and b < 8:
            classes, logits, attn, sels = orc.milnet_forward(bags[b][0].cpu(), sd, h, "relu", 200, 0.0, depth)
            assert np.array_equal(sels[0].numpy(), top[b, :n].cpu().numpy())
            assert (y1[0].cpu() - logits).abs().max().item() <= 1e-4 * max(1.0, logits.abs().max().item())
            assert (a1[0].cpu() - attn).abs().max().item() <= 1e-5


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_forward_bags_mixes_uniform_and_ragged_groups(precision):
    sizes = [1000, 150, 2000, 60, 180, 700, 5000]
    net = _net(384, 6, 200, 0.0, 1, precision)
    bags = _bags(sizes, 384, seed=13)
    with torch.no_grad():
        groups = net._pack_groups(bags)
        assert groups == [([0, 2, 5, 6], False), ([1, 3, 4], True)]
        got = net.forward_bags(bags)
        net.configure(graph_max_patches=1 << 16)
        net.forward_bags(bags)                        # first sight of a composition: eager (remembered)
        got_graph = net.forward_bags(bags)            # second: captured and replayed
        got_graph2 = net.forward_bags(bags)           # third: replay only
        assert sum(1 for k in net._graphs if k and k[0] == "bags") == 2      # one graph per group
        net.configure(graph_max_patches=0)
        ref = [net(x) for x in bags]
    tol = 2e-5 if precision == "fp32" else 2e-2
    for b, ((c0, y0, a0), (c1, y1, a1), (c2, y2, a2), (c3, y3, a3)) in enumerate(zip(ref, got, got_graph, got_graph2)):
        assert a1.shape == a0.shape
        assert torch.equal(c0, c1)
        assert (y0 - y1).abs().max().item() <= tol * max(1.0, y0.abs().max().item()), b
        assert (a0 - a1).abs().max().item() <= tol
        assert torch.equal(y1, y2) and torch.equal(a1, a2) and torch.equal(y2, y3)      # graph replay == eager issue, bit for bit


def test_forward_bags_splits_large_batches_into_row_capped_chunks():
    net = _net(384, 6, 200, 0.0, 1, "bf16")
    net._PACK_MAX_ROWS = 5000
    sizes = [2000, 2500, 1000, 3000, 1500, 4000, 900, 800]
    bags = _bags(sizes, 384, seed=14)
    with torch.no_grad():
        groups = net._pack_groups(bags)
        assert groups == [([0, 1], False), ([2, 3], False), ([5, 6], False)]       # 4 and 7 end up alone: per-bag forwards
        got = net.forward_bags(bags)
        ref = [net(x) for x in bags]
    for (c0, y0, a0), (c1, y1, a1) in zip(ref, got):
        assert torch.equal(c0, c1) and (y0 - y1).abs().max().item() <= 2e-2 and (a0 - a1).abs().max().item() <= 2e-2


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("SNF_FUZZ_SEEDS", "10")))))
def test_forward_bags_random_compositions(seed):
    """Random models and batch compositions (bag lengths from 1 patch up, below and above Lambda, head widths inside and outside
    the MFMA kernels, both arithmetics, depth 1-3): forward_bags against the per-bag forwards."""
    rs = np.random.RandomState(1000 + seed)
    d, h = [(64, 1), (128, 2), (384, 6), (768, 6), (166, 2), (256, 4), (384, 3), (96, 2)][rs.randint(8)]
    lam = int(rs.choice([16, 64, 200, 224]))
    depth = int(rs.randint(1, 4))
    precision = ["fp32", "bf16"][rs.randint(2)]
    if d // h not in (64, 128) and lam > 200:
        lam = 200
    nb = int(rs.randint(2, 9))
    sizes = [int(v) for v in np.clip(np.round(np.exp(rs.uniform(0, np.log(6000), nb))), 1, 6000)]
    net = _net(d, h, lam, 0.0, depth, precision, seed=seed)
    bags = _bags(sizes, d, seed=100 + seed)
    with torch.no_grad():
        got = net.forward_bags(bags)
        ref = [net(x) for x in bags]
    tol = 5e-5 if precision == "fp32" else 3e-2
    sd = {k: v.detach().cpu().double() for k, v in net.state_dict().items()}
    for b, ((c0, y0, a0), (c1, y1, a1)) in enumerate(zip(ref, got)):
        what = (d, h, lam, depth, precision, sizes, b)
        assert c1.shape == c0.shape and y1.shape == y0.shape and a1.shape == a0.shape, what
        assert torch.equal(c0, c1)
        assert (y0 - y1).abs().max().item() <= tol * max(1.0, y0.abs().max().item()), what
        da = (a0 - a1).abs().max().item()
        if depth == 1 or da <= tol:
            assert da <= tol, what
        else:
            # deeper stacks of these random nets (peaked softmaxes) amplify the last-place differences of the first layer: both
            # paths then sit ~1e-4 .. 1e-3 from the fp64 result on A (logits stay within 2e-6) -- the packed path must not be
            # further from the truth than the per-bag one
            _, _, a64, _ = orc.milnet_forward(bags[b][0].cpu().double(), sd, h, "relu", lam, 0.0, depth)
            e_ref = (a0[0].cpu().double() - a64).abs().max().item()
            e_got = (a1[0].cpu().double() - a64).abs().max().item()
            assert e_got <= 3 * max(e_ref, tol), (what, e_got, e_ref)
