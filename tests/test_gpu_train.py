"""Training path on the GPU: gradients / AdamW step against the golden vectors (F3 / F4), trainer classes end to end."""
import argparse

import numpy as np
import pytest
import torch

from tests.helpers import build_amd_milnet, golden_files, load_case

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("path", golden_files("f3_"), ids=lambda p: p.split("/")[-1][:-4])
def test_f3_f4_gradients_and_adamw_step(path):
    z, sd = load_case(path)
    N, D, h, lam, depth, seed = [int(v) for v in z["cfg"]]
    net = build_amd_milnet(D, h, str(z["act"]), lam, float(z["r"]), depth)
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).eval()                       # eval: dropouts off (the fixture was captured that way), grads on
    net.configure(precision="fp32", return_attention=False)
    x = torch.from_numpy(z["x"]).to(DEV)
    y = torch.from_numpy(z["y"]).to(DEV)
    w = torch.tensor(0.5, requires_grad=True, device=DEV)
    crit = torch.nn.BCEWithLogitsLoss()
    np.random.seed(seed)
    ins, logits, A = net(x)
    assert A is None
    max_pred, _ = torch.max(ins, 1)
    loss = w * crit(logits.view(1, -1), y.view(1, -1)) + (1 - w) * crit(max_pred.view(1, -1), y.view(1, -1))
    bag_pred = ((1 - w) * torch.sigmoid(max_pred) + w * torch.sigmoid(logits)).squeeze()
    np.testing.assert_allclose(loss.item(), float(z["loss"]), rtol=0, atol=1e-5)
    np.testing.assert_allclose(bag_pred.item(), float(z["bag_pred"]), rtol=0, atol=1e-5)
    loss.backward()
    np.testing.assert_allclose(w.grad.item(), float(z["w_grad"]), rtol=0, atol=1e-5)
    for k, p in net.named_parameters():
        g = z["grad." + k]
        scale = max(1e-6, float(np.abs(g).max()))
        np.testing.assert_allclose(p.grad.cpu().numpy(), g, rtol=0, atol=3e-4 * scale + 1e-8, err_msg=k)
    opt = torch.optim.AdamW([{"params": w, "lr": 2e-4 * 0.1}, {"params": net.parameters()}], lr=2e-4, betas=(0.5, 0.9),
                            weight_decay=5e-3)
    opt.step()
    for k, p in net.named_parameters():
        # Adam normalises each element's step to ~lr: where the gradient is rounding noise (|g| << max|g|, and the whole
        # key bias, whose true gradient is zero) the step direction is noise too -> allow 2*lr there, 6e-6 elsewhere.
        g = np.abs(z["grad." + k])
        noisy = g < 1e-4 * max(1e-12, g.max())
        if k.endswith("self_attn.linears.1.bias"):
            noisy = np.ones_like(noisy)
        diff = np.abs(p.detach().cpu().numpy() - z["post." + k])
        assert diff[~noisy].max(initial=0.0) < 6e-6, k
        assert diff[noisy].max(initial=0.0) < 4.2e-4, k


def tiny_args(**kw):
    from snuffy_amd.train import get_args_parser
    a = get_args_parser().parse_args([])
    a.feats_size, a.num_heads, a.big_lambda, a.optimizer, a.num_epochs = 64, 2, 16, 'adamw', 2
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def toy_data(n_bags=8, d=64, seed=0):
    g = np.random.RandomState(seed)
    labels, feats = [], []
    for i in range(n_bags):
        y = float(i % 2)
        f = g.randn(int(g.randint(30, 90)), d).astype(np.float32)
        if y:
            f[:5] += 2.0                               # a few "tumour" patches
        labels.append(np.array([y], dtype=np.float32))
        feats.append(f)
    return labels, feats, None, None


def test_snuffy_trainer_epoch_runs_and_learns():
    from snuffy_amd.train import Snuffy
    from snuffy_amd.utils import stage_bags
    torch.manual_seed(0)
    np.random.seed(0)
    tr = Snuffy(tiny_args(lr=2e-3))
    assert str(tr) == 'Snuffy_k16_sa0_depth1'
    data = toy_data()
    first = tr.train(data, 1)['epoch_train_loss']
    for e in range(2, 9):
        last = tr.train(data, e)['epoch_train_loss']
    assert np.isfinite(first) and last < first
    res = tr.valid(data)
    assert res['predictions'].shape == (8, 1) and np.isfinite(res['epoch_valid_loss'])
    assert 0.0 <= float(tr.single_weight_parameter) <= 1.0
    # resident bags give the same forward as host bags
    staged = stage_bags(data[1], DEV)
    res2 = tr.valid((data[0], staged, None, None))
    np.testing.assert_allclose(res2['predictions'], res['predictions'], atol=1e-6)


def test_training_mode_attention_dropout_is_active_like_the_reference():
    """train.py never overrides MultiHeadedAttention's dropout (p = 0.1): train-mode forwards differ run to run."""
    net = build_amd_milnet(64, 2, "relu", 16, 0.0, 1).to(DEV).train().configure(return_attention=False)
    x = torch.randn(1, 200, 64, device=DEV)
    torch.manual_seed(1)
    a = net(x)[1].item()
    torch.manual_seed(2)
    b = net(x)[1].item()
    assert a != b
    net.eval()
    with torch.no_grad():
        assert net(x)[1].item() == net(x)[1].item()


def test_bf16_autocast_training_tracks_fp32():
    """precision='bf16' in training = library GEMMs under autocast; gradients stay close to the fp32 path."""
    z, sd = load_case(golden_files("f3_g_n600")[0])
    N, D, h, lam, depth, seed = [int(v) for v in z["cfg"]]
    grads = {}
    for precision in ("fp32", "bf16"):
        net = build_amd_milnet(D, h, str(z["act"]), lam, float(z["r"]), depth)
        net.load_state_dict(sd, strict=True)
        net = net.to(DEV).eval().configure(precision=precision, return_attention=False)
        x = torch.from_numpy(z["x"]).to(DEV)
        ins, logits, _ = net(x)
        (logits.sum() + ins.max()).backward()
        grads[precision] = {k: p.grad.float().clone() for k, p in net.named_parameters()}
    for k in grads["fp32"]:
        if k.endswith("self_attn.linears.1.bias"):
            continue                                   # mathematically zero gradient: pure rounding noise in both paths
        a, b = grads["bf16"][k].double(), grads["fp32"][k].double()
        rel = float((a - b).norm() / b.norm().clamp_min(1e-12))
        assert rel < 0.1, (k, rel)


@pytest.mark.gpu
def test_multiclass_gradients_match_oracle_autograd():
    """snuffy_multiclass (B = 2, C = 2, depth 2): loss.backward() through the HIP kernels vs autograd through the CPU oracle
    with the same (replayed) random draws -- every parameter gradient."""
    import copy

    import numpy as np
    from oracle import snuffy_oracle as orc
    from snuffy_amd import snuffy_multiclass as smc
    torch.manual_seed(3)
    B, N, D, h, lam, r, depth = 2, 90, 64, 2, 12, 0.5, 2
    attn = smc.MultiHeadedAttention(h, D, dropout=0.0)
    ff = smc.PositionwiseFeedForward(D, D * 4, "gelu", dropout=0.0)
    net = smc.MILNet(smc.FCLayer(D, 2), smc.BClassifier(
        smc.Encoder(smc.EncoderLayer(D, copy.deepcopy(attn), copy.deepcopy(ff), 2, 0.0, lam, r), depth), 2, D))
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    x = torch.randn(B, N, D)
    y = torch.tensor([[1.0, 0.0], [0.0, 1.0]])
    # oracle: forward + backward on the CPU (fp32)
    np.random.seed(11)
    _, logits_ref, _ = orc.milnet_forward_multiclass(x, sd, h, "gelu", lam, r, depth)
    torch.nn.functional.binary_cross_entropy_with_logits(logits_ref, y).backward()
    # ours: same seed -> same np.random.choice draws
    net = net.to(DEV).train().configure(precision="fp32", return_attention=False)
    np.random.seed(11)
    _, logits, _ = net(x.to(DEV))
    torch.nn.functional.binary_cross_entropy_with_logits(logits, y.to(DEV)).backward()
    assert (logits.detach().cpu() - logits_ref.detach()).abs().max() < 1e-4
    checked = 0
    for name, p in net.named_parameters():
        g_ref = sd[name].grad
        if g_ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        denom = max(float(g_ref.abs().max()), 1e-6)
        assert float((p.grad.cpu() - g_ref).abs().max()) / denom < 2e-3, name
        checked += 1
    assert checked >= 20


def _ln_reference(x, dy, gamma, eps, residual):
    x = x.double().requires_grad_(True)
    g = gamma.double().requires_grad_(True)
    b = torch.zeros_like(g).requires_grad_(True)
    y = torch.nn.functional.layer_norm(x, (x.shape[1],), g, b, eps)
    y.backward(dy.double().expand_as(y))
    dx = x.grad + (residual.double() if residual is not None else 0)
    return dx, g.grad, b.grad


@pytest.mark.gpu
@pytest.mark.parametrize("n,d", [(1000, 768), (333, 384), (4096, 64), (77, 100), (5000, 1536)])
def test_layernorm_rows_bwd_kernel(n, d):
    """snf_layernorm_rows_bwd_f32 vs fp64 autograd: per-row dy (f32 / bf16), ONE broadcast row, residual, dgamma / dbeta."""
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(n + d)
    x = (torch.randn(n, d, generator=g) * 2 + 0.5).to(DEV)
    gamma = (1 + 0.3 * torch.randn(d, generator=g)).to(DEV)
    dy = torch.randn(n, d, generator=g).to(DEV)
    res = torch.randn(n, d, generator=g).to(DEV)
    for dyv, resv, tol in ((dy, None, 2e-5), (dy, res, 2e-5), (dy.to(torch.bfloat16), None, 2e-5), (dy[0].contiguous(), res, 2e-5)):
        dx, dxb, dgam, dbet = ops.layernorm_rows_bwd(x, dyv, gamma, 1e-5, residual=resv, want_dx_bf16=True)
        rdx, rg, rb = _ln_reference(x, dyv.float() if dyv.dim() == 2 else dyv.float().unsqueeze(0), gamma, 1e-5, resv)
        scale = rdx.abs().max().item()
        assert (dx.double() - rdx).abs().max().item() <= tol * scale
        assert torch.equal(dxb, dx.to(torch.bfloat16))
        assert (dgam.double() - rg).abs().max().item() <= 1e-4 * max(1.0, rg.abs().max().item())
        assert (dbet.double() - rb).abs().max().item() <= 1e-4 * max(1.0, rb.abs().max().item())
    # no gamma, no partial sums
    dx, _, a, b = ops.layernorm_rows_bwd(x, dy, None, 1e-5, want_param_grads=False)
    rdx, _, _ = _ln_reference(x, dy, torch.ones(d, device=DEV), 1e-5, None)
    assert a is None and b is None and (dx.double() - rdx).abs().max().item() <= 2e-5 * rdx.abs().max().item()


@pytest.mark.gpu
def test_head_function_gradients():
    """HeadFn (Linear(mean_n LayerNorm(z)), snuffy.py:86,71) vs fp64 autograd: dz, dgamma, dbeta, dW, db."""
    from snuffy_amd import autograd as SA
    torch.manual_seed(3)
    n, d, c = 3000, 384, 2
    z = (torch.randn(n, d, device=DEV) * 1.5).requires_grad_(True)
    norm = torch.nn.LayerNorm(d).to(DEV)
    lin = torch.nn.Linear(d, c).to(DEV)
    with torch.no_grad():
        norm.weight.add_(0.3 * torch.randn(d, device=DEV))
        norm.bias.add_(0.2 * torch.randn(d, device=DEV))
    w_out = torch.tensor([1.5, -0.7], device=DEV)
    (SA.head_train(z, norm, lin) * w_out).sum().backward()
    got = [z.grad.clone(), norm.weight.grad.clone(), norm.bias.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()]
    zd = z.detach().double().requires_grad_(True)
    nd, ld = torch.nn.LayerNorm(d).to(DEV).double(), torch.nn.Linear(d, c).to(DEV).double()
    nd.load_state_dict({k: v.double() for k, v in norm.state_dict().items()})
    ld.load_state_dict({k: v.double() for k, v in lin.state_dict().items()})
    (ld(nd(zd).mean(0)) * w_out.double()).sum().backward()
    want = [zd.grad, nd.weight.grad, nd.bias.grad, ld.weight.grad, ld.bias.grad]
    for a, b in zip(got, want):
        assert (a.double() - b).abs().max().item() <= 2e-5 * max(b.abs().max().item(), 1e-30), (a.shape,)


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,h,lam", [(700, 256, 4, 50), (3000, 384, 6, 200), (2048, 768, 6, 200)])
def test_fused_bf16_layer0_training_matches_generic_and_fp32(n, d, h, lam, monkeypatch):
    """EncoderLayer0Bf16Fn (hand-ordered forward / backward of the first layer, bf16) against the generic autograd chain of
    the same precision and against the fp32 training path: every parameter gradient, eval mode and train mode (the
    in-kernel dropout mask is a function of (seed, offset), so both bf16 chains see the same mask)."""
    from snuffy_amd import autograd as SA
    torch.manual_seed(n)
    ref = build_amd_milnet(d, h, "relu", lam, 0.0, 1)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.add_(0.3 * torch.randn(d))
                m.bias.add_(0.2 * torch.randn(d))
            if isinstance(m, torch.nn.Linear) and m.bias is not None:
                m.bias.add_(0.1 * torch.randn_like(m.bias))
    sd = ref.state_dict()
    x = torch.randn(1, n, d, device=DEV)
    for train_mode in (False, True):
        grads, outs = {}, {}
        for tag, precision, fused in (("fp32", "fp32", True), ("generic", "bf16", False), ("fused", "bf16", True)):
            if train_mode and tag == "fp32":
                continue                           # the fp32 chain draws its mask for a materialised P: same stream, other layout
            monkeypatch.setattr(SA, "FUSED_BF16_TRAINING", fused)
            net = build_amd_milnet(d, h, "relu", lam, 0.0, 1)
            net.load_state_dict(sd, strict=True)
            net = net.to(DEV).configure(precision=precision, return_attention=False)
            net.train(train_mode)
            torch.manual_seed(11)
            ins, logits, _ = net(x)
            (logits.sum() * 3 + ins.max()).backward()
            grads[tag] = {k: p.grad.float().clone() for k, p in net.named_parameters()}
            outs[tag] = logits.detach().clone()
        base = "generic"
        assert (outs["fused"] - outs[base]).abs().max().item() <= 2e-2 * max(1.0, outs[base].abs().max().item())
        for k in grads["fused"]:
            if k.endswith("self_attn.linears.1.bias"):
                continue                           # mathematically zero gradient
            a = grads["fused"][k].double()
            for other, bound in (("generic", 0.08), ("fp32", 0.1)):
                if other not in grads:
                    continue
                b = grads[other][k].double()
                rel = float((a - b).norm() / b.norm().clamp_min(1e-12))
                assert rel < bound, (train_mode, k, other, rel)


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,h,lam,share", [(700, 256, 4, 50, 0.0), (3000, 384, 6, 200, 0.0), (2048, 768, 6, 200, 0.0), (4100, 384, 4, 300, 0.5),
                                             (16384, 768, 6, 200, 0.0)])
def test_fused_x3_layer0_training_matches_the_generic_fp32_chain(n, d, h, lam, share, monkeypatch):
    """EncoderLayer0X3Fn (round 5: hand-ordered forward / backward of the first layer in fp32-class arithmetic) against the generic
    autograd chain of the same precision (LinearX3Fn + LayerNormRowsFn + SparseAttnFn): logits and every parameter gradient, eval
    mode and train mode (all draw the same Philox mask tensor for the attention dropout of reference snuffy.py:166-167), and against
    the generic chain with fp32 library GEMMs (FP32_GEMM = "library")."""
    from snuffy_amd import autograd as SA
    from snuffy_amd import functional as SF
    torch.manual_seed(n)
    ref = build_amd_milnet(d, h, "relu", lam, share, 1)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.add_(0.3 * torch.randn(d))
                m.bias.add_(0.2 * torch.randn(d))
            if isinstance(m, torch.nn.Linear) and m.bias is not None:
                m.bias.add_(0.1 * torch.randn_like(m.bias))
    sd = ref.state_dict()
    x = torch.randn(1, n, d, device=DEV)
    for train_mode in (False, True):
        grads, outs = {}, {}
        # (16384, 768): the bag is large enough for the one-pass chain (hl images, gemm_hl, gemm_tn: round 6) -- "fused" is that chain
        # there, "fused_cat" the concatenated-K chain of round 5 on the same bag
        runs = [("library", False, "library", True), ("generic", False, "x3", True), ("fused", True, "x3", True)]
        if n >= 16384:
            monkeypatch.setattr(SA, "X3_TRAIN_HL", True)
            assert SA._x3_train_hl_ok(n, d, 4 * d)
            runs.append(("fused_cat", True, "x3", False))
        for tag, fused, gemm, hl_chain in runs:
            monkeypatch.setattr(SA, "FUSED_X3_TRAINING", fused)
            monkeypatch.setattr(SA, "X3_TRAIN_HL", hl_chain)
            monkeypatch.setattr(SF, "FP32_GEMM", gemm)
            net = build_amd_milnet(d, h, "relu", lam, share, 1)
            net.load_state_dict(sd, strict=True)
            net = net.to(DEV).configure(precision="fp32", return_attention=False)
            net.train(train_mode)
            torch.manual_seed(11)
            np.random.seed(5)
            ins, logits, _ = net(x)
            (logits.sum() * 3 + ins.max()).backward()
            grads[tag] = {k: p.grad.float().clone() for k, p in net.named_parameters()}
            outs[tag] = logits.detach().clone()
        # A ReLU gate whose pre-activation is within rounding of zero may open in one arithmetic and not in the other (0 - 3 of the
        # N x F hidden elements here): one such element moves the FFN-in gradients by ~1e-4 of their norm at these bag sizes.
        # Such an element in one of the K selected rows also moves everything upstream of LayerNorm 1 of those rows: ~3e-5 of the
        # V / output projection gradients and, the query / key gradients being 100 - 1000x smaller in norm, ~2e-3 of those -- in
        # either chain (measured: fused 2e-3 / generic 7e-6 at (2048, 768), fused 3e-6 / generic 5e-4 at (4100, 384)).
        gate_keys = ("feed_forward.w_1.weight", "feed_forward.w_1.bias", "sublayer.1.norm.weight", "sublayer.1.norm.bias")
        small_keys = ("linears.0.weight", "linears.0.bias", "linears.1.weight", "sublayer.0.norm.weight")
        others = [("library", 5e-5, 2e-4), ("generic", 5e-5, 5e-4 if train_mode else 2e-4)]
        if "fused_cat" in outs:
            others.append(("fused_cat", 5e-5, 2e-4))
        for other, bound_out, bound in others:
            assert (outs["fused"] - outs[other]).abs().max().item() <= bound_out * max(1.0, outs[other].abs().max().item())
            for k in grads["fused"]:
                if k.endswith("self_attn.linears.1.bias"):
                    continue                           # mathematically zero gradient
                a, b = grads["fused"][k].double(), grads[other][k].double()
                rel = float((a - b).norm() / b.norm().clamp_min(1e-12))
                assert rel < (5e-3 if k.endswith(small_keys) else 3e-3 if k.endswith(gate_keys) else bound), (train_mode, k, other, rel)


@pytest.mark.gpu
@pytest.mark.parametrize("n,d", [(1000, 768), (4097, 3072), (33, 8), (700, 1536), (5000, 4096)])
def test_colsum_fused_kernel(n, d):
    """snf_colsum_fused: plain / weighted / gated column sums, the bf16 copy, the in-place form (vs torch in fp64)."""
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(n + d)
    x = torch.randn(n, d, generator=g).to(DEV)
    w2 = torch.randn(n, 2, generator=g).to(DEV)
    gate = torch.randn(n, d, generator=g).to(DEV).to(torch.bfloat16)
    gate[0, :8] = 0.0                                                  # a zero gate closes (ReLU' at 0 is 0, as threshold_backward)
    tol = lambda ref: 2e-5 * max(1.0, ref.abs().max().item()) * (n ** 0.5)
    s, c = ops.colsum_fused(x)
    assert c is None and (s.double() - x.double().sum(0)).abs().max().item() <= tol(x.double().sum(0))
    s, c = ops.colsum_fused(x, want_bf16=True)
    assert torch.equal(c, x.to(torch.bfloat16)) and (s.double() - c.double().sum(0)).abs().max().item() <= tol(c.double().sum(0))
    s, _ = ops.colsum_fused(x, row_weight=w2[:, 1])                    # strided weight column
    ref = (x.double() * w2[:, 1:2].double()).sum(0)
    assert (s.double() - ref).abs().max().item() <= tol(ref)
    xb = x.to(torch.bfloat16)
    want = torch.ops.aten.threshold_backward(xb, gate, 0)
    s, c = ops.colsum_fused(xb.clone(), gate=gate, want_bf16=True)
    assert torch.equal(c, want) and (s.double() - want.double().sum(0)).abs().max().item() <= tol(want.double().sum(0))
    y = xb.clone()
    s2, c2 = ops.colsum_fused(y, gate=gate, inplace=True)
    assert c2 is y and torch.equal(y, want) and torch.equal(s2, s)


@pytest.mark.gpu
@pytest.mark.parametrize("m,k", [(1000, 768), (4097, 3072), (33, 8), (700, 1536), (300, 8192)])
def test_split3_colsum_kernel(m, k):
    """snf_split3_colsum_f32 (round 5): split image + ReLU gate + column sums of a gradient matrix in one pass -- bit-exact against
    snf_split3_f32 of the gated matrix; sums against fp64; strided input / gate views; two matrices sharing one wider image."""
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(m + k)
    xw = torch.randn(m, k + 8, generator=g).to(DEV)
    x = xw[:, 8:] if k % 4 == 0 and (8 * 4) % 16 == 0 else xw[:, :k]
    act = torch.relu(torch.randn(m, k, generator=g)).to(DEV)
    act3 = ops.split3_rows(act)
    gate = act3[:, k:2 * k]
    img, cs = ops.split3_colsum(x)
    assert torch.equal(img, ops.split3_rows(x.contiguous()))
    assert (cs.double().cpu() - x.double().sum(0).cpu()).abs().max() <= 1e-5 * max(1.0, float(x.abs().sum(0).max()))
    xg = x * (gate > 0)
    img, cs = ops.split3_colsum(x, gate=gate)
    assert torch.equal(img, ops.split3_rows(xg.contiguous()))
    assert (cs.double().cpu() - xg.double().sum(0).cpu()).abs().max() <= 1e-5 * max(1.0, float(xg.abs().sum(0).max()))
    y = torch.randn(m, k, generator=g).to(DEV)
    wide = torch.full((m, 6 * k), 7.0, dtype=torch.bfloat16, device=DEV)
    _, none = ops.split3_colsum(x, out=wide, col=0, want_colsum=False)
    assert none is None
    ops.split3_colsum(y, out=wide, col=k)
    assert torch.equal(wide, ops.split3_rows(torch.cat([x, y], dim=1)))


@pytest.mark.gpu
@pytest.mark.parametrize("r,c", [(768, 768), (3072, 768), (100, 384), (33, 100), (1536, 2048)])
def test_fold_and_unfold_linear_kernels(r, c):
    """snf_fold_linear_f32 / snf_unfold_linear_f32 vs the torch expressions they replace (fp64)."""
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(r + c)
    w = (torch.randn(r, c, generator=g) / c ** 0.5).to(DEV)
    gam, bet = (1 + 0.3 * torch.randn(c, generator=g)).to(DEV), (0.2 * torch.randn(c, generator=g)).to(DEV)
    bias = torch.randn(r, generator=g).to(DEV)
    big = torch.zeros(r + 5, c + 8, dtype=torch.bfloat16, device=DEV)
    wf = big[3:3 + r, :c]                                             # a strided destination (rows of a larger buffer)
    bf, bfh = torch.empty(r, device=DEV), torch.empty(r, dtype=torch.bfloat16, device=DEV)
    ops.fold_linear(w, gam, bet, bias, wf, bf, bfh)
    assert torch.equal(wf, (w * gam).to(torch.bfloat16))
    ref_b = w.double() @ bet.double() + bias.double()
    assert (bf.double() - ref_b).abs().max().item() <= 2e-6 * max(1.0, ref_b.abs().max().item())
    assert torch.equal(bfh, bf.to(torch.bfloat16))
    assert float(big[:3].abs().sum()) == 0 and float(big[3 + r:].abs().sum()) == 0 and float(big[:, c:].abs().sum()) == 0
    ops.fold_linear(w, gam, bet, None, wf)                            # no bias, no bias outputs
    dwf = torch.randn(r, c, generator=g).to(DEV)
    dbf = torch.randn(r, generator=g).to(DEV)
    dw, dgam, dbet = ops.unfold_linear(dwf, w, gam, bet, dbf)
    ref_dw = dwf.double() * gam.double() + torch.outer(dbf.double(), bet.double())
    ref_dg = (dwf.double() * w.double()).sum(0)
    ref_db = dbf.double() @ w.double()
    assert (dw.double() - ref_dw).abs().max().item() <= 2e-6 * ref_dw.abs().max().item()
    assert (dgam.double() - ref_dg).abs().max().item() <= 1e-5 * max(1.0, ref_dg.abs().max().item())
    assert (dbet.double() - ref_db).abs().max().item() <= 1e-5 * max(1.0, ref_db.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 2])
def test_snuffy_multiclass_trainer_run_model_and_step(B):
    """train.SnuffyMulticlass (reference train.py:922-983) end to end: the trainer's own constructor builds the multi-class
    MILNet, `_run_model` returns (bag_prediction, loss, sigmoid(c)) and one `loss.backward()` +
    `_after_run_model_in_training_mode` step moves the weights -- all against the CPU oracle (same replayed random draws, oracle
    autograd, torch's Adam on the CPU)."""
    import numpy as np
    from oracle import snuffy_oracle as orc
    from snuffy_amd import train
    args = train.get_args_parser().parse_args([])
    args.arch, args.num_classes, args.feats_size, args.num_heads = "snuffy_multiclass", 2, 64, 2
    args.big_lambda, args.random_patch_share, args.depth, args.activation = 12, 0.5, 1, "relu"
    args.optimizer, args.lr, args.weight_decay, args.soft_average = "adam", 1e-3, 5e-3, 1
    torch.manual_seed(5)
    tr = train.ARCH_REGISTRY["snuffy_multiclass"](args)
    assert isinstance(tr, train.SnuffyMulticlass) and str(tr) == "Snuffy_Multiclass_k12_sa1_depth1"
    for layer in tr.milnet.b_classifier.encoder.layers:
        layer.self_attn.dropout.p = 0.0              # the reference leaves p = 0.1 on in train mode: not reproducible across devices
        layer.feed_forward.dropout.p = 0.0           # ... and the multi-class FFN's default 0.1 as well (train.py:935-939)
    N = 70
    x = torch.randn(B, N, 64)
    y = torch.tensor([[1.0, 0.0], [0.0, 1.0]][:B])
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in tr.milnet.state_dict().items()}
    w_ref = torch.tensor(0.5, requires_grad=True)
    # ---- oracle: forward, the trainer's loss mix (train.py:828-846), backward, Adam
    np.random.seed(21)
    ins_ref, logits_ref, _ = orc.milnet_forward_multiclass(x, sd, 2, "relu", 12, 0.5, 1)
    max_ref, _ = torch.max(ins_ref, 1)
    bce = torch.nn.BCEWithLogitsLoss()
    loss_ref = w_ref * bce(logits_ref.view(1, -1), y.view(1, -1)) + (1 - w_ref) * bce(max_ref.view(1, -1), y.view(1, -1))
    pred_ref = ((1 - w_ref) * torch.sigmoid(max_ref) + w_ref * torch.sigmoid(logits_ref)).detach().squeeze()
    loss_ref.backward()
    names = [k for k, _ in tr.milnet.named_parameters()]
    # what Adam normalises is g + weight_decay * w (L2 form, torch.optim.Adam): kept to tell apart the entries where it nearly cancels
    grads_ref = {k: (sd[k].grad + args.weight_decay * sd[k]).detach().clone() for k in names}
    opt_ref = torch.optim.Adam([{"params": [w_ref], "lr": args.lr * args.single_weight__lr_multiplier},
                                {"params": [sd[k] for k in names]}], lr=args.lr, betas=(0.5, 0.9), weight_decay=args.weight_decay)
    opt_ref.step()
    # ---- the trainer on the GPU
    tr.milnet.train()
    np.random.seed(21)
    pred, loss, ins = tr._run_model(x.to(DEV), y.to(DEV))
    assert ins.shape == (B * N * 2, 1)
    assert float((ins.detach().cpu().view(-1) - torch.sigmoid(ins_ref.detach()).reshape(-1)).abs().max()) < 1e-5
    assert abs(float(loss) - float(loss_ref)) < 1e-5
    assert float((pred.cpu().reshape(-1) - pred_ref.reshape(-1)).abs().max()) < 1e-5
    loss.backward()
    tr._after_run_model_in_training_mode(step=0, num_bags=1, batch_idx=0)
    for k, p in tr.milnet.named_parameters():
        ref = sd[k].detach()
        # the key bias has a mathematically zero gradient (softmax is shift-invariant): its Adam step is +-lr of rounding noise
        tol = 2.1 * args.lr if k.endswith("self_attn.linears.1.bias") else 2e-5 * max(1.0, float(ref.abs().max()))
        # Adam's first step is lr * g' / (|g'| + 1e-8) with g' = g + weight_decay * w: an entry where g' is of the order of eps moves by
        # a fraction of lr that the last bits of g decide (the summation order of a kernel) -- those entries are held to the step size,
        # the rest to 2e-5
        tiny = grads_ref[k].abs() < 1e-6
        diff = (p.detach().cpu() - ref).abs()
        assert float(diff[~tiny].max() if (~tiny).any() else 0.0) <= tol, k
        assert float(diff[tiny].max() if tiny.any() else 0.0) <= 2.1 * args.lr, k
    assert abs(float(tr.single_weight_parameter) - float(w_ref.detach().clamp(0, 1))) < 1e-6


@pytest.mark.gpu
def test_runner_epoch_loop_checkpoints_and_test_pass(tmp_path):
    """train.Runner / main (reference train.py:682-794, 1004-1039): validation before epoch 1, train / valid / scheduler step per
    epoch, `{epoch}.pth` + `thresholds_{epoch}.txt` + `single_weight_parameter_{epoch}` every epoch, best-AUC bookkeeping, the test
    pass with the stored thresholds, clean-up of the other epochs; the saved state dict reloads into a fresh MILNet bit for bit."""
    import json
    import numpy as np
    from snuffy_amd import train
    rs = np.random.RandomState(0)

    def split(n):
        labels = [np.array([float(i % 2)], dtype=np.float32) for i in range(n)]
        feats = [(rs.randn(40 + 3 * i, 32) + (1.5 if i % 2 else 0.0)).astype(np.float32) for i in range(n)]
        return labels, feats, [None] * n, [None] * n
    data = (split(12), split(6), split(6))
    np.random.seed(0)
    torch.manual_seed(0)
    argv = ["--dataset", "synthetic", "--feats_size", "32", "--num_heads", "2", "--big_lambda", "10", "--num_epochs", "3",
            "--lr", "2e-3", "--optimizer", "adamw", "--save_path", str(tmp_path / "run")]
    res = train.main(argv, data=data)
    assert set(res) == {"best_auc", "last_epoch"} and "last_epoch_loss" in res["last_epoch"]
    run = tmp_path / "run"
    with open(run / "train_metrics.json") as f:
        tm = json.load(f)
    assert tm["best_auc_epochs"] and 0.0 <= tm["best_auc"] <= 1.0
    keep = {min(tm["best_auc_epochs"]), 3}
    for epoch in (1, 2, 3):
        assert (run / f"{epoch}.pth").exists() == (epoch in keep) and (run / f"thresholds_{epoch}.txt").exists() == (epoch in keep)
    sd = torch.load(run / "3.pth", map_location="cpu")
    from snuffy_amd.snuffy import build_milnet
    twin = build_milnet(32, 2, "relu", 10, 0.0, 1)
    twin.load_state_dict(sd, strict=True)                       # the reference's key names, strict
    w = torch.load(run / "single_weight_parameter_3", map_location="cpu")
    assert 0.0 <= float(w) <= 1.0


@pytest.mark.parametrize("n,k,m", [(4096, 768, 1536), (1000, 384, 384), (8200, 256, 1024)])
def test_linear_x3_fn_forward_and_gradients_vs_fp64(n, k, m):
    """autograd.LinearX3Fn (fp32-class linear of the fp32 training path): y, dx, dW, db against fp64 -- products to 2^-17, fp32
    accumulation over up to 8 k rows; dW as three block GEMMs over the images' hi / lo column blocks."""
    from snuffy_amd import autograd as SA
    g = torch.Generator().manual_seed(21)
    x = torch.randn(n, k, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(m, k, generator=g) / k ** 0.5).to(DEV).requires_grad_(True)
    b = torch.randn(m, generator=g).to(DEV).requires_grad_(True)
    dy = torch.randn(n, m, generator=g).to(DEV)
    assert SA.linear_x3_ok(x, w)
    y = SA.LinearX3Fn.apply(x, w, b)
    y.backward(dy)
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yd = torch.nn.functional.linear(xd, wd, bd)
    yd.backward(dy.double())
    for name, got, ref in (("y", y, yd), ("dx", x.grad, xd.grad), ("dw", w.grad, wd.grad), ("db", b.grad, bd.grad)):
        err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 3e-5, (name, err)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_forward_after_fused_optimizer_steps_uses_the_new_weights(precision):
    """torch's fused AdamW (what the trainers use on the GPU) does not bump the parameters' version counters, which the folded /
    split / bf16 weight caches and the captured graphs key on: every optimizer step advances SF.param_key's epoch instead.  Losses of
    consecutive steps -- and an eval forward afterwards -- must equal those of a run that drops every cache after every step."""
    import torch.nn as nn
    from snuffy_amd.train import BagParallelStepper
    from tests.helpers import build_amd_milnet

    def run(inval):
        torch.manual_seed(0)
        net = build_amd_milnet(384, 6, "relu", 200, 0.0, 1).to(DEV)
        for m in net.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
        g = torch.Generator().manual_seed(42)
        bags = [torch.randn(1, 4096, 384, generator=g).to(DEV) for _ in range(2)]
        labels = [torch.tensor([float(i % 2)], device=DEV) for i in range(2)]
        st = BagParallelStepper(net, world_size=1, dist=None, device=DEV, precision=precision)
        v0 = net.b_classifier.encoder.layers[0].feed_forward.w_1.weight._version
        losses = []
        for s in range(4):
            losses.append(float(st.step(bags[s % 2], labels[s % 2])))
            if inval:
                net.invalidate()
        net.eval()
        with torch.no_grad():
            ev = net(bags[0])[1].clone()
        return losses, ev, net.b_classifier.encoder.layers[0].feed_forward.w_1.weight._version - v0

    la, ea, dv = run(False)
    lb, eb, _ = run(True)
    assert la == lb and torch.equal(ea, eb), (la, lb, dv)
    assert abs(la[2] - la[0]) > 1e-3      # the weights did move between the two sightings of bag 0



@pytest.mark.gpu
@pytest.mark.parametrize("n,c,pw,three_d", [(32768, 1, False, True), (5000, 2, True, False), (777, 5, True, True), (1, 1, False, False)])
def test_fused_mil_loss_head_matches_the_torch_formulation(n, c, pw, three_d):
    """autograd.MilLossFn (snf_mil_loss_f32: max over the instance scores, both BCEWithLogits terms, their mix by the single weight, the bag
    prediction -- one launch each way) against the reference formulation of train.py's _run_model: loss, bag prediction, every gradient."""
    from snuffy_amd import autograd as SA
    g = torch.Generator().manual_seed(n + c)
    ins0 = torch.randn(n, c, generator=g).to(DEV)
    logit0 = torch.randn(1, c, generator=g).to(DEV)
    label = (torch.rand(c, generator=g) > 0.5).float().to(DEV)
    # (positional = `weight`, as the reference constructs it; the second case also carries a pos_weight)
    crit = torch.nn.BCEWithLogitsLoss(torch.rand(c, generator=g).to(DEV) * 3 + 0.2 if pw else None,
                                      pos_weight=torch.rand(c, generator=g).to(DEV) * 2 + 0.5 if (pw and c == 2) else None)
    res = {}
    for tag in ("ref", "fused"):
        ins = (ins0.view(1, n, c) if three_d else ins0).clone().requires_grad_(True)
        logits = logit0.clone().requires_grad_(True)
        w = torch.tensor(0.3, device=DEV, requires_grad=True)
        if tag == "fused":
            loss, bag_pred = SA.mil_loss(ins, logits, label, w, crit)
        else:
            mx, _ = torch.max(ins, 1 if three_d else 0)
            loss = w * crit(logits.view(1, -1), label.view(1, -1)) + (1 - w) * crit(mx.view(1, -1), label.view(1, -1))
            bag_pred = ((1 - w) * torch.sigmoid(mx) + w * torch.sigmoid(logits)).detach().reshape(-1)
        (loss * 1.7).backward()
        res[tag] = (loss.detach(), bag_pred.detach().reshape(-1), ins.grad.reshape(n, c), logits.grad.reshape(-1), w.grad)
    for a, b in zip(res["fused"], res["ref"]):
        assert a.shape == b.shape and torch.allclose(a, b, rtol=2e-6, atol=2e-7), (a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,h,dk", [(32768, 200, 6, 128), (3000, 128, 4, 64), (1000, 516, 2, 96)])
def test_exact_attention_backward_regenerates_the_dropout_mask(n, k, h, dk):
    """snf_sparse_attn_bwd_dropout_f32: the backward kernels regenerate the forward's Philox mask instead of reading the [h, n, k] tensor --
    bit-identical gradients to the mask-tensor route."""
    from snuffy_amd import ops
    assert ops.attn_bwd_dropout_supported(k, dk)
    g = torch.Generator().manual_seed(n + k)
    d = h * dk
    q, v = torch.randn(n, d, generator=g).to(DEV), torch.randn(n, d, generator=g).to(DEV)
    kp = (torch.randn(k, d, generator=g) / dk ** 0.5).to(DEV)
    p = torch.softmax(torch.randn(h, n, k, generator=g), -1).to(DEV)
    dout = torch.randn(k, d, generator=g).to(DEV)
    drop = (0.1, 1234567, 42)
    mask = ops.dropout_mask(h, n, k, *drop, DEV)
    ref = ops.sparse_attn_bwd(q, kp, v, p, dout, h, mask=mask)
    out = ops.sparse_attn_bwd(q, kp, v, p, dout, h, dropout=drop)
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    plain = ops.sparse_attn_bwd(q, kp, v, p, dout, h)
    assert not torch.equal(plain[0], out[0])
