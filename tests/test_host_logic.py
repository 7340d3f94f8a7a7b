"""Host-side logic of the drop-in (no GPU): constructor / state-dict compatibility with the reference's checkpoints,
the k1/k2 arithmetic, and the random-share draw on the global numpy RNG."""
import inspect

import numpy as np
import pytest
import torch

from tests.helpers import build_amd_milnet, golden_files, load_case


def test_state_dict_keys_and_shapes_match_reference_checkpoints():
    for path in golden_files("f1_"):
        z, sd = load_case(path)
        N, D, h, lam, depth, seed = [int(v) for v in z["cfg"]]
        net = build_amd_milnet(D, h, str(z["act"]), lam, float(z["r"]), depth)
        own = net.state_dict()
        assert list(own.keys()) == list(sd.keys())           # same names, same ORDER (positional loaders rely on it)
        for k in own:
            assert tuple(own[k].shape) == tuple(sd[k].shape), k
        net.load_state_dict(sd, strict=True)


def test_constructor_signatures_match_reference():
    from snuffy_amd import snuffy
    want = {
        "FCLayer": ["in_size", "out_size"],
        "IClassifier": ["feature_extractor", "feature_size", "output_class"],
        "BClassifier": ["encoder", "num_classes", "input_size"],
        "Encoder": ["layer", "N"],
        "SublayerConnection": ["size", "dropout"],
        "EncoderLayer": ["size", "self_attn", "feed_forward", "dropout", "big_lambda", "random_patch_share"],
        "MultiHeadedAttention": ["h", "d_model", "dropout"],
        "PositionwiseFeedForward": ["d_model", "d_ff", "activation", "dropout"],
        "MILNet": ["i_classifier", "b_classifier"],
    }
    for cls, params in want.items():
        got = list(inspect.signature(getattr(snuffy, cls).__init__).parameters)[1:]
        assert got == params, (cls, got)
    assert snuffy.MultiHeadedAttention(2, 8).dropout.p == 0.1        # reference default, train.py never overrides it
    with pytest.raises(AssertionError):
        snuffy.MultiHeadedAttention(3, 8)                             # d_model % h (snuffy.py:176)
    with pytest.raises(KeyError):
        snuffy.PositionwiseFeedForward(8, 32, "swish")                # unknown activation (snuffy.py:215-221)
    assert list(inspect.signature(snuffy.attention).parameters) == ["query", "key", "value", "dropout"]


@pytest.mark.parametrize("path", golden_files("f1_"), ids=lambda p: p.split("/")[-1][:-4])
def test_random_share_draw_replays_reference(path):
    """Given the reference's top indices, EncoderLayer.select() reproduces its np.random.choice draws layer by layer."""
    z, sd = load_case(path)
    N, D, h, lam, depth, seed = [int(v) for v in z["cfg"]]
    net = build_amd_milnet(D, h, str(z["act"]), lam, float(z["r"]), depth)
    top = torch.from_numpy(z["top"])
    np.random.seed(seed)
    for l, layer in enumerate(net.b_classifier.encoder.layers):
        t, rnd = layer.select(None, N, top)
        assert t is top
        if int(z["n_rnd"]):
            assert np.array_equal(rnd.numpy(), z[f"rnd{l}"])
        else:
            assert rnd is None


def test_k_split_table():
    from snuffy_amd import snuffy
    z = np.load(golden_files("f2_")[0])
    import math
    for lam, r, n, k1, k2 in z["k_table"]:
        layer = snuffy.EncoderLayer(8, snuffy.MultiHeadedAttention(2, 8), snuffy.PositionwiseFeedForward(8, 32, "relu"),
                                    0.0, int(lam), float(r))
        assert min(math.ceil(layer.big_lambda * layer.top_big_lambda_share), int(n)) == int(k1)
        top = torch.arange(int(k1))
        np.random.seed(0)
        _, rnd = layer.select(None, int(n), top)
        assert (0 if rnd is None else rnd.numel()) == int(k2)
        if rnd is not None:
            assert len(set(rnd.tolist()) & set(top.tolist())) == 0 and len(set(rnd.tolist())) == int(k2)


def test_configure_is_shared_and_validated():
    net = build_amd_milnet(64, 2, "relu", 10, 0.0, 3)
    net.configure(precision="bf16", return_attention=False)
    for m in net.b_classifier.modules():
        if hasattr(m, "cfg"):
            assert m.cfg is net.b_classifier.cfg
    assert net.b_classifier.encoder.layers[2].cfg.precision == "bf16"
    with pytest.raises(ValueError):
        net.configure(precision="fp8")
    import copy
    net2 = copy.deepcopy(net)
    assert net2.b_classifier.encoder.layers[0].cfg is net2.b_classifier.cfg
    assert net2.b_classifier.cfg is not net.b_classifier.cfg


def test_bf16_setting_runs_deep_stacks_on_the_fp32_class_kernels(monkeypatch):
    """RuntimeConfig.compute: precision="bf16" is literal for ONE encoder layer (the benchmark model) and means the fp32-class kernels
    for deeper stacks (reference roi.py:318-339 builds depth 5): the 1e-2 gate is flat, not scaled with depth."""
    import copy

    from snuffy_amd import functional as SF
    from snuffy_amd import snuffy_multiclass as smc
    one, deep = build_amd_milnet(64, 2, "relu", 10, 0.0, 1), build_amd_milnet(64, 2, "relu", 10, 0.0, 5)
    for net in (one, deep):
        net.configure(precision="bf16")
    assert one.b_classifier.cfg.compute == "bf16" and deep.b_classifier.cfg.compute == "fp32"
    assert deep.b_classifier.cfg.precision == "bf16"                       # the user's setting is kept as given
    assert copy.deepcopy(deep).b_classifier.cfg.compute == "fp32"
    assert deep.b_classifier.encoder.layers[3].cfg.compute == "fp32"       # every module of the stack sees the same answer
    deep.configure(precision="fp32")
    assert deep.b_classifier.cfg.compute == "fp32"
    deep.configure(precision="bf16")
    monkeypatch.setattr(SF, "BF16_DEEP_STACKS", "bf16")
    assert deep.b_classifier.cfg.compute == "bf16"
    monkeypatch.setattr(SF, "BF16_DEEP_STACKS", "fp32")
    mc = smc.build_milnet(64, 2, "relu", 10, 0.0, 2, 2) if hasattr(smc, "build_milnet") else None
    if mc is not None:
        mc.configure(precision="bf16")
        assert mc.b_classifier.cfg.compute == "fp32"


def test_forward_refuses_cpu_tensors():
    from snuffy_amd import SnuffyHipError
    net = build_amd_milnet(64, 2, "relu", 10, 0.0, 1)
    with pytest.raises(SnuffyHipError):
        net(torch.zeros(1, 5, 64))


@pytest.mark.parametrize("camelyon16", [False, True])
def test_binary_sidecar_loads_exactly_what_the_csv_parses(tmp_path, camelyon16):
    """the .npz twin of a feature CSV must give get_bag_feats the same rows in the same shuffled order, the
    same '%.4f'-rounded float32 values, the same label / position columns, and leave the global numpy RNG in the same state."""
    import types

    import pandas as pd
    from snuffy_amd import compute_feats, utils
    rng = np.random.RandomState(3)
    n, d = 257, 24
    feats = (rng.randn(n, d) * 3).astype(np.float32)
    feats[0, 0], feats[1, 1] = 0.00005, -1.23455          # half-way cases of the 4-decimal rounding
    labels = rng.randint(0, 2, n).astype(np.float64) if camelyon16 else None
    positions = ["(%d, %d)" % (i % 17, i // 17) for i in range(n)] if camelyon16 else None
    csv = str(tmp_path / "slide_a.csv")
    compute_feats.write_bag_csv(csv, feats, labels, positions, camelyon16=camelyon16, sidecar=True)
    assert (tmp_path / "slide_a.csv.npz").exists()
    args = types.SimpleNamespace(num_classes=1)
    row = pd.Series([csv, 1])

    np.random.seed(11)
    lab_b, f_b, fl_b, pos_b = utils.get_bag_feats(row, args)          # binary twin
    after_b = np.random.rand()
    (tmp_path / "slide_a.csv.npz").unlink()
    np.random.seed(11)
    lab_c, f_c, fl_c, pos_c = utils.get_bag_feats(row, args)          # text path (the reference's)
    after_c = np.random.rand()

    assert f_b.dtype == np.float32 and f_b.shape == (n, d)
    assert np.array_equal(f_b, f_c) and np.array_equal(lab_b, lab_c)
    assert after_b == after_c
    if camelyon16:
        assert np.array_equal(fl_b, fl_c) and list(pos_b) == list(pos_c)
    else:
        assert fl_b is None and pos_b is None and fl_c is None and pos_c is None
    # a CSV rewritten after its twin invalidates the twin (mtime rule)
    compute_feats.write_bag_sidecar(csv)
    import os
    import time
    os.utime(csv, (time.time() + 5, time.time() + 5))
    np.random.seed(11)
    _, f_d, _, _ = utils.get_bag_feats(row, args)
    assert np.array_equal(f_d, f_c)


def test_weight_gradient_split_k_matches_plain_contraction():
    """autograd._tn_mm (X^T Y over the bag axis as one batched GEMM over row chunks + fp32 sum): same result as the plain
    contraction for bag sizes that are / are not a multiple of the chunk count, and the small-bag shortcut."""
    import torch

    from snuffy_amd import autograd as SA
    g = torch.Generator().manual_seed(0)
    for n in (64, 4096, 4099, 8192 + 5):
        a = torch.randn(n, 24, generator=g).to(torch.bfloat16)
        b = torch.randn(n, 40, generator=g).to(torch.bfloat16)
        ref = a.double().t() @ b.double()
        out = SA._tn_mm(a, b)
        assert out.dtype == torch.float32 and out.shape == (24, 40)
        # partials are rounded to bf16 once (2^-9 of their own scale ~ sqrt(n / chunks)) before the fp32 sum
        assert (out.double() - ref).abs().max().item() <= 8e-3 * max(1.0, (n / 8) ** 0.5 * 3), n


def test_param_key_changes_with_every_optimizer_step():
    """Cache keys of everything derived from a parameter advance on ANY optimizer step (a global post-step hook): torch's fused
    optimizers write parameters without bumping their version counters."""
    import torch
    from snuffy_amd import functional as SF
    p = torch.nn.Parameter(torch.randn(4, 4))
    opt = torch.optim.AdamW([p], lr=1e-3)
    k0 = SF.param_key(p)
    assert SF.param_key(p) == k0
    p.grad = torch.randn(4, 4)
    opt.step()
    k1 = SF.param_key(p)
    assert k1 != k0 and k1[3] == k0[3] + 1
    SF.bump_param_epoch()
    assert SF.param_key(p)[3] == k1[3] + 1



# ---- round 6: length-aware work assignment of the bag-parallel path (snuffy_amd/balance.py) -------------------------
def _cam16_lengths(n_bags=400, mean=30000, sigma=0.5, seed=0):
    rs = np.random.RandomState(seed)
    return np.clip(np.round(rs.lognormal(np.log(mean) - sigma * sigma / 2, sigma, n_bags)), 1000, 100000).astype(int)


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("n_bags", [400, 399, 7])
def test_step_groups_visit_every_bag_once_and_balance_the_steps(world, n_bags):
    from snuffy_amd import balance
    lens = _cam16_lengths(n_bags)
    order = np.random.RandomState(3).permutation(n_bags)
    pos = balance.step_groups(lens, world, order)
    steps = (n_bags + world - 1) // world
    assert pos.shape == (steps * world,) and set(pos.tolist()) == set(range(n_bags))
    assert len(pos) - len(set(pos.tolist())) == (-n_bags) % world                   # only the pad of the last group repeats
    assert np.array_equal(pos, balance.step_groups(lens, world, order))               # a pure function: identical on every rank
    if world == 1:
        assert np.array_equal(pos, order)                                             # one rank: the reference's shuffle itself
    if n_bags >= 399:
        # sum over steps of the longest bag <= 1.1 x (sum N / W); positions r::W of a shuffle pay 1.3 ... 1.9
        assert balance.imbalance(lens, pos, world, True) <= 1.1
        other = np.random.RandomState(4).permutation(n_bags)
        assert not np.array_equal(balance.step_groups(lens, world, other), pos) or world == n_bags   # the epoch's shuffle still decides the order


@pytest.mark.parametrize("world", [2, 4, 8])
def test_lpt_assignment_partitions_and_balances(world):
    from snuffy_amd import balance
    lens = _cam16_lengths()
    shares = balance.lpt_assignment(lens, world)
    assert sorted(i for sh in shares for i in sh) == list(range(len(lens))) and all(sh == sorted(sh) for sh in shares)
    assert balance.imbalance(lens, shares, world, False) <= 1.01
    rr = [list(range(r, len(lens), world)) for r in range(world)]
    assert balance.imbalance(lens, shares, world, False) <= balance.imbalance(lens, rr, world, False)
    assert balance.lpt_assignment([5, 5, 5], 2) == [[0, 2], [1]]                      # ties: ascending index, lowest rank first


def test_round6_training_dispatch_stays_off_the_gpu_paths_for_cpu_tensors():
    """The new training-step entry points are gated on CUDA tensors: CPU operands never reach a kernel (the callers keep the torch formulation)."""
    import torch
    from snuffy_amd import autograd as SA
    from snuffy_amd import ops
    a = torch.zeros(2048, 3 * 256, dtype=torch.bfloat16)
    assert not ops.gemm_tn_supported(2048, 256, 256, a, a)
    ins, logits = torch.randn(50, 1, requires_grad=True), torch.randn(1, 1, requires_grad=True)
    assert SA.mil_loss(ins, logits, torch.ones(1), torch.tensor(0.5), torch.nn.BCEWithLogitsLoss()) is None
    # the weight-gradient contraction of the bf16 chain on CPU tensors: the plain product
    x, y = torch.randn(64, 8).to(torch.bfloat16), torch.randn(64, 16).to(torch.bfloat16)
    assert torch.allclose(SA._tn_mm(x, y), x.float().t() @ y.float(), rtol=2e-2, atol=5e-2)      # (a bf16 product below the chunking size)
