#!/usr/bin/env python3
"""Generate golden vectors from the UNMODIFIED reference (jafarinia/snuffy @ /root/reference).

Run ONLY in the build container (the reference cannot travel to the GPU box):

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

What is captured:
  F1  layer math    : MILNet forward in eval mode over a grid of shapes  -> classes, logits, A, S per layer
  F2  selection     : top-k index vectors for tie-free and tie-heavy scores, k1/k2 table
  F3  gradients     : loss.backward() of the SmallWeightTrainer loss (train.py:828-846) in eval mode
  F4  trainer step  : one AdamW step post-update weights (train.py:468-473, 809-826)
  F5  loader        : utils.dropout_patches on the seeded global numpy RNG (utils.py:244-250)
  F6  multiclass    : snuffy_multiclass.MILNet forward (C=2)

Only inputs/outputs (data) are stored -- no reference source text.
"""
import copy
import math
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
np.float = float  # pos_embed.py:57 (MAE only)
_sk = types.ModuleType("skimage")
for _n in ("exposure", "io", "img_as_ubyte", "transform"):
    setattr(_sk, _n, types.ModuleType("skimage." + _n))
sys.modules["skimage"] = _sk
_timm = types.ModuleType("timm")
_td = types.ModuleType("timm.data")
_td.IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
_td.IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
_timm.data = _td
sys.modules["timm"] = _timm
sys.modules["timm.data"] = _td
sys.path.insert(0, "/root/reference")

import snuffy as ref_snuffy  # noqa: E402
import snuffy_multiclass as ref_multi  # noqa: E402
import utils as ref_utils  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def build_ref(mod, D, C, h, mlp, act, enc_drop, big_lambda, r, depth, seed, multiclass=False):
    """Same construction sequence as train.py:861-911 (binary) / 923-971 (multiclass)."""
    torch.manual_seed(seed)
    i_classifier = mod.FCLayer(in_size=D, out_size=C)
    attn = mod.MultiHeadedAttention(h, D)
    if multiclass:
        ff = mod.PositionwiseFeedForward(D, D * mlp, act)
        layer = mod.EncoderLayer(D, copy.deepcopy(attn), copy.deepcopy(ff), C, enc_drop, big_lambda, r)
    else:
        ff = mod.PositionwiseFeedForward(D, D * mlp, act, enc_drop)
        layer = mod.EncoderLayer(D, copy.deepcopy(attn), copy.deepcopy(ff), enc_drop, big_lambda, r)
    b_classifier = mod.BClassifier(mod.Encoder(layer, depth), C, D)
    net = mod.MILNet(i_classifier, b_classifier)
    for _, p in net.named_parameters():
        if p.dim() > 1:
            torch.nn.init.xavier_normal_(p)
    # make biases / LN affine non-trivial so that the vectors discriminate (train.py zero-inits them;
    # any value is a legal state_dict)
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if p.dim() == 1:
                if "norm.weight" in name:
                    p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.2 * torch.randn(p.shape, generator=g))
    return net


def capture_choice():
    """Wrap np.random.choice to record the random index draws (snuffy.py:141-143)."""
    rec = []
    orig = np.random.choice

    def wrapped(a, size=None, replace=True, p=None):
        out = orig(a, size, replace, p)
        rec.append(np.array(out, dtype=np.int64).reshape(-1))
        return out

    return rec, orig, wrapped


def sd_to_np(net):
    return {"sd." + k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}


def run_f1():
    cases = [
        # name,        N,    D,  h, Lam,  r,                depth, act,         seed
        ("musk40",     40,   166, 2, 200, 0.0,               1, "relu",       1),
        ("n1",         1,    64,  2, 10,  0.0,               1, "relu",       2),
        ("n2_rand",    2,    64,  4, 10,  0.5,               1, "relu",       3),
        ("n5_rand",    5,    64,  2, 10,  0.5,               1, "relu",       4),
        ("n150_gelu",  150,  64,  4, 10,  0.0,               1, "gelu",       5),
        ("n1000",      1000, 96,  6, 200, 0.0,               1, "relu",       6),
        ("n1000_d2",   1000, 96,  4, 300, 0.5,               2, "gelu",       7),
        ("n3000_900",  3000, 96,  4, 900, 0.7777777777777778, 1, "leakyrelu", 8),
        ("n150_d5",    150,  64,  2, 10,  0.5,               5, "selu",       9),
        ("n777_h1",    777,  128, 1, 64,  0.25,              1, "relu",       10),
    ]
    for name, N, D, h, lam, r, depth, act, seed in cases:
        net = build_ref(ref_snuffy, D, 1, h, 4, act, 0.0, lam, r, depth, seed).eval()
        g = torch.Generator().manual_seed(1234 + seed)
        x = torch.randn(1, N, D, generator=g)
        rec, orig, wrapped = capture_choice()
        np.random.seed(seed)
        np.random.choice = wrapped
        try:
            with torch.no_grad():
                classes, logits, A = net(x)
        finally:
            np.random.choice = orig
        c = classes.reshape(-1)
        k1 = min(math.ceil(lam * (1.0 - r)), N)
        top = torch.sort(classes, 1, descending=True)[1][:, :k1, :].reshape(-1).numpy().astype(np.int64)
        out = dict(
            x=x.numpy(), classes=classes.numpy(), logits=logits.numpy(),
            cfg=np.array([N, D, h, lam, depth, seed], dtype=np.int64), r=np.float64(r), act=np.array(act),
            top=top,
        )
        A_np = A.numpy()
        if A_np.size <= 600_000:
            out["A"] = A_np
        else:
            rows = np.arange(0, N, 37)
            out["A_rows"] = rows.astype(np.int64)
            out["A_sub"] = A_np[:, :, rows, :]
            out["A_colsum"] = A_np.astype(np.float64).sum(axis=2)  # [1,h,K]
        for li, rnd in enumerate(rec):
            out[f"rnd{li}"] = rnd
        out["n_rnd"] = np.int64(len(rec))
        out.update(sd_to_np(net))
        np.savez_compressed(os.path.join(HERE, f"f1_{name}.npz"), **out)
        print(f"F1 {name}: logits={logits.reshape(-1).tolist()} A={tuple(A.shape)} rnd_layers={len(rec)}")


def run_f2():
    out = {}
    # k1/k2 table (snuffy.py:124,129,137-140), python float arithmetic
    tab = []
    for lam, r, N in [(900, 0.7777777777777778, 3000), (200, 0.0, 8192), (200, 0.0, 40), (300, 0.5, 1000),
                      (10, 0.5, 5), (10, 0.5, 2), (512, 0.0, 100000), (10, 0.3, 7), (200, 0.1, 150), (7, 0.9, 3)]:
        share = 1.0 - r
        k1 = min(math.ceil(lam * share), N)
        k2 = min(int(lam * r), max(0, N - math.ceil(lam * share)))
        tab.append((lam, r, N, k1, k2))
    out["k_table"] = np.array(tab, dtype=np.float64)
    # tie-free scores: reference sort order (snuffy.py:128-130)
    g = torch.Generator().manual_seed(77)
    for N in (5000, 32768):
        cu = torch.unique(torch.randn(2 * N, generator=g))
        cu = cu[torch.randperm(cu.numel(), generator=g)[:N]]
        assert torch.unique(cu).numel() == N, "scores must be tie-free"
        c = cu.view(1, N, 1)
        idx = torch.sort(c, 1, descending=True)[1].reshape(-1).numpy().astype(np.int64)
        out[f"tiefree_c_{N}"] = c.reshape(-1).numpy()
        out[f"tiefree_order_{N}"] = idx[:1024]
    # tie-heavy scores: the build's rule is descending score, ties by ascending index
    c = torch.randint(0, 50, (1, 4000, 1), generator=g).float() / 8.0
    ref_idx = torch.sort(c, 1, descending=True)[1].reshape(-1).numpy().astype(np.int64)
    stable_idx = torch.sort(c, dim=1, descending=True, stable=True)[1].reshape(-1).numpy().astype(np.int64)
    out["ties_c"] = c.reshape(-1).numpy()
    out["ties_ref_order"] = ref_idx[:1024]
    out["ties_stable_order"] = stable_idx[:1024]
    out["ties_ref_equals_stable"] = np.bool_(np.array_equal(ref_idx, stable_idx))
    # special values: +-0, +-inf, denormals
    sp = torch.tensor([0.0, -0.0, 1e-45, -1e-45, float("inf"), float("-inf"), 1.0, -1.0, 0.0, -0.0, 3.5, 3.5])
    out["special_c"] = sp.numpy()
    out["special_stable_order"] = torch.sort(sp.view(1, -1, 1), dim=1, descending=True, stable=True)[1].reshape(-1).numpy()
    out["special_ref_order"] = torch.sort(sp.view(1, -1, 1), 1, descending=True)[1].reshape(-1).numpy()
    np.savez_compressed(os.path.join(HERE, "f2_selection.npz"), **out)
    print("F2 ties_ref_equals_stable =", bool(out["ties_ref_equals_stable"]), "k_table rows", len(tab))


def trainer_loss(net, x, y, w, criterion):
    """train.py:828-846 restated around the reference model (train.py itself needs wandb/lightly)."""
    ins_prediction, bag_prediction, _ = net(x)
    if len(ins_prediction.shape) == 2:
        max_prediction, _ = torch.max(ins_prediction, 0)
    else:
        max_prediction, _ = torch.max(ins_prediction, 1)
    bag_loss = criterion(bag_prediction.view(1, -1), y.view(1, -1))
    max_loss = criterion(max_prediction.view(1, -1), y.view(1, -1))
    loss = w * bag_loss + (1 - w) * max_loss
    with torch.no_grad():
        bag_pred = ((1 - w) * torch.sigmoid(max_prediction) + w * torch.sigmoid(bag_prediction)).squeeze().cpu().numpy()
    return bag_pred, loss, ins_prediction


def run_f3_f4():
    for name, N, D, h, lam, r, depth, act, seed, label in [
        ("g_n600", 600, 64, 4, 50, 0.0, 1, "relu", 21, 1.0),
        ("g_n900_d2", 900, 96, 6, 120, 0.25, 2, "gelu", 22, 0.0),
        ("g_n300_selu", 300, 64, 2, 400, 0.0, 1, "selu", 23, 1.0),
        ("g_n500_lrelu", 500, 64, 4, 64, 0.0, 1, "leakyrelu", 24, 0.0),
    ]:
        net = build_ref(ref_snuffy, D, 1, h, 4, act, 0.0, lam, r, depth, seed).eval()  # eval: dropouts off
        g = torch.Generator().manual_seed(4321 + seed)
        x = torch.randn(1, N, D, generator=g)
        y = torch.tensor([label])
        w = torch.tensor(0.5, requires_grad=True)
        crit = torch.nn.BCEWithLogitsLoss()
        rec, orig, wrapped = capture_choice()
        np.random.seed(seed)
        np.random.choice = wrapped
        try:
            bag_pred, loss, ins = trainer_loss(net, x, y, w, crit)
        finally:
            np.random.choice = orig
        loss.backward()
        out = dict(x=x.numpy(), y=y.numpy(), cfg=np.array([N, D, h, lam, depth, seed], dtype=np.int64),
                   r=np.float64(r), act=np.array(act), loss=loss.detach().numpy(), bag_pred=np.array(bag_pred),
                   ins_sigmoid=torch.sigmoid(ins.detach().view(-1, 1)).numpy(), w_grad=w.grad.numpy())
        for li, rnd in enumerate(rec):
            out[f"rnd{li}"] = rnd
        out["n_rnd"] = np.int64(len(rec))
        out.update(sd_to_np(net))
        for k, p in net.named_parameters():
            out["grad." + k] = p.grad.numpy().copy()
        # F4: one AdamW step with the SmallWeightTrainer param groups (train.py:809-826), defaults of train.py:54-115
        opt = torch.optim.AdamW(
            params=[{"params": w, "lr": 2e-4 * 0.1}, {"params": net.parameters()}],
            lr=2e-4, betas=(0.5, 0.9), weight_decay=5e-3)
        opt.step()
        with torch.no_grad():
            w.data.clamp_(0, 1)
        for k, p in net.named_parameters():
            out["post." + k] = p.detach().numpy().copy()
        out["post_w"] = w.detach().numpy().copy()
        np.savez_compressed(os.path.join(HERE, f"f3_{name}.npz"), **out)
        print(f"F3/F4 {name}: loss={float(loss):.6f} bag_pred={float(bag_pred):.6f}")


def run_f5():
    out = {}
    g = np.random.RandomState(5)
    feats = g.randn(37, 6).astype(np.float32)
    out["feats"] = feats
    for p in (0.0, 0.2, 0.5):
        np.random.seed(11)
        res = ref_utils.dropout_patches(feats, p)
        out[f"out_p{p}"] = res
        out[f"next_rand_p{p}"] = np.float64(np.random.rand())  # pins how much of the RNG stream was consumed
    np.savez_compressed(os.path.join(HERE, "f5_loader.npz"), **out)
    print("F5 done")


def musk_like_bags(seed=7, n_bags=92, d_file=24):
    """Synthetic stand-in for the MUSK pickle (the real file is not in the reference repo): 92 bags of 2..40 instances, labels
    in {-1, 0, 1, 2} (the loader clips to {0, 1}), d_file columns of which feats_size are used
    (MUSK itself has 166 feature columns; the fixture keeps 20 of 24 to stay small)."""
    g = np.random.RandomState(seed)
    bags = []
    for _ in range(n_bags):
        n = int(g.randint(2, 41))
        lab = int(g.choice([-1, 0, 1, 2]))
        bags.append([lab, [g.randn(d_file).astype(np.float32) for _ in range(n)]])
    return bags


def run_f9():
    """F9 host-side loader / metrics / positional table of the hot path's callers: load_mil_data (utils.py:425-496),
    multi_label_roc / optimal_thresh / five_scores (utils.py:253-294), get_2d_sincos_pos_embed (pos_embed.py:21-66)."""
    import pickle
    import tempfile
    import argparse
    from utils_ssls_cf.pos_embed import get_2d_sincos_pos_embed
    out = {}
    bags = musk_like_bags()
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "Musk"))
        for folds, ratio, cur in [(10, 0.2, 3), (5, 0.1, 0)]:
            args = argparse.Namespace(dataset="musk1", cv_num_folds=folds, cv_valid_ratio=ratio, cv_current_fold=cur,
                                      feats_size=20)
            with open(os.path.join(td, "Musk", f"musk1norm_{folds}folds_{ratio}split.pkl"), "wb") as f:
                pickle.dump(bags, f)
            parts = ref_utils.load_mil_data(args, td)
            tag = f"mil_{folds}_{cur}"
            for name, (labels, feats, a, b) in zip(("train", "valid", "test"), parts):
                assert a is None and b is None
                out[f"{tag}.{name}.labels"] = np.stack(labels)
                out[f"{tag}.{name}.lens"] = np.array([f.shape[0] for f in feats], dtype=np.int64)
                out[f"{tag}.{name}.feats"] = np.concatenate(feats, axis=0)
    g = np.random.RandomState(3)
    labels = (g.rand(60, 2) > 0.5).astype(np.float64)
    preds = np.clip(labels * 0.3 + g.rand(60, 2) * 0.7, 0, 1)
    aucs, thr, thr_opt = ref_utils.multi_label_roc(labels, preds, 2)
    out["roc.labels"], out["roc.preds"] = labels, preds
    out["roc.aucs"], out["roc.thr_opt"] = np.array(aucs), np.array(thr_opt)
    out["roc.thr0"], out["roc.thr1"] = thr[0], thr[1]
    aucs1, _, thr1 = ref_utils.multi_label_roc(labels[:, :1], preds[:, 0], 1)
    out["roc1.aucs"], out["roc1.thr_opt"] = np.array(aucs1), np.array(thr1)
    aucsf, _, thrf = ref_utils.multi_label_roc(labels[:, 1], preds[:, 1], 1, for_feats=True)
    out["rocf.aucs"], out["rocf.thr_opt"] = np.array(aucsf), np.array(thrf)
    out["five"] = np.array(ref_utils.five_scores(labels[:, 0], preds[:, 0]))
    for d, gs in [(64, 4), (128, 7)]:
        out[f"sincos_{d}_{gs}"] = get_2d_sincos_pos_embed(d, gs, cls_token=True)
    np.savez_compressed(os.path.join(HERE, "f9_host.npz"), **out)
    print("F9 done", {k: v.shape for k, v in out.items() if k.startswith("mil_10")})


def run_f10():
    """F10 tile preprocessing (compute_feats.py:104-152,173-177).  torchvision is not importable here; its Resize(224) on a PIL
    image is PIL's own Image.resize(size, BILINEAR) (torchvision/transforms/_functional_pil.py), ToTensor is uint8 / 255 in fp32
    and NormalizeImage is (t - mean) / std per channel: captured with PIL + torch on seeded uint8 tiles."""
    from PIL import Image
    rs = np.random.RandomState(10)
    out = {}
    mean = torch.tensor((0.485, 0.456, 0.406)).view(3, 1, 1)
    std = torch.tensor((0.229, 0.224, 0.225)).view(3, 1, 1)
    for name, (h, w) in [("t256", (256, 256)), ("t96x80", (96, 80)), ("t40x70", (40, 70))]:
        # smooth image + noise: exercises the rounding of the fixed-point accumulation on non-trivial gradients
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([(np.sin(xx / 7.0 + c) + np.cos(yy / 5.0 - c)) * 60 + 128 for c in range(3)], axis=-1)
        img = np.clip(base + rs.randint(-40, 41, (h, w, 3)), 0, 255).astype(np.uint8)
        size = 224 if name == "t256" else 32
        if (w <= h and w == size) or (h <= w and h == size):
            oh, ow = h, w
        elif w < h:
            oh, ow = int(size * h / w), size
        else:
            oh, ow = size, int(size * w / h)
        res = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        t = torch.from_numpy(res.copy()).permute(2, 0, 1).float().div(255.0)
        out[name + ".img"] = img
        out[name + ".size"] = np.int64(size)
        out[name + ".resized_u8"] = res
        out[name + ".tensor"] = t.numpy() if name != "t256" else t[:, ::7, ::5].numpy()       # subsampled: keeps the file small
        tn = (t - mean) / std
        out[name + ".normalized"] = tn.numpy() if name != "t256" else tn[:, ::7, ::5].numpy()
    np.savez_compressed(os.path.join(HERE, "f10_tiles.npz"), **out)
    print("F10 done", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim})


def run_f6():
    for name, B, N, D, h, lam, r, depth, seed in [
        ("mc_b1_n100", 1, 100, 64, 4, 10, 0.0, 1, 31),
        ("mc_b2_n60", 2, 60, 64, 2, 6, 0.3, 1, 32),
        ("mc_b1_n8", 1, 8, 64, 2, 10, 0.0, 2, 33),
        ("mc_b1_n400", 1, 400, 96, 6, 40, 0.5, 1, 34),
    ]:
        net = build_ref(ref_multi, D, 2, h, 4, "relu", 0.0, lam, r, depth, seed, multiclass=True).eval()
        g = torch.Generator().manual_seed(999 + seed)
        x = torch.randn(B, N, D, generator=g)
        rec, orig, wrapped = capture_choice()
        np.random.seed(seed)
        np.random.choice = wrapped
        try:
            with torch.no_grad():
                classes, logits, A = net(x)
        finally:
            np.random.choice = orig
        out = dict(x=x.numpy(), classes=classes.numpy(), logits=logits.numpy(), A=A.numpy(),
                   cfg=np.array([B, N, D, h, lam, depth, seed], dtype=np.int64), r=np.float64(r))
        for li, rnd in enumerate(rec):
            out[f"rnd{li}"] = rnd
        out["n_rnd"] = np.int64(len(rec))
        out.update(sd_to_np(net))
        np.savez_compressed(os.path.join(HERE, f"f6_{name}.npz"), **out)
        print(f"F6 {name}: K={A.shape[-1]} logits={logits.reshape(-1).tolist()}")


if __name__ == "__main__":
    which = sys.argv[1:] or ["f1", "f2", "f3", "f5", "f6", "f9", "f10"]
    if "f1" in which:
        run_f1()
    if "f2" in which:
        run_f2()
    if "f3" in which:
        run_f3_f4()
    if "f5" in which:
        run_f5()
    if "f6" in which:
        run_f6()
    if "f9" in which:
        run_f9()
    if "f10" in which:
        run_f10()


# ----------------------------------------------------------------------------------------------------------------------
# F7 / F8: ViT patch-embedding extractors (compute_feats.py path), reference models imported unmodified
# ----------------------------------------------------------------------------------------------------------------------
def _randomize(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 1 and "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif name in ("pos_embed", "decoder_pos_embed"):
                continue                                    # keep the model's own (trunc-normal / sin-cos) table
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))


def run_f7_f8():
    from functools import partial
    import utils_ssls_cf.vision_transformer_with_adapter_dino_version as vit_adapter
    import utils_ssls_cf.vision_transformer_dino as vit_plain
    import utils_ssls_cf.models_adapter_mae as mae_adapter
    ln = partial(torch.nn.LayerNorm, eps=1e-6)
    cases = [
        ("f7_dino_adapter_p16", lambda: vit_adapter.VisionTransformer(
            patch_size=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, qkv_bias=True, norm_layer=ln,
            adapter_ffn_layernorm_option="none", adapter_ffn_init_option="lora", adapter_ffn_scalar="10",
            adapter_ffn_num=8, adapter_d_model=128), dict(kind="dino_adapter", patch=16, dim=128, depth=2, heads=2, ffn=8, scalar=10.0)),
        ("f7_dino_adapter_p32", lambda: vit_adapter.VisionTransformer(
            patch_size=32, embed_dim=64, depth=1, num_heads=1, mlp_ratio=4, qkv_bias=True, norm_layer=ln,
            adapter_ffn_layernorm_option="none", adapter_ffn_init_option="lora", adapter_ffn_scalar="0.1",
            adapter_ffn_num=16, adapter_d_model=64), dict(kind="dino_adapter", patch=32, dim=64, depth=1, heads=1, ffn=16, scalar=0.1)),
        ("f7_dino_plain_p16", lambda: vit_plain.VisionTransformer(
            patch_size=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, qkv_bias=True, norm_layer=ln),
            dict(kind="dino", patch=16, dim=128, depth=2, heads=2, ffn=0, scalar=0.0)),
        ("f8_mae_adapter_p16", lambda: mae_adapter.MaskedAutoencoderViT(
            img_size=224, patch_size=16, embed_dim=128, depth=2, num_heads=2, decoder_embed_dim=32, decoder_depth=1,
            decoder_num_heads=2, mlp_ratio=4, norm_layer=ln, adapter_ffn_scalar="1.0", adapter_ffn_num=8,
            adapter_d_model=128), dict(kind="mae_adapter", patch=16, dim=128, depth=2, heads=2, ffn=8, scalar=1.0)),
        # the reference's own extractor recipe for DINO-adapter is --patch_size=8 (README.md:552-565): 224 / 8 -> T = 785 tokens
        ("f7_dino_adapter_p8", lambda: vit_adapter.VisionTransformer(
            patch_size=8, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, qkv_bias=True, norm_layer=ln,
            adapter_ffn_layernorm_option="none", adapter_ffn_init_option="lora", adapter_ffn_scalar="10",
            adapter_ffn_num=32, adapter_d_model=128), dict(kind="dino_adapter", patch=8, dim=128, depth=2, heads=2, ffn=32, scalar=10.0)),
    ]
    for i, (name, ctor, meta) in enumerate(cases):
        torch.manual_seed(40 + i)
        model = ctor().eval()
        _randomize(model, 400 + i)
        g = torch.Generator().manual_seed(4000 + i)
        img_u8 = torch.randint(0, 256, (2, 3, 224, 224), generator=g, dtype=torch.uint8)   # what ToTensor() sees
        imgs = img_u8.float() / 255.0
        out = dict(imgs_u8=img_u8.numpy(), kind=np.array(meta["kind"]),
                   cfg=np.array([meta["patch"], meta["dim"], meta["depth"], meta["heads"], meta["ffn"]], dtype=np.int64),
                   scalar=np.float64(meta["scalar"]))
        with torch.no_grad():
            feats = model(imgs)
            out["feats"] = feats.numpy()
            if meta["kind"] != "mae_adapter":
                tok = model.prepare_tokens(imgs)
                out["tokens0"] = tok[:, ::13, :].numpy()                 # rows 0, 13, 26, ... of the token matrix
                out["block0"] = model.blocks[0](tok)[:, ::13, :].numpy()
                out["last_attn"] = model.get_last_selfattention(imgs)[:, :, ::29, :].numpy()
        sd = {k: v for k, v in model.state_dict().items() if not k.startswith("decoder") and k != "mask_token"}
        out.update({"sd." + k: v.detach().numpy().copy() for k, v in sd.items()})
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print(f"{name}: feats {tuple(feats.shape)} |feats|max={float(feats.abs().max()):.3f} params={sum(v.numel() for v in sd.values())}")


if "f7" in (sys.argv[1:] or ["f7"]) and __name__ == "__main__":
    run_f7_f8()


# ----------------------------------------------------------------------------------------------------------------------
# F11 / F12: adapter pre-training (row f-4): DINO self-distillation pieces and the MAE-adapter pre-training forward / loss /
# gradients, captured from the unmodified reference.  dino_adapter/ has modules named `utils` / `vision_transformer_with_adapter`
# that shadow the root ones, so F11 runs in a subprocess with its own sys.path; torchvision / wandb (not installed) are stubbed:
# the pieces captured here never touch them.
# ----------------------------------------------------------------------------------------------------------------------
_F11_CHILD = r'''
import sys, types, os
import numpy as np, torch
sys.dont_write_bytecode = True
for name in ("torchvision", "torchvision.datasets", "torchvision.transforms", "torchvision.models", "wandb"):
    sys.modules[name] = types.ModuleType(name)
tv = sys.modules["torchvision"]
tv.datasets, tv.transforms, tv.models = sys.modules["torchvision.datasets"], sys.modules["torchvision.transforms"], sys.modules["torchvision.models"]
tv.models.__dict__.update({})
sys.path.insert(0, "/root/reference/dino_adapter")
import torch.distributed as dist
_avail = torch.cuda.is_available
torch.cuda.is_available = lambda: True   # the script refuses to be imported without a GPU (main_dino_adapter.py:42-44); nothing captured here runs on one
import main_dino_adapter as M            # DINOLoss, train step semantics
torch.cuda.is_available = _avail
import utils as U                        # MultiCropWrapper, cosine_scheduler, clip_gradients, ...
import vision_transformer_with_adapter as V
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("gloo", rank=0, world_size=1)
out = {}
g = torch.Generator().manual_seed(1100)
# (a) DINOLoss: 2 global + 3 local crops, batch 4, out_dim 48; two consecutive calls (the centre carries over)
B, ncrops, od = 4, 5, 48
loss_mod = M.DINOLoss(od, ncrops, 0.04, 0.07, 3, 10)
for step, epoch in enumerate((0, 2)):
    s = torch.randn(ncrops * B, od, generator=g, requires_grad=True)
    t = torch.randn(2 * B, od, generator=g)
    l = loss_mod(s, t, epoch)
    l.backward()
    out[f"dl_s{step}"], out[f"dl_t{step}"], out[f"dl_epoch{step}"] = s.detach().numpy(), t.numpy(), np.int64(epoch)
    out[f"dl_loss{step}"], out[f"dl_grad{step}"], out[f"dl_center{step}"] = l.detach().numpy(), s.grad.numpy(), loss_mod.center.numpy().copy()
out["dl_temp_schedule"] = loss_mod.teacher_temp_schedule
# (b) schedules and gradient helpers
out["cos_a"] = U.cosine_scheduler(5e-4, 1e-6, 7, 11, warmup_epochs=2)
out["cos_b"] = U.cosine_scheduler(0.996, 1.0, 7, 11)
# (c) student = MultiCropWrapper(tiny adapter ViT, DINOHead); one self-distillation step at dropout 0
torch.manual_seed(1101)
def make():
    vit = V.VisionTransformer(patch_size=16, embed_dim=64, depth=2, num_heads=1, mlp_ratio=4, qkv_bias=True,
                              norm_layer=__import__("functools").partial(torch.nn.LayerNorm, eps=1e-6),
                              adapter_ffn_layernorm_option="none", adapter_ffn_init_option="lora", adapter_ffn_scalar="10",
                              adapter_ffn_num=8, adapter_d_model=64, img_size=[64])
    head = V.DINOHead(64, od, use_bn=False, norm_last_layer=True, nlayers=3, hidden_dim=32, bottleneck_dim=16)
    return U.MultiCropWrapper(vit, head)
student, teacher = make(), make()
gg = torch.Generator().manual_seed(1102)
with torch.no_grad():
    for n, p in student.named_parameters():
        if n.endswith("weight_g"):
            continue
        if p.dim() == 1 and "norm" in n and n.endswith("weight"):
            p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=gg))
        elif "pos_embed" not in n:
            p.copy_(0.05 * torch.randn(p.shape, generator=gg))
teacher.load_state_dict(student.state_dict())
for m in list(student.modules()) + list(teacher.modules()):
    if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
        m.dropout = 0.0
for p in teacher.parameters():
    p.requires_grad = False
# adapter tuning: everything frozen but the adapters and the head (main_dino_adapter.py:307-314)
for n, p in student.named_parameters():
    p.requires_grad = ("adaptmlp" in n) or n.startswith("head.")
student.head.last_layer.weight_g.requires_grad = False
student.train(); teacher.train()
crops = [torch.rand(3, 3, 64, 64, generator=gg), torch.rand(3, 3, 64, 64, generator=gg),
         torch.rand(3, 3, 32, 32, generator=gg), torch.rand(3, 3, 32, 32, generator=gg)]
out.update({f"step_crop{i}": c.numpy() for i, c in enumerate(crops)})
out.update({"sd." + k: v.detach().numpy().copy() for k, v in student.state_dict().items()})
lm = M.DINOLoss(od, 4, 0.04, 0.07, 3, 10)
t_out = teacher(crops[:2]); s_out = student(crops)
loss = lm(s_out, t_out, 1)
loss.backward()
out["step_student_out"], out["step_teacher_out"], out["step_loss"], out["step_center"] = s_out.detach().numpy(), t_out.detach().numpy(), loss.detach().numpy(), lm.center.numpy().copy()
norms = U.clip_gradients(student, 0.3)
out["step_clip_norms"] = np.array(norms)
U.cancel_gradients_last_layer(0, student, 1)
names = [n for n, p in student.named_parameters() if p.grad is not None]
out["step_grad_names"] = np.array(names)
out.update({"grad." + n: p.grad.numpy().copy() for n, p in student.named_parameters() if p.grad is not None})
groups = U.get_params_groups(student)
out["groups_sizes"] = np.array([len(groups[0]["params"]), len(groups[1]["params"])])
opt = torch.optim.AdamW(groups, lr=1e-3, weight_decay=0.04)
opt.step()
with torch.no_grad():
    for pq, pk in zip(student.parameters(), teacher.parameters()):
        pk.data.mul_(0.99).add_((1 - 0.99) * pq.detach().data)
out.update({"after_student." + k: v.detach().numpy().copy() for k, v in student.state_dict().items() if "adaptmlp" in k or k.startswith("head.")})
out.update({"after_teacher." + k: v.detach().numpy().copy() for k, v in teacher.state_dict().items() if "adaptmlp" in k or k.startswith("head.")})
np.savez_compressed(sys.argv[1], **out)
print("f11_dino_pretrain: loss", float(loss), "student_out", tuple(s_out.shape), "grads", len(names))
dist.destroy_process_group()
'''


def run_f11():
    import subprocess
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(_F11_CHILD)
        path = f.name
    try:
        subprocess.run([sys.executable, path, os.path.join(HERE, "f11_dino_pretrain.npz")], check=True)
    finally:
        os.unlink(path)


_F12_CHILD = r'''
import sys, types
import numpy as np, torch
sys.dont_write_bytecode = True
np.float = float                                                    # util/pos_embed.py (numpy 2.x)
timm = types.ModuleType("timm"); td = types.ModuleType("timm.data")   # the vendored timm_modified imports constants from timm.data
for n, v in (("IMAGENET_DEFAULT_MEAN", (0.485, 0.456, 0.406)), ("IMAGENET_DEFAULT_STD", (0.229, 0.224, 0.225)),
             ("IMAGENET_INCEPTION_MEAN", (0.5,) * 3), ("IMAGENET_INCEPTION_STD", (0.5,) * 3),
             ("IMAGENET_DPN_MEAN", (124 / 255, 117 / 255, 104 / 255)), ("IMAGENET_DPN_STD", (1 / (.0167 * 255),) * 3)):
    setattr(td, n, v)
timm.data = td; sys.modules["timm"] = timm; sys.modules["timm.data"] = td
sys.path.insert(0, "/root/reference/mae_adapter")
from functools import partial
import models_mae
ln = partial(torch.nn.LayerNorm, eps=1e-6)
torch.manual_seed(1200)
model = models_mae.MaskedAutoencoderViT(img_size=64, patch_size=16, embed_dim=64, depth=2, num_heads=2, decoder_embed_dim=32,
                                        decoder_depth=1, decoder_num_heads=2, mlp_ratio=4, norm_layer=ln, norm_pix_loss=True,
                                        adapter_ffn_scalar="1.0", adapter_ffn_num=8, adapter_d_model=64)
g = torch.Generator().manual_seed(1201)
with torch.no_grad():
    for name, p in model.named_parameters():
        if p.dim() == 1 and "norm" in name and name.endswith("weight"):
            p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
        elif name in ("pos_embed", "decoder_pos_embed"):
            continue
        else:
            p.copy_(0.05 * torch.randn(p.shape, generator=g))
for m in model.modules():
    if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
        m.dropout = 0.0
model.train()
g = torch.Generator().manual_seed(1202)
imgs = torch.rand(3, 3, 64, 64, generator=g)
out = dict(imgs=imgs.numpy(), cfg=np.array([64, 16, 64, 2, 2, 32, 1, 2, 8], dtype=np.int64), mask_ratio=np.float64(0.75))
out.update({"sd." + k: v.detach().numpy().copy() for k, v in model.state_dict().items()})
torch.manual_seed(1203)
noise = torch.rand(3, 16)                       # random_masking's first draw under this seed (models_mae.py:150)
torch.manual_seed(1203)
loss, pred, mask = model(imgs, mask_ratio=0.75)
loss.backward()
out.update(noise=noise.numpy(), loss=loss.detach().numpy(), pred=pred.detach().numpy(), mask=mask.numpy())
names = [n for n, p in model.named_parameters() if p.grad is not None]
out["grad_names"] = np.array(names)
out.update({"grad." + n: p.grad.numpy().copy() for n, p in model.named_parameters() if p.grad is not None})
model.norm_pix_loss = False                     # the same model without the per-patch normalisation of the target
torch.manual_seed(1203)
out["loss_plain"] = model(imgs, mask_ratio=0.75)[0].detach().numpy()
np.savez_compressed(sys.argv[1], **out)
print("f12_mae_pretrain: loss", float(loss), "pred", tuple(pred.shape), "kept", int((1 - mask).sum()), "grads", len(names))
'''


def run_f12():
    import subprocess
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(_F12_CHILD)
        path = f.name
    try:
        subprocess.run([sys.executable, path, os.path.join(HERE, "f12_mae_pretrain.npz")], check=True)
    finally:
        os.unlink(path)


if "f11" in (sys.argv[1:] or []) and __name__ == "__main__":
    run_f11()
if "f12" in (sys.argv[1:] or []) and __name__ == "__main__":
    run_f12()
