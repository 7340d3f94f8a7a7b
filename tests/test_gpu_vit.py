"""ViT extractor on the GPU: kernels vs the CPU oracle, models vs the golden vectors captured from the reference."""
import os
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vit_oracle as vorc
from tests.helpers import golden_files, load_case, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ops():
    from snuffy_amd import ops as o
    return o


@pytest.mark.parametrize("b,c,hw,ps", [(2, 3, 224, 16), (3, 3, 224, 32), (1, 3, 64, 8), (5, 1, 32, 16)])
def test_patchify_exact(b, c, hw, ps):
    g = torch.Generator().manual_seed(hw + ps)
    img = torch.rand(b, c, hw, hw, generator=g)
    gh = hw // ps
    ref = img.reshape(b, c, gh, ps, gh, ps).permute(0, 2, 4, 1, 3, 5).reshape(b * gh * gh, c * ps * ps)
    got = ops().vit_patchify(img.to(DEV), ps)
    assert torch.equal(got.cpu(), ref)
    gotb = ops().vit_patchify(img.to(DEV), ps, torch.bfloat16)
    assert torch.equal(gotb.cpu(), ref.to(torch.bfloat16))
    # and it is the conv: cols @ W^T + b == Conv2d
    w, bias = torch.randn(24, c, ps, ps, generator=g), torch.randn(24, generator=g)
    conv = F.conv2d(img, w, bias, stride=ps).flatten(2).transpose(1, 2).reshape(-1, 24)
    assert rel_err(got.cpu() @ w.reshape(24, -1).t() + bias, conv) < 1e-5


def test_assemble_and_residual_ln():
    g = torch.Generator().manual_seed(1)
    B, P, D = 3, 49, 192
    pe, cls, pos = torch.randn(B * P, D, generator=g), torch.randn(1, 1, D, generator=g), torch.randn(1, P + 1, D, generator=g)
    ref = torch.cat((cls.expand(B, -1, -1), pe.view(B, P, D)), dim=1) + pos
    got = ops().vit_assemble_tokens(pe.to(DEV), cls.to(DEV), pos[0].to(DEV), B)
    assert torch.equal(got.cpu().view(B, P + 1, D), ref)
    gotb = ops().vit_assemble_tokens(pe.to(DEV).to(torch.bfloat16), cls.to(DEV), pos[0].to(DEV), B)
    refb = torch.cat((cls.expand(B, -1, -1), pe.to(torch.bfloat16).float().view(B, P, D)), dim=1) + pos
    assert torch.equal(gotb.cpu().view(B, P + 1, D), refb)
    for d in (64, 192, 384, 768):
        n = 301
        x = torch.randn(n, d, generator=g)
        a1, a2 = torch.randn(n, d, generator=g).to(torch.bfloat16), torch.randn(n, d, generator=g).to(torch.bfloat16)
        gam, bet = torch.randn(d, generator=g), torch.randn(d, generator=g)
        xd = x.to(DEV)
        ln, xb = ops().vit_residual_ln_(xd, a1.to(DEV), a2.to(DEV), 10.0, gam.to(DEV), bet.to(DEV), 1e-6, True, True)
        xr = x + a1.float() + 10.0 * a2.float()
        assert (xd.cpu() - xr).abs().max() < 1e-5
        assert torch.equal(xb.cpu(), xd.cpu().to(torch.bfloat16))
        lr = F.layer_norm(xr, (d,), gam, bet, 1e-6)
        assert (ln.cpu().float() - lr).abs().max() < 3e-2 and rel_err(ln.cpu().float(), lr) < 1e-2
        x2 = x.to(DEV)
        ln2, _ = ops().vit_residual_ln_(x2, None, None, 1.0, gam.to(DEV), bet.to(DEV), 1e-6, True, False)
        assert torch.equal(x2.cpu(), x)


def ref_attention(qkv, b, t, h):
    d = qkv.shape[1] // 3
    dk = d // h
    q, k, v = qkv.double().view(b, t, 3, h, dk).permute(2, 0, 3, 1, 4)
    attn = ((q @ k.transpose(-2, -1)) * dk ** -0.5).softmax(-1)
    return (attn @ v).transpose(1, 2).reshape(b * t, d), attn


@pytest.mark.parametrize("b,t,h,dk", [(2, 197, 6, 64), (3, 50, 3, 64), (1, 1, 2, 32), (2, 300, 2, 64), (2, 65, 1, 128)])
def test_vit_attention_exact(b, t, h, dk):
    g = torch.Generator().manual_seed(t)
    qkv = torch.randn(b * t, 3 * h * dk, generator=g)
    o_ref, a_ref = ref_attention(qkv, b, t, h)
    o, attn = ops().vit_attention(qkv.to(DEV), b, t, h, need_attn=True)
    assert (attn.cpu().double() - a_ref).abs().max() < 1e-6
    assert rel_err(o.cpu(), o_ref) < 1e-5


@pytest.mark.parametrize("b,t,h", [(2, 197, 6), (3, 50, 3), (1, 1, 1), (2, 256, 2), (4, 100, 12), (2, 150, 2), (1, 33, 1),
                                   (2, 785, 6), (1, 257, 2), (3, 512, 1), (1, 1030, 2), (2, 300, 3)])   # > 256: keys in LDS chunks
def test_vit_attention_mfma(b, t, h):
    g = torch.Generator().manual_seed(t + h)
    qkv = (torch.randn(b * t, 3 * h * 64, generator=g) * 1.5).to(torch.bfloat16)
    o_ref, _ = ref_attention(qkv.float(), b, t, h)            # same bf16-rounded operands: tight comparison
    o, _ = ops().vit_attention(qkv.to(DEV), b, t, h)
    assert o.dtype == torch.bfloat16
    assert rel_err(o.cpu().float(), o_ref) < 1.5e-2           # P and O are rounded to bf16 (2^-8 relative)
    o2, _ = ops().vit_attention(qkv.to(DEV), b, t, h)
    assert torch.equal(o, o2)


@pytest.mark.parametrize("b,t,h", [(2, 197, 6), (3, 50, 3), (1, 1, 1), (2, 224, 2), (2, 225, 2), (4, 100, 12), (1, 33, 1),
                                   (2, 785, 6), (1, 449, 2), (1, 1030, 1)])
def test_vit_attention_x3(b, t, h):
    """fp32-class self-attention on the matrix cores (split-bf16 x3 products) against fp64: the aggregator's arithmetic class."""
    g = torch.Generator().manual_seed(7 * t + h)
    qkv = torch.randn(b * t, 3 * h * 64, generator=g) * 1.5
    o_ref, _ = ref_attention(qkv, b, t, h)
    o, attn = ops().vit_attention(qkv.to(DEV), b, t, h, arithmetic="x3")
    assert o.dtype == torch.float32 and attn is None
    e = rel_err(o.cpu(), o_ref)
    print("MEASURED vit x3 attention", (b, t, h), e)
    assert e < 5e-5
    o2, _ = ops().vit_attention(qkv.to(DEV), b, t, h, arithmetic="x3")
    assert torch.equal(o, o2)
    oe, _ = ops().vit_attention(qkv.to(DEV), b, t, h)                       # exact kernel: the cross-check on the device
    assert rel_err(o.cpu(), oe.cpu()) < 5e-5
    # the hl-image output (operand of the one-pass proj GEMM) is the split of the same fp32 rows, bit for bit
    img, _ = ops().vit_attention(qkv.to(DEV), b, t, h, arithmetic="x3", hl_out=True)
    assert torch.equal(img, ops().split_hl_rows(o))


def build(z):
    from snuffy_amd import vit
    patch, dim, depth, heads, ffn = [int(v) for v in z["cfg"]]
    kind, scale = str(z["kind"]), float(z["scalar"])
    ln = partial(torch.nn.LayerNorm, eps=1e-6)
    if kind == "mae_adapter":
        return vit.mae_adapter_encoder(224, patch, dim, depth, heads, 4, ln, repr(scale), ffn, dim)
    return vit.VisionTransformer(patch_size=patch, embed_dim=dim, depth=depth, num_heads=heads, mlp_ratio=4, qkv_bias=True,
                                 norm_layer=ln, adapter_ffn_scalar=repr(scale), adapter_ffn_num=max(ffn, 1),
                                 adapter_d_model=dim, use_adapter=(kind != "dino"))


@pytest.mark.parametrize("gemm", ["x3", "library"])
@pytest.mark.parametrize("path", golden_files("f7_") + golden_files("f8_"), ids=lambda p: p.split("/")[-1][:-4])
def test_vit_models_match_reference_goldens(path, gemm, monkeypatch):
    """fp32 path against the reference's own outputs, with the projections as split-bf16 x3 products on the matrix cores (the
    default: ~1e-5 per product, fp32 accumulate) and as plain fp32 library GEMMs."""
    from snuffy_amd import vit as vit_mod
    monkeypatch.setattr(vit_mod, "FP32_GEMM", gemm)
    tight = gemm == "library"
    z, sd = load_case(path)
    model = build(z)
    assert sorted(model.state_dict().keys()) == sorted(sd.keys())
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).eval()
    imgs = (torch.from_numpy(z["imgs_u8"]).float() / 255.0).to(DEV)
    feats = model(imgs)
    np.testing.assert_allclose(feats.cpu().numpy(), z["feats"], rtol=0, atol=1e-3)      # north-star fp32 tolerance
    print("MEASURED vit fp32 golden", gemm, path.split("/")[-1], float(np.abs(feats.cpu().numpy() - z["feats"]).max()))
    np.testing.assert_allclose(feats.cpu().numpy(), z["feats"], rtol=0, atol=5e-5 if tight else 2e-4)   # reference-class in fact
    if str(z["kind"]) != "mae_adapter":
        with torch.no_grad():
            tok = model.prepare_tokens(imgs)
            np.testing.assert_allclose(tok[:, ::13, :].cpu().numpy(), z["tokens0"], rtol=0, atol=1e-5 if tight else 5e-5)
            np.testing.assert_allclose(model.blocks[0](tok)[:, ::13, :].cpu().numpy(), z["block0"], rtol=0,
                                       atol=5e-5 if tight else 2e-4)
            attn = model.get_last_selfattention(imgs)
            np.testing.assert_allclose(attn[:, :, ::29, :].cpu().numpy(), z["last_attn"], rtol=0, atol=1e-5 if tight else 5e-5)
    ref = torch.from_numpy(z["feats"])
    for fused in (True, False):       # the fused block (LayerNorm folded into the GEMMs, round 6) and the round-2 block
        monkeypatch.setattr(vit_mod, "BF16_FUSED_BLOCK", fused)
        fb = model.configure("bf16")(imgs)
        e = rel_err(fb.cpu(), ref)
        print("MEASURED vit bf16 golden", "fused" if fused else "unfused", path.split("/")[-1] if isinstance(path, str) else "", e)
        assert e < 1e-2, e                                                             # north-star bf16 class, no depth allowance


def test_vit_small_shape_and_iclassifier():
    from snuffy_amd import vit
    torch.manual_seed(0)
    model = vit.vit_small(patch_size=16, adapter_ffn_scalar="10", adapter_ffn_num=32, adapter_d_model=384)
    emb = vit.IClassifier(model, 384, 2).to(DEV).eval()
    x = torch.rand(4, 3, 224, 224, device=DEV)
    with torch.no_grad():
        feats, c = emb(x)
        sd = {k: v.cpu() for k, v in model.state_dict().items()}
        ref = vorc.vit_forward(x.cpu(), sd, 16, 12, 6, 10.0, "dino_adapter")
    assert feats.shape == (4, 384) and c.shape == (4, 2)
    assert (feats.cpu() - ref).abs().max() < 1e-3
    from snuffy_amd import vit as vit_mod
    assert vit_mod.BF16_FUSED_BLOCK and model._fused_ok()
    fb = model.configure("bf16")(x)
    e = rel_err(fb.cpu(), ref)
    print("MEASURED vit_small bf16 depth 12", e)
    assert e < 1e-2, e                                                                 # north-star bf16 class at depth 12
    vit_mod.BF16_FUSED_BLOCK = False
    try:
        e2 = rel_err(model(x).cpu(), ref)
    finally:
        vit_mod.BF16_FUSED_BLOCK = True
    print("MEASURED vit_small bf16 depth 12, unfused block", e2)
    assert e2 < 1e-2, e2


def test_compute_feats_end_to_end(tmp_path):
    """Tile directory -> embedder -> CSV (%.4f) -> utils.get_bag_feats, the reference's file hand-off between stages."""
    import argparse

    import pandas as pd
    from PIL import Image

    from snuffy_amd import compute_feats as cf
    from snuffy_amd import utils, vit
    rng = np.random.RandomState(0)
    bag_dir = tmp_path / "single" / "tumor" / "slide_001"
    bag_dir.mkdir(parents=True)
    for r in range(3):
        for c in range(3):
            Image.fromarray(rng.randint(0, 255, (256, 256, 3), dtype=np.uint8)).save(bag_dir / f"{r}_{c}-17.jpeg", quality=95)
    args = argparse.Namespace(backbone="vit_tiny", embedder="DINO_adapter", patch_size=16, adapter_ffn_scalar="10",
                              ffn_num=8, num_classes=1, batch_size=4, num_workers=0, transform=0, dataset="camelyon16",
                              weights=None, precision="fp32")
    torch.manual_seed(0)
    backbone, nf = cf.get_embedder_backbone(args)
    assert nf == 192 and not any(p.requires_grad for p in backbone.parameters())
    embedder, _ = cf.get_embedder(args, backbone, nf)
    labels = {os.path.join("tumor", "slide_001", f"{r}_{c}-17.jpeg"): int((r + c) % 2) for r in range(3) for c in range(3)}
    cf.compute_feats(args, [str(bag_dir)], embedder, str(tmp_path / "out"), labels)
    csv = tmp_path / "out" / "single" / "tumor" / "slide_001.csv"
    df = pd.read_csv(csv)
    assert df.shape == (9, 192 + 2) and list(df.columns[-2:]) == ["label", "position"]
    body = open(csv).read().splitlines()[1].split(",")[:192]
    assert all(len(v.split(".")[1]) == 4 for v in body)                        # '%.4f'
    # the features are what the embedder gives for the same transformed tiles
    files = sorted(str(p) for p in bag_dir.glob("*.jpeg"))
    tf = cf.TileTransform(224)
    x = torch.stack([tf({"input": Image.open(f), "label": 0, "position": None})["input"] for f in files]).to(DEV)
    with torch.no_grad():
        feats, _ = embedder(x)
    np.testing.assert_allclose(df.iloc[:, :192].to_numpy(), feats.cpu().numpy(), atol=5.1e-5)
    # and the MIL loader reads it back (rows shuffled, label / position split off)
    a = argparse.Namespace(num_classes=1)
    lab, f, fl, pos = utils.get_bag_feats(pd.Series([str(csv), 1]), a)
    assert f.shape == (9, 192) and f.dtype == np.float32 and lab.tolist() == [1.0] and len(pos) == 9


def test_positional_weight_loading_like_reference(tmp_path):
    """compute_feats.py:477-480 maps checkpoint tensors to the embedder BY POSITION: our key order must be the reference's."""
    import argparse

    from snuffy_amd import compute_feats as cf
    z, sd = load_case(golden_files("f7_dino_adapter_p16")[0])
    ref_order = [k[3:] for k in z.files if k.startswith("sd.")]
    model = build(z)
    assert list(model.state_dict().keys()) == ref_order
    # a DINO checkpoint: {'teacher': {'backbone.<key>': tensor}} in the same order
    ckpt = {"teacher": {"backbone." + k: sd[k] for k in ref_order}}
    torch.save(ckpt, tmp_path / "checkpoint.pth")
    args = argparse.Namespace(embedder="DINO_adapter", weights=str(tmp_path / "checkpoint.pth"), num_classes=1)
    emb, _ = cf.get_embedder(args, model, 128)
    for k in ref_order:
        assert torch.equal(emb.state_dict()["feature_extractor." + k].cpu(), sd[k])


def test_tile_preprocess_kernel_bit_exact_vs_pil_fixtures():
    """snf_tile_preprocess_u8 (Resize + ToTensor + NormalizeImage for a batch of uint8 tiles in one launch) against the PIL /
    torch fixtures: the fp32 tensor bit for bit, the bf16 patch-embedding operand = patchify of that tensor."""
    from snuffy_amd import ops as o
    from snuffy_amd.tiles import preprocess_tiles
    z = np.load(golden_files("f10_")[0])
    for name in ("t256", "t96x80", "t40x70"):
        size = int(z[name + ".size"])
        img = torch.from_numpy(z[name + ".img"]).to(DEV)
        batch = torch.stack([img, img.flip(0), img.flip(1)])                      # three tiles, one launch
        for normalize, key in ((False, ".tensor"), (True, ".normalized")):
            out = preprocess_tiles(batch, size, normalize=normalize)
            got = out[0].cpu().numpy()
            want = z[name + key]
            if name == "t256":
                got = got[:, ::7, ::5]
            assert np.array_equal(got, want), (name, normalize)
        u8 = (preprocess_tiles(batch, size)[0] * 255.0).round().to(torch.uint8).permute(1, 2, 0).cpu().numpy()
        assert np.array_equal(u8, z[name + ".resized_u8"])
    # the GEMM-operand output: same values as patchify of the fp32 output, for the whole batch
    img = torch.from_numpy(z["t256.img"]).to(DEV)
    batch = torch.stack([img, img.flip(0)])
    f32, cols = preprocess_tiles(batch, 224, normalize=True, want="both", patch=16)
    assert torch.equal(cols, o.vit_patchify(f32, 16, torch.bfloat16))
    with pytest.raises(Exception):
        preprocess_tiles(batch.cpu(), 224)


def test_compute_feats_device_preprocess_equals_pil_path(tmp_path):
    """compute_feats with the batched on-device preprocessing (default) writes the same CSV as with the reference's per-tile
    PIL transforms (--device_preprocess 0): the two pipelines feed the embedder identical tensors."""
    import argparse

    import pandas as pd
    from PIL import Image

    from snuffy_amd import compute_feats as cf
    rng = np.random.RandomState(1)
    bag_dir = tmp_path / "single" / "normal" / "slide_007"
    bag_dir.mkdir(parents=True)
    for r in range(2):
        for c in range(3):
            Image.fromarray(rng.randint(0, 255, (256, 256, 3), dtype=np.uint8)).save(bag_dir / f"{r}_{c}.jpeg", quality=90)
    outs = {}
    for mode in (1, 0):
        args = argparse.Namespace(backbone="vit_tiny", embedder="DINO_adapter", patch_size=16, adapter_ffn_scalar="10",
                                  ffn_num=8, num_classes=1, batch_size=4, num_workers=0, transform=1, dataset="tcga",
                                  weights=None, precision="fp32", device_preprocess=mode)
        torch.manual_seed(0)
        backbone, nf = cf.get_embedder_backbone(args)
        embedder, _ = cf.get_embedder(args, backbone, nf)
        cf.compute_feats(args, [str(bag_dir)], embedder, str(tmp_path / f"out{mode}"))
        outs[mode] = pd.read_csv(tmp_path / f"out{mode}" / "single" / "normal" / "slide_007.csv").to_numpy()
    assert outs[1].shape == (6, 192) and np.array_equal(outs[0], outs[1])


def test_compute_feats_bf16_takes_patch_columns_straight_from_the_tile_kernel(tmp_path):
    """bf16 extractor + on-device preprocessing: the tile kernel writes the patch-embedding GEMM operand (im2col rows, bf16)
    and the ViT starts from it (VisionTransformer.forward_cols) -- same CSV as the PIL path through patchify."""
    import argparse

    import pandas as pd
    from PIL import Image

    from snuffy_amd import compute_feats as cf
    from snuffy_amd import vit
    rng = np.random.RandomState(2)
    bag_dir = tmp_path / "single" / "tumor" / "slide_011"
    bag_dir.mkdir(parents=True)
    for r in range(2):
        for c in range(2):
            Image.fromarray(rng.randint(0, 255, (256, 256, 3), dtype=np.uint8)).save(bag_dir / f"{r}_{c}.jpeg", quality=90)
    calls = {"cols": 0}
    orig = vit.VisionTransformer.forward_cols

    def counting(self, *a, **kw):
        calls["cols"] += 1
        return orig(self, *a, **kw)
    vit.VisionTransformer.forward_cols = counting
    try:
        outs = {}
        for mode in (1, 0):
            args = argparse.Namespace(backbone="vit_tiny", embedder="DINO_adapter", patch_size=16, adapter_ffn_scalar="10",
                                      ffn_num=8, num_classes=1, batch_size=4, num_workers=0, transform=1, dataset="tcga",
                                      weights=None, precision="bf16", device_preprocess=mode)
            torch.manual_seed(0)
            backbone, nf = cf.get_embedder_backbone(args)
            embedder, _ = cf.get_embedder(args, backbone, nf)
            before = calls["cols"]
            cf.compute_feats(args, [str(bag_dir)], embedder, str(tmp_path / f"out{mode}"))
            if mode == 1:
                assert calls["cols"] == before + 1                 # one batch of 4 tiles went in as columns
            outs[mode] = pd.read_csv(tmp_path / f"out{mode}" / "single" / "tumor" / "slide_011.csv").to_numpy()
    finally:
        vit.VisionTransformer.forward_cols = orig
    assert outs[1].shape == (4, 192) and np.array_equal(outs[0], outs[1])


def test_adapter_pretraining_steps_on_the_device():
    """snuffy_amd/ssl_pretrain.py on the GPU against the reference fixtures F11 / F12: the DINO self-distillation
    step (loss, adapter gradients) and the MAE-adapter reconstruction loss / gradients through PyTorch-ROCm autograd."""
    from tests import test_ssl_pretrain as T
    from snuffy_amd import ssl_pretrain as S
    z = T._npz("f11_dino_pretrain.npz")
    student, teacher = T._student(), T._student()
    student.load_state_dict(T._sd(z)), teacher.load_state_dict(T._sd(z))
    student, teacher = student.to(DEV).train(), teacher.to(DEV).train()
    for p in teacher.parameters():
        p.requires_grad = False
    S.freeze_for_adapter_tuning(student)
    crops = [torch.from_numpy(z[f"step_crop{i}"]).to(DEV) for i in range(4)]
    lm = S.DINOLoss(48, 4, 0.04, 0.07, 3, 10).to(DEV)
    with torch.no_grad():
        t_out = teacher(crops[:2])
    loss = lm(student(crops), t_out, 1)
    loss.backward()
    assert abs(float(loss) - float(z["step_loss"])) < 1e-4
    S.clip_gradients(student, 0.3)                       # the fixture holds the gradients behind the per-parameter clipping
    for n, p in student.named_parameters():
        if p.grad is not None and "adaptmlp" in n:
            ref = z["grad." + n]
            assert np.allclose(p.grad.cpu().numpy(), ref, rtol=2e-3, atol=1e-3 * np.abs(ref).max() + 1e-7), n
    z = T._npz("f12_mae_pretrain.npz")
    model = T._mae()
    model.load_state_dict(T._sd(z))
    model = model.to(DEV).train()
    loss, pred, mask = model(torch.from_numpy(z["imgs"]).to(DEV), 0.75, torch.from_numpy(z["noise"]).to(DEV))
    loss.backward()
    assert np.array_equal(mask.cpu().numpy(), z["mask"]) and abs(float(loss) - float(z["loss"])) < 1e-4
    ref = z["grad.decoder_pred.weight"]
    assert np.allclose(model.decoder_pred.weight.grad.cpu().numpy(), ref, rtol=2e-3, atol=1e-3 * np.abs(ref).max())
    # the trained model is still the extractor: under no_grad / eval its forward is the kernel path of compute_feats
    model.eval()
    with torch.no_grad():
        feats = model(torch.rand(2, 3, 64, 64, device=DEV))
    assert feats.shape == (2, 64) and bool(torch.isfinite(feats).all())
