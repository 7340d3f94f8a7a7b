"""Host-side callers of the hot path (no GPU): the PRODUCT's loader / metric / positional-table functions against
fixtures captured from the unmodified reference (tests/golden/make_golden.py f5, f9)."""
import argparse
import os
import pickle

import numpy as np
import pytest
import torch

from tests.helpers import golden_files


def _musk_like_bags(seed=7, n_bags=92, d_file=24):
    """Same generator as tests/golden/make_golden.py:musk_like_bags (inputs are re-created, outputs come from the fixture)."""
    g = np.random.RandomState(seed)
    bags = []
    for _ in range(n_bags):
        n = int(g.randint(2, 41))
        lab = int(g.choice([-1, 0, 1, 2]))
        bags.append([lab, [g.randn(d_file).astype(np.float32) for _ in range(n)]])
    return bags


def test_product_dropout_patches_replays_reference():
    """utils.dropout_patches (reference utils.py:244-250): same rows AND same consumption of the global numpy RNG."""
    from snuffy_amd.utils import dropout_patches
    z = np.load(golden_files("f5_")[0])
    feats = z["feats"]
    for p in (0.0, 0.2, 0.5):
        np.random.seed(11)
        out = dropout_patches(feats, p)
        assert np.array_equal(out, z[f"out_p{p}"])
        assert np.random.rand() == float(z[f"next_rand_p{p}"])          # the next draw of the stream is the reference's


def test_load_mil_data_matches_reference(tmp_path):
    """load_mil_data / cross_validation_set / format conversion (reference utils.py:425-496) on a MUSK-shaped pickle."""
    from snuffy_amd.utils import load_mil_data
    z = np.load(golden_files("f9_")[0])
    os.makedirs(tmp_path / "Musk")
    bags = _musk_like_bags()
    for folds, ratio, cur in [(10, 0.2, 3), (5, 0.1, 0)]:
        with open(tmp_path / "Musk" / f"musk1norm_{folds}folds_{ratio}split.pkl", "wb") as f:
            pickle.dump(bags, f)
        args = argparse.Namespace(dataset="musk1", cv_num_folds=folds, cv_valid_ratio=ratio, cv_current_fold=cur, feats_size=20)
        parts = load_mil_data(args, str(tmp_path))
        assert len(parts) == 3
        for name, (labels, feats, fl, pos) in zip(("train", "valid", "test"), parts):
            tag = f"mil_{folds}_{cur}.{name}"
            assert fl is None and pos is None
            assert np.array_equal(np.stack(labels), z[tag + ".labels"])
            assert labels[0].dtype == z[tag + ".labels"].dtype and labels[0].shape == (1,)
            assert np.array_equal(np.array([f.shape[0] for f in feats]), z[tag + ".lens"])
            assert np.array_equal(np.concatenate(feats, axis=0), z[tag + ".feats"])
    with pytest.raises(KeyError):
        load_mil_data(argparse.Namespace(dataset="camelyon16"), str(tmp_path))


def test_epoch_metrics_match_reference():
    """multi_label_roc / optimal_thresh / five_scores (reference utils.py:253-294)."""
    from snuffy_amd.utils import five_scores, multi_label_roc
    z = np.load(golden_files("f9_")[0])
    labels, preds = z["roc.labels"], z["roc.preds"]
    aucs, thr, thr_opt = multi_label_roc(labels, preds, 2)
    assert np.array_equal(np.array(aucs), z["roc.aucs"]) and np.array_equal(np.array(thr_opt), z["roc.thr_opt"])
    assert np.array_equal(thr[0], z["roc.thr0"]) and np.array_equal(thr[1], z["roc.thr1"])
    aucs1, _, thr1 = multi_label_roc(labels[:, :1], preds[:, 0], 1)       # 1-D predictions are promoted to one column
    assert np.array_equal(np.array(aucs1), z["roc1.aucs"]) and np.array_equal(np.array(thr1), z["roc1.thr_opt"])
    aucsf, _, thrf = multi_label_roc(labels[:, 1], preds[:, 1], 1, for_feats=True)
    assert np.array_equal(np.array(aucsf), z["rocf.aucs"]) and np.array_equal(np.array(thrf), z["rocf.thr_opt"])
    assert np.array_equal(np.array(five_scores(labels[:, 0], preds[:, 0])), z["five"])


def test_trainer_calc_metrics_is_the_reference_rule():
    """Trainer._calc_metrics (reference train.py:475-506) on the same labels / predictions: accuracy from the optimal
    thresholds, computed here independently from the fixture's thresholds."""
    from snuffy_amd.train import Trainer
    z = np.load(golden_files("f9_")[0])
    labels, preds = z["roc.labels"], z["roc.preds"]
    tr = Trainer.__new__(Trainer)
    tr.args = argparse.Namespace(num_classes=2)
    acc, aucs, thr = tr._calc_metrics(list(labels), list(preds))
    assert np.array_equal(np.array(aucs), z["roc.aucs"]) and np.array_equal(np.array(thr), z["roc.thr_opt"])
    hard = (preds >= z["roc.thr_opt"][None, :]).astype(float)
    assert acc == float(np.mean(np.all(hard == labels, axis=1)))
    tr.args = argparse.Namespace(num_classes=1)
    acc1, aucs1, thr1 = tr._calc_metrics(list(labels[:, :1]), list(preds[:, :1]))
    assert np.array_equal(np.array(thr1), z["roc1.thr_opt"])
    assert acc1 == float(np.mean((preds[:, 0] >= z["roc1.thr_opt"][0]).astype(float) == labels[:, 0]))


def test_sincos_positional_table_matches_reference():
    """vit.get_2d_sincos_pos_embed (reference utils_ssls_cf/pos_embed.py:21-66), and the MAE encoder starts from it."""
    from snuffy_amd import vit
    z = np.load(golden_files("f9_")[0])
    for d, g in [(64, 4), (128, 7)]:
        assert np.array_equal(vit.get_2d_sincos_pos_embed(d, g, cls_token=True), z[f"sincos_{d}_{g}"])
        assert np.array_equal(vit.get_2d_sincos_pos_embed(d, g), z[f"sincos_{d}_{g}"][1:])
    enc = vit.mae_adapter_encoder(img_size=112, patch_size=16, embed_dim=128, depth=1, num_heads=2, adapter_ffn_num=8,
                                  adapter_d_model=128)
    assert not enc.pos_embed.requires_grad and "pos_embed" in enc.state_dict()
    assert torch.equal(enc.pos_embed[0], torch.from_numpy(z["sincos_128_7"]).float())


def test_cosine_warmup_scheduler_shape():
    """train.py:189-194: linear warm-up over num_epochs / 20 epochs, cosine decay to 0.001 x lr afterwards."""
    from snuffy_amd.train import CosineWarmupScheduler
    lin = torch.nn.Linear(2, 2)
    opt = torch.optim.SGD(lin.parameters(), lr=1.0)
    sch = CosineWarmupScheduler(opt, warmup_epochs=5, max_epochs=100)
    lrs = []
    for _ in range(100):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    assert np.allclose(lrs[:5], [0.2, 0.4, 0.6, 0.8, 1.0])
    assert lrs[5] == 1.0 and all(a >= b for a, b in zip(lrs[5:], lrs[6:])) and abs(lrs[-1] - 0.001) < 1e-12


def test_arch_registry_and_parser():
    from snuffy_amd import train
    assert set(train.ARCH_REGISTRY) == {"snuffy", "snuffy_multiclass"}
    a = train.get_args_parser().parse_args(["--scheduler", "cosinewarmup"])
    assert a.scheduler == "cosinewarmup" and a.big_lambda == 200 and a.betas == [0.5, 0.9]


def test_pil_resampler_tables_reproduce_pil_fixtures():
    """snuffy_amd.tiles.resample_coeffs (Pillow's 8-bit bilinear resampler restated) applied in numpy == the PIL-resized
    fixtures, byte for byte -- the tables are what the HIP kernel consumes (reference compute_feats.py:173-177: Resize(224))."""
    from snuffy_amd.tiles import PRECISION_BITS, resample_coeffs, resize_target
    z = np.load(golden_files("f10_")[0])
    for name in ("t256", "t96x80", "t40x70"):
        img, want = z[name + ".img"], z[name + ".resized_u8"]
        h, w, _ = img.shape
        oh, ow = resize_target(h, w, int(z[name + ".size"]))
        assert (oh, ow) == want.shape[:2]
        hb, hc, _ = resample_coeffs(w, ow)
        vb, vc, _ = resample_coeffs(h, oh)
        tmp = np.zeros((h, ow, 3), dtype=np.uint8)
        for ox in range(ow):
            x0, n = hb[ox]
            acc = (img[:, x0:x0 + n, :].astype(np.int64) * hc[ox, :n].astype(np.int64)[None, :, None]).sum(1)
            tmp[:, ox, :] = np.clip((acc + (1 << (PRECISION_BITS - 1))) >> PRECISION_BITS, 0, 255)
        out = np.zeros((oh, ow, 3), dtype=np.uint8)
        for oy in range(oh):
            y0, n = vb[oy]
            acc = (tmp[y0:y0 + n].astype(np.int64) * vc[oy, :n].astype(np.int64)[:, None, None]).sum(0)
            out[oy] = np.clip((acc + (1 << (PRECISION_BITS - 1))) >> PRECISION_BITS, 0, 255)
        assert np.array_equal(out, want), name
    assert resize_target(224, 224, 224) == (224, 224) and resize_target(256, 256, 224) == (224, 224)
    assert resize_target(300, 260, 224) == (258, 224) and resize_target(200, 333, 224) == (224, 372)


def test_tile_transform_matches_pil_fixtures():
    """The PIL path kept for --device_preprocess 0 (compute_feats.TileTransform) against the same fixtures."""
    from PIL import Image

    from snuffy_amd.compute_feats import TileTransform
    z = np.load(golden_files("f10_")[0])
    for name in ("t96x80", "t40x70"):
        size = int(z[name + ".size"])
        for normalize, key in ((False, ".tensor"), (True, ".normalized")):
            out = TileTransform(size, normalize)({"input": Image.fromarray(z[name + ".img"]), "label": 0, "position": None})
            assert np.array_equal(out["input"].numpy(), z[name + key]), (name, normalize)
