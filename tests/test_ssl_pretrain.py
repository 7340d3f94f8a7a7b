"""Adapter pre-training: snuffy_amd/ssl_pretrain.py against fixtures captured from the unmodified reference
(tests/golden/make_golden.py f11 / f12: dino_adapter/main_dino_adapter.py + utils.py + vision_transformer_with_adapter.py, and
mae_adapter/models_mae.py).  Pure PyTorch autograd: runs without a GPU."""
import os
import socket
from functools import partial

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LN = partial(torch.nn.LayerNorm, eps=1e-6)


def _npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def _sd(z, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def test_dino_loss_centre_and_schedules_match_the_reference():
    from snuffy_amd import ssl_pretrain as S
    z = _npz("f11_dino_pretrain.npz")
    loss_mod = S.DINOLoss(48, 5, 0.04, 0.07, 3, 10)
    assert np.array_equal(loss_mod.teacher_temp_schedule, z["dl_temp_schedule"])
    for step in (0, 1):
        s = torch.from_numpy(z[f"dl_s{step}"]).requires_grad_(True)
        loss = loss_mod(s, torch.from_numpy(z[f"dl_t{step}"]), int(z[f"dl_epoch{step}"]))
        loss.backward()
        assert abs(float(loss) - float(z[f"dl_loss{step}"])) < 1e-6
        assert np.allclose(s.grad.numpy(), z[f"dl_grad{step}"], atol=1e-8)
        assert np.allclose(loss_mod.center.numpy(), z[f"dl_center{step}"], atol=1e-7)      # the centre carries over between the calls
    assert np.array_equal(S.cosine_scheduler(5e-4, 1e-6, 7, 11, warmup_epochs=2), z["cos_a"])
    assert np.array_equal(S.cosine_scheduler(0.996, 1.0, 7, 11), z["cos_b"])


def _student():
    from snuffy_amd import ssl_pretrain as S
    from snuffy_amd import vit
    backbone = vit.VisionTransformer(img_size=[64], patch_size=16, embed_dim=64, depth=2, num_heads=1, mlp_ratio=4, qkv_bias=True, norm_layer=LN,
                                     adapter_ffn_layernorm_option="none", adapter_ffn_init_option="lora", adapter_ffn_scalar="10",
                                     adapter_ffn_num=8, adapter_d_model=64)
    head = S.DINOHead(64, 48, norm_last_layer=True, nlayers=3, hidden_dim=32, bottleneck_dim=16)
    net = S.MultiCropWrapper(backbone, head)
    for m in net.modules():
        if isinstance(getattr(m, "dropout", None), float):
            m.dropout = 0.0
    return net


def test_dino_self_distillation_step_matches_the_reference():
    """MultiCropWrapper(adapter ViT, DINOHead) on 2 global + 2 local crops, adapter tuning, one step: outputs, loss, gradients of every
    trainable parameter, per-parameter clipping, the last-layer freeze, the AdamW step and the teacher EMA."""
    from snuffy_amd import ssl_pretrain as S
    z = _npz("f11_dino_pretrain.npz")
    student, teacher = _student(), _student()
    sd = _sd(z)
    assert sorted(student.state_dict().keys()) == sorted(sd.keys())                        # the reference's checkpoint keys
    student.load_state_dict(sd, strict=True)
    teacher.load_state_dict(sd, strict=True)
    for p in teacher.parameters():
        p.requires_grad = False
    trainable = S.freeze_for_adapter_tuning(student)
    assert all(("adaptmlp" in n) or n.startswith("head.") for n, p in student.named_parameters() if p.requires_grad) and trainable
    student.train(), teacher.train()
    crops = [torch.from_numpy(z[f"step_crop{i}"]) for i in range(4)]
    lm = S.DINOLoss(48, 4, 0.04, 0.07, 3, 10)
    with torch.no_grad():
        t_out = teacher(crops[:2])
    s_out = student(crops)
    assert np.allclose(t_out.numpy(), z["step_teacher_out"], atol=2e-6) and np.allclose(s_out.detach().numpy(), z["step_student_out"], atol=2e-6)
    loss = lm(s_out, t_out, 1)
    assert abs(float(loss) - float(z["step_loss"])) < 2e-6 and np.allclose(lm.center.numpy(), z["step_center"], atol=1e-6)
    loss.backward()
    norms = S.clip_gradients(student, 0.3)
    assert np.allclose(np.array(norms), z["step_clip_norms"], rtol=2e-4, atol=1e-7)
    S.cancel_gradients_last_layer(0, student, 1)
    names = [n for n, p in student.named_parameters() if p.grad is not None]
    assert names == [str(n) for n in z["step_grad_names"]]
    for n, p in student.named_parameters():
        if p.grad is not None:
            ref = z["grad." + n]
            assert np.allclose(p.grad.numpy(), ref, rtol=2e-4, atol=2e-7 + 1e-4 * np.abs(ref).max()), n
    groups = S.get_params_groups(student)
    assert [len(groups[0]["params"]), len(groups[1]["params"])] == z["groups_sizes"].tolist() and groups[1]["weight_decay"] == 0.
    torch.optim.AdamW(groups, lr=1e-3, weight_decay=0.04).step()
    S.ema_update(student, teacher, 0.99)
    for k, v in student.state_dict().items():
        if "adaptmlp" in k or k.startswith("head."):
            assert np.allclose(v.numpy(), z["after_student." + k], atol=2e-6), k
    for k, v in teacher.state_dict().items():
        if "adaptmlp" in k or k.startswith("head."):
            assert np.allclose(v.numpy(), z["after_teacher." + k], atol=2e-6), k


def test_dino_train_step_helper_runs_the_same_iteration():
    from snuffy_amd import ssl_pretrain as S
    z = _npz("f11_dino_pretrain.npz")
    student, teacher = _student(), _student()
    student.load_state_dict(_sd(z)), teacher.load_state_dict(_sd(z))
    for p in teacher.parameters():
        p.requires_grad = False
    S.freeze_for_adapter_tuning(student)
    student.train(), teacher.train()
    opt = torch.optim.AdamW(S.get_params_groups(student), lr=1e-3, weight_decay=0.04)
    crops = [torch.from_numpy(z[f"step_crop{i}"]) for i in range(4)]
    loss = S.dino_train_step(student, teacher, S.DINOLoss(48, 4, 0.04, 0.07, 3, 10), crops, opt, epoch=1, it=0,
                             momentum_schedule=np.array([0.99]), clip_grad=0.3, freeze_last_layer=1)
    assert abs(float(loss) - float(z["step_loss"])) < 2e-6
    # (epoch 1 >= freeze_last_layer: the last layer trains in this call, unlike the fixture's epoch-0 freeze -- compare the adapters)
    for k, v in student.state_dict().items():
        if "adaptmlp" in k:
            assert np.allclose(v.numpy(), z["after_student." + k], atol=2e-6), k


def _mae():
    from snuffy_amd import ssl_pretrain as S
    model = S.MAEPretrainModel(img_size=64, patch_size=16, embed_dim=64, depth=2, num_heads=2, decoder_embed_dim=32, decoder_depth=1,
                               decoder_num_heads=2, mlp_ratio=4, norm_layer=LN, norm_pix_loss=True, adapter_ffn_scalar="1.0",
                               adapter_ffn_num=8, adapter_d_model=64)
    for m in model.modules():
        if isinstance(getattr(m, "dropout", None), float):
            m.dropout = 0.0
    return model


def test_mae_adapter_pretraining_forward_loss_and_gradients_match_the_reference():
    z = _npz("f12_mae_pretrain.npz")
    model = _mae()
    sd = _sd(z)
    assert sorted(model.state_dict().keys()) == sorted(sd.keys())
    model.load_state_dict(sd, strict=True)
    model.train()
    imgs, noise = torch.from_numpy(z["imgs"]), torch.from_numpy(z["noise"])
    loss, pred, mask = model(imgs, mask_ratio=float(z["mask_ratio"]), noise=noise)
    assert np.array_equal(mask.numpy(), z["mask"])                                          # the same patches are removed
    assert abs(float(loss) - float(z["loss"])) < 2e-6 and np.allclose(pred.detach().numpy(), z["pred"], atol=5e-6)
    loss.backward()
    names = [n for n, p in model.named_parameters() if p.grad is not None]
    assert sorted(names) == sorted(str(n) for n in z["grad_names"])
    for n, p in model.named_parameters():
        if p.grad is not None:
            ref = z["grad." + n]
            assert np.allclose(p.grad.numpy(), ref, rtol=2e-4, atol=2e-7 + 1e-4 * np.abs(ref).max()), n
    model.norm_pix_loss = False
    assert abs(float(model(imgs, 0.75, noise)[0]) - float(z["loss_plain"])) < 2e-6
    # the reference's own draw: without `noise` the mask comes from torch.rand under the caller's seed
    torch.manual_seed(1203)
    assert np.array_equal(model(imgs, 0.75)[2].numpy(), z["mask"])


def test_mae_train_step_and_lr_schedule():
    from snuffy_amd import ssl_pretrain as S
    z = _npz("f12_mae_pretrain.npz")
    model = _mae()
    model.load_state_dict(_sd(z))
    model.train()
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.95))
    before = model.blocks[0].adaptmlp.down_proj.weight.detach().clone()
    loss = S.mae_train_step(model, torch.from_numpy(z["imgs"]), opt, 0.75, torch.from_numpy(z["noise"]), lr=5e-4)
    assert abs(float(loss) - float(z["loss"])) < 2e-6 and opt.param_groups[0]["lr"] == 5e-4
    assert not torch.equal(before, model.blocks[0].adaptmlp.down_proj.weight) and model.pos_embed.grad is None
    assert S.adjust_learning_rate(1e-3, 0.0, 0.5, 2, 10) == 1e-3 * 0.25
    assert abs(S.adjust_learning_rate(1e-3, 1e-5, 6.0, 2, 10) - (1e-5 + (1e-3 - 1e-5) * 0.5 * (1 + np.cos(np.pi * 0.5)))) < 1e-12


# ---- two ranks (gloo): the centre of the DINO loss is the mean over ALL ranks' teacher outputs, the gradients are averaged ----------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dino_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from snuffy_amd import ssl_pretrain as S
    from snuffy_amd.train import FlatGradAllReduce
    z = _npz("f11_dino_pretrain.npz")
    student, teacher = _student(), _student()
    student.load_state_dict(_sd(z)), teacher.load_state_dict(_sd(z))
    for p in teacher.parameters():
        p.requires_grad = False
    trainable = S.freeze_for_adapter_tuning(student)
    student.train(), teacher.train()
    g = torch.Generator().manual_seed(50 + rank)
    crops = [torch.rand(2, 3, 64, 64, generator=g), torch.rand(2, 3, 64, 64, generator=g), torch.rand(2, 3, 32, 32, generator=g)]
    lm = S.DINOLoss(48, 3, 0.04, 0.07, 3, 10, dist=dist, world_size=world)
    opt = torch.optim.AdamW(S.get_params_groups(student), lr=1e-3, weight_decay=0.04)
    loss = S.dino_train_step(student, teacher, lm, crops, opt, epoch=1, it=0, momentum_schedule=np.array([0.99]),
                             grad_sync=FlatGradAllReduce(trainable, dist, world))
    with torch.no_grad():
        t_out = teacher(crops[:2])        # (after the EMA: only for the single-process reference's bookkeeping)
    out[rank] = dict(loss=float(loss), center=lm.center.clone(), w={k: v.clone() for k, v in student.state_dict().items() if "adaptmlp" in k},
                     crops=crops)
    dist.barrier()
    dist.destroy_process_group()


def test_dino_step_on_two_ranks_equals_one_process_on_both_batches():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dino_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for k in out[0]["w"]:
        assert torch.equal(out[0]["w"][k], out[1]["w"][k]), k                               # replicas stay identical
    assert torch.equal(out[0]["center"], out[1]["center"])
    # single process: both ranks' losses, averaged gradient, centre over the union of the teacher outputs
    from snuffy_amd import ssl_pretrain as S
    z = _npz("f11_dino_pretrain.npz")
    student, teacher = _student(), _student()
    student.load_state_dict(_sd(z)), teacher.load_state_dict(_sd(z))
    S.freeze_for_adapter_tuning(student)
    student.train(), teacher.train()
    opt = torch.optim.AdamW(S.get_params_groups(student), lr=1e-3, weight_decay=0.04)
    t_all, total = [], 0.0
    for r in range(world):
        crops = out[r]["crops"]
        with torch.no_grad():
            t_out = teacher(crops[:2])
        t_all.append(t_out)
        lm = S.DINOLoss(48, 3, 0.04, 0.07, 3, 10)
        loss = lm(student(crops), t_out, 1)
        assert abs(float(loss) - out[r]["loss"]) < 1e-6
        (loss / world).backward()
    S.cancel_gradients_last_layer(1, student, 1)
    opt.step()
    center = torch.cat(t_all).mean(dim=0, keepdim=True) * (1 - 0.9)
    assert torch.allclose(center, out[0]["center"], atol=1e-7)
    for k, v in student.state_dict().items():
        if "adaptmlp" in k:
            assert torch.allclose(v, out[0]["w"][k], atol=1e-6), k


def test_pretraining_loops_learn_and_write_the_checkpoints_the_extractor_loads(tmp_path):
    """Two short runs on synthetic tiles: the losses fall, only the adapters (+ head) move, the checkpoints carry the keys
    compute_feats.py:493-504 reads ('teacher' / 'model') and load back into a fresh backbone."""
    from snuffy_amd import ssl_pretrain as S
    torch.manual_seed(3)
    g = torch.Generator().manual_seed(4)
    tiles = [torch.rand(4, 3, 64, 64, generator=g) for _ in range(3)]
    student, teacher = _student(), _student()
    with torch.no_grad():
        for m in student.modules():                      # LoRA start: the up-projection is zero -- give the adapters something to do
            if hasattr(m, "up_proj"):
                m.up_proj.weight.normal_(0, 0.02, generator=g)
    trunk_before = student.backbone.blocks[0].attn.qkv.weight.detach().clone()
    ad_before = student.backbone.blocks[0].adaptmlp.down_proj.weight.detach().clone()
    aug = S.MultiCropAugment(global_size=64, local_size=32, local_crops_number=2, generator=g)
    ck = str(tmp_path / "dino.pth")
    losses = S.pretrain_dino(student, teacher, lambda e: iter(tiles), epochs=4, niter_per_ep=3, out_dim=48, ncrops=4, lr=2e-3,
                             warmup_teacher_temp_epochs=0, augment=aug, checkpoint_path=ck)
    assert len(losses) == 12 and all(np.isfinite(losses))
    assert torch.equal(trunk_before, student.backbone.blocks[0].attn.qkv.weight)            # frozen trunk
    assert not torch.equal(ad_before, student.backbone.blocks[0].adaptmlp.down_proj.weight)
    saved = torch.load(ck, map_location="cpu")
    assert set(saved) == {"student", "teacher", "epoch"} and saved["epoch"] == 4
    fresh = _student()
    fresh.load_state_dict(saved["teacher"], strict=True)
    assert torch.equal(fresh.backbone.blocks[1].adaptmlp.up_proj.weight, teacher.backbone.blocks[1].adaptmlp.up_proj.weight)

    model = _mae()
    with torch.no_grad():
        for m in model.modules():
            if hasattr(m, "up_proj"):
                m.up_proj.weight.normal_(0, 0.02, generator=g)
    ck2 = str(tmp_path / "mae.pth")
    mlosses = S.pretrain_mae(model, lambda e: iter(tiles), epochs=6, niter_per_ep=3, lr=5e-3, checkpoint_path=ck2)
    assert np.mean(mlosses[-3:]) < np.mean(mlosses[:3])                                     # reconstruction improves on the three tiles
    saved = torch.load(ck2, map_location="cpu")
    assert set(saved) == {"model", "epoch"} and "decoder_pred.weight" in saved["model"] and "blocks.0.adaptmlp.down_proj.weight" in saved["model"]
