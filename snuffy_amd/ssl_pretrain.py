"""Adapter pre-training of the ViT extractor (last part): DINO self-distillation and MAE reconstruction with adapters.

Counterparts in the reference (what a user of `dino_adapter/main_dino_adapter.py` / `mae_adapter/main_pretrain_adapter.py` needs):

  differentiable forward of snuffy_amd.vit.VisionTransformer      dino_adapter/vision_transformer_with_adapter.py:97-127, 218-236;
                                                                  mae_adapter/timm_modified/models/vision_transformer.py:151-156
  DINOHead, MultiCropWrapper                                      vision_transformer_with_adapter.py:279-314, dino_adapter/utils.py:609-645
  DINOLoss (+ centre EMA over all ranks)                          main_dino_adapter.py:618-672
  cosine_scheduler / clip_gradients / cancel_gradients_last_layer /
  get_params_groups / teacher EMA                                 dino_adapter/utils.py:137-155, 192-203, 648-659; main_dino_adapter.py:548-552
  adapter tuning (everything frozen but adapters + head)          main_dino_adapter.py:307-314
  MAEPretrainModel (encoder + decoder + masked-patch loss)        mae_adapter/models_mae.py:21-246
  dino_train_step / mae_train_step                                main_dino_adapter.py:485-552, mae_adapter/engine_pretrain.py:21-81

This is a TRAINING path: it runs on PyTorch-ROCm autograd (library GEMMs); the hand-written inference kernels of the extractor
(snuffy_amd/vit.py under torch.no_grad) are not differentiable and are not used here.  State-dict keys are the reference's, so a
checkpoint written here ('teacher' / 'model') loads into compute_feats exactly like the reference's (compute_feats.py:441-504).
Several ranks: one process per GPU, gradients of the trainable (adapter + head) parameters averaged with ONE flat all-reduce per
step (train.FlatGradAllReduce; backend "nccl" = RCCL), the DINO centre with one all-reduce of [1, out_dim].
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import vit


# ----------------------------------------------------------------------------------------------------------------------
# differentiable ViT forward over the modules of snuffy_amd.vit (same parameters, torch ops)
# ----------------------------------------------------------------------------------------------------------------------
def _patch_tokens(model, x):
    """Conv2d patch embedding as a convolution: [B, 3, H, W] -> [B, P, D]."""
    pe = model.patch_embed
    return F.conv2d(x, pe.proj.weight, pe.proj.bias, stride=pe.patch_size).flatten(2).transpose(1, 2)


def _attention(attn, x):
    b, t, d = x.shape
    h = attn.num_heads
    qkv = F.linear(x, attn.qkv.weight, attn.qkv.bias).view(b, t, 3, h, d // h).permute(2, 0, 3, 1, 4)
    p = ((qkv[0] @ qkv[1].transpose(-2, -1)) * attn.scale).softmax(dim=-1)
    return F.linear((p @ qkv[2]).transpose(1, 2).reshape(b, t, d), attn.proj.weight, attn.proj.bias)


def _adapter(ad, x, training):
    """scale * up(dropout(ReLU(down(x)))) (adapter.py:74-94, add_residual=False, layernorm option "none")."""
    if ad.adapter_layernorm_option != "none":
        raise NotImplementedError("adapter pre-training supports adapter_layernorm_option='none' (the reference's recipes)")
    down = F.dropout(F.relu(F.linear(x, ad.down_proj.weight, ad.down_proj.bias)), p=float(ad.dropout), training=training)
    return F.linear(down, ad.up_proj.weight, ad.up_proj.bias) * ad.scale


def block_autograd(blk, x, training):
    """x + attn(norm1 x); then x + mlp(norm2 x) + adapter(x)  (drop_path is 0 in every recipe of the reference)."""
    n1, n2 = blk.norm1, blk.norm2
    x = x + _attention(blk.attn, F.layer_norm(x, (x.shape[-1],), n1.weight, n1.bias, n1.eps))
    ad = _adapter(blk.adaptmlp, x, training) if hasattr(blk, "adaptmlp") else 0.0
    y = F.layer_norm(x, (x.shape[-1],), n2.weight, n2.bias, n2.eps)
    y = F.linear(F.gelu(F.linear(y, blk.mlp.fc1.weight, blk.mlp.fc1.bias)), blk.mlp.fc2.weight, blk.mlp.fc2.bias)
    return x + y + ad


def vit_forward_autograd(model, x):
    """The DINO extractor's forward with a graph: CLS token of LayerNorm(blocks(tokens))  (…with_adapter.py:218-236)."""
    b, _, w, h = x.shape
    tok = _patch_tokens(model, x)
    tok = torch.cat((model.cls_token.expand(b, -1, -1), tok), dim=1)
    tok = tok + model.interpolate_pos_encoding(tok, w, h)
    for blk in model.blocks:
        tok = block_autograd(blk, tok, model.training)
    return F.layer_norm(tok, (tok.shape[-1],), model.norm.weight, model.norm.bias, model.norm.eps)[:, 0]


class TrainableBackbone(nn.Module):
    """A snuffy_amd.vit.VisionTransformer behind an autograd forward (its own forward() is the no_grad kernel path)."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, x):
        return vit_forward_autograd(self.model, x)


# ----------------------------------------------------------------------------------------------------------------------
# DINO
# ----------------------------------------------------------------------------------------------------------------------
class DINOHead(nn.Module):
    """MLP (nlayers, GELU) -> l2-normalise -> weight-normalised linear without bias.  Keys: mlp.{0,2,4}.*, last_layer.weight_{g,v}."""

    def __init__(self, in_dim, out_dim, use_bn=False, norm_last_layer=True, nlayers=3, hidden_dim=2048, bottleneck_dim=256):
        super().__init__()
        if use_bn:
            raise NotImplementedError("DINOHead: use_bn is off in every recipe of the reference")
        widths = [in_dim] + [hidden_dim] * (max(nlayers, 1) - 1) + [bottleneck_dim]
        if len(widths) == 2:
            self.mlp = nn.Linear(in_dim, bottleneck_dim)
        else:
            layers = []
            for i in range(len(widths) - 1):
                layers.append(nn.Linear(widths[i], widths[i + 1]))
                if i + 2 < len(widths):
                    layers.append(nn.GELU())
            self.mlp = nn.Sequential(*layers)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                nn.init.zeros_(m.bias)
        self.last_layer = nn.utils.weight_norm(nn.Linear(bottleneck_dim, out_dim, bias=False))
        self.last_layer.weight_g.data.fill_(1)
        if norm_last_layer:
            self.last_layer.weight_g.requires_grad = False

    def forward(self, x):
        return self.last_layer(F.normalize(self.mlp(x), dim=-1, p=2))


class MultiCropWrapper(nn.Module):
    """backbone over the crops, one pass per run of equal resolutions, head over the concatenated features.  Keys: backbone.*, head.*
    (the backbone's parameters sit directly under `backbone.`, as in the reference's checkpoints)."""

    def __init__(self, backbone, head):
        super().__init__()
        self.backbone = backbone            # a snuffy_amd.vit.VisionTransformer
        self.head = head

    def forward(self, crops):
        crops = crops if isinstance(crops, (list, tuple)) else [crops]
        feats, start = [], 0
        while start < len(crops):
            end = start + 1
            while end < len(crops) and crops[end].shape[-1] == crops[start].shape[-1]:
                end += 1
            feats.append(vit_forward_autograd(self.backbone, torch.cat(list(crops[start:end]))))
            start = end
        return self.head(torch.cat(feats))


class DINOLoss(nn.Module):
    """Cross-entropy between the centred, sharpened teacher distribution of each global view and the student distribution of every
    OTHER view; the centre is an EMA of the teacher outputs' mean over all ranks."""

    def __init__(self, out_dim, ncrops, warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs, nepochs, student_temp=0.1,
                 center_momentum=0.9, dist=None, world_size=1):
        super().__init__()
        self.student_temp, self.center_momentum, self.ncrops = student_temp, center_momentum, ncrops
        self.dist, self.world_size = dist, world_size
        self.register_buffer("center", torch.zeros(1, out_dim))
        self.teacher_temp_schedule = np.concatenate((np.linspace(warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs),
                                                     np.ones(nepochs - warmup_teacher_temp_epochs) * teacher_temp))

    def forward(self, student_output, teacher_output, epoch):
        log_s = [F.log_softmax(c, dim=-1) for c in (student_output / self.student_temp).chunk(self.ncrops)]
        q_all = F.softmax((teacher_output - self.center) / self.teacher_temp_schedule[epoch], dim=-1).detach().chunk(2)
        total, terms = 0, 0
        for iq, q in enumerate(q_all):
            for v, ls in enumerate(log_s):
                if v != iq:                                   # never the same view on both sides
                    total = total + torch.sum(-q * ls, dim=-1).mean()
                    terms += 1
        self.update_center(teacher_output)
        return total / terms

    @torch.no_grad()
    def update_center(self, teacher_output):
        batch_center = torch.sum(teacher_output, dim=0, keepdim=True)
        if self.dist is not None and self.world_size > 1:
            self.dist.all_reduce(batch_center)
        batch_center = batch_center / (len(teacher_output) * self.world_size)
        self.center = self.center * self.center_momentum + batch_center * (1 - self.center_momentum)


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0):
    """Per-iteration schedule: linear warm-up, then half a cosine from base_value to final_value."""
    warm = warmup_epochs * niter_per_ep
    head = np.linspace(start_warmup_value, base_value, warm) if warmup_epochs > 0 else np.array([])
    it = np.arange(epochs * niter_per_ep - warm)
    tail = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * it / len(it)))
    out = np.concatenate((head, tail))
    assert len(out) == epochs * niter_per_ep
    return out


def clip_gradients(model, clip):
    """Per-parameter gradient-norm clipping (NOT a global norm): returns the norms before clipping."""
    norms = []
    for _, p in model.named_parameters():
        if p.grad is not None:
            n = p.grad.data.norm(2)
            norms.append(n.item())
            coef = clip / (n + 1e-6)
            if coef < 1:
                p.grad.data.mul_(coef)
    return norms


def cancel_gradients_last_layer(epoch, model, freeze_last_layer):
    if epoch < freeze_last_layer:
        for n, p in model.named_parameters():
            if "last_layer" in n:
                p.grad = None


def get_params_groups(model):
    """[weights (regularised), biases and 1-D parameters (weight_decay 0)] of the trainable parameters."""
    reg, plain = [], []
    for name, p in model.named_parameters():
        if p.requires_grad:
            (plain if name.endswith(".bias") or p.dim() == 1 else reg).append(p)
    return [{"params": reg}, {"params": plain, "weight_decay": 0.}]


def freeze_for_adapter_tuning(student):
    """Continued pre-training with adapters: the pre-trained trunk is frozen, the adapters and the projection head train."""
    for n, p in student.named_parameters():
        p.requires_grad = ("adaptmlp" in n) or n.startswith("head.")
    lg = getattr(student.head.last_layer, "weight_g", None)
    if lg is not None:
        lg.requires_grad = False
    return [p for p in student.parameters() if p.requires_grad]


@torch.no_grad()
def ema_update(student, teacher, momentum):
    for ps, pt in zip(student.parameters(), teacher.parameters()):
        pt.data.mul_(momentum).add_((1 - momentum) * ps.detach().data)


def dino_train_step(student, teacher, loss_mod, crops, optimizer, epoch, it=None, lr_schedule=None, wd_schedule=None,
                    momentum_schedule=None, clip_grad=0.0, freeze_last_layer=1, grad_sync=None):
    """One iteration of main_dino_adapter.train_one_epoch (fp32): schedules -> teacher on the 2 global views, student on all ->
    loss -> backward -> (all-reduce) -> per-parameter clipping -> last-layer freeze -> step -> teacher EMA.  Returns the loss."""
    if it is not None and lr_schedule is not None:
        for i, group in enumerate(optimizer.param_groups):
            group["lr"] = lr_schedule[it]
            if i == 0 and wd_schedule is not None:
                group["weight_decay"] = wd_schedule[it]
    with torch.no_grad():
        t_out = teacher(crops[:2])
    loss = loss_mod(student(crops), t_out, epoch)
    if not math.isfinite(float(loss.detach())):
        raise FloatingPointError("DINO loss is %r" % float(loss.detach()))
    optimizer.zero_grad()
    loss.backward()
    if grad_sync is not None:
        grad_sync()
    if clip_grad:
        clip_gradients(student, clip_grad)
    cancel_gradients_last_layer(epoch, student, freeze_last_layer)
    optimizer.step()
    if momentum_schedule is not None and it is not None:
        ema_update(student, teacher, momentum_schedule[it])
    return loss.detach()


# ----------------------------------------------------------------------------------------------------------------------
# MAE
# ----------------------------------------------------------------------------------------------------------------------
class MAEPretrainModel(vit.VisionTransformer):
    """Masked autoencoder over the adapter ViT: the encoder is the extractor's own VisionTransformer (same keys: patch_embed.*,
    cls_token, pos_embed, blocks.*, norm.*), plus the reference's light decoder (decoder_embed, mask_token, decoder_pos_embed,
    decoder_blocks.*, decoder_norm, decoder_pred).  forward(imgs, mask_ratio, noise=None) -> (loss, pred, mask)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16, decoder_embed_dim=512,
                 decoder_depth=8, decoder_num_heads=16, mlp_ratio=4., norm_layer=nn.LayerNorm, norm_pix_loss=False,
                 adapter_ffn_layernorm_option="none", adapter_ffn_init_option="lora", adapter_ffn_scalar="0.1", adapter_ffn_num=64,
                 adapter_d_model=768):
        super().__init__(img_size=[img_size], patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim, depth=depth,
                         num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=True, norm_layer=norm_layer,
                         adapter_ffn_layernorm_option=adapter_ffn_layernorm_option, adapter_ffn_init_option=adapter_ffn_init_option,
                         adapter_ffn_scalar=adapter_ffn_scalar, adapter_ffn_num=adapter_ffn_num, adapter_d_model=adapter_d_model,
                         pool="mean_patches")
        num_patches = self.patch_embed.num_patches
        self.in_chans, self.norm_pix_loss = in_chans, norm_pix_loss
        dec_ffn = int(adapter_ffn_num / adapter_d_model * decoder_embed_dim)        # models_mae.py:45-46
        self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, decoder_embed_dim), requires_grad=False)
        self.decoder_blocks = nn.ModuleList([
            vit.Block(dim=decoder_embed_dim, num_heads=decoder_num_heads, mlp_ratio=mlp_ratio, qkv_bias=True, norm_layer=norm_layer,
                      adapter_ffn_layernorm_option=adapter_ffn_layernorm_option, adapter_ffn_init_option=adapter_ffn_init_option,
                      adapter_ffn_scalar=adapter_ffn_scalar, adapter_ffn_num=dec_ffn, d_model=decoder_embed_dim,
                      adapter_d_model=decoder_embed_dim) for _ in range(decoder_depth)])
        self.decoder_norm = norm_layer(decoder_embed_dim)
        self.decoder_pred = nn.Linear(decoder_embed_dim, patch_size ** 2 * in_chans, bias=True)
        side = int(num_patches ** .5)
        with torch.no_grad():                                                       # fixed sin-cos tables, frozen
            self.pos_embed.copy_(torch.from_numpy(vit.get_2d_sincos_pos_embed(embed_dim, side, cls_token=True)).float().unsqueeze(0))
            self.decoder_pos_embed.copy_(torch.from_numpy(vit.get_2d_sincos_pos_embed(decoder_embed_dim, side, cls_token=True)).float().unsqueeze(0))
            w = self.patch_embed.proj.weight
            nn.init.xavier_uniform_(w.view(w.shape[0], -1))
            nn.init.normal_(self.cls_token, std=.02)
            nn.init.normal_(self.mask_token, std=.02)
        self.pos_embed.requires_grad = False

    def patchify(self, imgs):
        """[N, C, H, W] -> [N, L, p*p*C] with the pixel order (row in patch, column in patch, channel)."""
        p = self.patch_embed.patch_size
        n, c, hh, ww = imgs.shape
        assert hh == ww and hh % p == 0
        g = hh // p
        return imgs.reshape(n, c, g, p, g, p).permute(0, 2, 4, 3, 5, 1).reshape(n, g * g, p * p * c)

    @staticmethod
    def random_masking(x, mask_ratio, noise=None):
        """Keep the int(L (1 - ratio)) tokens with the smallest noise per sample.  -> (kept tokens, mask [N, L] with 1 = removed,
        ids_restore).  noise [N, L] (nullable: drawn with torch.rand, the reference's own draw)."""
        n, length, d = x.shape
        keep = int(length * (1 - mask_ratio))
        if noise is None:
            noise = torch.rand(n, length, device=x.device)
        shuffle = torch.argsort(noise, dim=1)
        restore = torch.argsort(shuffle, dim=1)
        kept = torch.gather(x, 1, shuffle[:, :keep].unsqueeze(-1).expand(-1, -1, d))
        mask = torch.ones(n, length, device=x.device)
        mask[:, :keep] = 0
        return kept, torch.gather(mask, 1, restore), restore

    def forward_encoder(self, imgs, mask_ratio, noise=None):
        x = _patch_tokens(self, imgs) + self.pos_embed[:, 1:, :]
        x, mask, restore = self.random_masking(x, mask_ratio, noise)
        x = torch.cat(((self.cls_token + self.pos_embed[:, :1, :]).expand(x.shape[0], -1, -1), x), dim=1)
        for blk in self.blocks:
            x = block_autograd(blk, x, self.training)
        return F.layer_norm(x, (x.shape[-1],), self.norm.weight, self.norm.bias, self.norm.eps), mask, restore

    def forward_decoder(self, latent, restore):
        x = self.decoder_embed(latent)
        n, length = restore.shape
        fill = self.mask_token.expand(n, length + 1 - x.shape[1], -1)
        body = torch.gather(torch.cat([x[:, 1:, :], fill], dim=1), 1, restore.unsqueeze(-1).expand(-1, -1, x.shape[2]))
        x = torch.cat([x[:, :1, :], body], dim=1) + self.decoder_pos_embed
        for blk in self.decoder_blocks:
            x = block_autograd(blk, x, self.training)
        x = F.layer_norm(x, (x.shape[-1],), self.decoder_norm.weight, self.decoder_norm.bias, self.decoder_norm.eps)
        return self.decoder_pred(x)[:, 1:, :]

    def forward_loss(self, imgs, pred, mask):
        target = self.patchify(imgs)
        if self.norm_pix_loss:
            target = (target - target.mean(dim=-1, keepdim=True)) / (target.var(dim=-1, keepdim=True) + 1.e-6) ** .5
        per_patch = ((pred - target) ** 2).mean(dim=-1)
        return (per_patch * mask).sum() / mask.sum()                                 # mean over the REMOVED patches

    def forward(self, imgs, mask_ratio=0.75, noise=None):
        if not torch.is_grad_enabled() and not self.training:
            return super().forward(imgs)                                             # the extractor (compute_feats path)
        latent, mask, restore = self.forward_encoder(imgs, mask_ratio, noise)
        pred = self.forward_decoder(latent, restore)
        return self.forward_loss(imgs, pred, mask), pred, mask


def mae_train_step(model, imgs, optimizer, mask_ratio=0.75, noise=None, grad_sync=None, lr=None, accum_iter=1, update=True):
    """One iteration of mae_adapter/engine_pretrain.train_one_epoch (fp32, no loss scaler): loss / accum_iter -> backward ->
    (all-reduce) -> step every accum_iter iterations.  Returns the (un-divided) loss."""
    if lr is not None:
        for group in optimizer.param_groups:
            group["lr"] = lr * group.get("lr_scale", 1.0)
    loss, _, _ = model(imgs, mask_ratio=mask_ratio, noise=noise)
    if not math.isfinite(float(loss.detach())):
        raise FloatingPointError("MAE loss is %r" % float(loss.detach()))
    (loss / accum_iter).backward()
    if update:
        if grad_sync is not None:
            grad_sync()
        optimizer.step()
        optimizer.zero_grad()
    return loss.detach()


def adjust_learning_rate(base_lr, min_lr, epoch_float, warmup_epochs, epochs):
    """Half-cycle cosine after a linear warm-up, evaluated per iteration (mae_adapter/util/lr_sched.py)."""
    if epoch_float < warmup_epochs:
        return base_lr * epoch_float / warmup_epochs
    return min_lr + (base_lr - min_lr) * 0.5 * (1. + math.cos(math.pi * (epoch_float - warmup_epochs) / (epochs - warmup_epochs)))


# ----------------------------------------------------------------------------------------------------------------------
# loops and checkpoints (main_dino_adapter.train_dino:159-483, main_pretrain_adapter.main:137-420 without wandb / torchvision)
# ----------------------------------------------------------------------------------------------------------------------
class MultiCropAugment:
    """Stand-in for DataAugmentationDINO (main_dino_adapter.py:674-745) on tile TENSORS [B, 3, H, W] in [0, 1]: 2 global crops
    (scale in global_scale, 224-class size) + n local crops (scale in local_scale, 96-class size), random resized crop + horizontal
    flip, on the device.  The reference's colour jitter / blur / solarisation are torchvision PIL transforms (not available here);
    the training step is indifferent to where its crops come from."""

    def __init__(self, global_size=224, local_size=96, local_crops_number=8, global_scale=(0.4, 1.0), local_scale=(0.05, 0.4), generator=None):
        self.gs, self.ls, self.n = global_size, local_size, local_crops_number
        self.gscale, self.lscale, self.gen = global_scale, local_scale, generator

    def _crop(self, x, size, scale):
        b, _, h, w = x.shape
        area = float(torch.empty(1).uniform_(scale[0], scale[1], generator=self.gen)) * h * w
        ratio = math.exp(float(torch.empty(1).uniform_(math.log(3 / 4), math.log(4 / 3), generator=self.gen)))
        cw, ch = min(w, max(1, int(round(math.sqrt(area * ratio))))), min(h, max(1, int(round(math.sqrt(area / ratio)))))
        x0 = int(torch.randint(0, w - cw + 1, (1,), generator=self.gen))
        y0 = int(torch.randint(0, h - ch + 1, (1,), generator=self.gen))
        out = F.interpolate(x[:, :, y0:y0 + ch, x0:x0 + cw], size=(size, size), mode="bilinear", align_corners=False)
        return out.flip(-1) if float(torch.rand(1, generator=self.gen)) < 0.5 else out

    def __call__(self, x):
        return [self._crop(x, self.gs, self.gscale) for _ in range(2)] + [self._crop(x, self.ls, self.lscale) for _ in range(self.n)]


def pretrain_dino(student, teacher, batches, epochs, niter_per_ep, out_dim, ncrops, lr=5e-4, min_lr=1e-6, weight_decay=0.04,
                  weight_decay_end=0.4, momentum_teacher=0.996, warmup_epochs=0, warmup_teacher_temp=0.04, teacher_temp=0.04,
                  warmup_teacher_temp_epochs=0, clip_grad=3.0, freeze_last_layer=1, augment=None, dist=None, world_size=1,
                  checkpoint_path=None, log=None):
    """Adapter tuning by self-distillation: `batches(epoch)` yields tile tensors (already on the device) -- or lists of crops when
    augment is None --, niter_per_ep of them per epoch.  Teacher starts from the student's weights and follows it by EMA; only the
    adapters and the head train; one flat gradient all-reduce per step across `world_size` ranks.  Writes {'student', 'teacher',
    'epoch'} after every epoch (the extractor loads 'teacher', compute_feats.py:493-504).  Returns the per-iteration losses."""
    from .train import FlatGradAllReduce
    teacher.load_state_dict(student.state_dict())
    for p in teacher.parameters():
        p.requires_grad = False
    trainable = freeze_for_adapter_tuning(student)
    dev = next(student.parameters()).device
    loss_mod = DINOLoss(out_dim, ncrops, warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs, epochs, dist=dist,
                        world_size=world_size).to(dev)
    optimizer = torch.optim.AdamW(get_params_groups(student))
    lr_s = cosine_scheduler(lr, min_lr, epochs, niter_per_ep, warmup_epochs=warmup_epochs)
    wd_s = cosine_scheduler(weight_decay, weight_decay_end, epochs, niter_per_ep)
    mom_s = cosine_scheduler(momentum_teacher, 1, epochs, niter_per_ep)
    sync = FlatGradAllReduce(trainable, dist, world_size)
    losses = []
    student.train(), teacher.train()
    for epoch in range(epochs):
        for i, batch in zip(range(niter_per_ep), batches(epoch)):
            crops = augment(batch) if augment is not None else batch
            losses.append(float(dino_train_step(student, teacher, loss_mod, crops, optimizer, epoch, epoch * niter_per_ep + i, lr_s, wd_s,
                                                mom_s, clip_grad, freeze_last_layer, sync)))
        if log is not None:
            log("epoch %d  loss %.4f" % (epoch, float(np.mean(losses[-niter_per_ep:]))))
        if checkpoint_path is not None and (dist is None or world_size == 1 or dist.get_rank() == 0):
            torch.save({"student": student.state_dict(), "teacher": teacher.state_dict(), "epoch": epoch + 1}, checkpoint_path)
    return losses


def pretrain_mae(model, batches, epochs, niter_per_ep, lr=1.5e-4, min_lr=0.0, warmup_epochs=0, weight_decay=0.05, mask_ratio=0.75,
                 dist=None, world_size=1, checkpoint_path=None, log=None, tune_adapters_only=True):
    """MAE reconstruction with adapters: per-iteration half-cosine learning rate, AdamW (0.9, 0.95), one flat gradient all-reduce per
    step.  tune_adapters_only: the pre-trained trunk is frozen, adapters train (main_pretrain_adapter.py freezes everything the
    loaded checkpoint provides).  Writes {'model', 'epoch'} after every epoch (the extractor loads 'model')."""
    from .train import FlatGradAllReduce
    if tune_adapters_only:
        for n, p in model.named_parameters():
            p.requires_grad = "adaptmlp" in n
    trainable = [p for p in model.parameters() if p.requires_grad]
    optimizer = torch.optim.AdamW(trainable, lr=lr, betas=(0.9, 0.95), weight_decay=weight_decay)
    sync = FlatGradAllReduce(trainable, dist, world_size)
    losses = []
    model.train()
    for epoch in range(epochs):
        for i, imgs in zip(range(niter_per_ep), batches(epoch)):
            cur = adjust_learning_rate(lr, min_lr, epoch + i / niter_per_ep, warmup_epochs, epochs)
            losses.append(float(mae_train_step(model, imgs, optimizer, mask_ratio, grad_sync=sync, lr=cur)))
        if log is not None:
            log("epoch %d  loss %.4f" % (epoch, float(np.mean(losses[-niter_per_ep:]))))
        if checkpoint_path is not None and (dist is None or world_size == 1 or dist.get_rank() == 0):
            torch.save({"model": model.state_dict(), "epoch": epoch + 1}, checkpoint_path)
    return losses
