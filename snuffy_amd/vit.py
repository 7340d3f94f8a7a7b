"""MI355X-native ViT patch-embedding extractors (the L2 stage of the reference: compute_feats.py -> IClassifier ->
VisionTransformer.forward).

Mirrors the module API and state-dict keys of the reference's inference-time model files:
  utils_ssls_cf/vision_transformer_with_adapter_dino_version.py   Mlp:51 Attention:70 Block:97 PatchEmbed:130
                                                                  VisionTransformer:149 vit_tiny/small/base:258-276
  utils_ssls_cf/vision_transformer_dino.py                        same without the adapter
  utils_ssls_cf/adapter.py                                        Adapter:34-94
  utils_ssls_cf/models_adapter_mae.py                             MaskedAutoencoderViT encoder:174-195
(keys: cls_token, pos_embed, patch_embed.proj.*, blocks.{i}.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2,
 adaptmlp.down_proj,adaptmlp.up_proj}.*, norm.*).

Inference only (parameters of the extractor are frozen in the reference, compute_feats.py:432-433).  Hand-written HIP:
patchify, token assembly, LayerNorm, residual+LayerNorm fusion, GELU epilogue, multi-head self-attention (exact fp32 and
bf16 MFMA).  Every dense contraction of a block -- patch-embed, qkv, proj, fc1 (+ erf GELU), fc2 and the adapter's down / up projections
(a bottleneck below 64, the DINO recipe's 32 included, is zero-padded to the kernel's 64-deep minimum) -- runs on the hand-written
MFMA GEMM (csrc/gemm.hip), in both precisions.  ``configure(precision=...)``: "bf16" = bf16 GEMM / MFMA operands with an fp32
residual stream (round 6: the blocks' LayerNorms folded into the consumer GEMMs, the residual adds into the producers' epilogues);
"fp32" = fp32 tensors, self-attention and projections as split-bf16 x3 products on the matrix cores
(FP32_GEMM = "x3": every operand as hi + lo bf16 halves over a tripled K axis, fp32 accumulate -- fp32-class, ~1e-5 per product;
FP32_GEMM = "library" restores plain fp32 library GEMMs).
"""
import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as SF
from . import ops
from ._ffi import SnuffyHipError

# fp32 path, the dense projections: "x3" = split-bf16 x3 products on the hand-written MFMA GEMM (as the aggregator's fp32 path,
# functional.FP32_GEMM), "library" = plain fp32 library GEMMs
FP32_GEMM = "x3"
# bf16 path: the blocks with the LayerNorms folded into the consumer GEMMs and the residual adds into the producers' epilogues
# (VisionTransformer._blocks_fused, round 6); False = the round-2 block (residual_ln passes between the GEMMs)
BF16_FUSED_BLOCK = True


def _image_of(weight, fmt):
    """Split image of a parameter ("hl": interleaved [hi(32) | lo(32)], ops.split_hl_weight; "cat": [Wh | Wl | Wh],
    ops.split3_weight), cached on the parameter until it is written again."""
    key = SF.param_key(weight)
    name = "_snf_img_" + fmt
    hit = getattr(weight, name, None)
    if hit is None or hit[0] != key:
        w2 = weight.detach().reshape(weight.shape[0], -1)
        hit = (key, ops.split_hl_weight(w2) if fmt == "hl" else ops.split3_weight(w2))
        setattr(weight, name, hit)
    return hit[1]


class SplitImage:
    """An fp32 activation held as its bf16 split image: fmt "hl" ([m, 2 k], interleaved) or "cat" ([m, 3 k] = [hi | hi | lo])."""

    def __init__(self, data, fmt, k):
        self.data, self.fmt, self.k = data, fmt, k

    def to_f32(self):
        m = self.data.shape[0]
        if self.fmt == "hl":
            v = self.data.view(m, self.k // 32, 2, 32).float()
            return (v[:, :, 0] + v[:, :, 1]).reshape(m, self.k)
        return self.data[:, :self.k].float() + self.data[:, 2 * self.k:].float()


def image_format(m, weight):
    """Format of the split image a projection with `weight` wants for m rows: "hl" (one-pass kernel) where 256 x 256 tiles fill the
    chip, "cat" (concatenated K) where only the 32-deep-step kernel applies, None = plain fp32 (library GEMM)."""
    n, k = weight.shape[0], weight[0].numel()
    if FP32_GEMM != "x3":
        return None
    if ops.hl_eligible(m, n, k):
        return "hl"
    return "cat" if (k % 8 == 0 and ops.gemm_x3_supported(m, n, k)) else None


def split_image(x, fmt):
    x = x.float() if x.dtype != torch.float32 else x
    return SplitImage(ops.split_hl_rows(x) if fmt == "hl" else ops.split3_rows(x), fmt, x.shape[1])


def layernorm_image(x2, norm, fmt):
    """LayerNorm(x2) as the image a following projection wants (written by the LayerNorm kernel itself), or plain fp32."""
    if fmt == "hl" and x2.shape[1] % 32 == 0:
        return SplitImage(ops.layernorm_rows_hl(x2, norm.weight, norm.bias, norm.eps), "hl", x2.shape[1])
    if fmt == "cat":
        return SplitImage(ops.layernorm_rows_split3(x2, norm.weight, norm.bias, norm.eps), "cat", x2.shape[1])
    y = ops.layernorm_rows(x2, norm.weight, norm.bias, norm.eps)
    return split_image(y, fmt) if fmt else y


def linear_f32(x, weight, bias, act="none", image_for=None, resid=None, scale=None):
    """act(x W^T + b) of the fp32 path.  x: [m, k] f32 or a SplitImage.  FP32_GEMM == "x3": split-bf16 x3 products on the hand-written
    MFMA GEMMs -- the one-pass kernel on interleaved images where the shape fills the chip, the concatenated form otherwise -- else
    (or for shapes outside both kernels' domains) the fp32 library GEMM + the activation kernel.  image_for = the weight of the
    NEXT projection: the result is returned as the split image that projection wants, straight from this GEMM's epilogue (it never
    exists in fp32).  resid [m, n] f32 (fp32 results only): added to the result -- in the one-pass kernel's epilogue, so a block's
    residual adds cost no pass of their own (round 6).  scale: the result is scale * (x W^T + b) (the adapter's scalar; act "none")."""
    m = x.data.shape[0] if isinstance(x, SplitImage) else x.shape[0]
    n = weight.shape[0]
    if scale is not None:
        weight, bias = _scaled(weight, bias, float(scale))
    fmt = image_format(m, weight)
    out_fmt = image_format(m, image_for) if image_for is not None else None
    if out_fmt == "hl" and n % 32:
        out_fmt = "cat"
    b = None if bias is None else bias.detach().float().contiguous()
    if fmt is not None:
        if not isinstance(x, SplitImage):
            x = split_image(x, fmt)
        elif x.fmt != fmt:
            x = split_image(x.to_f32(), fmt)
        w_img = _image_of(weight, fmt)
        if fmt == "hl":
            if out_fmt == "hl":
                return SplitImage(ops.gemm_hl(x.data, w_img, b, act, hl_out=True), "hl", n)
            y = ops.gemm_hl(x.data, w_img, b, act, resid=resid if out_fmt is None else None)
            if resid is not None and out_fmt is not None:
                y = y + resid
        else:
            if out_fmt == "cat" and resid is None:
                return SplitImage(ops.gemm_x3(x.data, w_img, b, act, split3=True), "cat", n)
            y = ops.gemm_x3(x.data, w_img, b, act, out_dtype=torch.float32)
            if resid is not None:
                y += resid
        return split_image(y, out_fmt) if out_fmt else y
    if isinstance(x, SplitImage):
        x = x.to_f32()
    h = torch.mm(x.float(), weight.reshape(n, -1).t())
    if act != "none" or bias is not None:
        if act == "none":
            h += bias
        else:
            ops.bias_act_(h, bias, act)
    if resid is not None:
        h += resid
    return split_image(h, out_fmt) if out_fmt else h


def _scaled(weight, bias, scale):
    """(scale * W, scale * b), cached on the weight until it (or the bias) is written again: the adapter's `up * scale`
    (adapter.py:88) folded into its up-projection."""
    key = (SF.param_key(weight), None if bias is None else SF.param_key(bias), scale)
    hit = getattr(weight, "_snf_scaled", None)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            w = (weight.detach() * scale).contiguous()
            b = None if bias is None else (bias.detach() * scale).contiguous()
        hit = (key, w, b)
        try:
            weight._snf_scaled = hit
        except (AttributeError, RuntimeError):
            pass
    return hit[1], hit[2]


class Adapter(nn.Module):
    """scale * up(ReLU(down(x))) bottleneck adapter.  Reference utils_ssls_cf/adapter.py:34-94."""

    def __init__(self, adapter_d_model, d_model=None, bottleneck=None, dropout=0.0, init_option="bert",
                 adapter_scalar="1.0", adapter_layernorm_option="in"):
        super().__init__()
        self.n_embd = adapter_d_model if d_model is None else d_model
        self.down_size = bottleneck
        self.adapter_layernorm_option = adapter_layernorm_option
        self.adapter_layer_norm_before = None
        if adapter_layernorm_option in ("in", "out"):
            self.adapter_layer_norm_before = nn.LayerNorm(self.n_embd)
        if adapter_scalar == "learnable_scalar":
            self.scale = nn.Parameter(torch.ones(1))
        else:
            self.scale = float(adapter_scalar)
        self.down_proj = nn.Linear(self.n_embd, self.down_size)
        self.non_linear_func = nn.ReLU()
        self.up_proj = nn.Linear(self.down_size, self.n_embd)
        self.dropout = dropout
        if init_option == "lora":
            self._init_lora()
        elif init_option == "bert":
            raise NotImplementedError("init_option='bert' is not implemented (neither is it in the reference, adapter.py:63)")

    @torch.no_grad()
    def _init_lora(self):
        """LoRA-style start: random down-projection, ZERO up-projection -> the adapter starts as the identity branch."""
        nn.init.kaiming_normal_(self.down_proj.weight, a=math.sqrt(5))
        for t in (self.down_proj.bias, self.up_proj.weight, self.up_proj.bias):
            t.zero_()

    def forward(self, x, add_residual=True, residual=None):
        residual = x if residual is None else residual
        if self.adapter_layernorm_option == 'in':
            x = self.adapter_layer_norm_before(x)
        shp = x.shape
        if x.is_cuda and not torch.is_grad_enabled():
            a = linear_f32(x.reshape(-1, shp[-1]).float().contiguous(), self.down_proj.weight, self.down_proj.bias, "relu",
                           image_for=self.up_proj.weight)
            up = linear_f32(a, self.up_proj.weight, self.up_proj.bias).view(shp) * self.scale
        else:
            up = self.up_proj(F.relu(self.down_proj(x))) * self.scale   # eval: the reference's dropout is inactive
        if self.adapter_layernorm_option == 'out':
            up = self.adapter_layer_norm_before(up)
        return up + residual if add_residual else up


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x, resid=None):
        """x: [..., D] f32, or the SplitImage of the normalised tokens (Block.forward).  resid (nullable, [m, D] f32): added to the
        output in fc2's epilogue."""
        image_in = isinstance(x, SplitImage)
        x2 = x if image_in else x.reshape(-1, x.shape[-1]).float().contiguous()
        h = linear_f32(x2, self.fc1.weight, self.fc1.bias, "gelu", image_for=self.fc2.weight)
        out = linear_f32(h, self.fc2.weight, self.fc2.bias, resid=resid)
        return out if image_in else out.view(*x.shape[:-1], -1)


class Attention(nn.Module):
    """Reference …dino_version.py:70-94: returns (x, attn)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        return self._run(x.reshape(-1, x.shape[-1]).float().contiguous(), x.shape[0], x.shape[1])

    def _run(self, x2, B, N, need_attn=True, resid=None):
        """x2: [B * N, C] f32 or its SplitImage.  resid [B * N, C] (nullable): added to the projection's output in its epilogue."""
        C = self.proj.weight.shape[0]
        qkv = linear_f32(x2, self.qkv.weight, self.qkv.bias)
        # fp32-class arithmetic (the projections' own): the attention on the matrix cores too; exact fp32 with the plain-fp32 GEMMs.
        # Where the proj GEMM is the one-pass kernel the attention writes its operand image directly (no fp32 O, no split pass)
        x3 = FP32_GEMM == "x3" and not need_attn and ops.vit_mfma_attention_supported(N, C // self.num_heads)
        hl = x3 and C % 32 == 0 and image_format(B * N, self.proj.weight) == "hl"
        o, attn = ops.vit_attention(qkv, B, N, self.num_heads, self.scale, need_attn=need_attn,
                                    arithmetic="x3" if FP32_GEMM == "x3" else "exact", hl_out=hl)
        if hl:
            o = SplitImage(o, "hl", C)
        return linear_f32(o, self.proj.weight, self.proj.bias, resid=resid).view(B, N, C), attn


class Block(nn.Module):
    """x += Attn(LN1 x); x = x + MLP(LN2 x) [+ Adapter(x)].  Reference …dino_version.py:97-127 / vision_transformer_dino.py:96."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 adapter_ffn_layernorm_option="none", adapter_ffn_init_option="lora",
                 adapter_ffn_scalar="0.1", adapter_ffn_num=64, d_model=768, adapter_d_model=768, use_adapter=True):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        if use_adapter:
            self.adaptmlp = Adapter(bottleneck=adapter_ffn_num, dropout=0.1, adapter_d_model=adapter_d_model,
                                    d_model=d_model, init_option=adapter_ffn_init_option,
                                    adapter_scalar=adapter_ffn_scalar,
                                    adapter_layernorm_option=adapter_ffn_layernorm_option)

    def forward(self, x, return_attention=False):
        B, N, C = x.shape
        x2 = x.reshape(B * N, C).float().contiguous()
        # x3: the LayerNorm writes its output straight as the split image the projection GEMM reads
        ln1 = layernorm_image(x2, self.norm1, image_format(B * N, self.attn.qkv.weight))
        if return_attention:
            return self.attn._run(ln1, B, N, need_attn=True)[1]
        # round 6: every residual add of the block rides in the epilogue of the GEMM that produces the addend (one-pass kernel's
        # `resid`), the adapter's scalar is folded into its up-projection: no elementwise pass is left between the GEMMs
        x2 = self.attn._run(ln1, B, N, need_attn=False, resid=x2)[0].reshape(B * N, C)            # x + attn(norm1(x))
        ln2 = layernorm_image(x2, self.norm2, image_format(B * N, self.mlp.fc1.weight))
        if hasattr(self, "adaptmlp") and self.adaptmlp.adapter_layernorm_option == "none" and not torch.is_grad_enabled() \
                and not isinstance(self.adaptmlp.scale, nn.Parameter):
            ad = self.adaptmlp
            a = linear_f32(x2, ad.down_proj.weight, ad.down_proj.bias, "relu", image_for=ad.up_proj.weight)
            y = self.mlp(ln2, resid=x2)                                                            # x + mlp(norm2(x))
            return linear_f32(a, ad.up_proj.weight, ad.up_proj.bias, resid=y, scale=ad.scale).view(B, N, C)   # + s * up(relu(down(x)))
        if hasattr(self, "adaptmlp"):
            return (self.mlp(ln2, resid=x2) + self.adaptmlp(x2, add_residual=False)).view(B, N, C)
        return self.mlp(ln2, resid=x2).view(B, N, C)


class PatchEmbed(nn.Module):
    """Image to patch embedding: Conv2d(in_chans, embed_dim, patch, patch) run as patchify + GEMM."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = (img_size // patch_size) * (img_size // patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        B = x.shape[0]
        cols = ops.vit_patchify(x.float().contiguous(), self.patch_size)
        return linear_f32(cols, self.proj.weight, self.proj.bias).view(B, -1, self.proj.weight.shape[0])


class VisionTransformer(nn.Module):
    """DINO ViT, optionally with adapters.  Reference …dino_version.py:149-256 (adapter) / vision_transformer_dino.py."""

    def __init__(self, img_size=[224], patch_size=16, in_chans=3, num_classes=0, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer=nn.LayerNorm, adapter_ffn_layernorm_option="none",
                 adapter_ffn_init_option="lora", adapter_ffn_scalar="0.1",
                 adapter_ffn_num=64, adapter_d_model=768, use_adapter=True, pool="cls", **kwargs):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.pool = pool                                   # "cls" (DINO) or "mean_patches" (MAE encoder)
        self.precision = "fp32"
        self.patch_embed = PatchEmbed(img_size=img_size[0] if isinstance(img_size, (list, tuple)) else img_size,
                                      patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, drop_path=0., norm_layer=norm_layer,
                  adapter_ffn_layernorm_option=adapter_ffn_layernorm_option,
                  adapter_ffn_init_option=adapter_ffn_init_option, adapter_ffn_scalar=adapter_ffn_scalar,
                  adapter_ffn_num=adapter_ffn_num, d_model=adapter_d_model, adapter_d_model=adapter_d_model,
                  use_adapter=use_adapter)
            for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        self._bf16_cache = None

    def configure(self, precision="fp32"):
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self.precision = precision
        return self

    def invalidate(self):
        """Forget everything derived from the current weights (the split images cached on the parameters): call it after editing
        parameters through `.data` -- an edit that changes neither data_ptr nor _version, the keys of those caches."""
        from . import functional as SF
        SF.drop_param_caches(self)
        return self

    # -- reference helper API ---------------------------------------------------------------------------------------
    def interpolate_pos_encoding(self, x, w, h):
        """Positional table for a w x h pixel input (reference ...dino_version.py:196-216): the stored table at the training
        resolution; otherwise its patch rows resampled bicubically on the sqrt(N) grid (the reference's +0.1 guard against
        floor rounding of the scale factor included), class row untouched."""
        table = self.pos_embed
        n_train = table.shape[1] - 1
        if x.shape[1] - 1 == n_train and w == h:
            return table
        side = int(math.sqrt(n_train))
        ps = self.patch_embed.patch_size
        gw, gh = w // ps, h // ps
        grid = table[:, 1:].reshape(1, side, side, table.shape[-1]).permute(0, 3, 1, 2)
        grid = F.interpolate(grid, scale_factor=((gw + 0.1) / math.sqrt(n_train), (gh + 0.1) / math.sqrt(n_train)),
                             mode='bicubic')
        assert tuple(grid.shape[-2:]) == (gw, gh)
        return torch.cat((table[:, :1], grid.permute(0, 2, 3, 1).reshape(1, gw * gh, table.shape[-1])), dim=1)

    def prepare_tokens(self, x):
        """[B, 3, H, W] -> tokens [B, T, D] (fp32): patch embedding, cls concat, + positional table."""
        if not x.is_cuda:
            raise SnuffyHipError("input must be a GPU tensor: snuffy_amd has no CPU fallback")
        B, nc, w, h = x.shape
        pe = self.patch_embed(x)                                                  # [B, P, D]
        pos = self.interpolate_pos_encoding(torch.empty(1, pe.shape[1] + 1, pe.shape[2], device="meta"), w, h)
        tok = ops.vit_assemble_tokens(pe.reshape(-1, pe.shape[2]).contiguous(), self.cls_token.detach(), pos.detach()[0], B)
        return tok.view(B, pe.shape[1] + 1, pe.shape[2])

    def get_last_selfattention(self, x):
        """Attention probabilities [B, h, T, T] of the LAST block (reference ...dino_version.py:238-245)."""
        tokens = self.prepare_tokens(x)
        *body, last = self.blocks
        for blk in body:
            tokens = blk(tokens)
        return last(tokens, return_attention=True)

    def get_intermediate_layers(self, x, n=1):
        """Normalised token maps of the last n blocks (reference ...dino_version.py:247-255)."""
        tokens = self.prepare_tokens(x)
        first_kept = len(self.blocks) - n
        kept = []
        for i, blk in enumerate(self.blocks):
            tokens = blk(tokens)
            if i >= first_kept:
                B, T, C = tokens.shape
                kept.append(ops.layernorm_rows(tokens.reshape(B * T, C).contiguous(), self.norm.weight, self.norm.bias,
                                               self.norm.eps).view(B, T, C))
        return kept

    # -- forward -----------------------------------------------------------------------------------------------------
    def _pool(self, x, B, T):
        D = x.shape[-1]
        xv = x.view(B, T, D)
        if self.pool == "cls":
            rows = xv[:, 0].contiguous()                                          # LN is per row: only the CLS rows matter
        else:
            rows = xv[:, 1:, :].mean(dim=1)                                       # models_adapter_mae.py:192
        return ops.layernorm_rows(rows, self.norm.weight, self.norm.bias, self.norm.eps)

    @torch.no_grad()
    def forward(self, x):
        if self.precision == "bf16":
            return self._forward_bf16(x)
        x = self.prepare_tokens(x)
        for blk in self.blocks:
            x = blk(x)
        B, T, D = x.shape
        return self._pool(x.reshape(B * T, D), B, T)

    def _weights_bf16(self):
        params = list(self.parameters())
        key = tuple(SF.param_key(p) for p in params)
        if self._bf16_cache is not None and self._bf16_cache[0] == key:
            return self._bf16_cache[1]
        bf = torch.bfloat16
        f32 = torch.float32
        w = dict(pe_w=self.patch_embed.proj.weight.reshape(self.embed_dim, -1).to(bf).contiguous(),
                 pe_b=self.patch_embed.proj.bias.to(bf), pe_bf=self.patch_embed.proj.bias.to(f32), blocks=[])
        for blk in self.blocks:
            qkv_bias = (blk.attn.qkv.bias if blk.attn.qkv.bias is not None else
                        torch.zeros(3 * self.embed_dim, device=blk.attn.qkv.weight.device))
            d = dict(qkv_w=blk.attn.qkv.weight.to(bf), qkv_b=qkv_bias.to(bf), qkv_bf=qkv_bias.to(f32),
                     proj_w=blk.attn.proj.weight.to(bf), proj_b=blk.attn.proj.bias.to(bf), proj_bf=blk.attn.proj.bias.to(f32),
                     fc1_w=blk.mlp.fc1.weight.to(bf), fc1_b=blk.mlp.fc1.bias.to(bf), fc1_bf=blk.mlp.fc1.bias.to(f32),
                     fc2_w=blk.mlp.fc2.weight.to(bf), fc2_b=blk.mlp.fc2.bias.to(bf), fc2_bf=blk.mlp.fc2.bias.to(f32))
            if hasattr(blk, "adaptmlp"):
                dn, up = blk.adaptmlp.down_proj, blk.adaptmlp.up_proj
                dn_w, dn_b, up_w = dn.weight.detach(), dn.bias.detach(), up.weight.detach()
                pad = (-dn_w.shape[0]) % 64 if dn_w.shape[0] < 64 else 0
                if pad:
                    # a bottleneck below the GEMM's 64-deep minimum (the DINO recipe uses 32): zero rows of the down-projection
                    # (ReLU(0 + 0) = 0) against zero columns of the up-projection -- the same product, two K steps
                    dn_w = torch.cat([dn_w, dn_w.new_zeros(pad, dn_w.shape[1])])
                    dn_b = torch.cat([dn_b, dn_b.new_zeros(pad)])
                    up_w = torch.cat([up_w, up_w.new_zeros(up_w.shape[0], pad)], dim=1)
                d.update(dn_w=dn_w.to(bf).contiguous(), dn_b=dn_b.to(bf), dn_bf=dn_b.to(f32).contiguous(),
                         up_w=up_w.to(bf).contiguous(), up_b=up.bias.to(bf), up_bf=up.bias.to(f32))
            d.update(self._folded_block(blk, d))
            w["blocks"].append(d)
        self._bf16_cache = (key, w)
        return w

    def _folded_block(self, blk, d):
        """Operands of the fused bf16 block (round 6, forward_cols): the LayerNorms folded into the qkv / fc1 weights
        (LN(x) W^T + b = rstd (x (W diag(gamma))^T - mean colsum) + (W beta + b), colsum of the ROUNDED product), the adapter's
        up-projection riding as extra K columns of fc2 ([W2 | s Wup], bias b2 + s b_up)."""
        bf, f64 = torch.bfloat16, torch.float64
        out = {}
        for tag, lin, norm, bias in (("qkv", blk.attn.qkv, blk.norm1, blk.attn.qkv.bias), ("fc1", blk.mlp.fc1, blk.norm2, blk.mlp.fc1.bias)):
            w64 = lin.weight.detach().to(f64)
            wf = (w64 * norm.weight.detach().to(f64)).to(torch.float32).to(bf).contiguous()
            b0 = torch.zeros(w64.shape[0], dtype=f64, device=w64.device) if bias is None else bias.detach().to(f64)
            out[tag + "_wf"] = wf
            out[tag + "_cs"] = wf.to(f64).sum(dim=1).to(torch.float32).contiguous()
            out[tag + "_bfold"] = (w64 @ norm.bias.detach().to(f64) + b0).to(torch.float32).contiguous()
        if "dn_w" in d:
            sc = float(blk.adaptmlp.scale)
            up_w = blk.adaptmlp.up_proj.weight.detach()
            bott = d["dn_w"].shape[0]
            up_pad = torch.cat([up_w, up_w.new_zeros(up_w.shape[0], bott - up_w.shape[1])], dim=1) if bott != up_w.shape[1] else up_w
            out["fc2cat_w"] = torch.cat([blk.mlp.fc2.weight.detach(), sc * up_pad], dim=1).to(bf).contiguous()
            out["fc2cat_b"] = (blk.mlp.fc2.bias.detach() + sc * blk.adaptmlp.up_proj.bias.detach()).to(torch.float32).contiguous()
        return out

    def _fused_ok(self):
        """The fused bf16 block needs the epilogue variants' domain (widths % 64, K >= 96) and equal-width blocks."""
        blk = self.blocks[0]
        bott = 0
        if hasattr(blk, "adaptmlp"):
            bott = blk.adaptmlp.down_proj.weight.shape[0]
            bott += (-bott) % 64 if bott < 64 else 0
            if not ops.gemm_supported(1 << 20, bott, self.embed_dim):
                return False
        return _fused_ok_dims(self.embed_dim, blk.mlp.fc1.weight.shape[0], bott)

    def _blocks_fused(self, xt, W, B, T):
        """The blocks of the bf16 path with no LayerNorm / residual / adapter-sum pass (round 6): per block four GEMMs (+ the adapter's
        down-projection) and the attention.  qkv and fc1 take the RAW residual stream (bf16 copy) and apply the LayerNorm in their
        epilogue; proj and fc2 update the fp32 residual stream in place in theirs and leave its bf16 copy and the rows' moments for the
        next LayerNorm; the adapter's up-projection rides as 64 extra K columns of fc2 ([gelu(fc1) | ReLU(down)] x [W2 | s Wup]).
        Reference: ...dino_version.py:120-127, adapter.py:74-94."""
        R, D = xt.shape
        heads = self.num_heads
        hidden = self.blocks[0].mlp.fc1.weight.shape[0]
        has_ad = "dn_w" in W["blocks"][0]
        bott = W["blocks"][0]["dn_w"].shape[0] if has_ad else 0
        part = torch.empty(R, D // 32, 2, dtype=torch.float32, device=xt.device)
        hbuf = torch.empty(R, hidden + bott, dtype=torch.bfloat16, device=xt.device)      # [gelu(fc1) | ReLU(down)]
        stats, xb = ops.vit_row_stats(xt, eps=self.blocks[0].norm1.eps, want_bf16=True)
        for i, blk in enumerate(self.blocks):
            wb = W["blocks"][i]
            qkv = ops.gemm_bf16_lnfold(xb, wb["qkv_wf"], wb["qkv_cs"], wb["qkv_bfold"], stats)       # [R, 3D] bf16
            o, _ = ops.vit_attention(qkv, B, T, heads, blk.attn.scale)
            ops.gemm_bf16_resid_(xt, o, wb["proj_w"], wb["proj_bf"], xb, part)                       # x += proj(attn)
            stats, _ = ops.vit_row_stats(part=part, d=D, eps=blk.norm2.eps)
            ops.gemm_bf16_lnfold(xb, wb["fc1_wf"], wb["fc1_cs"], wb["fc1_bfold"], stats, "gelu", out=hbuf[:, :hidden])
            if has_ad:
                ops.gemm_bf16(xb, wb["dn_w"], wb["dn_bf"], "relu", out=hbuf[:, hidden:], tile_n=128)
                ops.gemm_bf16_resid_(xt, hbuf, wb["fc2cat_w"], wb["fc2cat_b"], xb, part)             # x += mlp + s adapter
            else:
                ops.gemm_bf16_resid_(xt, hbuf, wb["fc2_w"], wb["fc2_bf"], xb, part)
            if i + 1 < len(self.blocks):
                stats, _ = ops.vit_row_stats(part=part, d=D, eps=self.blocks[i + 1].norm1.eps)
        return self._pool(xt, B, T)

    def _forward_bf16(self, x):
        """bf16 GEMM / MFMA operands, fp32 residual stream, LayerNorm fused with the residual adds."""
        if not x.is_cuda:
            raise SnuffyHipError("input must be a GPU tensor: snuffy_amd has no CPU fallback")
        B, nc, w_, h_ = x.shape
        ps = self.patch_embed.patch_size
        cols = ops.vit_patchify(x.float().contiguous(), ps, torch.bfloat16)
        return self.forward_cols(cols, B, w_, h_)

    @torch.no_grad()
    def forward_cols(self, cols, B, w_, h_):
        """bf16 path from the patch-embedding GEMM operand: cols [B * P, 3 * patch * patch] bf16 (im2col rows of B images of
        w_ x h_ pixels), as snuffy_amd.tiles.preprocess_tiles(want="cols") writes them straight from the uint8 tiles -- the
        fp32 image tensor and the patchify pass never exist."""
        if not (cols.is_cuda and cols.dtype == torch.bfloat16 and cols.dim() == 2):
            raise SnuffyHipError("forward_cols: need the bf16 im2col matrix on the GPU (no CPU fallback)")
        W = self._weights_bf16()
        # patch embedding = im2col rows x conv weight on the hand-written MFMA kernel                 [B*P, D] bf16
        pe = ops.linear_bf16(cols, W["pe_w"], W["pe_bf"], W["pe_b"], prefer_native=True)
        P = pe.shape[0] // B
        T = P + 1
        pos = self.interpolate_pos_encoding(torch.empty(1, T, self.embed_dim, device="meta"), w_, h_)
        xt = ops.vit_assemble_tokens(pe, self.cls_token, pos[0], B)                              # [B*T, D] fp32
        heads = self.num_heads
        dk = self.embed_dim // heads
        use_mfma = ops.vit_mfma_attention_supported(T, dk)
        if BF16_FUSED_BLOCK and use_mfma and self._fused_ok():
            return self._blocks_fused(xt, W, B, T)
        n1 = self.blocks[0].norm1
        ln = ops.layernorm_rows(xt, n1.weight, n1.bias, n1.eps, out_dtype=torch.bfloat16)
        for i, blk in enumerate(self.blocks):
            wb = W["blocks"][i]
            qkv = ops.linear_bf16(ln, wb["qkv_w"], wb["qkv_bf"], wb["qkv_b"])                    # [B*T, 3D] bf16
            if use_mfma:
                o, _ = ops.vit_attention(qkv, B, T, heads, blk.attn.scale)
            else:
                o, _ = ops.vit_attention(qkv.float(), B, T, heads, blk.attn.scale)
                o = o.to(torch.bfloat16)
            y = ops.linear_bf16(o, wb["proj_w"], wb["proj_bf"], wb["proj_b"])                    # [B*T, D] bf16
            has_ad = "dn_w" in wb
            ln2, xb = ops.vit_residual_ln_(xt, add1=y, gamma=blk.norm2.weight, beta=blk.norm2.bias, eps=blk.norm2.eps,
                                           want_ln=True, want_x_bf16=has_ad)                     # x += attn ; LN2(x)
            # bias + GELU (erf form, as nn.GELU) in the epilogue of the hand-written GEMM: one rounding, no extra pass over the
            # [B*T, 4D] tensor
            hdn = ops.linear_bf16(ln2, wb["fc1_w"], wb["fc1_bf"], wb["fc1_b"], "gelu")
            # fc2 (K = 4 D) on the hand-written kernel too (the library is ~15 % ahead on this long-K shape; the block's
            # contractions are ours end to end)
            m = ops.linear_bf16(hdn, wb["fc2_w"], wb["fc2_bf"], wb["fc2_b"], prefer_native=True)
            u, s2 = None, 1.0
            if has_ad:
                # the adapter's skinny projections (D -> bottleneck -> D): 128-wide tiles of the hand-written GEMM are 1.5-1.8x the
                # library here (tools/adapter_bench.py); bottlenecks below 64 arrive zero-padded to 64 (_weights_bf16)
                bott = wb["dn_w"].shape[0]
                if ops.gemm_supported(xb.shape[0], bott, xb.shape[1]) and ops.gemm_supported(xb.shape[0], xb.shape[1], bott):
                    a = ops.gemm_bf16(xb, wb["dn_w"], wb["dn_bf"], "relu", tile_n=128)           # ReLU(down(x))
                    u = ops.gemm_bf16(a, wb["up_w"], wb["up_bf"], "none", tile_n=128)
                else:
                    a = torch._addmm_activation(wb["dn_b"], xb, wb["dn_w"].t())
                    u = torch.addmm(wb["up_b"], a, wb["up_w"].t())
                s2 = float(blk.adaptmlp.scale)
            last = i == len(self.blocks) - 1
            nxt = self.norm if last else self.blocks[i + 1].norm1
            ln, _ = ops.vit_residual_ln_(xt, add1=m, add2=u, scale2=s2, gamma=nxt.weight, beta=nxt.bias, eps=nxt.eps,
                                         want_ln=not last)                                       # x += mlp + s*adapter
        return self._pool(xt, B, T)


def _fused_ok_dims(d, hidden, bott):
    return (ops.gemm_lnfold_supported(3 * d, d) and ops.gemm_lnfold_supported(d, d) and ops.gemm_lnfold_supported(hidden, d)
            and ops.gemm_lnfold_supported(d, hidden + bott) and d <= 1024)


def vit_tiny(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=192, depth=12, num_heads=3, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_small(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_base(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    """Fixed 2-D sine-cosine positional table of the MAE encoder, [grid*grid (+1), embed_dim] float64 (reference
    utils_ssls_cf/pos_embed.py:21-66).  Column layout: [sin(x w) | cos(x w) | sin(y w) | cos(y w)] with D/4 frequencies
    w_i = 10000^(-i / (D/4)), x = column and y = row of the patch; an all-zero first row stands for the class token."""
    import numpy as np
    if embed_dim % 4:
        raise AssertionError("embed_dim must be divisible by 4")
    quarter = embed_dim // 4
    freq = 1.0 / 10000 ** (np.arange(quarter, dtype=np.float64) / quarter)
    ys, xs = np.divmod(np.arange(grid_size * grid_size), grid_size)           # row-major patch order
    ax = np.outer(xs.astype(np.float32), freq)
    ay = np.outer(ys.astype(np.float32), freq)
    table = np.concatenate([np.sin(ax), np.cos(ax), np.sin(ay), np.cos(ay)], axis=1)
    if cls_token:
        table = np.concatenate([np.zeros((1, embed_dim)), table], axis=0)
    return table


def mae_adapter_encoder(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.,
                        norm_layer=partial(nn.LayerNorm, eps=1e-6), adapter_ffn_scalar="0.1", adapter_ffn_num=64,
                        adapter_d_model=768, **kwargs):
    """Encoder half of models_adapter_mae.MaskedAutoencoderViT (the part compute_feats.py runs): same keys, output
    LN(mean of the patch tokens) (models_adapter_mae.py:174-195).  The decoder is training-only and out of scope."""
    model = VisionTransformer(img_size=[img_size], patch_size=patch_size, embed_dim=embed_dim, depth=depth,
                              num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=True, norm_layer=norm_layer,
                              adapter_ffn_scalar=adapter_ffn_scalar, adapter_ffn_num=adapter_ffn_num,
                              adapter_d_model=adapter_d_model, pool="mean_patches", **kwargs)
    # the MAE encoder's positional table is the fixed sin-cos one (models_adapter_mae.py:89-92), frozen
    table = get_2d_sincos_pos_embed(embed_dim, int(model.patch_embed.num_patches ** .5), cls_token=True)
    with torch.no_grad():
        model.pos_embed.copy_(torch.from_numpy(table).float().unsqueeze(0))
    model.pos_embed.requires_grad = False
    return model


class IClassifier(nn.Module):
    """feature_extractor + Linear, returns (feats, c).  Reference dsmil.py:39-50 (what compute_feats.py:442 builds)."""

    def __init__(self, feature_extractor, feature_size, output_class):
        super().__init__()
        self.feature_extractor = feature_extractor
        self.fc = nn.Linear(feature_size, output_class)

    def forward(self, x):
        feats = self.feature_extractor(x)
        feats = feats.view(feats.shape[0], -1)
        c = ops.critic(feats.float().contiguous(), self.fc.weight, self.fc.bias)
        return feats, c
