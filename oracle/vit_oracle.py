"""CPU ORACLE for the ViT patch-embedding extractors -- TEST INFRASTRUCTURE, NOT PRODUCT (only tests/ may import it).

Restatement, written from the math, of the inference forward of
  * DINO ViT (plain / with adapter)   utils_ssls_cf/vision_transformer_dino.py:96-127,
                                      utils_ssls_cf/vision_transformer_with_adapter_dino_version.py:70-94,97-127,130-146,218-236
  * Adapter                           utils_ssls_cf/adapter.py:74-94  (add_residual=False, layernorm option "none")
  * MAE-adapter encoder               utils_ssls_cf/models_adapter_mae.py:174-195
over a plain state dict with the reference's key names.

PARITY PIN: tests/golden/f7_*.npz, f8_*.npz captured from the unmodified reference (tests/golden/make_golden.py);
checked by tests/test_oracle_vit.py.
"""
import torch
import torch.nn.functional as F


def patch_embed(imgs, w, b, patch):
    """Conv2d(3, D, patch, patch) as im2col + GEMM.  imgs [B,3,H,W] -> [B, P, D]  (…dino_version.py:141-146)."""
    bsz, c, h, wd = imgs.shape
    gh, gw = h // patch, wd // patch
    cols = imgs.reshape(bsz, c, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5).reshape(bsz, gh * gw, c * patch * patch)
    return cols @ w.reshape(w.shape[0], -1).t() + b


def attention(x, sd, pre, heads):
    """Attention.forward (…dino_version.py:82-94).  x [B,T,D] -> (out [B,T,D], attn [B,h,T,T])."""
    b, t, d = x.shape
    dk = d // heads
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd.get(pre + "qkv.bias")).reshape(b, t, 3, heads, dk).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = ((q @ k.transpose(-2, -1)) * dk ** -0.5).softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(b, t, d)
    return F.linear(out, sd[pre + "proj.weight"], sd[pre + "proj.bias"]), attn


def block(x, sd, pre, heads, adapter_scale, eps=1e-6, return_attention=False):
    """Block.forward (…dino_version.py:120-127): x += Attn(LN1 x); x = x + MLP(LN2 x) + Adapter(x)."""
    d = x.shape[-1]
    y, attn = attention(F.layer_norm(x, (d,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], eps), sd, pre + "attn.", heads)
    if return_attention:
        return attn
    x = x + y
    ad = 0.0
    if pre + "adaptmlp.down_proj.weight" in sd:                                  # adapter.py:74-94
        down = F.relu(F.linear(x, sd[pre + "adaptmlp.down_proj.weight"], sd[pre + "adaptmlp.down_proj.bias"]))
        ad = F.linear(down, sd[pre + "adaptmlp.up_proj.weight"], sd[pre + "adaptmlp.up_proj.bias"]) * adapter_scale
    xn = F.layer_norm(x, (d,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], eps)
    m = F.linear(F.gelu(F.linear(xn, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])),
                 sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    return x + m + ad


def prepare_tokens(imgs, sd, patch, mae=False):
    x = patch_embed(imgs, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], patch)
    pos = sd["pos_embed"]
    if mae:                                                                     # models_adapter_mae.py:176-186
        x = x + pos[:, 1:, :]
        cls = (sd["cls_token"] + pos[:, :1, :]).expand(x.shape[0], -1, -1)
        return torch.cat((cls, x), dim=1)
    cls = sd["cls_token"].expand(x.shape[0], -1, -1)                            # …dino_version.py:218-229
    return torch.cat((cls, x), dim=1) + pos


def vit_forward(imgs, sd, patch, depth, heads, adapter_scale=0.0, kind="dino", eps=1e-6):
    """-> features [B, D].  kind: 'dino' / 'dino_adapter' (CLS token of LN(x)) or 'mae_adapter' (LN(mean of patch tokens))."""
    x = prepare_tokens(imgs, sd, patch, mae=(kind == "mae_adapter"))
    for i in range(depth):
        x = block(x, sd, f"blocks.{i}.", heads, adapter_scale, eps)
    d = x.shape[-1]
    if kind == "mae_adapter":
        return F.layer_norm(x[:, 1:, :].mean(dim=1), (d,), sd["norm.weight"], sd["norm.bias"], eps)
    return F.layer_norm(x, (d,), sd["norm.weight"], sd["norm.bias"], eps)[:, 0]
