"""CPU emulation of the bf16 ViT block's roundings: today's order (LayerNorm in fp32, rounded to bf16, then the GEMM) against the
LayerNorm FOLDED into the consumer GEMM (round the raw residual stream to bf16, y = rstd (x W'^T - mean colsum(W')) + b', the
residual adds in the producers' fp32 epilogues, adapter down-projection riding as extra columns of fc1 and its up-projection as
extra K columns of fc2).  Development tool (round 6): decides whether the fold stays inside the 1e-2 bf16 class before any
kernel is written.  Uses the ViT oracle's math (oracle/vit_oracle.py) for the exact reference.

    python tools/vit_fold_emulation.py            # vit_small + adapter (ffn 32, scalar 10), depth 12, 4 images, seeds 0..2
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import vit_oracle as vorc  # noqa: E402


def r(x):
    return x.to(torch.bfloat16).float()


def rel_err(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def attn_core(qkv, b, t, heads):
    d = qkv.shape[-1] // 3
    dk = d // heads
    q, k, v = qkv.view(b, t, 3, heads, dk).permute(2, 0, 3, 1, 4)
    p = ((q @ k.transpose(-2, -1)) * dk ** -0.5).softmax(-1)
    return r((r(p) @ v).transpose(1, 2).reshape(b * t, d))


def forward(imgs, sd, patch, depth, heads, scale, fold, eps=1e-6):
    x = vorc.prepare_tokens(imgs, sd, patch)
    b, t, d = x.shape
    x = x.reshape(b * t, d)
    for i in range(depth):
        p = f"blocks.{i}."
        g1, b1 = sd[p + "norm1.weight"], sd[p + "norm1.bias"]
        g2, b2 = sd[p + "norm2.weight"], sd[p + "norm2.bias"]
        wq, bq = sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]
        wp, bp = sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"]
        w1, bb1 = sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]
        w2, bb2 = sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"]
        wd, bd = sd[p + "adaptmlp.down_proj.weight"], sd[p + "adaptmlp.down_proj.bias"]
        wu, bu = sd[p + "adaptmlp.up_proj.weight"], sd[p + "adaptmlp.up_proj.bias"]
        if not fold:
            ln = r(F.layer_norm(x, (d,), g1, b1, eps))
            qkv = r(ln @ r(wq).t() + bq)
            o = attn_core(qkv, b, t, heads)
            x = x + r(o @ r(wp).t() + bp)
            ln2, xb = r(F.layer_norm(x, (d,), g2, b2, eps)), r(x)
            h = r(F.gelu(ln2 @ r(w1).t() + bb1))
            m = r(h @ r(w2).t() + bb2)
            a = r(F.relu(xb @ r(wd).t() + bd))
            u = r(a @ r(wu).t() + bu)
            x = x + m + scale * u
        else:
            def folded(x, w, bias, g, bt):
                mu = x.mean(-1, keepdim=True)
                rstd = (x.var(-1, unbiased=False, keepdim=True) + eps).rsqrt()
                wf = r(w * g)                                    # W' = W diag(gamma), rounded once
                return rstd * (r(x) @ wf.t() - mu * wf.sum(-1)) + (w @ bt + bias)
            qkv = r(folded(x, wq, bq, g1, b1))
            o = attn_core(qkv, b, t, heads)
            x = x + (o @ r(wp).t() + bp)                         # residual in the fp32 epilogue
            h = r(F.gelu(folded(x, w1, bb1, g2, b2)))
            a = r(scale * F.relu(r(x) @ r(wd).t() + bd))
            x = x + (torch.cat([h, a], 1) @ torch.cat([r(w2), r(wu)], 1).t() + bb2 + scale * bu)
    return F.layer_norm(x.view(b, t, d), (d,), sd["norm.weight"], sd["norm.bias"], eps)[:, 0]


def main():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from snuffy_amd import vit
    for seed in range(3):
        torch.manual_seed(seed)
        model = vit.vit_small(patch_size=16, adapter_ffn_scalar="10", adapter_ffn_num=32, adapter_d_model=384).eval()
        if seed == 2:   # trained-like statistics: non-trivial LayerNorm affine and a row mean comparable to the spread
            with torch.no_grad():
                for blk in model.blocks:
                    blk.norm1.weight.uniform_(0.5, 2.0), blk.norm1.bias.normal_(0, 0.3)
                    blk.norm2.weight.uniform_(0.5, 2.0), blk.norm2.bias.normal_(0, 0.3)
                model.cls_token.add_(0.5), model.pos_embed.add_(0.3)
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        imgs = torch.rand(4, 3, 224, 224)
        with torch.no_grad():
            ref = vorc.vit_forward(imgs, sd, 16, 12, 6, 10.0, "dino_adapter")
            a = forward(imgs, sd, 16, 12, 6, 10.0, fold=False)
            f = forward(imgs, sd, 16, 12, 6, 10.0, fold=True)
        print("seed %d  today's order %.3e   folded %.3e" % (seed, rel_err(a, ref), rel_err(f, ref)))


if __name__ == "__main__":
    main()
