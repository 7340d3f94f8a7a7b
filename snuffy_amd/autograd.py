"""Training path of the Snuffy aggregator (loss.backward() of reference train.py:259).

The FORWARD of every custom op runs on the hand-written HIP kernels.  The attention BACKWARD is hand-written too
(snf_sparse_attn_bwd_f32 exact, snf_sparse_attn_bwd_mfma on the matrix cores: P recomputed from the saved log-sum-exp, dQ /
dV streamed, dKp reduced); the LayerNorm backward and the dense projections are library GPU ops inside / next to the
``torch.autograd.Function``s below.  Everything stays on the GPU; nothing here touches the CPU oracle.

Semantics follow the reference in train mode:
  * attention dropout p = MultiHeadedAttention.dropout.p (0.1 -- train.py:866-869 never overrides it) on P, snuffy.py:166-167
  * encoder dropout (default 0) after the attention output, inside the FFN and after it, snuffy.py:108,110,225
  * no gradient flows through the selection (snuffy.py:128-147); x itself is data (no grad) in layer 0
"""
import math

import torch
import torch.nn.functional as F

from . import ops

_ACT = {
    "relu": F.relu,
    "gelu": F.gelu,
    "leakyrelu": lambda t: F.leaky_relu(t, 0.01),
    "selu": F.selu,
}


class LayerNormRowsFn(torch.autograd.Function):
    """y = LayerNorm(x) * gamma + beta (forward: snf_layernorm_rows_f32, statistics saved for backward)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, gamma, beta, eps):
        y, mean, rstd = ops.layernorm_rows(x, gamma, beta, eps, want_stats=True)
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        xhat = (x - mean.unsqueeze(1)) * rstd.unsqueeze(1)
        dgamma = (dy * xhat).sum(0)
        dbeta = dy.sum(0)
        dx = None
        if ctx.needs_input_grad[0]:
            dxhat = dy * gamma
            m1 = dxhat.mean(1, keepdim=True)
            m2 = (dxhat * xhat).mean(1, keepdim=True)
            dx = (dxhat - m1 - xhat * m2) * rstd.unsqueeze(1)
        return dx, dgamma, dbeta, None


class ScatterRowsFn(torch.autograd.Function):
    """y = x.clone(); y[sel] = rows  (snuffy.py:154-155; forward: snf_scatter_rows_f32)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, sel, rows):
        ctx.save_for_backward(sel)
        return ops.scatter_rows(x, sel, rows)

    @staticmethod
    def backward(ctx, dy):
        (sel,) = ctx.saved_tensors
        drows = dy.index_select(0, sel)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = dy.clone()
            dx.index_fill_(0, sel, 0.0)
        return dx, None, drows


def draw_dropout_state():
    """(seed, offset) of one dropout mask: the seed of torch's default CPU generator and a 62-bit offset drawn from it --
    reproducible under torch.manual_seed, no device synchronisation."""
    return int(torch.initial_seed()), int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


class SparseAttnFn(torch.autograd.Function):
    """O = dropout(softmax_K(Q Kp^T / sqrt(dk)))^T V per head (snuffy.py:160-168).

    bf16-autocast training (dk = 128 / 64, K within one LDS image): forward AND backward on the MFMA kernels; the dropout
    mask is regenerated in registers from (seed, offset) in both (csrc/philox.h) -- no [h, N, K] tensor is stored for it and
    the forward's O is the kernel's own.  Otherwise: fp32-class forward (split-bf16 x 3 on the matrix cores where the shape
    allows, else the exact vector-ALU kernel) with P materialised for the exact backward kernel snf_sparse_attn_bwd_f32; the
    mask is the same Philox stream written out as a tensor.  Returns (O, P): P is the dropped P in train mode, as the
    reference's attention() returns it."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, q, kp, v, h, dropout_p, bf16_operands=False, want_p=True):
        n, d = q.shape
        k = kp.shape[0]
        dk = d // h
        drop = (float(dropout_p),) + draw_dropout_state() if dropout_p > 0.0 else None
        ctx.h, ctx.drop = h, drop
        ctx.fast_bwd = bool(bf16_operands and ops.mfma_attn_supported(k, dk) and ops.mfma_attn_bwd_supported(k, dk))
        if ctx.fast_bwd:
            q16, v16 = q.to(torch.bfloat16), v.to(torch.bfloat16)
            out, p, lse = ops.sparse_attn_fwd_mfma(q16, v16, kp, n, h, need_attn=want_p, need_lse=True, dropout=drop)
            ctx.save_for_backward(q16, kp, v16, lse)        # P is recomputed from lse, the mask from (seed, offset)
            return out, p
        if bf16_operands and ops.mfma_attn_supported(k, dk):
            out, p, _ = ops.sparse_attn_fwd_mfma(q.to(torch.bfloat16), v.to(torch.bfloat16), kp, n, h, need_attn=True)
        elif ops.x3_attn_supported(k, dk):
            out, p, _ = ops.sparse_attn_fwd_x3(q, v, kp, h, need_attn=True)
        else:
            out, p, _ = ops.sparse_attn_fwd(q, kp, v, h, need_attn=True)
        mask = None
        if drop is not None:
            mask = ops.dropout_mask(h, n, k, drop[0], drop[1], drop[2], q.device)
            vh = v.float().view(n, h, dk).transpose(0, 1)
            out = torch.bmm((p * mask).transpose(1, 2), vh).transpose(0, 1).reshape(k, d)
        ctx.save_for_backward(q, kp, v, p, mask)
        return out, (p * mask if mask is not None else p)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dout, _dp_unused):
        h = ctx.h
        if ctx.fast_bwd:
            q, kp, v, lse = ctx.saved_tensors
            dq, dkp, dv = ops.sparse_attn_bwd_mfma(q, v, kp, dout.float().contiguous(), lse, h, dropout=ctx.drop)
            return dq, dkp, dv, None, None, None, None
        q, kp, v, p, mask = ctx.saved_tensors
        n, d = q.shape
        dk = d // h
        scale = 1.0 / math.sqrt(dk)
        # K7-bwd on the HIP kernels (exact fp32): dS never leaves the workspace, P / dP are not re-materialised by bmm's
        dq, dkp, dv = ops.sparse_attn_bwd(q, kp, v, p, dout.float().contiguous(), h, mask=mask, scale=scale)
        return dq, dkp, dv, None, None, None, None


def critic_train(feats2, w, b):
    """Critic scores with autograd (library GEMV); the selection itself uses them detached."""
    return F.linear(feats2, w, b)


def encoder_layer_train(x2, sel, layer, need_attn, precision):
    """Differentiable EncoderLayer.forward (snuffy.py:126-157).  Returns (functional.Parts, A)."""
    from .functional import Parts
    # precision "bf16": the dense projections run under torch.autocast (bf16 operands, fp32 accumulate, fp32 master
    # weights); LayerNorm, softmax / attention and the residual stream stay fp32.
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(precision == "bf16")):
        return _encoder_layer_train(x2, sel, layer, need_attn)


def _encoder_layer_train(x2, sel, layer, need_attn):
    from .functional import Parts
    mha, ff = layer.self_attn, layer.feed_forward
    n0, n1 = layer.sublayer[0].norm, layer.sublayer[1].norm
    drop0, drop1 = layer.sublayer[0].dropout, layer.sublayer[1].dropout
    lq, lk, lv, lo = mha.linears
    training = layer.training
    if sel.numel() == 0:
        y = x2
        attn = torch.empty(1, mha.h, x2.shape[0], 0, device=x2.device) if need_attn else None
    else:
        xs = x2.index_select(0, sel)                                            # snuffy.py:131,145-147
        xn = LayerNormRowsFn.apply(x2, n0.weight, n0.bias, n0.eps)              # snuffy.py:107
        q = F.linear(xn, lq.weight, lq.bias)
        kp = F.linear(xs, lk.weight, lk.bias)
        v = F.linear(xn, lv.weight, lv.bias)
        p_drop = mha.dropout.p if training else 0.0
        o, p = SparseAttnFn.apply(q, kp, v, mha.h, p_drop, torch.is_autocast_enabled(), need_attn)
        delta = F.linear(o, lo.weight, lo.bias)                                 # snuffy.py:205
        if training and drop0.p > 0:
            delta = F.dropout(delta, drop0.p, True)
        x_sel = xs + delta.float()                                              # snuffy.py:108
        y = ScatterRowsFn.apply(x2, sel, x_sel)                                 # snuffy.py:154-155
        attn = p.detach().unsqueeze(0) if (need_attn and p is not None) else None
    yn = LayerNormRowsFn.apply(y, n1.weight, n1.bias, n1.eps)
    hid = _ACT[ff.activation_name](F.linear(yn, ff.w_1.weight, ff.w_1.bias))    # snuffy.py:224-225
    if training and ff.dropout.p > 0:
        hid = F.dropout(hid, ff.dropout.p, True)
    f = F.linear(hid, ff.w_2.weight, ff.w_2.bias)
    if training and drop1.p > 0:
        f = F.dropout(f, drop1.p, True)
    z = y + f.float()                                                           # snuffy.py:110
    return Parts(z), attn


def head_train(z, norm, linear):
    """logits = Linear(mean_n LayerNorm(z)) with autograd (snuffy.py:86,71)."""
    zn = LayerNormRowsFn.apply(z, norm.weight, norm.bias, norm.eps)
    return F.linear(zn.mean(dim=0), linear.weight, linear.bias)
