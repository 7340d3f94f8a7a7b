"""Fused forward pipeline of the Snuffy aggregator on top of the HIP kernels (ops.py).

Data layout in HBM for one bag (N patches, D features, K selected rows, F = mlp_multiplier * D):
  x      [N, D]  fp32   the bag (never cloned; the K updated rows live in small [K, D] side buffers)
  c      [N]     fp32   critic scores
  S      [K]     int64  selected rows (top-Lambda ++ random)
  fp32 path : Xn [N, D] fp32, Q / V [N, D] fp32, hidden [N, F] fp32, z [N, D] fp32
  bf16 path : xhat [N, D] bf16 (row-normalised x, affine folded into the GEMM weights; shared by the Q/V projection and,
              after re-normalising the K patched rows in place, by the FFN), [Q | V] [N, 2D] bf16 out of ONE fused
              projection GEMM (the attention kernel takes the two halves in place through their row pitch),
              hidden [N, F] bf16, FFN out [N, D] bf16; the fp32 residual stream is re-assembled inside the final
              LayerNorm + mean + head kernel.
"""
import math

import torch
import torch.nn.functional as F

from . import ops
from ._ffi import SnuffyHipError

ACTIVATIONS = ("relu", "gelu", "leakyrelu", "selu")      # reference snuffy.py:215-220
# attention kernel of the fp32 path: "x3" = split-bf16 x 3 on the matrix cores where the shape allows, "exact" = fp32 FMA on the
# vector ALUs for every shape.  What "x3" means numerically: every operand is hi + lo with hi = bf16(v), lo = bf16(v - hi), i.e.
# 16-17 mantissa bits, and lo * lo is dropped -- ~1e-5 relative per PRODUCT (fp32: 6e-8), accumulated in fp32.  Measured against
# fp64 this leaves P within 3.5e-6, O within 1e-5 relative and a bag's logits within 3e-5 of the reference goldens: inside
# north_star's 1e-3 fp32 class by 30x, but NOT bit-level fp32 -- "exact" / "library" are the plain-fp32 settings.
FP32_ATTENTION = "x3"
# with FP32_ATTENTION == "x3": the pipelined kernel on pre-split operands (snf_sparse_attn_fwd_x3_hl) wherever the layer runs on the
# one-pass hl GEMMs and the shape allows (ops.x3_hl_attn_supported); False keeps the round-3 kernel on fp32 operands
X3_HL_ATTENTION = True
# ... and the key projection in front of it writes the kernel's Kp fragment image itself (no fp32 Kp, no prep launch)
X3_HL_KPFRAG = True
# ... and gathers its input rows itself: gather + row -> slot map + key projection in one launch (round 6); False = gather_slot_map first
X3_HL_KPFRAG_GATHER = True
# fp32 path, the [N, .] projections: "x3" = split-bf16 products on the hand-written MFMA GEMM (fp32-class: logits within
# ~1e-5 of the exact path), "library" = fp32 library GEMMs.
FP32_GEMM = "x3"
# precision="bf16" on a stack of MORE than one encoder layer: "fp32" = run the fp32-class kernels (snuffy.RuntimeConfig.compute:
# the K-way softmax of a later layer amplifies the bf16 re-roundings of the layers before it past the 1e-2 class), "bf16" = literal.
BF16_DEEP_STACKS = "fp32"
# fp32 path on the one-pass hl GEMMs: ONE normalised image xhat = (x - mean) rstd serves both sublayers (the LayerNorm affines are
# folded into Wq | Wv and W1, in fp64, rounded once; after the attention only the K patched rows are re-normalised into the image)
# instead of a second full LayerNorm pass over the bag -- what the bf16 path has always done.  Needs equal eps in both LayerNorms.
FP32_SHARED_NORM = True


# ----------------------------------------------------------------------------------------------------------------------
# shape helpers
# ----------------------------------------------------------------------------------------------------------------------
def as_2d(x):
    if x.dim() == 3:
        if x.shape[0] != 1:
            # the reference's binary model indexes with a 1-D index tensor and breaks for B > 1 (snuffy.py:130-131)
            raise IndexError("snuffy (binary) supports a single bag per forward: got batch %d" % x.shape[0])
        x = x[0]
    if x.dim() != 2:
        raise ValueError("expected [1, N, D] or [N, D], got %s" % (tuple(x.shape),))
    if not x.is_cuda:
        raise SnuffyHipError("input must be a GPU tensor: snuffy_amd has no CPU fallback")
    if x.dtype != torch.float32:
        x = x.float()
    return x.contiguous()


def check_bag(x, c):
    x2 = as_2d(x)
    n = x2.shape[0]
    if c.numel() != n:
        raise IndexError("binary snuffy needs one critic score per patch: c has %d values for %d patches"
                         % (c.numel(), n))
    c1 = c.reshape(-1)
    if c1.dtype != torch.float32:
        c1 = c1.float()
    return x2, c1


def critic_scores(feats, w, b):
    lead = feats.shape[:-1]
    f2 = feats.reshape(-1, feats.shape[-1])
    if not f2.is_cuda:
        raise SnuffyHipError("input must be a GPU tensor: snuffy_amd has no CPU fallback")
    f2 = f2.float().contiguous()
    if torch.is_grad_enabled() and (w.requires_grad or f2.requires_grad):
        from . import autograd as SA
        s = SA.critic_train(f2, w, b)                    # training: library GEMV with autograd
    else:
        fused = ops.critic_select(f2, w, b)              # one class, large bag: the selector's histogram rides in the pass
        s = fused[0] if fused is not None else ops.critic(f2, w, b)
    return s.view(*lead, w.shape[0])


# ----------------------------------------------------------------------------------------------------------------------
# When is a tensor derived from a parameter (folded / split / bf16 copies, captured graphs) stale?  The caches below key on
# (data_ptr, _version) of the parameter -- but torch's FUSED optimizers (AdamW(fused=True): what train.py uses on the GPU) update the
# parameters WITHOUT bumping their version counters (measured, torch 2.10: version 0 -> 0 across step(); foreach / default: +2).
# Every optimizer step anywhere in the process therefore advances an epoch that is part of every key (a global post-step hook).
# Edits through ``p.data`` are still invisible: invalidate() is the documented way to announce those.
# ----------------------------------------------------------------------------------------------------------------------
_PARAM_EPOCH = [0]


def _on_optimizer_step(optimizer, args, kwargs):
    _PARAM_EPOCH[0] += 1


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post_hook
    _reg_post_hook(_on_optimizer_step)
except ImportError:      # a torch without global optimizer hooks: the trainers of this package bump the epoch themselves
    _reg_post_hook = None    # (train.Trainer / BagParallelStepper call bump_param_epoch() after every optimizer.step())


def bump_param_epoch():
    """Announce that parameters were written in a way the version counters do not show (an optimizer this package cannot hook)."""
    _PARAM_EPOCH[0] += 1


def after_optimizer_step():
    """What the trainers call behind every optimizer.step(): advances the epoch when torch has no global post-step hook to do it."""
    if _reg_post_hook is None:
        bump_param_epoch()


def param_key(p):
    """Cache key of anything derived from parameter p (see above)."""
    return (id(p), p.data_ptr(), p._version, _PARAM_EPOCH[0])


def split3_cached(weight, transposed=False):
    """[Wh | Wl | Wh] image of a parameter (ops.split3_weight) -- or of its transpose -- cached on the parameter until it is
    written again (optimizer step, load_state_dict)."""
    key = param_key(weight)
    name = "_snf_x3t" if transposed else "_snf_x3"
    hit = getattr(weight, name, None)
    if hit is None or hit[0] != key:
        w = weight.detach()
        hit = (key, ops.split3_weight(w.t().contiguous() if transposed else w))
        setattr(weight, name, hit)
    return hit[1]


def _needs_grad(*tensors):
    """True when autograd has to see this op: grad mode on and some input / parameter asks for a gradient."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


# bf16 eval path: the critic pass can hand the first encoder layer its normalised input (one read of the bag instead of
# two).  The hand-over lives on the receiving layer object, is keyed on the bag's storage and version and is consumed at
# most once.
def critic_scores_with_xhat(feats, w, b, eps, layer):
    """critic_scores() that also leaves xhat = (x - mean) * rstd (bf16) on `layer` for encoder_layer()'s bf16 path."""
    lead = feats.shape[:-1]
    f2 = feats.reshape(-1, feats.shape[-1])
    if not f2.is_cuda:
        raise SnuffyHipError("input must be a GPU tensor: snuffy_amd has no CPU fallback")
    if f2.dtype != torch.float32 or not f2.is_contiguous():
        layer._xhat_offer = None
        return critic_scores(feats, w, b)             # a converted copy would not be the tensor the encoder sees
    s, xhat = ops.critic_select(f2, w, b, eps) or ops.critic_ln(f2, w, b, eps)
    layer._xhat_offer = (f2.data_ptr(), tuple(f2.shape), f2._version, float(eps), xhat)
    return s.view(*lead, w.shape[0])


def hl_layer_eligible(layer, n, d):
    """The fp32-class encoder layer takes the one-pass GEMMs on interleaved images for a bag of n rows (encoder_layer's test)."""
    ff = layer.feed_forward
    f = ff.w_1.weight.shape[0]
    return (FP32_GEMM == "x3" and d % 32 == 0 and f % 32 == 0 and ops.gemm_supported(n, d, 3 * ff.w_2.weight.shape[1])
            and ops.gemm_supported(n, 2 * d, 3 * d) and ops.hl_eligible(n, 2 * d, d) and ops.hl_eligible(n, f, d)
            and ops.hl_eligible(n, d, f))


def critic_scores_with_hl(feats, w, b, layer):
    """critic_scores() of the fp32-class path that also leaves LayerNorm_0(x) -- with its affine, as the interleaved hi / lo image
    of the one-pass GEMM -- on `layer` for encoder_layer(): one read of the bag instead of two (snf_critic_ln_hl_f32)."""
    lead = feats.shape[:-1]
    f2 = feats.reshape(-1, feats.shape[-1])
    n0 = layer.sublayer[0].norm
    if (not f2.is_cuda or f2.dtype != torch.float32 or not f2.is_contiguous() or n0.weight is None or n0.bias is None
            or not hl_layer_eligible(layer, f2.shape[0], f2.shape[1])):
        layer._xn3_offer = None
        return critic_scores(feats, w, b)
    if shared_norm_layer(layer):
        s, img = ops.critic_ln_hl(f2, w, b, None, None, n0.eps)         # xhat itself: the affine lives in the folded weights
    else:
        s, img = ops.critic_ln_hl(f2, w, b, n0.weight, n0.bias, n0.eps)
    layer._xn3_offer = (_xn3_key(f2, n0, shared_norm_layer(layer)), img)
    return s.view(*lead, w.shape[0])


def shared_norm_layer(layer):
    """encoder_layer()'s fp32 hl branch keeps ONE normalised image for both sublayers of this layer (FP32_SHARED_NORM)."""
    n0, n1 = layer.sublayer[0].norm, layer.sublayer[1].norm
    return (FP32_SHARED_NORM and n0.eps == n1.eps and n0.weight is not None and n0.bias is not None and n1.weight is not None
            and n1.bias is not None)


def _xn3_key(x2, n0, shared):
    if shared:
        return (x2.data_ptr(), tuple(x2.shape), x2._version, float(n0.eps), "xhat")
    return (x2.data_ptr(), tuple(x2.shape), x2._version, float(n0.eps)) + param_key(n0.weight) + param_key(n0.bias)


def _take_xn3(layer, x2, n0, shared=False):
    offer = getattr(layer, "_xn3_offer", None)
    layer._xn3_offer = None
    if offer is not None and offer[0] == _xn3_key(x2, n0, shared):
        return offer[1]
    return None


def _take_xhat(layer, x2, eps):
    offer = getattr(layer, "_xhat_offer", None)
    layer._xhat_offer = None
    if offer is not None and offer[:4] == (x2.data_ptr(), tuple(x2.shape), x2._version, float(eps)):
        return offer[4]
    return None


def select_top(c1, big_lambda, top_share, n):
    k1 = min(math.ceil(big_lambda * top_share), n)          # python float arithmetic, snuffy.py:129
    if k1 < 1:
        return torch.empty(0, dtype=torch.int64, device=c1.device)
    return ops.topk(c1.detach(), k1)                         # no gradient through the selection (snuffy.py:128-130)


def gather(x2, idx):
    return ops.gather_rows(x2, idx)


def layer_norm(x2, norm):
    if _needs_grad(x2, norm.weight, norm.bias):
        from . import autograd as SA
        return SA.LayerNormRowsFn.apply(x2, norm.weight, norm.bias, norm.eps)
    return ops.layernorm_rows(x2, norm.weight, norm.bias, norm.eps)


def act_name(ff):
    return ff.activation_name


# ----------------------------------------------------------------------------------------------------------------------
# attention building blocks (API-fidelity entry points)
# ----------------------------------------------------------------------------------------------------------------------
def _drop_p(dropout):
    return float(dropout.p) if dropout is not None and getattr(dropout, "training", False) else 0.0


def attention_4d(query, key, value, dropout=None):
    """attention() of snuffy.py:160-168 on [1, h, N, dk] tensors (dropout: an nn.Dropout module, active in train mode)."""
    b, h, n, dk = query.shape
    if b != 1:
        raise IndexError("single bag only")
    k = key.shape[2]
    q2 = query[0].transpose(0, 1).reshape(n, h * dk).contiguous()
    k2 = key[0].transpose(0, 1).reshape(k, h * dk).contiguous()
    v2 = value[0].transpose(0, 1).reshape(n, h * dk).contiguous()
    p_drop = _drop_p(dropout)
    if p_drop > 0.0 or _needs_grad(query, key, value):
        from . import autograd as SA
        out, attn = SA.SparseAttnFn.apply(q2, k2, v2, h, p_drop, False)
    else:
        out, attn, _ = ops.sparse_attn_fwd(q2, k2, v2, h, need_attn=True)
    return out.view(k, h, dk).transpose(0, 1).unsqueeze(0), attn.unsqueeze(0)


def mha_forward(mha, q_in, key_in, v_in, need_attn, precision):
    """MultiHeadedAttention.forward (snuffy.py:183-205) on 2-D inputs; fp32 projections, exact attention kernel.  With grad
    mode on (or the module in train mode: its dropout is active) the call goes through the autograd functions."""
    lq, lk, lv, lo = mha.linears
    q = F.linear(q_in, lq.weight, lq.bias)
    kp = F.linear(key_in, lk.weight, lk.bias)
    v = F.linear(v_in, lv.weight, lv.bias)
    p_drop = _drop_p(mha.dropout)
    if p_drop > 0.0 or _needs_grad(q, kp, v):
        from . import autograd as SA
        o, attn = SA.SparseAttnFn.apply(q, kp, v, mha.h, p_drop, False)
    else:
        o, attn, _ = ops.sparse_attn_fwd(q, kp, v, mha.h, need_attn=need_attn)
    out = F.linear(o, lo.weight, lo.bias)
    return out, (attn.unsqueeze(0) if attn is not None else None)


def ffn_forward(ff, x2, precision):
    """PositionwiseFeedForward.forward (snuffy.py:224-225); autograd / dropout semantics as the reference's module."""
    p_drop = _drop_p(ff.dropout)
    if p_drop > 0.0 or _needs_grad(x2, ff.w_1.weight, ff.w_1.bias, ff.w_2.weight, ff.w_2.bias):
        from . import autograd as SA
        hid = SA._ACT[ff.activation_name](F.linear(x2, ff.w_1.weight, ff.w_1.bias))
        return F.linear(ff.dropout(hid), ff.w_2.weight, ff.w_2.bias)
    hid = torch.mm(x2, ff.w_1.weight.t())
    ops.bias_act_(hid, ff.w_1.bias, ff.activation_name)
    return F.linear(hid, ff.w_2.weight, ff.w_2.bias)


# ----------------------------------------------------------------------------------------------------------------------
# fused encoder layer
# ----------------------------------------------------------------------------------------------------------------------
class Parts:
    """z = base (+ add_bf16) (+ add_bias) (+ delta[slot]) -- the residual stream after a layer, not yet assembled."""

    __slots__ = ("base", "add_bf16", "add_bias", "slot", "delta")

    def __init__(self, base, add_bf16=None, add_bias=None, slot=None, delta=None):
        self.base, self.add_bf16, self.add_bias, self.slot, self.delta = base, add_bf16, add_bias, slot, delta

    @property
    def plain(self):
        return self.add_bf16 is None and self.add_bias is None and self.slot is None


_MATERIALIZE_CONST = {}


def materialize(parts):
    if parts.plain:
        return parts.base
    d = parts.base.shape[1]
    key = (parts.base.device, d)
    const = _MATERIALIZE_CONST.get(key)
    if const is None:          # the head kernel assembles z on its way; its own outputs are not wanted here: unit gamma, zero head
        const = (torch.zeros(1, d, dtype=torch.float32, device=parts.base.device),
                 torch.ones(d, dtype=torch.float32, device=parts.base.device))
        _MATERIALIZE_CONST[key] = const
    dummy_w, ones = const
    _, _, z = ops.ln_mean_head(parts.base, ones, dummy_w[0], 1e-5, dummy_w, None, parts.add_bf16, parts.add_bias,
                               parts.slot, parts.delta, want_z=True)
    return z


def head(parts, norm, linear, packed=None):
    """logits = Linear(mean_n LayerNorm(z))  (snuffy.py:86,71), residual assembly fused into the read.  packed (ops.PackedBags):
    the rows are B packed bags -> logits [B, C], one mean per bag."""
    if packed is not None:
        logits, _ = ops.ln_mean_head_varlen(parts.base, packed, norm.weight, norm.bias, norm.eps, linear.weight, linear.bias,
                                            parts.add_bf16, parts.add_bias, parts.slot, parts.delta)
        return logits
    if torch.is_grad_enabled() and (parts.base.requires_grad or norm.weight.requires_grad or linear.weight.requires_grad):
        from . import autograd as SA
        return SA.head_train(materialize(parts), norm, linear)
    logits, _, _ = ops.ln_mean_head(parts.base, norm.weight, norm.bias, norm.eps, linear.weight, linear.bias,
                                    parts.add_bf16, parts.add_bias, parts.slot, parts.delta)
    return logits


def _folded(layer, dkp=None):
    """bf16 path: LayerNorm affine folded into the following projection, cached ON THE LAYER until a parameter changes.

    LN(x) W^T + b = xhat (W * gamma)^T + (W beta + b)   with xhat = (x - mean) * rstd.
    The cache key is (id, storage pointer, version counter) of every folded parameter: optimizer steps, load_state_dict and
    parameter replacement refresh it.  An edit through ``p.data`` does not bump the version counter -- call
    ``invalidate_folded(layer)`` (or ``MILNet.invalidate()``) after such an edit.
    """
    n0, n1 = layer.sublayer[0].norm, layer.sublayer[1].norm
    lq, lk, lv, lo = layer.self_attn.linears
    ff = layer.feed_forward
    plist = [n0.weight, n0.bias, n1.weight, n1.bias, lq.weight, lq.bias, lv.weight, lv.bias, ff.w_1.weight,
             ff.w_1.bias, ff.w_2.weight, lk.weight, lk.bias]
    hh = layer.self_attn.h
    dk0 = lq.weight.shape[0] // hh
    dkp = dk0 if dkp is None else dkp                     # head width of the Q | V columns (head_pad; dk0 = not padded)
    key = tuple(param_key(p) for p in plist) + (dkp,)
    ent = getattr(layer, "_fold", None)
    if ent is not None and ent[0] == key:
        return ent[1]
    if dkp != dk0:
        # padded heads (README recipes, dk = 96 -> 128): fold the zero-padded projections; the key projection and the output
        # projection of the padded form come from _padded_heads (fp32, split-bf16 x3 like the unpadded K-row projections)
        with torch.no_grad():
            bf = torch.bfloat16
            g0, b0, g1, b1 = (t.detach().float() for t in (n0.weight, n0.bias, n1.weight, n1.bias))
            wqv, bqv = _qv_weights(layer, dkp)
            w1 = ff.w_1.weight.detach().float()
            bqv_f = (wqv.float() @ b0 + bqv.float()).contiguous()
            b1f = (w1 @ b1 + ff.w_1.bias.detach().float()).contiguous()
            out = dict(wqv=(wqv.float() * g0).to(bf).contiguous(), bqv=bqv_f.to(bf), bqv_f=bqv_f,
                       w1=(w1 * g1).to(bf).contiguous(), b1=b1f, b1h=b1f.to(bf), w2=ff.w_2.weight.detach().to(bf),
                       wk=None, bk=None)
        layer._fold = (key, out)
        return out
    with torch.no_grad():
        g0, b0, g1, b1 = n0.weight, n0.bias, n1.weight, n1.bias
        d, f = lq.weight.shape[1], ff.w_1.weight.shape[0]
        bf = torch.bfloat16
        if lq.weight.is_cuda and d <= 2048 and lq.weight.dtype == torch.float32:
            # one kernel per projection (snf_fold_linear_f32) instead of ~20 small library ops -- the training chain refolds
            # after every optimizer step
            dev = lq.weight.device
            wqv = torch.empty(2 * d, d, dtype=bf, device=dev)
            bqv_f = torch.empty(2 * d, dtype=torch.float32, device=dev)
            bqv = torch.empty(2 * d, dtype=bf, device=dev)
            ops.fold_linear(lq.weight, g0, b0, lq.bias, wqv[:d], bqv_f[:d], bqv[:d])
            ops.fold_linear(lv.weight, g0, b0, lv.bias, wqv[d:], bqv_f[d:], bqv[d:])
            w1 = torch.empty(f, d, dtype=bf, device=dev)
            b1f = torch.empty(f, dtype=torch.float32, device=dev)
            b1h = torch.empty(f, dtype=bf, device=dev)
            ops.fold_linear(ff.w_1.weight, g1, b1, ff.w_1.bias, w1, b1f, b1h)
            out = dict(wqv=wqv, bqv=bqv, bqv_f=bqv_f, w1=w1, b1=b1f, b1h=b1h, w2=ff.w_2.weight.to(bf),
                       wk=lk.weight.to(bf).contiguous(), bk=lk.bias.to(bf).contiguous())
        else:
            out = dict(
                # Q and V projections as ONE GEMM over the shared normalised input: output [N, 2D] = [Q | V]
                wqv=torch.cat([lq.weight * g0, lv.weight * g0]).to(torch.bfloat16).contiguous(),
                bqv=torch.cat([lq.weight @ b0 + lq.bias, lv.weight @ b0 + lv.bias]).to(torch.bfloat16).contiguous(),
                bqv_f=torch.cat([lq.weight @ b0 + lq.bias, lv.weight @ b0 + lv.bias]).float().contiguous(),
                w1=(ff.w_1.weight * g1).to(torch.bfloat16), b1=(ff.w_1.weight @ b1 + ff.w_1.bias).contiguous(),
                b1h=(ff.w_1.weight @ b1 + ff.w_1.bias).to(torch.bfloat16),
                w2=ff.w_2.weight.to(torch.bfloat16),
                # key projection of the K raw selected rows (snuffy.py:190), bf16 operands like Q and V
                wk=lk.weight.to(torch.bfloat16).contiguous(), bk=lk.bias.to(torch.bfloat16).contiguous(),
            )
    layer._fold = (key, out)
    return out


def head_pad(dk):
    """Head width the pipelined fp32-class attention runs a head of true width dk at, or None.  The kernels are built for dk = 64 and
    128; other widths ride zero-padded -- the README recipes train with h = 4 (dk = 96 at D = 384, reference README.md:609-643):
    zero rows in Wq / Wk / Wv give zero columns in Q, Kp and V, which add nothing to Q Kp^T and produce zero columns of O that zero
    columns of Wo ignore: bit-for-bit the products of the true width (one third more matrix work and Q | V bytes at dk = 96)."""
    if dk in (64, 128):
        return dk
    if dk % 16 == 0 and 64 < dk < 128:
        return 128
    if dk % 16 == 0 and 16 <= dk < 64:
        return 64
    return None


def _pad_heads_out(w, b, h, dk, dkp):
    """Rows of w [h dk, d] (and entries of b) regrouped per head and zero-padded to dkp per head -> ([h dkp, d], [h dkp])."""
    d_in = w.shape[1]
    wp = w.new_zeros(h, dkp, d_in)
    wp[:, :dk] = w.detach().view(h, dk, d_in)
    bp = w.new_zeros(h, dkp)
    if b is not None:
        bp[:, :dk] = b.detach().view(h, dk)
    return wp.view(h * dkp, d_in), bp.view(h * dkp)


def _pad_heads_in(w, h, dk, dkp):
    """Columns of w [n, h dk] regrouped per head and zero-padded -> [n, h dkp]."""
    n = w.shape[0]
    wp = w.new_zeros(n, h, dkp)
    wp[:, :, :dk] = w.detach().view(n, h, dk)
    return wp.view(n, h * dkp)


def _qv_weights(layer, dkp):
    """(cat[Wq; Wv], cat[bq; bv]) of the layer, its heads zero-padded to dkp columns when that differs from the true width."""
    lq, lk, lv, lo = layer.self_attn.linears
    h = layer.self_attn.h
    dk = lq.weight.shape[0] // h
    if dkp == dk:
        return torch.cat([lq.weight, lv.weight]).detach(), torch.cat([lq.bias, lv.bias]).detach()
    wq, bq = _pad_heads_out(lq.weight, lq.bias, h, dk, dkp)
    wv, bv = _pad_heads_out(lv.weight, lv.bias, h, dk, dkp)
    return torch.cat([wq, wv]), torch.cat([bq, bv])


def _padded_heads(layer, dkp):
    """Zero-padded key / output projections of a layer whose head width rides padded to dkp (head_pad): (wk [h dkp, d], bk, wo
    [d, h dkp]), cached on the layer like the other derived weights."""
    lq, lk, lv, lo = layer.self_attn.linears
    key = tuple(param_key(p) for p in (lk.weight, lk.bias, lo.weight)) + (dkp,)
    ent = getattr(layer, "_fold_pad", None)
    if ent is not None and ent[0] == key:
        return ent[1]
    h = layer.self_attn.h
    dk = lk.weight.shape[0] // h
    with torch.no_grad():
        wk, bk = _pad_heads_out(lk.weight, lk.bias, h, dk, dkp)
        out = dict(wk=wk.contiguous(), bk=bk.contiguous(), wo=_pad_heads_in(lo.weight, h, dk, dkp).contiguous())
    layer._fold_pad = (key, out)
    return out


def _split_weights(layer, dkp=None):
    """fp32 path on the matrix cores: every projection weight as its split image [Wh | Wl | Wh] (ops.split3_weight), cached
    on the layer like the bf16 fold (same key rule)."""
    lq, lk, lv, lo = layer.self_attn.linears
    ff = layer.feed_forward
    plist = [lq.weight, lq.bias, lv.weight, lv.bias, ff.w_1.weight, ff.w_1.bias, ff.w_2.weight, ff.w_2.bias]
    dkp = lq.weight.shape[0] // layer.self_attn.h if dkp is None else dkp          # head width of the Q | V columns (head_pad)
    key = tuple(param_key(p) for p in plist) + (dkp,)
    ent = getattr(layer, "_fold3", None)
    if ent is not None and ent[0] == key:
        return ent[1]
    with torch.no_grad():
        wqv, bqv = _qv_weights(layer, dkp)
        out = dict(
            wqv=ops.split3_weight(wqv),                                                # [2D, 3D]: output [Q | V]
            bqv=bqv.float().contiguous(),
            w1=ops.split3_weight(ff.w_1.weight), b1=ff.w_1.bias.detach().float().contiguous(),
            w2=ops.split3_weight(ff.w_2.weight), b2=ff.w_2.bias.detach().float().contiguous(),
        )
    out["_wqv_f32"] = wqv            # kept for the interleaved images of the one-pass kernel (_hl_weights), built on first use
    layer._fold3 = (key, out)
    return out


def _hl_weights(layer, fw):
    """Interleaved ("hl") images of the layer's projection weights for the one-pass fp32-class GEMM (ops.gemm_hl): built once per
    parameter version, by the first forward that takes that branch (plain tensors in the layer's cache: the module stays picklable)."""
    if "hl" not in fw:
        ff = layer.feed_forward
        with torch.no_grad():
            fw["hl"] = dict(wqv=ops.split_hl_weight(fw["_wqv_f32"]), w1=ops.split_hl_weight(ff.w_1.weight),
                            w2=ops.split_hl_weight(ff.w_2.weight))
    return fw["hl"]


def _hl_weights_folded(layer, dkp=None):
    """hl images of Wq | Wv and W1 with the LayerNorm affines folded in (FP32_SHARED_NORM):  LN(x) W^T + b = xhat (W * gamma)^T +
    (W beta + b).  The fold is done in fp64 and rounded once to fp32 before the hi / lo split; cached like _split_weights."""
    n0, n1 = layer.sublayer[0].norm, layer.sublayer[1].norm
    lq, lk, lv, lo = layer.self_attn.linears
    ff = layer.feed_forward
    plist = [n0.weight, n0.bias, n1.weight, n1.bias, lq.weight, lq.bias, lv.weight, lv.bias, ff.w_1.weight, ff.w_1.bias]
    dkp = lq.weight.shape[0] // layer.self_attn.h if dkp is None else dkp
    key = tuple(param_key(p) for p in plist) + (dkp,)
    ent = getattr(layer, "_fold3f", None)
    if ent is not None and ent[0] == key:
        return ent[1]
    with torch.no_grad():
        g0, b0, g1, b1 = (t.double() for t in (n0.weight, n0.bias, n1.weight, n1.bias))
        wqv, bqv = (t.double() for t in _qv_weights(layer, dkp))
        w1, bb1 = ff.w_1.weight.double(), ff.w_1.bias.double()
        out = dict(wqv=ops.split_hl_weight((wqv * g0).float().contiguous()), bqv=(wqv @ b0 + bqv).float().contiguous(),
                   w1=ops.split_hl_weight((w1 * g1).float().contiguous()), b1=(w1 @ b1 + bb1).float().contiguous())
    layer._fold3f = (key, out)
    return out


_PARAM_CACHES = ("_snf_x3", "_snf_x3t", "_snf_img_hl", "_snf_img_cat")


def drop_param_caches(module):
    """Forget the split images cached ON the parameters of `module` (split3_cached here, vit._image_of): they are keyed on
    (data_ptr, _version), and an edit through `p.data` changes neither -- invalidate() is the documented way to say so."""
    for prm in module.parameters():
        for name in _PARAM_CACHES:
            if hasattr(prm, name):
                delattr(prm, name)


def invalidate_folded(layer):
    """Drop the layer's folded bf16 weights and any pending critic hand-over (after editing parameters through .data)."""
    layer._fold = None
    layer._fold3 = None
    layer._fold3f = None
    layer._fold3t = None
    layer._fold3t_hl = None
    layer._fold_pad = None
    layer._xhat_offer = None
    layer._xn3_offer = None
    drop_param_caches(layer)


def _rows_linear(x, lin):
    """x W^T + b of an fp32 [rows, D] operand, fp32-class on the matrix cores (FP32_GEMM == "x3"): the skinny kernel for the K selected
    rows of one bag, the tile kernel over split images for thousands of rows (the B x K selected rows of a packed batch); the fp32
    library GEMM for FP32_GEMM == "library", under autograd and for shapes outside both kernels."""
    m, k = x.shape
    n = lin.weight.shape[0]
    if FP32_GEMM == "x3" and not _needs_grad(x, lin.weight, lin.bias) and x.dtype == torch.float32 and lin.weight.dtype == torch.float32:
        if m >= 2048 and ops.gemm_x3_supported(m, n, k):
            return ops.gemm_x3(ops.split3_rows(x), split3_cached(lin.weight), None if lin.bias is None else lin.bias.detach(),
                               out_dtype=torch.float32)
        if m < 2048 and ops.linear_rows_x3_supported(m, n, k):
            # the K rows of one bag: skinny kernel, operands split in registers (the fp32 library GEMM took ~10 us per projection)
            return ops.linear_rows_x3(x, lin.weight.detach(), None if lin.bias is None else lin.bias.detach())
    return F.linear(x, lin.weight, lin.bias)


def encoder_layer(x2, sel, layer, need_attn, precision, packed=None, ragged=None, last=False):
    """EncoderLayer.forward (snuffy.py:126-157) for x2 [N, D] and selected rows sel [K].  Returns (Parts, A).

    packed (ops.PackedBags, inference only): x2 holds the rows of B bags and sel the B x K selected rows in packed coordinates
    (bag b's K rows at sel[b K : (b + 1) K]).  Everything row-wise runs once over the packed rows; only the attention changes
    kernel entry (one varlen launch: every bag attends to its own K keys).  A is then [1, h, T, K] over the packed rows.
    ragged (ops.RaggedKeys, with packed): the bags select different numbers of rows (sel is their concatenation) -- small bags,
    exact-fp32 ragged attention kernel, A [1, h, T, Kmax] with bag b's columns 0 .. K_b - 1 valid.
    last: the result goes straight to head() (no layer follows): the fp32 paths then leave the K patched rows to the head kernel's
    read (Parts with slot / delta) instead of scattering them into z with a launch of their own."""
    if packed is None and torch.is_grad_enabled() and (x2.requires_grad or any(p.requires_grad for p in layer.parameters())):
        from . import autograd as SA  # training path (custom backward kernels); also when only the input asks for a gradient
        return SA.encoder_layer_train(x2, sel, layer, need_attn, precision)
    n, d = x2.shape
    mha, ff = layer.self_attn, layer.feed_forward
    h = mha.h
    k = sel.shape[0]
    kb = k // packed.bags if packed is not None else k            # keys of ONE bag
    if ragged is not None:
        if packed is None or ragged.total != k or not ops.ragged_attn_supported(ragged.kmax, d // h):
            raise SnuffyHipError("ragged packed bags: %d keys (max %d per bag) at head width %d is outside the ragged attention "
                                 "kernel" % (k, ragged.kmax, d // h))
    elif packed is not None and (kb < 1 or not ops.varlen_attn_supported(precision, kb, d // h)):
        raise SnuffyHipError("packed bags: %d keys per bag at head width %d is outside the varlen attention kernels" % (kb, d // h))
    n0, n1 = layer.sublayer[0].norm, layer.sublayer[1].norm
    lq, lk, lv, lo = mha.linears
    if k == 0:
        # nothing selected (possible only when N == 0 rows qualify): the attention sublayer is the identity
        attn = torch.empty(1, h, n, 0, device=x2.device) if need_attn else None
        yn = ops.layernorm_rows(x2, n1.weight, n1.bias, n1.eps)
        hid = torch.mm(yn, ff.w_1.weight.t())
        ops.bias_act_(hid, ff.w_1.bias, ff.activation_name)
        z = torch.addmm(x2, hid, ff.w_2.weight.t())
        ops.bias_act_(z, ff.w_2.bias, "none")
        return Parts(z), attn

    if precision == "fp32" and FP32_GEMM == "x3" and ops.gemm_supported(n, d, 3 * ff.w_2.weight.shape[1]) \
            and ops.gemm_supported(n, 2 * d, 3 * d):
        # fp32-class arithmetic on the matrix cores: every [N, .] projection is ONE bf16 GEMM over a tripled K axis
        # (activations [hi | hi | lo], weights [Wh | Wl | Wh]; products to ~2^-17, fp32 accumulate) -- the fp32 library
        # GEMMs these replace were 85 % of a config-B bag (profiles/r02_bench_cfgB_fp32_kernel_stats.csv)
        f = ff.w_1.weight.shape[0]
        dk = d // h
        # pre-split operands for the pipelined attention kernel: the projection's epilogue writes [Q | V] as its hl image (the same
        # hi / lo values the attention kernel would derive from the fp32 tensor, the same 4 bytes per element); bags too small for
        # the one-pass GEMM get the same image out of the concatenated-K GEMM's epilogue.  Head widths between the kernel's (the
        # README recipes: h = 4, dk = 96) ride zero-padded to the next one (head_pad): dkp columns per head in Q, Kp, V and O.
        dkp = head_pad(dk) if (FP32_ATTENTION == "x3" and X3_HL_ATTENTION) else None
        hl_attn = (ragged is None and packed is None and dkp is not None and d % 16 == 0 and ops.x3_hl_attn_supported(k, dkp)
                   and lk.weight.dtype == torch.float32 and (dkp == dk or ops.gemm_supported(n, 2 * h * dkp, 3 * d)))
        if not hl_attn:
            dkp = dk
        dp = h * dkp                                                                # width of Q, V, Kp, O (d unless padded)
        fw = _split_weights(layer, dkp)
        # large bags: the one-pass kernel on interleaved [hi(32) | lo(32)] images (one full-line DMA per operand row and K step,
        # 96 MFMAs per 24 fragment reads); otherwise the concatenated form over [hi | hi | lo]
        hl = d % 32 == 0 and f % 32 == 0 and ops.hl_eligible(n, 2 * dp, d) and ops.hl_eligible(n, f, d) and ops.hl_eligible(n, d, f)
        fh = _hl_weights(layer, fw) if hl else None
        shared = hl and shared_norm_layer(layer)
        fhf = _hl_weights_folded(layer, dkp) if shared else None
        xn3 = _take_xn3(layer, x2, n0, shared)                                      # left by the critic pass, if any
        fp = _padded_heads(layer, dkp) if dkp != dk else None
        wk, bk = (lk.weight.detach(), None if lk.bias is None else lk.bias.detach()) if fp is None else (fp["wk"], fp["bk"])
        scale = 1.0 / math.sqrt(dk)                                                 # of the TRUE head width (snuffy.py:162)
        # keys = RAW selected rows (K rows: fp32).  For the pipelined kernel the projection writes its fragment image directly -- and
        # gathers its rows itself (round 6: gather, row -> slot map and key projection are ONE launch)
        frag_ok = hl_attn and X3_HL_KPFRAG and ops.x3_hl_kpfrag_supported(k, h, dkp) and x2.dtype == torch.float32
        fused_gather = frag_ok and X3_HL_KPFRAG_GATHER and d % 16 == 0
        if fused_gather:
            kp, xs, slot = ops.gather_linear_rows_x3_kpfrag(x2, sel, wk, bk, h, scale=scale)   # snuffy.py:131,145-147,190
        else:
            xs, slot = ops.gather_slot_map(x2, sel)                                 # snuffy.py:131,145-147 (+ row -> slot map)
        if fused_gather:
            pass
        elif frag_ok:
            kp = ops.linear_rows_x3_kpfrag(xs, wk, bk, h, scale=scale)
        elif fp is not None:
            kp = ops.linear_rows_x3(xs, wk, bk)
        else:
            kp = _rows_linear(xs, lk)
        if hl:
            if xn3 is None:
                xn3 = ops.layernorm_rows_hl(x2, None if shared else n0.weight, None if shared else n0.bias, n0.eps)   # snuffy.py:107
            if shared:
                qv = ops.gemm_hl(xn3, fhf["wqv"], fhf["bqv"], hl_out=hl_attn)
            else:
                qv = ops.gemm_hl(xn3, fh["wqv"], fw["bqv"], hl_out=hl_attn)         # [N, 2D] f32 = [Q | V] (or its [N, 4D] image)
        else:
            xn3 = ops.layernorm_rows_split3(x2, n0.weight, n0.bias, n0.eps)
            qv = ops.gemm_x3(xn3, fw["wqv"], fw["bqv"], out_dtype=torch.float32, hl_out=hl_attn)
        if not shared:
            del xn3
        q, v = (None, None) if hl_attn else (qv[:, :d], qv[:, d:])
        if hl_attn:
            o, attn, _ = ops.sparse_attn_fwd_x3_hl(qv[:, :2 * dp], qv[:, 2 * dp:], kp, h, need_attn=need_attn,
                                                   scale=None if isinstance(kp, ops.KpFrag) else scale)   # snuffy.py:160-168
        elif ragged is not None:
            o, attn, _ = ops.sparse_attn_fwd_ragged(q, v, kp, packed, ragged, h, need_attn=need_attn)
        elif packed is not None:
            o, attn, _ = ops.sparse_attn_fwd_x3_varlen(q, v, kp, packed, kb, h, need_attn=need_attn)
        elif FP32_ATTENTION == "x3" and ops.x3_attn_supported(k, d // h):
            o, attn, _ = ops.sparse_attn_fwd_x3(q, v, kp, h, need_attn=need_attn)    # snuffy.py:160-168
        elif FP32_ATTENTION == "x3" and ops.x3u_attn_supported(k, d // h):
            # head widths no pipelined kernel takes (dk = 192: the README recipe D = 768 / h = 4): the unfused fp32-class pair
            o, attn, _ = ops.sparse_attn_fwd_x3u(q, kp, v, h, need_attn=need_attn)      # q, v: halves of the fused projection, in place
        else:
            o, attn, _ = ops.sparse_attn_fwd(q.contiguous(), kp, v.contiguous(), h, need_attn=need_attn)
        del q, v, qv
        x_sel = None
        if fp is not None:                                              # padded heads: the zero columns of O meet zero columns of Wo
            delta = ops.linear_rows_x3(o, fp["wo"], None if lo.bias is None else lo.bias.detach())
        elif (not shared and FP32_GEMM == "x3" and lo.weight.dtype == torch.float32 and o.shape[0] < 2048
                and ops.linear_rows_x3_supported(o.shape[0], lo.weight.shape[0], o.shape[1])):
            # the output projection also writes x_sel = xs + delta (snuffy.py:205, 108): one launch instead of two (round 6)
            delta, x_sel = ops.linear_rows_x3_resid(o, lo.weight.detach(), None if lo.bias is None else lo.bias.detach(), xs)
        else:
            delta = _rows_linear(o, lo)                                 # snuffy.py:205
        if shared:
            # y differs from x in the K selected rows only (x_sel = xs + delta, snuffy.py:108): re-normalise those into the image
            # the first sublayer used -- one small launch: sum, statistics, hl rows written at their place
            ops.layernorm_rows_hl_patch_(xn3, sel, xs, delta, eps=n1.eps)
            hid3 = ops.gemm_hl(xn3, fhf["w1"], fhf["b1"], ff.activation_name, hl_out=True)   # snuffy.py:224-225, [N, 2F] image
            del xn3
            z = ops.gemm_hl(hid3, fh["w2"], fw["b2"], resid=x2)                    # x + W2 hid + b2: the residual rides in the epilogue
        elif hl:
            if x_sel is None:
                x_sel = xs + delta                                                  # snuffy.py:108
            yn3 = ops.layernorm_rows_hl(x2, n1.weight, n1.bias, n1.eps, slot=slot, patch_rows=x_sel)   # LN(y), y never built
            hid3 = ops.gemm_hl(yn3, fh["w1"], fw["b1"], ff.activation_name, hl_out=True)   # snuffy.py:224-225, [N, 2F] image
            del yn3
            z = ops.gemm_hl(hid3, fh["w2"], fw["b2"], resid=x2)                    # x + W2 hid + b2: the residual rides in the epilogue
        else:
            if x_sel is None:
                x_sel = xs + delta                                                  # snuffy.py:108
            yn3 = ops.layernorm_rows_split3(x2, n1.weight, n1.bias, n1.eps, slot=slot, patch_rows=x_sel)
            hid3 = ops.gemm_x3(yn3, fw["w1"], fw["b1"], ff.activation_name, split3=True)   # [N, 3F] image
            del yn3
            z = ops.gemm_x3(hid3, fw["w2"], fw["b2"], out_dtype=torch.float32, resid=x2)   # x + W2 hid + b2: the residual rides in the epilogue
        del hid3
        if last:                                                                    # rows S: x -> x_sel (snuffy.py:155), in the head's read
            return Parts(z, slot=slot, delta=delta), (attn.unsqueeze(0) if attn is not None else None)
        ops.scatter_add_rows_(z, sel, delta)                                        # rows S: x -> x_sel (snuffy.py:155)
        return Parts(z), (attn.unsqueeze(0) if attn is not None else None)

    if precision == "fp32":
        xs, slot = ops.gather_slot_map(x2, sel)                                     # snuffy.py:131,145-147 (+ row -> slot map)
        kp = _rows_linear(xs, lk)                                       # keys = RAW selected rows
        xn = ops.layernorm_rows(x2, n0.weight, n0.bias, n0.eps)                     # snuffy.py:107
        q = F.linear(xn, lq.weight, lq.bias)
        v = F.linear(xn, lv.weight, lv.bias)
        if ragged is not None:
            o, attn, _ = ops.sparse_attn_fwd_ragged(q, v, kp, packed, ragged, h, need_attn=need_attn)
        elif packed is not None:
            o, attn, _ = ops.sparse_attn_fwd_x3_varlen(q, v, kp, packed, kb, h, need_attn=need_attn)
        elif FP32_ATTENTION == "x3" and ops.x3_attn_supported(k, d // h):
            o, attn, _ = ops.sparse_attn_fwd_x3(q, v, kp, h, need_attn=need_attn)    # snuffy.py:160-168, fp32-class on MFMA
        else:
            o, attn, _ = ops.sparse_attn_fwd(q, kp, v, h, need_attn=need_attn)       # exact fp32 on the vector ALUs
        del q, v, xn
        delta = _rows_linear(o, lo)                                     # snuffy.py:205
        x_sel = xs + delta                                                          # snuffy.py:108
        yn = ops.layernorm_rows(x2, n1.weight, n1.bias, n1.eps, slot=slot, patch_rows=x_sel)  # LN(y), y never built
        hid = torch.mm(yn, ff.w_1.weight.t())
        del yn
        ops.bias_act_(hid, ff.w_1.bias, ff.activation_name)                         # snuffy.py:224-225
        z = torch.addmm(x2, hid, ff.w_2.weight.t())                                 # x + W2 hid  (residual in the GEMM)
        del hid
        ops.bias_act_(z, ff.w_2.bias, "none")
        ops.scatter_add_rows_(z, sel, delta)                                        # rows S: x -> x_sel (snuffy.py:155)
        return Parts(z), (attn.unsqueeze(0) if attn is not None else None)

    # ---- bf16 path -------------------------------------------------------------------------------------------------
    if n0.eps != n1.eps:
        raise NotImplementedError("bf16 path shares one normalisation between both sublayers: eps must match")
    dk = d // h
    # head widths between the MFMA kernel's ride zero-padded (head_pad: the README recipes' dk = 96 -> 128), single bags only
    dkp = head_pad(dk) if (ragged is None and packed is None and lk.weight.dtype == torch.float32) else dk
    if dkp is None or (dkp != dk and not (ops.mfma_attn_supported(k, dkp, n, 2 * h * dkp) and ops.linear_rows_x3_supported(k, h * dkp, d)
                                         and ops.gemm_supported(n, 2 * h * dkp, d))):
        dkp = dk
    dp = h * dkp
    fw = _folded(layer, dkp)
    fp = _padded_heads(layer, dkp) if dkp != dk else None
    xhat = _take_xhat(layer, x2, n0.eps)                                                 # left by the critic pass, if any
    if xhat is None:
        xhat = torch.empty(n, d, dtype=torch.bfloat16, device=x2.device)
        ops.layernorm_rows(x2, None, None, n0.eps, out=xhat)
    qv = ops.linear_bf16(xhat, fw["wqv"], fw["bqv_f"], fw["bqv"])                   # [N, 2D] bf16 = [Q | V], bias epilogue
    q, v = qv[:, :dp], qv[:, dp:]                                                   # row-strided views, used in place
    if ragged is None and (packed is not None or ops.mfma_attn_supported(k, dkp, n, qv.stride(0))):
        # keys = RAW selected rows: the gather also leaves them in bf16, the projection runs like Q | V (bf16 operands,
        # fp32 accumulate, bf16 out) and the attention kernel reads Kp as it is
        xs, slot, xs16 = ops.gather_slot_map(x2, sel, bf16_copy=True)               # snuffy.py:131,145-147 (+ row -> slot map)
        if fp is not None:
            kp = ops.linear_rows_x3(xs, fp["wk"], fp["bk"], out_dtype=torch.bfloat16)
        elif (FP32_GEMM == "x3" and xs.shape[0] < 2048 and ops.linear_rows_x3_supported(xs.shape[0], d, d)
                and lk.weight.dtype == torch.float32):
            # keys of one bag: fp32-class product of the fp32 rows, rounded once to the bf16 the attention kernel reads
            kp = ops.linear_rows_x3(xs, lk.weight.detach(), lk.bias.detach(), out_dtype=torch.bfloat16)
        else:
            kp = torch.addmm(fw["bk"], xs16, fw["wk"].t())
        if packed is not None:
            o, attn, _ = ops.sparse_attn_fwd_mfma_varlen(q, v, kp, packed, kb, h, need_attn=need_attn)
        else:
            o, attn, _ = ops.sparse_attn_fwd_mfma(q, v, kp, n, h, scale=1.0 / math.sqrt(dk), need_attn=need_attn)
    else:
        xs, slot = ops.gather_slot_map(x2, sel)
        kp = _rows_linear(xs, lk)
        if ragged is not None:
            qvf = qv.float()
            o, attn, _ = ops.sparse_attn_fwd_ragged(qvf[:, :d], qvf[:, d:], kp, packed, ragged, h, need_attn=need_attn)
            del qvf
        else:
            o, attn, _ = ops.sparse_attn_fwd(q.float(), kp, v.float(), h, need_attn=need_attn)
    del q, v, qv
    if fp is not None:
        delta = ops.linear_rows_x3(o, fp["wo"], None if lo.bias is None else lo.bias.detach())
    else:
        delta = _rows_linear(o, lo)
    x_sel = xs + delta
    ops.layernorm_rows(x_sel, None, None, n1.eps, out=xhat, out_row_idx=sel)        # re-normalise the K rows in place
    # W1 + bias + activation in the GEMM epilogue: [N, F] bf16.  GELU is the reference's erf form (nn.GELU(), snuffy.py:218) --
    # the library's fused epilogue only has the tanh form, so gelu / leakyrelu / selu always run on the native kernel
    hid = ops.linear_bf16(xhat, fw["w1"], fw["b1"], fw["b1h"], ff.activation_name)
    zb = ops.linear_bf16(hid, fw["w2"])                                             # [N, D] bf16
    del hid
    parts = Parts(x2, add_bf16=zb, add_bias=ff.w_2.bias, slot=slot, delta=delta)
    return parts, (attn.unsqueeze(0) if attn is not None else None)
