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

from . import functional as SF
from . import ops

# bf16 training: the first encoder layer as one hand-ordered forward / backward chain (EncoderLayer0Bf16Fn) instead of the generic
# autograd graph; False keeps the generic chain everywhere (tests compare the two)
FUSED_BF16_TRAINING = True
# fp32 training: the same for the fp32-class arithmetic (EncoderLayer0X3Fn, round 5)
FUSED_X3_TRAINING = True
# ... on the one-pass kernels (round 6): activations and gradients as interleaved hl images (4 bytes per element instead of the 6 of
# [hi | hi | lo]), projections on gemm_hl, weight gradients on gemm_tn(hl=True); False: the concatenated-K chain of round 5
X3_TRAIN_HL = True
# ... and, in that chain, the ReLU mask of the FFN input gradient in the GEMM's epilogue (snf_gemm_hl_gated_bf16); False: gemm_hl + split pass
X3_TRAIN_GATED_GEMM = True

_ACT = {
    "relu": F.relu,
    "gelu": F.gelu,
    "leakyrelu": lambda t: F.leaky_relu(t, 0.01),
    "selu": F.selu,
}


class LayerNormRowsFn(torch.autograd.Function):
    """y = LayerNorm(x) * gamma + beta (forward: snf_layernorm_rows_f32; backward: snf_layernorm_rows_bwd_f32, one pass that
    recomputes the statistics from x and leaves per-workgroup partial sums for dgamma / dbeta)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, gamma, beta, eps):
        y = ops.layernorm_rows(x, gamma, beta, eps)
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dx, _, dgamma, dbeta = ops.layernorm_rows_bwd(x, dy.float(), gamma, ctx.eps, want_dx=ctx.needs_input_grad[0])
        return dx, dgamma, dbeta, None


FUSED_MIL_LOSS = True   # the loss head of a training step as one launch each way (snf_mil_loss_f32); False: the torch formulation


class MilLossFn(torch.autograd.Function):
    """loss = w BCE(logits, y) + (1 - w) BCE(max_n ins, y) and the bag prediction (reference train.py, _run_model of the single-weight
    trainer) in ONE launch, its backward in one more: the torch formulation is ~30 scalar-sized launches per step -- 3 % of an fp32-class
    step's GPU time, a tenth of the host-bound bf16 step."""

    @staticmethod
    def forward(ctx, ins, logits, label, w, pos_weight, weight):
        n, c = ins.shape
        lib = ops._ffi.load()
        out = torch.empty(2 + 4 * c, dtype=torch.float32, device=ins.device)
        arg = torch.empty(c, dtype=torch.int64, device=ins.device)
        ops.check(lib.snf_mil_loss_f32(ops._p(ins), n, c, ops._p(logits), ops._p(label), ops._p(w), ops._p(pos_weight), ops._p(weight), ops._p(out), ops._p(arg),
                                       ops._stream()), "snf_mil_loss_f32")
        ctx.save_for_backward(out, arg)
        ctx.shape = (n, c)
        ctx.mark_non_differentiable(arg)
        bag_pred = out[2:2 + c]
        ctx.mark_non_differentiable(bag_pred)
        return out[0], bag_pred, arg

    @staticmethod
    def backward(ctx, go, _gp, _ga):
        out, arg = ctx.saved_tensors
        n, c = ctx.shape
        d_ins = torch.empty(n, c, dtype=torch.float32, device=out.device)
        small = torch.empty(c + 1, dtype=torch.float32, device=out.device)
        go = go.reshape(1).float().contiguous()
        ops.check(ops._ffi.load().snf_mil_loss_bwd_f32(ops._p(go), ops._p(out), ops._p(arg), n, c, ops._p(d_ins), ops._p(small), ops._stream()),
                  "snf_mil_loss_bwd_f32")
        return d_ins, small[:c], None, small[c].reshape(()), None, None


def mil_loss(ins_prediction, bag_prediction, bag_label, w, criterion):
    """(loss, bag_pred) of the single-weight loss head -- fused where it applies (fp32 scores on the GPU, <= 8 classes, a plain
    BCEWithLogitsLoss with mean reduction), else None: the caller keeps the reference formulation."""
    if not FUSED_MIL_LOSS or not isinstance(criterion, torch.nn.BCEWithLogitsLoss) or criterion.reduction != "mean":
        return None
    ins = ins_prediction if ins_prediction.dim() == 2 else ins_prediction.reshape(-1, ins_prediction.shape[-1])
    if ins_prediction.dim() == 3 and ins_prediction.shape[0] != 1:
        return None
    c = ins.shape[1]
    if not ins.is_cuda or ins.dtype != torch.float32 or ins.shape[0] < 1 or c > 8 or bag_prediction.numel() != c or bag_label.numel() != c \
            or bag_prediction.dtype != torch.float32:
        return None
    vecs = []
    for v in (criterion.pos_weight, criterion.weight):      # (the reference hands its class weights over positionally: `weight`, train.py:246)
        if v is not None:
            if v.numel() == 1:
                v = v.reshape(1).expand(c)
            if v.numel() != c:
                return None
            v = v.reshape(c).to(device=ins.device, dtype=torch.float32).contiguous()
        vecs.append(v)
    pw, cw = vecs
    wt = w if torch.is_tensor(w) else torch.tensor(float(w), device=ins.device)
    wt = wt.to(device=ins.device, dtype=torch.float32)
    loss, bag_pred, _ = MilLossFn.apply(ins.contiguous(), bag_prediction.reshape(c).contiguous(),
                                        bag_label.reshape(c).to(device=ins.device, dtype=torch.float32).contiguous(), wt.reshape(()), pw, cw)
    return loss, bag_pred


class HeadFn(torch.autograd.Function):
    """logits = Linear(mean_n LayerNorm(z)) (snuffy.py:86,71): forward on the fused column-reduction kernels
    (snf_ln_mean_head_f32); backward in ONE pass over z -- every row receives the same gradient row d pooled / N, which
    snf_layernorm_rows_bwd_f32 broadcasts."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, z, gamma, beta, w, b, eps):
        logits, pooled, _ = ops.ln_mean_head(z, gamma, beta, eps, w, b)
        ctx.save_for_backward(z, gamma, w, pooled)
        ctx.eps, ctx.has_bias = eps, b is not None
        return logits

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dlogits):
        z, gamma, w, pooled = ctx.saved_tensors
        dlogits = dlogits.float().reshape(-1)
        dpooled = dlogits @ w                                   # [D]
        dw = torch.outer(dlogits, pooled)
        dy_row = (dpooled / z.shape[0]).contiguous()            # d loss / d LN(z)_i, identical for every row i
        dz, _, dgamma, dbeta = ops.layernorm_rows_bwd(z, dy_row, gamma, ctx.eps, want_dx=ctx.needs_input_grad[0])
        return dz, dgamma, dbeta, dw, (dlogits if ctx.has_bias else None), None


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
        in_kernel = False
        if bf16_operands and ops.mfma_attn_supported(k, dk):
            out, p, _ = ops.sparse_attn_fwd_mfma(q.to(torch.bfloat16), v.to(torch.bfloat16), kp, n, h, need_attn=True)
        elif ops.x3_attn_supported(k, dk) and SF.FP32_ATTENTION != "exact":
            in_kernel = drop is not None and ops.x3_attn_dropout_supported(k, dk)       # mask applied to P in registers, P written undropped
            out, p, _ = ops.sparse_attn_fwd_x3(q, v, kp, h, need_attn=True, dropout=drop if in_kernel else None)
        else:
            out, p, _ = ops.sparse_attn_fwd(q, kp, v, h, need_attn=True)
        mask = None
        # (round 6) nobody asked for the dropped P and the forward applied the mask itself: the backward regenerates it too, no tensor
        ctx.regen = bool(in_kernel and not want_p and ops.attn_bwd_dropout_supported(k, dk))
        if drop is not None and not ctx.regen:
            mask = ops.dropout_mask(h, n, k, drop[0], drop[1], drop[2], q.device)
            if not in_kernel:
                vh = v.float().reshape(n, h, dk).transpose(0, 1)
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
        dq, dkp, dv = ops.sparse_attn_bwd(q, kp, v, p, dout.float().contiguous(), h, mask=mask, scale=scale,
                                          dropout=ctx.drop if ctx.regen else None)
        return dq, dkp, dv, None, None, None, None


def _tn_mm(a, b, chunks=8):
    """a^T b for a [n, p], b [n, q] bf16 -> [p, q] f32: the weight-gradient contraction over the bag axis.  The output is small
    (p q <= 2.4 M) and the contraction long (n = bag size), so the library's one-tile-per-workgroup kernels leave most CUs idle
    (36 tiles of 256 x 256 at config B: 311 us); split over `chunks` row blocks as ONE batched GEMM plus an fp32 sum of the
    partials (180-190 us, profiles/r02_tn_gemm.txt)."""
    n = a.shape[0]
    if a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and ops.gemm_tn_supported(n, a.shape[1], b.shape[1], a, b):
        # round 6: one bf16 product on snf_gemm_tn_f32 (fp32 partials summed in part order: no bf16 rounding of the chunk results)
        return ops.gemm_tn(a, b, a.shape[1], b.shape[1])
    m = (n // chunks) * chunks
    if n < 4096 or m == 0:
        return torch.mm(a.t(), b).float()
    av = a[:m].view(chunks, m // chunks, a.shape[1]).transpose(1, 2)
    out = torch.bmm(av, b[:m].view(chunks, m // chunks, b.shape[1])).sum(0, dtype=torch.float32)
    if m < n:
        out += torch.mm(a[m:].t(), b[m:]).float()
    return out


_MM_F32_OUT = None      # does this torch build take out_dtype on bf16 GEMMs (fp32 results without a bf16 rounding)?


def _tn_mm_f32(a, b, chunks=8):
    """a^T b for bf16 a [n, p], b [n, q] (row-strided views allowed) -> [p, q] f32 with NO bf16 rounding of the result: the
    contraction over the bag axis of the fp32-class weight gradients.  Library GEMM with an fp32 output (out_dtype), split
    over row chunks like _tn_mm; a build without out_dtype falls back to the fp32 library GEMM."""
    global _MM_F32_OUT
    n = a.shape[0]
    if _MM_F32_OUT is None:
        try:
            torch.mm(a[:8].t(), b[:8], out_dtype=torch.float32)
            _MM_F32_OUT = True
        except (TypeError, RuntimeError, NotImplementedError):
            _MM_F32_OUT = False
    if not _MM_F32_OUT:
        return torch.mm(a.float().t(), b.float())
    m = (n // chunks) * chunks
    if n < 4096 or m == 0:
        return torch.mm(a.t(), b, out_dtype=torch.float32)
    av = a[:m].view(chunks, m // chunks, a.shape[1]).transpose(1, 2)
    out = torch.bmm(av, b[:m].view(chunks, m // chunks, b.shape[1]), out_dtype=torch.float32).sum(0)
    if m < n:
        out += torch.mm(a[m:].t(), b[m:], out_dtype=torch.float32)
    return out


class LinearX3Fn(torch.autograd.Function):
    """y = x W^T + b with fp32 tensors and fp32-class arithmetic on the matrix cores, forward AND backward (reference train.py:259
    differentiates nn.Linear in fp32; an fp32 library GEMM runs at 60-100 TFLOP/s here).  Every product is taken as
    hi hi + hi lo + lo hi of the operands' bf16 halves with fp32 accumulation (~1e-5 per product, as functional.FP32_GEMM = "x3"):
      forward   the split image [hi | hi | lo] of x against [Wh | Wl | Wh]                     (ops.gemm_bf16, one launch)
      dx        the split image of dy against the same image of W^T                            (ops.gemm_bf16)
      dW        dyh^T xh + dyh^T xl + dyl^T xh over the bag axis -- column blocks of the two images, in place -- as three library
                GEMMs with fp32 results (lo lo is never computed)
    The image of x (bf16 [n, 3 k]) is what is saved for the backward."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        img = ops.split3_rows(x.float() if x.dtype != torch.float32 else x)
        y = ops.gemm_x3(img, SF.split3_cached(weight), None if bias is None else bias.detach().float().contiguous(),
                          out_dtype=torch.float32)
        ctx.save_for_backward(img, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        img, weight = ctx.saved_tensors
        n_out, k = weight.shape
        dimg = ops.split3_rows(dy.float().contiguous())
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_x3(dimg, SF.split3_cached(weight, transposed=True), None, out_dtype=torch.float32)
        if ctx.needs_input_grad[1]:
            # dyh^T xh + dyh^T xl + dyl^T xh over the bag axis: three fp32-output GEMMs on column blocks of the two images, in place
            # (one GEMM over [dyh | dyl]^T [xh | xl] also computes the dropped lo lo block: a quarter more work, measured 682 vs 3 x 170 us)
            # (round 6: snf_gemm_tn_f32 where the shape allows -- the three products out of each staged step of the two images)
            dw = _tn3(dimg, img, n_out, k)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.float().sum(0)
        return dx, dw, db


def linear_x3_ok(x, weight):
    """fp32-class linear on the matrix cores applies: fp32 training path with FP32_GEMM = "x3", shapes in the GEMM's domain."""
    n, k = x.shape
    return (SF.FP32_GEMM == "x3" and not torch.is_autocast_enabled() and x.is_cuda and x.dtype == torch.float32 and n >= 256
            and k % 8 == 0 and weight.shape[0] % 8 == 0 and ops.gemm_x3_supported(n, weight.shape[0], k)
            and ops.gemm_x3_supported(n, k, weight.shape[0]))


def _linear(x, lin):
    if linear_x3_ok(x, lin.weight):
        return LinearX3Fn.apply(x, lin.weight, lin.bias)
    return F.linear(x, lin.weight, lin.bias)


class EncoderLayer0Bf16Fn(torch.autograd.Function):
    """EncoderLayer.forward + backward (snuffy.py:126-157) for the FIRST layer of a bf16 training step, as one hand-ordered
    chain: the bag x is data there (no gradient), so nothing upstream of the K selected rows needs d/dx and the backward
    collapses to the weight gradients plus the K-row chain through the attention output.

    Forward = the inference pipeline of functional.encoder_layer (LayerNorm affine folded into the projection weights,
    one normalised bf16 copy of the bag, fused Q|V projection, MFMA attention with in-kernel dropout, FFN with the
    activation in the GEMM epilogue).  Backward: dz -> FFN weight gradients (three [N, .] GEMMs) -> the K selected rows
    through LayerNorm 1 / the output projection -> MFMA attention backward -> Q|V weight gradients; the folded-weight
    gradients are unfolded into (W, b, gamma, beta) at the end.  Replaces ~150 autograd nodes (casts, adds, reductions:
    2.6 ms of a 5.4 ms step at config B) by 9 large kernels and a handful of [K, D] / [4D, D] ones."""

    @staticmethod
    def forward(ctx, x2, sel, layer, need_attn, g0, b0, g1, b1, wq, bq, wk, bk, wv, bv, wo, bo, w1, bb1, w2, bb2):
        from . import functional as SF
        n, d = x2.shape
        mha = layer.self_attn
        h = mha.h
        eps = layer.sublayer[0].norm.eps
        fw = SF._folded(layer)
        xhat = SF._take_xhat(layer, x2, eps)                                           # left by the critic pass, if any
        if xhat is None:
            xhat = torch.empty(n, d, dtype=torch.bfloat16, device=x2.device)
            ops.layernorm_rows(x2, None, None, eps, out=xhat)
        qv = ops.linear_bf16(xhat, fw["wqv"], fw["bqv_f"], fw["bqv"])
        q, v = qv[:, :d], qv[:, d:]
        xs, slot, xs16 = ops.gather_slot_map(x2, sel, bf16_copy=True)
        kp = torch.addmm(fw["bk"], xs16, fw["wk"].t()).float()
        p_drop = mha.dropout.p if layer.training else 0.0
        drop = (float(p_drop),) + draw_dropout_state() if p_drop > 0.0 else None
        o, attn, lse = ops.sparse_attn_fwd_mfma(q, v, kp, n, h, need_attn=need_attn, need_lse=True, dropout=drop)
        delta = torch.addmm(bo, o, wo.t())
        x_sel = xs + delta
        xhat0_sel = xhat.index_select(0, sel)
        ops.layernorm_rows(x_sel, None, None, eps, out=xhat, out_row_idx=sel)
        hid = ops.linear_bf16(xhat, fw["w1"], fw["b1"], fw["b1h"], "relu")
        zb = ops.linear_bf16(hid, fw["w2"])
        z = SF.materialize(SF.Parts(x2, add_bf16=zb, add_bias=bb2, slot=slot, delta=delta))
        ctx.save_for_backward(sel, xhat0_sel, qv, kp, lse, o, xs, x_sel, hid, g0, b0, g1, b1, wq, wk, wv, wo, w1,
                              fw["w1"], fw["w2"])
        ctx.xhat = xhat          # read-only in backward; outside save_for_backward because the forward wrote rows S in place after the
        #                          Q|V projection had read it (the version check would reject it)
        ctx.h, ctx.eps, ctx.drop = h, eps, drop
        ctx.mark_non_differentiable(*([attn] if attn is not None else []))
        return z, attn

    @staticmethod
    def backward(ctx, dz, _dattn):
        (sel, xhat0_sel, qv, kp, lse, o, xs, x_sel, hid, g0, b0, g1, b1, wq, wk, wv, wo, w1, w1f, w2f) = ctx.saved_tensors
        xhat = ctx.xhat
        n, d = xhat.shape
        h, eps = ctx.h, ctx.eps
        f32 = torch.float32
        dz = dz.float().contiguous()
        # ---- FFN: z = y + act(xhat1 W1'^T + b1') W2^T + b2                                                (snuffy.py:224-225)
        db2, dz16 = ops.colsum_fused(dz, want_bf16=True)                              # bias gradient + the GEMM operand, one pass
        dw2 = _tn_mm(dz16, hid)                                                        # [D, F]
        dhid = torch.mm(dz16, w2f)                                                   # [N, F] bf16
        db1f, _ = ops.colsum_fused(dhid, gate=hid, inplace=True)                     # ReLU mask from the output + bias gradient
        dw1f = _tn_mm(dhid, xhat)                                                     # [F, D] gradient of the FOLDED weight
        # ---- the K selected rows: y[S] = x_sel = xs + o Wo^T + bo; every other row of y is data             (snuffy.py:108,152-155)
        dyn_s = torch.mm(dhid.index_select(0, sel), w1f, out_dtype=f32)              # d loss / d xhat1[S] (bf16 operands as they are)
        # dy[S] = dz[S] + LayerNorm-1 backward of dyn_s at the rows x_sel (affine folded away): one kernel
        dy_s, _, _, _ = ops.layernorm_rows_bwd(x_sel, dyn_s, None, eps, residual=dz.index_select(0, sel), want_param_grads=False)
        dbo = dy_s.sum(0)
        dwo = dy_s.t() @ o
        do = dy_s @ wo
        # ---- attention                                                                                     (snuffy.py:160-168)
        q, v = qv[:, :d], qv[:, d:]
        dq, dkp, dv = ops.sparse_attn_bwd_mfma(q, v, kp, do.contiguous(), lse, h, dropout=ctx.drop, fused_bf16_grads=True)
        dqv = dq._base                                                                # [N, 2D] bf16 = [dQ | dV]
        del dq, dv
        # xhat holds LayerNorm 1's rows at S since the forward re-normalised them in place; the Q|V projection saw LayerNorm 0's.
        # The buffer is only READ here: dW = dqv^T xhat + dqv[S]^T (xhat0[S] - xhat1[S]), a K-row correction of the big product
        dwqvf = _tn_mm(dqv, xhat)                                                     # [2D, D] folded
        corr = (xhat0_sel.float() - xhat.index_select(0, sel).float())               # [K, D]
        dwqvf += dqv.index_select(0, sel).float().t() @ corr
        dbqvf, _ = ops.colsum_fused(dqv)
        dwk = dkp.t() @ xs
        dbk = dkp.sum(0)
        # ---- unfold  W' = W * gamma,  b' = W beta + b  (one kernel per projection: dW, and the partial sums of dgamma / dbeta)
        dbq, dbv = dbqvf[:d].contiguous(), dbqvf[d:].contiguous()
        if d <= 2048:
            dwq, dgq, dbq0 = ops.unfold_linear(dwqvf[:d], wq, g0, b0, dbq)
            dwv, dgv, dbv0 = ops.unfold_linear(dwqvf[d:], wv, g0, b0, dbv)
            dg0, db0 = dgq + dgv, dbq0 + dbv0
            dw1, dg1, db1 = ops.unfold_linear(dw1f, w1, g1, b1, db1f)
        else:
            dwqf, dwvf = dwqvf[:d], dwqvf[d:]
            dg0 = (dwqf * wq).sum(0) + (dwvf * wv).sum(0)
            db0 = dbq @ wq + dbv @ wv
            dg1 = (dw1f * w1).sum(0)
            db1 = db1f @ w1
            dwq = dwqf * g0 + torch.outer(dbq, b0)
            dwv = dwvf * g0 + torch.outer(dbv, b0)
            dw1 = dw1f * g1 + torch.outer(db1f, b1)
        return (None, None, None, None, dg0, db0, dg1, db1, dwq, dbq, dwk, dbk, dwv, dbv, dwo, dbo, dw1, db1f, dw2, db2)


def fused_layer0_shape_ok(layer, n, d, k=None):
    """Settings / shapes EncoderLayer0Bf16Fn covers (k = number of selected rows; default: the layer's Lambda capped by n)."""
    if not FUSED_BF16_TRAINING:
        return False
    mha, ff = layer.self_attn, layer.feed_forward
    n0, n1 = layer.sublayer[0].norm, layer.sublayer[1].norm
    if ff.activation_name != "relu" or n0.eps != n1.eps or d % mha.h:
        return False
    if layer.training and (layer.sublayer[0].dropout.p > 0 or layer.sublayer[1].dropout.p > 0 or ff.dropout.p > 0):
        return False
    if k is None:
        k = min(int(layer.big_lambda), n)
    dk = d // mha.h
    return (k >= 1 and ops.mfma_attn_supported(k, dk, n, 2 * d) and k <= (224 if dk == 128 else 256)
            and ops.mfma_attn_bwd_supported(k, dk) and all(p.requires_grad for p in layer.parameters()))


def fused_layer0_ok(x2, sel, layer, precision):
    """The first-layer chain applies: bf16, the bag is data (no gradient flows into x2), supported shape / settings;
    everything else keeps the generic autograd chain below."""
    if precision != "bf16" or x2.requires_grad or sel.numel() == 0:
        return False
    return fused_layer0_shape_ok(layer, x2.shape[0], x2.shape[1], sel.numel())


def _x3_train_weights(layer, hl=False):
    """Operands of EncoderLayer0X3Fn, cached on the layer per parameter version: the LayerNorm affines folded into the following
    projection in fp32 (LN(x) W^T + b = xhat (W * gamma)^T + (W beta + b)) and the split images [Wh | Wl | Wh] (hl: the interleaved
    images) of the folded Q | V and FFN-in weights, of W2 and of W2^T (the operand of the FFN-out input gradient)."""
    n0, n1 = layer.sublayer[0].norm, layer.sublayer[1].norm
    lq, lk, lv, lo = layer.self_attn.linears
    ff = layer.feed_forward
    plist = [n0.weight, n0.bias, n1.weight, n1.bias, lq.weight, lq.bias, lv.weight, lv.bias, ff.w_1.weight, ff.w_1.bias, ff.w_2.weight]
    key = tuple(SF.param_key(p) for p in plist)
    attr = "_fold3t_hl" if hl else "_fold3t"
    ent = getattr(layer, attr, None)
    if ent is not None and ent[0] == key:
        return ent[1]
    with torch.no_grad():
        wqv = torch.cat([lq.weight, lv.weight]).float()
        w1 = ff.w_1.weight.float()
        w1f = (w1 * n1.weight).contiguous()
        out = dict(bqv=(wqv @ n0.bias + torch.cat([lq.bias, lv.bias])).float().contiguous(), w1f=w1f,
                   b1=(w1 @ n1.bias + ff.w_1.bias).float().contiguous())
        if hl:
            w2 = ff.w_2.weight.float()
            out.update(wqv_hl=ops.split_hl_rows((wqv * n0.weight).contiguous()), w1_hl=ops.split_hl_rows(w1f),
                       w2_hl=ops.split_hl_rows(w2.contiguous()), w2t_hl=ops.split_hl_rows(w2.t().contiguous()))
        else:
            out.update(wqv3=ops.split3_weight(wqv, n0.weight), w1_3=ops.split3_weight(w1f), w2_3=ops.split3_weight(ff.w_2.weight),
                       w2t_3=ops.split3_weight(ff.w_2.weight.t().contiguous()))
    setattr(layer, attr, (key, out))
    return out


def _hl_planes_sum(img):
    """fp32 value hi + lo of every element of a (small) hl image [r, 2 k] -> [r, k]."""
    v = img.view(img.shape[0], -1, 2, 32).float()
    return (v[:, :, 0] + v[:, :, 1]).reshape(img.shape[0], -1)


def _x3_train_hl_ok(n, d, f):
    """The one-pass chain applies: hl images (32-column groups), every projection fills the chip with 256 x 256 tiles, the three weight
    gradients are in gemm_tn's domain."""
    return (X3_TRAIN_HL and d % 32 == 0 and f % 32 == 0 and n % 32 == 0 and ops.hl_eligible(n, 2 * d, d) and ops.hl_eligible(n, f, d)
            and ops.hl_eligible(n, d, f) and ops.GEMM_TN and n >= 1024 and d * f >= 65536)


_BMM_OUT = None         # does this torch build take out= together with out_dtype on bmm?


def _tn3(a3, b3, p, q, chunks=None):
    """a^T b over the bag axis for the split images a3 [n, 3 p] = [hi | hi | lo], b3 [n, 3 q] -> [p, q] f32, fp32-class:
    ah^T bh + ah^T bl + al^T bh on column blocks of the images, in place (lo lo is never computed).  Each product is a batched
    library GEMM over `chunks` row blocks (see _tn_mm); the 3 x chunks partial results land in ONE buffer and are summed in one
    pass, in a fixed order.  chunks: 4 for the large outputs (the FFN weights at config B: 48 library tiles x 4 chunks fill the chip as
    well as x 8 and leave half the partials -- 577 -> 540 us, 562 -> 511 us), 8 below 2 M elements (288 vs 369 us; tools/tn_chunks_bench.py)."""
    global _BMM_OUT
    if ops.gemm_tn_supported(a3.shape[0], p, q, a3, b3):
        # round 6: the contraction on the matrix cores straight from the row-major images (snf_gemm_tn_f32: transposing LDS reads, the
        # three products per staged step, tiles cut into row parts and summed in part order)
        return ops.gemm_tn(a3, b3, p, q, (p, 2 * p), (q, 2 * q))
    if chunks is None:
        chunks = 4 if p * q >= (1 << 21) else 8
    ah, al = a3[:, p:2 * p], a3[:, 2 * p:]
    bh, bl = b3[:, q:2 * q], b3[:, 2 * q:]
    n = a3.shape[0]
    m = (n // chunks) * chunks
    if n >= 4096 and m == n and _MM_F32_OUT is not False and _BMM_OUT is not False:
        buf = torch.empty(3 * chunks, p, q, dtype=torch.float32, device=a3.device)
        r = n // chunks
        try:
            for i, (a, b) in enumerate(((ah, bh), (ah, bl), (al, bh))):
                torch.bmm(a.view(chunks, r, p).transpose(1, 2), b.view(chunks, r, q), out_dtype=torch.float32,
                          out=buf[i * chunks:(i + 1) * chunks])
            _BMM_OUT = True
            return buf.sum(0)
        except (TypeError, RuntimeError, NotImplementedError) as exc:
            if "out of memory" in str(exc).lower():
                raise
            _BMM_OUT = False       # this torch build has no bmm(out_dtype=, out=): three separate products below
    out = _tn_mm_f32(ah, bh, chunks)
    out += _tn_mm_f32(ah, bl, chunks)
    out += _tn_mm_f32(al, bh, chunks)
    return out


class EncoderLayer0X3Fn(torch.autograd.Function):
    """EncoderLayer.forward + backward (snuffy.py:126-157) for the FIRST layer of an fp32 training step in fp32-class arithmetic (every
    large product as split-bf16 x 3 on the matrix cores), as one hand-ordered chain -- the fp32 twin of EncoderLayer0Bf16Fn (round 5).

    The bag x is data there, so nothing upstream of the K selected rows needs d/dx: of the three input-gradient GEMMs of the generic
    autograd graph only the FFN-out one is left, the LayerNorm backward runs on K rows, and the ~150 elementwise / reduction nodes
    between the kernels (2 ms of a 7.6 ms step at config B) disappear.  Activations live as split images [hi | hi | lo]: the
    LayerNorm kernel writes the normalised bag as one (affines folded into the weights, ONE image for both sublayers with the K
    selected rows re-normalised in place), the FFN-in GEMM writes the hidden layer as one in its epilogue; the weight gradients
    contract column blocks of those images over the bag axis (three fp32-output library GEMMs per product).  The attention backward
    is the exact kernel on the materialised P (dropout: the same Philox mask tensor as SparseAttnFn)."""

    @staticmethod
    def forward(ctx, x2, sel, layer, need_attn, g0, b0, g1, b1, wq, bq, wk, bk, wv, bv, wo, bo, w1, bb1, w2, bb2):
        n, d = x2.shape
        mha = layer.self_attn
        h = mha.h
        dk = d // h
        k = sel.numel()
        eps = layer.sublayer[0].norm.eps
        hl = _x3_train_hl_ok(n, d, w1.shape[0])
        fw = _x3_train_weights(layer, hl)
        if hl:
            xn3 = ops.layernorm_rows_hl(x2, None, None, eps)                           # xhat = LN_0(x) without its affine, hl image [N, 2D]
            qv = ops.gemm_hl(xn3, fw["wqv_hl"], fw["bqv"])                             # [N, 2D] f32 = [Q | V]
        else:
            xn3 = ops.layernorm_rows_split3(x2, None, None, eps)                      # ... as [hi | hi | lo], [N, 3D]
            qv = ops.gemm_x3(xn3, fw["wqv3"], fw["bqv"])
        q, v = qv[:, :d], qv[:, d:]
        xs, _slot = ops.gather_slot_map(x2, sel)
        # the K-row projections in plain fp32 (10 us each, as the generic chain runs them under autograd): the gradient that reaches the
        # attention through LayerNorm 1 of these rows is a small difference of large terms -- a 1e-5 error on x_sel moved it by 4e-4
        kp = F.linear(xs, wk, bk)
        p_drop = mha.dropout.p if layer.training else 0.0
        drop = (float(p_drop),) + draw_dropout_state() if p_drop > 0.0 else None
        x3 = ops.x3_attn_supported(k, dk) and SF.FP32_ATTENTION != "exact"
        in_kernel = drop is not None and x3 and ops.x3_attn_dropout_supported(k, dk)
        if x3:
            # training: the kernel applies the Philox mask to P in registers (O = (P o M)^T V) and writes the undropped P for the backward
            o, p, _ = ops.sparse_attn_fwd_x3(q, v, kp, h, need_attn=True, dropout=drop if in_kernel else None)
        else:
            o, p, _ = ops.sparse_attn_fwd(q, kp, v, h, need_attn=True)
        mask = None
        # the backward regenerates the mask from the same Philox state where it can (round 6: no [h, n, k] tensor written and read twice)
        regen = in_kernel and not need_attn and ops.attn_bwd_dropout_supported(k, dk)
        if drop is not None and not regen:
            mask = ops.dropout_mask(h, n, k, drop[0], drop[1], drop[2], x2.device)         # the same mask as a tensor: the backward reads it
            if not in_kernel:
                o = torch.bmm((p * mask).transpose(1, 2), v.reshape(n, h, dk).transpose(0, 1)).transpose(0, 1).reshape(k, d)
        delta = F.linear(o, wo, bo)
        x_sel = xs + delta                                                             # snuffy.py:108 at the K rows
        xhat0_sel = xn3.index_select(0, sel)
        # LayerNorm_1 sees y = x with the K rows replaced: the image is re-normalised at those rows, every other row is shared
        if hl:
            ops.layernorm_rows_hl_patch_(xn3, sel, x_sel, None, eps=eps)
            hid3 = ops.gemm_hl(xn3, fw["w1_hl"], fw["b1"], "relu", hl_out=True)        # [N, 2F] image
            z = ops.gemm_hl(hid3, fw["w2_hl"], bb2.detach().float().contiguous(), resid=x2)
        else:
            xn3.index_copy_(0, sel, ops.split3_rows(ops.layernorm_rows(x_sel, None, None, eps)))
            hid3 = ops.gemm_x3(xn3, fw["w1_3"], fw["b1"], "relu", split3=True)         # [N, 3F]
            z = ops.gemm_x3(hid3, fw["w2_3"], bb2.detach().float().contiguous(), resid=x2)  # x + f + b2: the residual rides in the epilogue
        z.index_add_(0, sel, delta)                                                    # snuffy.py:110,154-155
        ctx.save_for_backward(sel, xhat0_sel, qv, kp, p, mask, o, xs, x_sel, hid3, g0, b0, g1, b1, wq, wv, wo, w1)
        ctx.xn3, ctx.fw = xn3, fw       # xn3: written in place after the Q | V projection read it (outside the version check)
        ctx.h, ctx.eps, ctx.hl = h, eps, hl
        ctx.drop = drop if regen else None
        attn = (p * mask if mask is not None else p) if need_attn else None
        if attn is not None:
            ctx.mark_non_differentiable(attn)
        return z, attn

    @staticmethod
    def backward(ctx, dz, _dattn):
        sel, xhat0_sel, qv, kp, p, mask, o, xs, x_sel, hid3, g0, b0, g1, b1, wq, wv, wo, w1 = ctx.saved_tensors
        xn3, fw = ctx.xn3, ctx.fw
        d = xs.shape[1]
        hl = ctx.hl
        f = hid3.shape[1] // (2 if hl else 3)
        h, eps = ctx.h, ctx.eps
        dk = d // h
        dz = dz.float().contiguous()
        # ---- FFN: z = y + relu(xhat1 W1'^T + b1') W2^T + b2                                                  (snuffy.py:224-225)
        if hl:
            dz3, db2 = ops.split_hl_colsum(dz)                                         # operand image + bias gradient, one pass
            dw2 = ops.gemm_tn(dz3, hid3, d, f, hl=True)                                # [D, F]
            if X3_TRAIN_GATED_GEMM and 2 * f <= 8192:
                # the ReLU mask (hi values of the output image) in the GEMM's epilogue, which writes the gated gradient as its hl image: no
                # fp32 dhid, no split pass; the bias gradient is one column-sum pass over the image, the K rows come back out of it
                dhid3 = ops.gemm_hl_gated(dz3, fw["w2t_hl"], hid3)
                del dz3
                db1f = ops.hl_colsum(dhid3)
                dw1f = ops.gemm_tn(dhid3, xn3, f, d, hl=True)                          # [F, D], gradient of the FOLDED weight
                dhid_s, gate_s = _hl_planes_sum(dhid3.index_select(0, sel)), None
                del dhid3
            else:
                dhid = ops.gemm_hl(dz3, fw["w2t_hl"])                                  # [N, F] f32, not yet gated
                del dz3
                dhid3, db1f = ops.split_hl_colsum(dhid, gate_hl=hid3)
                dw1f = ops.gemm_tn(dhid3, xn3, f, d, hl=True)
                del dhid3
                dhid_s = dhid.index_select(0, sel)
                gate_s = hid3.index_select(0, sel).view(sel.numel(), f // 32, 2, 32)[:, :, 0].reshape(sel.numel(), f)
                del dhid
        else:
            dz3, db2 = ops.split3_colsum(dz)
            dw2 = _tn3(dz3, hid3, d, f)
            dhid = ops.gemm_x3(dz3, fw["w2t_3"])
            del dz3
            gate = hid3[:, f:2 * f]                                                    # ReLU mask from the hi plane of the output image
            dhid3, db1f = ops.split3_colsum(dhid, gate=gate)
            dw1f = _tn3(dhid3, xn3, f, d)
            del dhid3
            dhid_s, gate_s = dhid.index_select(0, sel), gate.index_select(0, sel)
            del dhid
        # ---- the K selected rows: y[S] = x_sel = xs + o Wo^T + bo; every other row of y is data               (snuffy.py:108,152-155)
        dyn_s = (dhid_s if gate_s is None else dhid_s * (gate_s > 0)) @ fw["w1f"]     # d loss / d xhat1[S]
        dy_s = ops.layernorm_rows_bwd(x_sel, dyn_s, None, eps, residual=dz.index_select(0, sel), want_param_grads=False)[0]
        dbo = dy_s.sum(0)
        dwo = dy_s.t() @ o
        do = dy_s @ wo
        # ---- attention, exact backward on the materialised P                                                 (snuffy.py:160-168)
        dq, dkp, dv = ops.sparse_attn_bwd(qv[:, :d], kp, qv[:, d:], p, do.contiguous(), h, mask=mask, scale=1.0 / math.sqrt(dk),
                                          dropout=ctx.drop)
        # xn3 holds LayerNorm 1's rows at S since the forward re-normalised them in place; the Q | V projection saw LayerNorm 0's:
        # dW = dqv^T xn3 + dqv[S]^T (xhat0[S] - xhat1[S]), a K-row correction of the big product
        x1_sel = xn3.index_select(0, sel)
        if hl:
            corr = _hl_planes_sum(xhat0_sel) - _hl_planes_sum(x1_sel)
            dqv3 = torch.empty(xn3.shape[0], 4 * d, dtype=torch.bfloat16, device=dq.device)   # ONE image of [dQ | dV]
            _, dbq = ops.split_hl_colsum(dq, out=dqv3, col=0)
            _, dbv = ops.split_hl_colsum(dv, out=dqv3, col=d)
            dwqvf = ops.gemm_tn(dqv3, xn3, 2 * d, d, hl=True)                          # [2D, D] folded
        else:
            corr = (xhat0_sel[:, d:2 * d].float() + xhat0_sel[:, 2 * d:].float()) - (x1_sel[:, d:2 * d].float() + x1_sel[:, 2 * d:].float())
            dqv3 = torch.empty(xn3.shape[0], 6 * d, dtype=torch.bfloat16, device=dq.device)
            _, dbq = ops.split3_colsum(dq, out=dqv3, col=0)
            _, dbv = ops.split3_colsum(dv, out=dqv3, col=d)
            dwqvf = _tn3(dqv3, xn3, 2 * d, d)
        del dqv3
        dwqvf += torch.cat([dq.index_select(0, sel), dv.index_select(0, sel)], dim=1).t() @ corr
        dwqf, dwvf = dwqvf[:d], dwqvf[d:]
        dwk = dkp.t() @ xs
        dbk = dkp.sum(0)
        # ---- unfold  W' = W * gamma,  b' = W beta + b
        if d <= 2048:
            dwq, dgq, dbq0 = ops.unfold_linear(dwqf, wq, g0, b0, dbq)
            dwv, dgv, dbv0 = ops.unfold_linear(dwvf, wv, g0, b0, dbv)
            dg0, db0 = dgq + dgv, dbq0 + dbv0
            dw1, dg1, db1 = ops.unfold_linear(dw1f, w1, g1, b1, db1f)
        else:
            dg0 = (dwqf * wq).sum(0) + (dwvf * wv).sum(0)
            db0 = dbq @ wq + dbv @ wv
            dg1 = (dw1f * w1).sum(0)
            db1 = db1f @ w1
            dwq = dwqf * g0 + torch.outer(dbq, b0)
            dwv = dwvf * g0 + torch.outer(dbv, b0)
            dw1 = dw1f * g1 + torch.outer(db1f, b1)
        return (None, None, None, None, dg0, db0, dg1, db1, dwq, dbq, dwk, dbk, dwv, dbv, dwo, dbo, dw1, db1f, dw2, db2)


def fused_layer0_x3_ok(x2, sel, layer, precision):
    """EncoderLayer0X3Fn applies: fp32 with FP32_GEMM = "x3", the bag is data, ReLU FFN, no encoder dropout (attention dropout is
    handled), shapes in the tile GEMM's domain; everything else keeps the generic autograd chain."""
    if (not FUSED_X3_TRAINING or precision != "fp32" or SF.FP32_GEMM != "x3" or x2.requires_grad or sel.numel() == 0
            or torch.is_autocast_enabled() or x2.dtype != torch.float32 or not x2.is_contiguous()):
        return False
    mha, ff = layer.self_attn, layer.feed_forward
    n0, n1 = layer.sublayer[0].norm, layer.sublayer[1].norm
    n, d = x2.shape
    f = ff.w_1.weight.shape[0]
    if ff.activation_name != "relu" or n0.eps != n1.eps or d % mha.h or d % 8 or f % 8 or n < 256:
        return False
    if f > 8192 or d > 8192:        # the backward's ops.split3_colsum runs over dhid [N, F] and dz [N, D]: k % 8 == 0, k <= 8192
        return False
    if layer.training and (layer.sublayer[0].dropout.p > 0 or layer.sublayer[1].dropout.p > 0 or ff.dropout.p > 0):
        return False
    if any(t is None for t in (n0.weight, n0.bias, n1.weight, n1.bias, ff.w_1.bias, ff.w_2.bias) + tuple(l.bias for l in mha.linears)):
        return False
    return (ops.gemm_x3_supported(n, 2 * d, d) and ops.gemm_x3_supported(n, f, d) and ops.gemm_x3_supported(n, d, f)
            and all(p.requires_grad for p in layer.parameters()))


class CriticFn(torch.autograd.Function):
    """scores = x w^T + b (FCLayer, snuffy.py:39-41) on the one-pass critic kernel; with `layer` (bf16 training) the same pass
    leaves the normalised bf16 copy of the bag for the first encoder layer, as in inference.  Backward: dW = dS^T x (the loss
    reaches the scores through a max over the bag, so dS is one-hot per class, but nothing here relies on that)."""

    @staticmethod
    def forward(ctx, x2, w, b, layer, eps):
        if layer is not None:
            s, xhat = ops.critic_select(x2, w, b, eps) or ops.critic_ln(x2, w, b, eps)
            layer._xhat_offer = (x2.data_ptr(), tuple(x2.shape), x2._version, float(eps), xhat)
        else:
            fused = ops.critic_select(x2, w, b)
            s = fused[0] if fused is not None else ops.critic(x2, w, b)
        ctx.save_for_backward(x2, w)
        ctx.has_bias = b is not None
        return s

    @staticmethod
    def backward(ctx, ds):
        x2, w = ctx.saved_tensors
        ds = ds.float()
        if ds.shape[1] <= 4 and x2.shape[1] % 8 == 0 and x2.shape[1] <= 8192:
            dw = torch.stack([ops.colsum_fused(x2, row_weight=ds[:, c])[0] for c in range(ds.shape[1])])   # one pass per class
        else:
            dw = ds.t() @ x2
        dx = ds @ w if ctx.needs_input_grad[0] else None
        return dx, dw, (ds.sum(0) if ctx.has_bias else None), None, None


def critic_train(feats2, w, b, layer=None, eps=1e-5):
    """Critic scores with autograd; the selection itself uses them detached."""
    return CriticFn.apply(feats2, w, b, layer, eps)


def encoder_layer_train(x2, sel, layer, need_attn, precision):
    """Differentiable EncoderLayer.forward (snuffy.py:126-157).  Returns (functional.Parts, A)."""
    from .functional import Parts
    if fused_layer0_ok(x2, sel, layer, precision):
        n0, n1 = layer.sublayer[0].norm, layer.sublayer[1].norm
        lq, lk, lv, lo = layer.self_attn.linears
        ff = layer.feed_forward
        z, attn = EncoderLayer0Bf16Fn.apply(x2, sel, layer, bool(need_attn), n0.weight, n0.bias, n1.weight, n1.bias, lq.weight,
                                            lq.bias, lk.weight, lk.bias, lv.weight, lv.bias, lo.weight, lo.bias,
                                            ff.w_1.weight, ff.w_1.bias, ff.w_2.weight, ff.w_2.bias)
        return Parts(z), (attn.unsqueeze(0) if attn is not None else None)
    if fused_layer0_x3_ok(x2, sel, layer, precision):
        layer._xhat_offer = None
        n0, n1 = layer.sublayer[0].norm, layer.sublayer[1].norm
        lq, lk, lv, lo = layer.self_attn.linears
        ff = layer.feed_forward
        z, attn = EncoderLayer0X3Fn.apply(x2, sel, layer, bool(need_attn), n0.weight, n0.bias, n1.weight, n1.bias, lq.weight,
                                          lq.bias, lk.weight, lk.bias, lv.weight, lv.bias, lo.weight, lo.bias,
                                          ff.w_1.weight, ff.w_1.bias, ff.w_2.weight, ff.w_2.bias)
        return Parts(z), (attn.unsqueeze(0) if attn is not None else None)
    layer._xhat_offer = None        # chain declined: a normalised copy the critic may have left must not outlive this bag
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
        if (linear_x3_ok(xn, lq.weight) and lq.bias is not None and lv.bias is not None
                and ops.gemm_x3_supported(xn.shape[0], 2 * lq.weight.shape[0], xn.shape[1])      # the fused [2d, d] projection and its dx
                and ops.gemm_x3_supported(xn.shape[0], xn.shape[1], 2 * lq.weight.shape[0])):
            # fp32: split-bf16 x3 on the MFMA GEMM, forward and backward; Q | V as ONE projection (one image of xn, one dx)
            d = xn.shape[1]
            qv = LinearX3Fn.apply(xn, torch.cat([lq.weight, lv.weight]), torch.cat([lq.bias, lv.bias]))
            q, v = qv[:, :d], qv[:, d:]
        else:
            q = F.linear(xn, lq.weight, lq.bias)
            v = F.linear(xn, lv.weight, lv.bias)
        kp = F.linear(xs, lk.weight, lk.bias)
        p_drop = mha.dropout.p if training else 0.0
        o, p = SparseAttnFn.apply(q, kp, v, mha.h, p_drop, torch.is_autocast_enabled(), need_attn)
        delta = F.linear(o, lo.weight, lo.bias)                                 # snuffy.py:205
        if training and drop0.p > 0:
            delta = F.dropout(delta, drop0.p, True)
        x_sel = xs + delta.float()                                              # snuffy.py:108
        y = ScatterRowsFn.apply(x2, sel, x_sel)                                 # snuffy.py:154-155
        attn = p.detach().unsqueeze(0) if (need_attn and p is not None) else None
    yn = LayerNormRowsFn.apply(y, n1.weight, n1.bias, n1.eps)
    hid = _ACT[ff.activation_name](_linear(yn, ff.w_1))                         # snuffy.py:224-225
    if training and ff.dropout.p > 0:
        hid = F.dropout(hid, ff.dropout.p, True)
    f = _linear(hid, ff.w_2)
    if training and drop1.p > 0:
        f = F.dropout(f, drop1.p, True)
    z = y + f.float()                                                           # snuffy.py:110
    return Parts(z), attn


def head_train(z, norm, linear):
    """logits = Linear(mean_n LayerNorm(z)) with autograd (snuffy.py:86,71)."""
    return HeadFn.apply(z, norm.weight, norm.bias, linear.weight, linear.bias, norm.eps)
