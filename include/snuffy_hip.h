/*
 * snuffy_hip.h -- C ABI of libsnuffy_hip.so: hand-written HIP (gfx950 / CDNA4) kernels for the hot path of
 * jafarinia/snuffy: the sparse-attention MIL aggregator (snuffy.py) and, in later rounds, the ViT extractor.
 *
 * The reference has NO native layer: its boundary is the Python nn.Module API.  This header is
 * the FFI a maintainer would bind *underneath* that API (INTEGRATION.md shows the ctypes stub).  Each entry point
 * names the reference code it replaces (file:line, relative to the reference repo).
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes; every buffer (inputs, outputs, workspace) is DEVICE memory owned by the caller;
 *   - row-major, contiguous unless a leading dimension is passed; index tensors are int64 (torch.long);
 *   - asynchronous on `stream` (a hipStream_t passed as void*); never synchronises, never allocates;
 *   - returns 0 on success, a negative SNF_E* code on failure; message via snf_last_error() (thread-local);
 *   - stateless, re-entrant, thread-safe; results are deterministic run-to-run (no float atomics).  The ONE exception is the block
 *     of snf_debug_* switches at the end of this header: process-wide kernel-variant selectors for the parity tests and the A / B
 *     timing tools (atomic ints read once at the entry of a call; every variant computes the same function, so a racing switch
 *     can change which kernel runs, never what a call returns beyond the documented rounding order).  Nothing in snuffy_amd/
 *     touches them; an embedding that wants the contract above without the footnote simply never calls them.
 */
#ifndef SNUFFY_HIP_H
#define SNUFFY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* snf_stream_t; /* hipStream_t */

#define SNF_OK 0
#define SNF_EINVAL (-1)      /* bad argument (null pointer, unsupported shape) */
#define SNF_ELAUNCH (-2)     /* hipLaunch / runtime error */
#define SNF_EUNSUPPORTED (-3) /* shape outside what the fast kernel handles; caller picks the generic entry */
#define SNF_EWORKSPACE (-4)  /* workspace too small */

/* activation codes: PositionwiseFeedForward activation_dictionary, snuffy.py:215-221 */
#define SNF_ACT_RELU 0
#define SNF_ACT_GELU 1       /* erf form (nn.GELU default) */
#define SNF_ACT_LEAKYRELU 2  /* slope 0.01 (nn.LeakyReLU default) */
#define SNF_ACT_SELU 3
#define SNF_ACT_NONE 4

/* element types of q / v^T operands of the MFMA attention kernel */
#define SNF_DT_F32 0
#define SNF_DT_BF16 1
#define SNF_DT_BF16_SPLIT3 2 /* [hi | hi | lo] bf16 image of an fp32 value (snf_gemm_bf16 output, snf_layernorm_rows_split3_f32) */
#define SNF_DT_BF16_HL 3     /* interleaved bf16 image: every 32 columns as [hi(32) | lo(32)], 2 n columns (snf_gemm_hl_bf16) */

const char* snf_version(void);
const char* snf_last_error(void);
/* Number of compute units of the current device (grid sizing on the host side); <0 on error. */
int snf_device_cu_count(void);

/* ---------------------------------------------------------------------------------------------------------
 * K1  critic scores  c = x w^T + b            replaces FCLayer.forward, snuffy.py:39-41 (nn.Linear(D, C))
 *     plus column max over the bag             replaces torch.max(ins_prediction, 1), train.py:831-834
 *   x [n, d] f32, w [c_out, d], b [c_out] (nullable) -> scores [n, c_out]
 *   colmax_val [c_out] f32 / colmax_idx [c_out] i64 nullable (first index on ties, as torch.max on CPU).
 * --------------------------------------------------------------------------------------------------------- */
int snf_critic_f32(const float* x, int64_t n, int d, const float* w, const float* b, int c_out, float* scores,
                   float* colmax_val, int64_t* colmax_idx, snf_stream_t stream);
/* K1 + K5 in one pass over the bag: the critic scores AND xhat [n, d] bf16 = (x - mean) * rstd, the affine-free LayerNorm
 * of SublayerConnection 'attn' (snuffy.py:97,107) whose gamma/beta the bf16 path folds into the Q|V projection weights.
 * Same arithmetic as snf_critic_f32 and snf_layernorm_rows_f32(gamma = beta = NULL, out_bf16): identical outputs. */
int snf_critic_ln_f32(const float* x, int64_t n, int d, const float* w, const float* b, int c_out, float* scores,
                      float eps, void* xhat_bf16, snf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K2  top-k patch selector                    replaces torch.sort(c,1,descending=True)[:k], snuffy.py:128-130
 *   scores: n values at stride `stride` (elements).  idx_out [k] int64: the k largest, in descending score
 *   order, ties by ascending index (== torch.sort(stable=True, descending=True)[:k]); -0.0 == +0.0; NaN sorts
 *   first like torch.  Requires 1 <= k <= n, k <= SNF_TOPK_MAX_K.  One-workgroup radix select on the orderable 32-bit keys (three counting
 *   passes over register-resident keys, LDS integer atomics) + rank sort of the survivors: exact, deterministic.
 *   Above 65536 scores (and with a workspace of snf_topk_workspace_bytes): histogram launch + the multi-workgroup
 *   select of the fused selector below.
 * K4  fused gather  xs[j,:] = x[idx[j],:]      replaces index_select/cat, snuffy.py:131,145-147,103-106
 *   snf_topk_gather_f32 = selector then gather in one call (x, xs nullable -> selector only).
 * --------------------------------------------------------------------------------------------------------- */
#define SNF_TOPK_MAX_K 2048
size_t snf_topk_workspace_bytes(int64_t n, int k); /* bytes the selector needs for this shape (may be 0) */
int snf_topk_f32(const float* scores, int64_t n, int64_t stride, int k, int64_t* idx_out, void* workspace,
                 size_t workspace_bytes, snf_stream_t stream);
int snf_topk_gather_f32(const float* scores, int64_t n, int64_t stride, int k, int64_t* idx_out, const float* x,
                        int d, float* xs, void* workspace, size_t workspace_bytes, snf_stream_t stream);
int snf_gather_rows_f32(const float* x, int64_t n, int d, const int64_t* idx, int k, float* out,
                        snf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K1 + K2 fused selector                         FCLayer.forward + torch.sort(...)[:k], snuffy.py:39-41,128-130
 *   The critic pass already touches every score it writes: snf_critic_select_f32 (one class; xhat_bf16 nullable, as
 *   snf_critic_ln_f32 when given) also counts the first radix digit (top 11 bits of the orderable key) of each score
 *   into a small replicated histogram inside `selector_state` -- integer atomics, exact and order-independent.
 *   snf_topk_select_f32 then runs ceil(n / 4096) workgroups: each reads the histogram, finds the threshold digit,
 *   classifies its slice and appends the scores above / inside the threshold bin to two short lists; the workgroup
 *   that arrives last finishes the radix selection on the lists and writes idx_out (same order and tie rule as
 *   snf_topk_f32, bit for bit).  A crowded threshold bin (> 4096 keys: massive ties) or a histogram whose total is
 *   not n falls back, inside the same launch, to the exact one-workgroup selection on the scores.
 *   selector_state: snf_selector_state_bytes() of device memory, 16-byte aligned, zeroed ONCE by the caller; every
 *   snf_topk_select_f32 leaves it zeroed for the next bag (no memset in the pipeline).  One state per stream.
 * --------------------------------------------------------------------------------------------------------- */
size_t snf_selector_state_bytes(void);
int snf_critic_select_f32(const float* x, int64_t n, int d, const float* w, const float* b, float* scores, float eps,
                          void* xhat_bf16, void* selector_state, snf_stream_t stream);
int snf_topk_select_f32(const float* scores, int64_t n, int k, int64_t* idx_out, void* selector_state,
                        snf_stream_t stream);
/* the same selection for scores that did NOT come out of snf_critic_select_f32 (any stride): clears the counted part of
 * `state` (scratch, snf_selector_state_bytes), counts the first digit in its own launch, then selects.  snf_topk_f32
 * takes this route above 65536 scores. */
int snf_topk_hist_select_f32(const float* scores, int64_t n, int64_t stride, int k, int64_t* idx_out, void* state,
                             snf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K9  scatter of the K updated rows            replaces y = x.clone(); y[:, S, :] = x_sel, snuffy.py:152-155
 *   snf_scatter_rows_f32 : y = x (skipped when y == x) then y[idx[j],:] = rows[j,:]
 *   snf_scatter_add_rows_f32 : z[idx[j],:] += delta[j,:]   (idx distinct)
 *   snf_slot_map_i32 : map[i] = j if idx[j] == i else -1   (row -> slot lookup used by the fused row passes)
 * --------------------------------------------------------------------------------------------------------- */
int snf_scatter_rows_f32(const float* x, int64_t n, int d, const int64_t* idx, int k, const float* rows, float* y,
                         snf_stream_t stream);
int snf_scatter_add_rows_f32(float* z, int64_t n, int d, const int64_t* idx, int k, const float* delta,
                             snf_stream_t stream);
int snf_slot_map_i32(const int64_t* idx, int k, int64_t n, int32_t* map, snf_stream_t stream);
/* K4 + slot map in one launch: xs[j] = x[idx[j]] and map[i] = j if idx[j] == i else -1 (k <= 2048, idx duplicate-free).
 * Same results as snf_gather_rows_f32 + snf_slot_map_i32.  xs_bf16 (nullable): a second copy of xs rounded to bf16
 * [k, d] -- the operand of the bf16 key projection that feeds snf_sparse_attn_fwd_mfma. */
int snf_gather_slot_map_f32(const float* x, int64_t n, int d, const int64_t* idx, int k, float* xs, int32_t* map,
                            void* xs_bf16, snf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K5  LayerNorm over rows, with the K9 scatter fused into the read
 *     replaces SublayerConnection.norm, snuffy.py:97,107,110 (nn.LayerNorm(size), eps 1e-5)
 *   src row i = (slot_map && slot_map[i] >= 0) ? patch_rows[slot_map[i]] : x[i]      (slot_map nullable)
 *   out = (src - mean) * rstd * gamma + beta   (gamma/beta nullable -> plain normalisation, affine folded into
 *   the following GEMM's weights by the host).  Any of out_f32 / out_bf16 / mean / rstd may be null.
 *   out_row_idx (nullable): output row of input row i is out_row_idx[i] (re-normalising the K patched rows in place).
 * --------------------------------------------------------------------------------------------------------- */
int snf_layernorm_rows_f32(const float* x, int64_t n, int d, const int32_t* slot_map, const float* patch_rows,
                           const float* gamma, const float* beta, float eps, float* out_f32, void* out_bf16,
                           float* mean, float* rstd, const int64_t* out_row_idx, snf_stream_t stream);
/* Same rows, written as the split image out_bf16 [n, 3 d] = [hi | hi | lo] (hi = bf16(v), lo = bf16(v - hi)): the A operand
 * of the fp32-class projection  v W^T ~ hi Wh^T + hi Wl^T + lo Wh^T  run as ONE bf16 GEMM over the tripled K axis against
 * W3 = [Wh | Wl | Wh] (snf_gemm_bf16; products to 2^-17 relative, fp32 accumulate). */
/* Backward of the same LayerNorm (loss.backward() of train.py:259 through snuffy.py:86,97,107,110); statistics are recomputed
 * from x, nothing is saved by the forward:
 *   dxhat = dy * gamma;  dx = residual + rstd * (dxhat - mean_d(dxhat) - xhat * mean_d(dxhat * xhat))
 *   dy [n, d] f32 or bf16 (dy_dtype) with row pitch dy_stride; dy_stride = 0 broadcasts ONE row (the mean-pooled head).
 *   gamma, residual, dx, dx_bf16, partials nullable.  partials [snf_layernorm_bwd_blocks(n), 2, d] f32: per-workgroup
 *   (sum dy * xhat, sum dy) in a fixed order -- dgamma / dbeta are their sums over the first axis. */
int snf_layernorm_bwd_blocks(int64_t n);
int snf_layernorm_rows_bwd_f32(const float* x, int64_t n, int d, const void* dy, int dy_dtype, int64_t dy_stride,
                               const float* gamma, float eps, const float* residual, float* dx, void* dx_bf16,
                               float* partials, snf_stream_t stream);
/* bf16 path: the LayerNorm affine of SublayerConnection (snuffy.py:97,107,110) folded into the projection that follows it
 * (LN(x) W^T + b = xhat (W * gamma)^T + (W beta + b)), and the gradients of the folded weights taken back to (W, b, gamma, beta):
 *   fold:    wf_bf16[r, c] = bf16(w[r, c] * gamma[c]) (row pitch ldwf);  bf[r] = sum_c w[r, c] * beta[c] + bias[r] (f32 / bf16, nullable)
 *   unfold:  dw[r, c] = dwf[r, c] * gamma[c] + dbf[r] * beta[c];  partial [snf_fold_blocks(r), 2, c] = per-workgroup
 *            (sum_r dwf * w, sum_r dbf[r] * w): dgamma / dbeta are their sums over the first axis.   c <= 2048. */
int snf_fold_blocks(int r);
int snf_fold_linear_f32(const float* w, int r, int c, const float* gamma, const float* beta, const float* bias, void* wf_bf16,
                        int64_t ldwf, float* bf_f32, void* bf_bf16, snf_stream_t stream);
int snf_unfold_linear_f32(const float* dwf, const float* w, int r, int c, const float* gamma, const float* beta, const float* dbf,
                          float* dw, float* partial, snf_stream_t stream);
/* x [m, k] f32 (row pitch ldx) -> out [m, 3 k] bf16 = [hi | hi | lo]: the activation image of an fp32-class projection whose
 * producer is not one of the kernels that can emit it directly (ViT fp32 path: patch columns, attention output). */
int snf_split3_f32(const float* x, int64_t ldx, int64_t m, int k, void* out_bf16, snf_stream_t stream);
/* w [m, k] f32 (row pitch ldw; the nn.Linear layout) -> out [m, 3 k] bf16 = [Wh | Wl | Wh], the weight operand of the same products;
 * colscale [k] (nullable): W' = W diag(colscale) formed in fp32 first (a LayerNorm gamma folded into the projection that follows it). */
/* The loss head of a training step in one launch each way (reference train.py: _run_model of the single-weight trainer -- max over the instance
 * scores, two BCEWithLogits terms (optional pos_weight) mixed by the weight w, the bag prediction):
 *   ins [n, c] f32 contiguous, logits [c], label [c], w [1], pos_weight [c] and weight [c] of BCEWithLogitsLoss (nullable; the reference passes
 *   its class weights POSITIONALLY, i.e. as `weight`: train.py:246), c <= 8
 *   out [2 + 4 c] = {loss, d loss / d w, bag_pred [c], d loss / d logits [c], d loss / d max [c], max [c]};  argmax [c] (first index on ties)
 *   _bwd: d_ins [n, c] = grad_out * (d loss / d max) at the argmax rows, 0 elsewhere; d_small [c + 1] = grad_out * {d loss / d logits, d loss / d w}.
 * A NaN instance score never wins the max (torch.max would return it): documented in DESIGN.md section 7. */
int snf_mil_loss_f32(const float* ins, int64_t n, int c, const float* logits, const float* label, const float* w, const float* pos_weight,
                     const float* weight, float* out, int64_t* argmax, snf_stream_t stream);
int snf_mil_loss_bwd_f32(const float* grad_out, const float* fwd_out, const int64_t* argmax, int64_t n, int c, float* d_ins, float* d_small,
                         snf_stream_t stream);
int snf_split3_weight_f32(const float* w, int64_t ldw, int64_t m, int k, const float* colscale, void* out_bf16, snf_stream_t stream);
/* Column sums of a [n, d] matrix fused with the elementwise step of the same pass (training: bias gradients next to the ReLU
 * mask / the bf16 cast / the critic's weight gradient; backward of snuffy.py:39-41, 224-225):
 *   v = src[i, c] (f32 or bf16) * row_weight[i * weight_stride] (nullable) ; v = 0 where gate_bf16[i, c] <= 0 (nullable: the ReLU
 *   mask from the activation output) ; dst_bf16[i, c] = bf16(v) (nullable, may alias src) ; partial[b, c] = sum over the rows
 *   of workgroup b (of the rounded values when dst is given).  partial [snf_colsum_blocks(n), d] f32, summed by the caller.
 *   d % 8 == 0, d <= 8192, 16-byte aligned buffers. */
int snf_colsum_blocks(int64_t n);
int snf_colsum_fused(const void* src, int src_dtype, int64_t n, int d, const float* row_weight, int64_t weight_stride,
                     const void* gate_bf16, void* dst_bf16, float* partial, snf_stream_t stream);
/* The backward of an fp32-class training step (train.py:259 through snuffy.py:187-190, 224-225) needs each gradient matrix as a split
 * image (operand of the next GEMM and of the weight-gradient contractions) AND its column sums (the bias gradient), the FFN one behind
 * the ReLU mask: one pass.   v = x[i, c] (f32, row pitch ldx) ; v = 0 where gate_bf16[i, c] <= 0 (nullable; row pitch ldg: the hi
 * plane of the activation's own split image) ; out[i, c] = out[i, plane + c] = hi(v), out[i, 2 plane + c] = lo(v) (bf16, row pitch
 * ldo: with plane = k and ldo = 3 k the image of snf_split3_f32; a wider plane lets several matrices share one image, e.g.
 * [dQ | dV]) ; partial[b, c] (nullable) = sum of v over the rows of workgroup b, [snf_colsum_blocks(m), k] f32, summed by the caller.
 * k % 8 == 0, k <= 8192, 16-byte aligned rows. */
int snf_split3_colsum_f32(const float* x, int64_t ldx, int64_t m, int k, const void* gate_bf16, int64_t ldg, void* out_bf16,
                          int64_t ldo, int64_t plane, float* partial, snf_stream_t stream);
/* The same pass writing the INTERLEAVED image ([hi(32) | lo(32)] per 32 columns, 2 k columns: the operand format of snf_gemm_hl_bf16 and of
 * snf_gemm_tn_f32's hl layout); gate_hl (nullable): the activation's own hl image, its hi values decide.  k % 32 == 0. */
int snf_split_hl_colsum_f32(const float* x, int64_t ldx, int64_t m, int k, const void* gate_hl, int64_t ldg, void* out_hl, int64_t ldo,
                            float* partial, snf_stream_t stream);
/* Skinny fp32-class projection for the K selected rows of a bag (key / output projections, snuffy.py:190, 205; the tile GEMMs
 * need thousands of rows): out [r, c] (f32 or bf16: out_dtype) = x [r, k] f32 . w [c, k]^T f32 + bias, every product as split-bf16 x3
 * on the matrix cores with the split done in registers.  r <= 8192, k % 16 == 0, rows 16-byte aligned. */
int snf_linear_rows_x3_f32(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int r, int c, int k, void* out,
                           int64_t ldo, int out_dtype, snf_stream_t stream);
/* The same with a second output out2 [r, c] = out + resid [r, c]: delta = o Wo^T + bo and x_sel = xs + delta (snuffy.py:205, 108) in
 * one launch (round 6) */
int snf_linear_rows_x3_resid_f32(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* resid, int64_t ldr,
                                 int r, int c, int k, float* out, int64_t ldo, float* out2, int64_t ldo2, snf_stream_t stream);
/* critic scores + LayerNorm (with affine) of the same rows in one pass, the normalised rows as the interleaved hi / lo image
 * (= snf_critic_f32 + snf_layernorm_rows_hl_f32 with one read of x; FCLayer.forward snuffy.py:39-41 + SublayerConnection.norm
 * snuffy.py:107).  d % 32 == 0; selector_state nullable (one class: also counts the selector's first radix digit); gamma and beta
 * both NULL = the affine-free image (x - mean) * rstd (one normalised image for both sublayers, affines folded into the projections). */
int snf_critic_ln_hl_f32(const float* x, int64_t n, int d, const float* w, const float* b, int c_out, float* scores,
                         const float* gamma, const float* beta, float eps, void* out_hl, void* selector_state,
                         snf_stream_t stream);
int snf_layernorm_rows_split3_f32(const float* x, int64_t n, int d, const int32_t* slot_map, const float* patch_rows,
                                  const float* gamma, const float* beta, float eps, void* out_bf16, snf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K10 epilogue  h = act(h + bias) in place     replaces activation(w_1(x)) of snuffy.py:224-225
 *   h [n, f] f32 or bf16 (dtype = SNF_DT_*), bias [f] f32 nullable.
 * --------------------------------------------------------------------------------------------------------- */
int snf_bias_act(void* h, int dtype, int64_t n, int f, const float* bias, int act, snf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K11 head: logits = W_head * mean_n(LayerNorm_f(z)) + b_head
 *     replaces Encoder.norm + x.mean(dim=1) + BClassifier.linear, snuffy.py:86,71
 *   row value = z[i] (+ add_bf16[i]) (+ add_bias) (+ delta_rows[slot_map[i]] when slot_map[i] >= 0): the FFN
 *   residual  z = y + W2(...) + b2  of snuffy.py:110 and the K9 scatter are fused into this read; z_out (nullable)
 *   receives the assembled rows (needed only when another layer follows).
 *   workspace: snf_ln_mean_head_workspace_bytes(d) bytes.  pooled [d] nullable output (= mean_n LN(z) before the
 *   head).  Deterministic two-stage column reduction.
 * --------------------------------------------------------------------------------------------------------- */
size_t snf_ln_mean_head_workspace_bytes(int d);
int snf_ln_mean_head_f32(const float* z, int64_t n, int d, const void* add_bf16, const float* add_bias,
                         const int32_t* slot_map, const float* delta_rows, float* z_out, const float* gamma,
                         const float* beta, float eps, const float* w_head, const float* b_head, int c_out,
                         float* logits, float* pooled, void* workspace, size_t workspace_bytes, snf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K7  sparse attention forward                 replaces attention(), snuffy.py:160-168 (+ head split/merge of
 *                                              MultiHeadedAttention.forward, snuffy.py:187-203)
 *   For every head a (column block a*dk..(a+1)*dk):
 *       P_a = softmax_j( Q_a Kp_a^T * scale )   [n, k]   softmax over the k selected keys
 *       O_a = P_a^T V_a                         [k, dk]
 *   out [k, d] = heads concatenated.  attn [h, n, k] f32 (nullable) = P;  lse [h, n] f32 (nullable) =
 *   log-sum-exp of the scaled scores (saved for backward).
 *
 *   snf_sparse_attn_fwd_f32 : exact fp32 arithmetic, any n/k/h/dk.  q, v [n, d]; kp [k, d].
 *   snf_sparse_attn_fwd_mfma: bf16 MFMA (fp32 accumulate), softmax in fp32.  q [n, ldq] and v [n, ldv] row-major
 *       (ldq, ldv >= d in elements, rows 16-byte aligned: q and v may be the two column halves of ONE fused
 *       projection output [n, 2d]), both of dtype qv_dtype (f32 converted in registers, or bf16); kp [k, d] of
 *       dtype kp_dtype: bf16 is read as it is, f32 is rounded to bf16 into the workspace first (one small launch).
 *       One launch holds 256 (dk == 64) / 224 (dk == 128) keys (Kp + P + V images share the 160 KiB LDS); more keys --
 *       up to 8 such chunks, k <= 2048 / 1792 -- run as key chunks: a statistics launch per chunk (row max / sum) and a
 *       full launch per chunk normalising with the statistics of all chunks, so the softmax stays exact.
 *       Other dk / larger k: SNF_EUNSUPPORTED, the caller picks snf_sparse_attn_fwd_f32.
 *   workspace: deterministic cross-workgroup reduction of the [h, k, dk] accumulators.
 * --------------------------------------------------------------------------------------------------------- */
size_t snf_sparse_attn_fwd_workspace_bytes(int64_t n, int k, int h, int dk, int mfma);
int snf_sparse_attn_fwd_f32(const float* q, const float* kp, const float* v, int64_t n, int k, int h, int dk,
                            float scale, float* out, float* attn, float* lse, void* workspace,
                            size_t workspace_bytes, snf_stream_t stream);
/* The same attention with the scores in the fp32-CLASS arithmetic of snf_sparse_attn_fwd_x3 (every product as three bf16 MFMAs of the
 * operands' hi / lo halves, fp32 accumulate / softmax) for head widths the pipelined kernels do not take: dk % 16 == 0, dk <= 256 (the
 * README recipe D = 768 / h = 4 of reference README.md:661-669: dk = 192), k <= 1024; other shapes SNF_EUNSUPPORTED.  P^T V is exact
 * fp32 on the f32 matrix cores.  Two launches + the fixed-order reduction like snf_sparse_attn_fwd_f32 (same workspace:
 * snf_sparse_attn_fwd_workspace_bytes(.., 0)), operands split on the fly; P is materialised (`attn`, or in the workspace).  q / v take
 * row pitches ldq / ldv (elements, >= h dk, ldq % 4 == 0): the halves of a fused [Q | V] projection output are used in place. */
int snf_sparse_attn_fwd_x3u_f32(const float* q, int64_t ldq, const float* kp, const float* v, int64_t ldv, int64_t n, int k, int h, int dk,
                                float scale, float* out, float* attn, float* lse, void* workspace, size_t workspace_bytes,
                                snf_stream_t stream);
int snf_sparse_attn_fwd_mfma(const void* q, int64_t ldq, const void* v, int64_t ldv, int qv_dtype, const void* kp,
                             int kp_dtype, int64_t n, int k, int h, int dk, float scale, float* out, float* attn,
                             float* lse, void* workspace, size_t workspace_bytes, snf_stream_t stream);

/* Attention dropout (training; nn.Dropout(p = 0.1) on p_attn, snuffy.py:166-167,173).  The keep-mask is a pure function of
 * (seed, offset, head, row, key) -- Philox4x32-10, one call per 4 consecutive keys of a row (csrc/philox.h; host restatement
 * oracle/philox_ref.py) -- regenerated in registers by the forward and the backward kernel, never stored:
 *   forward : the normalised probabilities are multiplied by the mask (0 or 1 / (1 - p)) before they are pooled; `out` and the
 *             returned `attn` are those of the dropped P, as in the reference.  Needs lse (or attn) requested.
 *   backward: pass mask == NULL and the forward's (dropout_p, seed, offset).
 * snf_dropout_mask_f32 writes the same mask as a tensor [h, n, k] (exact-fp32 training path, tests). */
int snf_sparse_attn_fwd_mfma_dropout(const void* q, int64_t ldq, const void* v, int64_t ldv, int qv_dtype, const void* kp,
                                     int kp_dtype, int64_t n, int k, int h, int dk, float scale, float* out, float* attn,
                                     float* lse, float dropout_p, uint64_t seed, uint64_t offset, void* workspace,
                                     size_t workspace_bytes, snf_stream_t stream);
int snf_sparse_attn_bwd_mfma_dropout(const void* q, int64_t ldq, const void* v, int64_t ldv, int qv_dtype, const float* kp,
                                     const float* dout, const float* lse, const float* mask, float dropout_p, uint64_t seed,
                                     uint64_t offset, int64_t n, int k, int h, int dk, float scale, float* dq, float* dv,
                                     void* ds, int ds_dtype, snf_stream_t stream);
int snf_dropout_mask_f32(float dropout_p, uint64_t seed, uint64_t offset, int h, int64_t n, int k, float* mask,
                         snf_stream_t stream);

/* K7, fp32-class on the matrix cores: q, v, kp fp32; every operand split into hi + lo bf16 halves and every product taken as
 * three bf16 MFMAs (ah bh + ah bl + al bh, fp32 accumulate); softmax / normalisation fp32.  Same outputs as
 * snf_sparse_attn_fwd_f32 to ~3e-6 on P, 1e-5 on O (the reference's own arithmetic class, north-star bound 1e-3), ~10x its speed.
 * q [n, ldq], v [n, ldv] row-major (row pitches in elements, 16-byte aligned rows: column halves of a fused projection are
 * taken in place); dk in {64, 128}; 256 / 224 keys fit one launch, more keys (up to 8 x 256 / 8 x 224) run as key chunks: a
 * statistics pass per chunk (row max / sum), then the chunks' main passes with the softmax exact over all keys -- other shapes
 * SNF_EUNSUPPORTED (the caller keeps the exact kernel).
 * workspace: snf_sparse_attn_fwd_x3_workspace_bytes (deterministic cross-workgroup reduction, as the bf16 kernel). */
size_t snf_sparse_attn_fwd_x3_workspace_bytes(int64_t n, int k, int h, int dk);
int snf_sparse_attn_fwd_x3(const float* q, int64_t ldq, const float* v, int64_t ldv, const float* kp, int64_t n, int k, int h,
                           int dk, float scale, float* out, float* attn, float* lse, void* workspace, size_t workspace_bytes,
                           snf_stream_t stream);
/* Training forward (snuffy.py:166-167: nn.Dropout on p_attn): the same launch with the Philox keep-mask of snf_dropout_mask_f32 -- the
 * same (seed, offset) give the same mask, bit for bit -- applied to P in registers: out = (P o M)^T V, while `attn` receives the
 * UNDROPPED probabilities P (what snf_sparse_attn_bwd_f32 wants next to the mask tensor).  One key chunk only (k <= 224 at dk = 128,
 * <= 256 at dk = 64; more keys: SNF_EUNSUPPORTED, the caller multiplies and contracts itself); dropout_p == 0 is snf_sparse_attn_fwd_x3. */
int snf_sparse_attn_fwd_x3_dropout(const float* q, int64_t ldq, const float* v, int64_t ldv, const float* kp, int64_t n, int k, int h,
                                   int dk, float scale, float dropout_p, uint64_t seed, uint64_t offset, float* out, float* attn,
                                   float* lse, void* workspace, size_t workspace_bytes, snf_stream_t stream);

/* K7 (fp32-class, pre-split operands)    the same attention() of snuffy.py:160-168, same arithmetic (split-bf16 x3, fp32
 * softmax / accumulate), for operands that already ARE split: q_hl, v_hl = interleaved "hl" images [n, ld] bf16 (every 32 true
 * columns c .. c + 31 of the fp32 tensor stored as [hi(32) | lo(32)], hi = bf16(x), lo = bf16(x - hi): what snf_gemm_hl_bf16
 * writes with out_dtype SNF_DT_BF16_HL and snf_split_hl_f32 makes of an fp32 tensor; 4 bytes per element like the fp32 tensor);
 * row pitches in bf16 elements (>= 2 h dk, 16-byte aligned rows: the halves of a fused [Q | V] image are taken in place);
 * kp [k, h dk] f32.  Software-pipelined kernel: LDS-DMA of full lines, one wave per key block, GEMM1 / GEMM2 MFMAs interleaved
 * with the softmax vector work in every wave's own instruction stream (csrc/sparse_attn_x3p.hip).  dk = 128, 97 <= k <= 2048: up to 256 keys
 * (4 .. 8 key blocks) in one launch, more as up to 8 key chunks -- a statistics pass per chunk, then the chunks' main passes; other shapes SNF_EUNSUPPORTED (the caller keeps snf_sparse_attn_fwd_x3).  Outputs as snf_sparse_attn_fwd_x3. */
size_t snf_sparse_attn_fwd_x3_hl_workspace_bytes(int64_t n, int k, int h, int dk);
int snf_sparse_attn_fwd_x3_hl(const void* q_hl, int64_t ldq, const void* v_hl, int64_t ldv, const float* kp, int64_t n, int k, int h,
                              int dk, float scale, float* out, float* attn, float* lse, void* workspace, size_t workspace_bytes,
                              snf_stream_t stream);
/* The same with the key projection fused in front (snuffy.py:190 + 160-168): snf_linear_rows_x3_kpfrag_f32 computes
 * Kp = x w^T + bias for the k selected rows (fp32-class, as snf_linear_rows_x3_f32) and writes it, scaled by scale * log2(e) and
 * split into bf16 hi / lo, as the MFMA fragment image the attention kernel keeps in registers -- no fp32 Kp tensor, no prep
 * launch; snf_sparse_attn_fwd_x3_hl_kpfrag then takes that image instead of kp (the scale is already in it).  Results are
 * bit-identical to the two-step form.  snf_sparse_attn_x3_hl_kpfrag_bytes: size of the image, 0 where the fused form does not
 * apply (dk != 128, k outside 97 .. 2048, or key chunks that do not start on 32-key boundaries: use the two-step form). */
size_t snf_sparse_attn_x3_hl_kpfrag_bytes(int k, int h, int dk);
int snf_linear_rows_x3_kpfrag_f32(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int k_keys, int h, int dk,
                                  int kdim, float scale, void* kp_frag, size_t kp_frag_bytes, snf_stream_t stream);
/* The same with the gather of the selected rows fused in (snuffy.py:131,145-147 + 190 in ONE launch, round 6): input row i of the
 * projection is row idx[i] of the bag x [n, ldx].  xs (nullable) [k_keys, kdim] receives the gathered rows, slot_map (nullable) [n] the
 * row -> slot map of snf_gather_slot_map_f32 -- the two launches this call replaces write the same bytes. */
int snf_gather_linear_rows_x3_kpfrag_f32(const float* x, int64_t ldx, int64_t n, const int64_t* idx, const float* w, int64_t ldw,
                                         const float* bias, int k_keys, int h, int dk, int kdim, float scale, void* kp_frag,
                                         size_t kp_frag_bytes, float* xs, int32_t* slot_map, snf_stream_t stream);
int snf_sparse_attn_fwd_x3_hl_kpfrag(const void* q_hl, int64_t ldq, const void* v_hl, int64_t ldv, const void* kp_frag, int64_t n, int k,
                                     int h, int dk, float* out, float* attn, float* lse, void* workspace, size_t workspace_bytes,
                                     snf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K7-bwd  sparse attention backward, exact fp32    replaces autograd through attention(), snuffy.py:160-168
 *   p [h, n, k] = the probabilities the forward returned (attn);  mask [h, n, k] nullable = dropout keep-mask already
 *   divided by (1 - p_drop) (the forward used p o mask);  dout [k, d] = gradient of the attention output.
 *       dV = (P o M) dO        dS = P o (dP - rowsum(dP o P)) * scale,  dP = (V dO^T) o M
 *       dQ = dS Kp             dKp = dS^T Q   (deterministic slice reduction, as the forward's P^T V)
 *   dq, dv [n, d], dkp [k, d] f32.  workspace: snf_sparse_attn_bwd_workspace_bytes (holds dS [h, n, k]).
 * --------------------------------------------------------------------------------------------------------- */
size_t snf_sparse_attn_bwd_workspace_bytes(int64_t n, int k, int h, int dk);
int snf_sparse_attn_bwd_f32(const float* q, const float* kp, const float* v, const float* p, const float* mask,
                            const float* dout, int64_t n, int k, int h, int dk, float scale, float* dq, float* dkp,
                            float* dv, void* workspace, size_t workspace_bytes, snf_stream_t stream);
/* The same with row pitches for q / v (elements, >= h dk; round 5): the halves of a fused [Q | V] projection output are read in place.
 * Pitches other than h dk need the matrix-core kernels (dk % 8 == 0, k <= 1024, 16-byte aligned rows), else SNF_EUNSUPPORTED. */
int snf_sparse_attn_bwd_ld_f32(const float* q, int64_t ldq, const float* kp, const float* v, int64_t ldv, const float* p, const float* mask,
                               const float* dout, int64_t n, int k, int h, int dk, float scale, float* dq, float* dkp, float* dv,
                               void* workspace, size_t workspace_bytes, snf_stream_t stream);
/* ... with the forward's dropout mask REGENERATED in the kernels from its Philox state (dropout_p, seed, offset: the arguments
 * snf_sparse_attn_fwd_x3_dropout was given) instead of read from a [h, n, k] tensor (round 6: the tensor was 157 MB written once and
 * read twice per config-B training step).  Only on the matrix-core route (k <= 1024, k % 4 == 0, dk % 8 == 0, 16-byte aligned operands);
 * SNF_EUNSUPPORTED otherwise -- hand the mask over as a tensor then. */
int snf_sparse_attn_bwd_dropout_f32(const float* q, int64_t ldq, const float* kp, const float* v, int64_t ldv, const float* p, float dropout_p,
                                    uint64_t seed, uint64_t offset, const float* dout, int64_t n, int k, int h, int dk, float scale, float* dq,
                                    float* dkp, float* dv, void* workspace, size_t workspace_bytes, snf_stream_t stream);
/* Fast form of the backward for dk == 128 with k <= 224 or dk == 64 with k <= 256 (bf16 MFMA operands, fp32 accumulate): P is recomputed from q, kp
 * and the forward's lse [h, n] (never read back), dQ / dV rows are owned by one wave (no reduction), dS [h, n, k] is
 * written out (ds_dtype f32, or bf16) -- the caller contracts it with q for dKp (snf_sparse_attn_dkp_f32 on f32, or a
 * bf16 library batched GEMM).  q, v as in
 * snf_sparse_attn_fwd_mfma (row-strided views allowed); mask as in snf_sparse_attn_bwd_f32.  Other shapes:
 * SNF_EUNSUPPORTED, the caller uses snf_sparse_attn_bwd_f32. */
int snf_sparse_attn_bwd_mfma(const void* q, int64_t ldq, const void* v, int64_t ldv, int qv_dtype, const float* kp,
                             const float* dout, const float* lse, const float* mask, int64_t n, int k, int h, int dk,
                             float scale, float* dq, float* dv, void* ds, int ds_dtype, snf_stream_t stream);
/* Same kernel with the gradients written where the caller wants them: dq / dv f32 or bf16 (dqv_dtype) with row pitch ldd
 * (elements) -- e.g. the two column halves of ONE [n, 2 d] bf16 buffer, the operand of the fused Q|V weight-gradient GEMM. */
int snf_sparse_attn_bwd_mfma_ex(const void* q, int64_t ldq, const void* v, int64_t ldv, int qv_dtype, const float* kp,
                                const float* dout, const float* lse, const float* mask, float dropout_p, uint64_t seed,
                                uint64_t offset, int64_t n, int k, int h, int dk, float scale, void* dq, void* dv, int64_t ldd,
                                int dqv_dtype, void* ds, int ds_dtype, snf_stream_t stream);
/* dKp [k, d] = dS^T Q per head (deterministic slice reduction); ds [h, n, k], q [n, d] f32.
 * workspace: snf_sparse_attn_bwd_workspace_bytes. */
int snf_sparse_attn_dkp_f32(const float* ds, const float* q, int64_t n, int k, int h, int dk, float* dkp, void* workspace,
                            size_t workspace_bytes, snf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K6 / K10 / K13  dense projection on the matrix cores      replaces nn.Linear (+ activation):
 *     snuffy.py:187-190 (Q | V and key projections), snuffy.py:224-225 (w_1 + activation, w_2),
 *     utils_ssls_cf/vision_transformer_with_adapter_dino_version.py:51-67 (Mlp), :82-94 (qkv / proj), :141-146 (patch embed)
 *   C[m, n] = act(A[m, k] W[n, k]^T + bias[n])   A, W bf16 row-major with row pitches lda / ldw (elements), bias f32 [n]
 *   (nullable), act = SNF_ACT_* (GELU = erf form, as nn.GELU), C bf16 or f32 (out_dtype = SNF_DT_*), row pitch ldc.
 *   Hand-written v_mfma_f32_16x16x32_bf16 kernel, fp32 accumulate; 256 x 256 or 256 x 128 output tiles (tile_n = 256 / 128,
 *   anything else = chosen from the shape).  Domain: k % 32 == 0, k >= 64, n % 8 == 0, rows 16-byte aligned, m * lda and
 *   n * ldw < 2^31; outside it SNF_EUNSUPPORTED (the caller keeps its library GEMM).
 *   out_dtype = SNF_DT_BF16_SPLIT3: C is the bf16 image [hi | hi | lo] of the fp32 result, [m, 3 n] (ldc >= 3 n) -- the
 *   hidden activations of the fp32-class FFN go from the first GEMM to the second without an fp32 round trip.
 *   out_dtype = SNF_DT_BF16_HL: C is the interleaved hl image of the fp32 result, [m, 2 n] (n % 32 == 0, ldc >= 2 n): the Q | V
 *   projection of bags too small for the one-pass kernel hands its result to snf_sparse_attn_fwd_x3_hl in this form.
 * --------------------------------------------------------------------------------------------------------- */
int snf_gemm_bf16(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, int64_t m, int n, int k,
                  int act, void* c, int64_t ldc, int out_dtype, int tile_n, snf_stream_t stream);
/* fp32 output with a residual in the epilogue: c = act(a w^T + bias) + resid [m, ldr] -- z = x + W2 act(W1 LN(y)) of snuffy.py:110 for
 * bags too small for the one-pass kernel (round 6; snf_gemm_hl_resid_bf16 is the large-bag twin) */
int snf_gemm_bf16_resid_f32(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, const float* resid, int64_t ldr,
                            int64_t m, int n, int k, int act, float* c, int64_t ldc, int tile_n, snf_stream_t stream);
/* The ViT block without LayerNorm / residual passes (round 6; vd:97-127 "x = x + attn(norm1(x)); x = x + mlp(norm2(x)) + adapter(x)"):
 *   snf_gemm_bf16_lnfold   the consumer's LayerNorm folded into its GEMM: a = the RAW rows x rounded to bf16, w = W0 diag(gamma) (bf16),
 *                          c [m, ldc] bf16 = act(rstd[m] (a w^T - mean[m] colsum[n]) + bias[n]);  colsum[n] = sum_k w[n, k] of the ROUNDED w,
 *                          bias = W0 beta + b0, rowstats [m][2] = (mean, rstd) of the fp32 rows (snf_vit_row_stats).  act: none, gelu.
 *   snf_gemm_bf16_resid    the residual stream updated by the producer: x [m, ldx] fp32 IN PLACE  x += a w^T + bias; the bf16 copy of
 *                          the new x goes to x_bf16 [m, ldxb]; stats_part [m][n / 32][2] receives (sum, sum of squares) of the new
 *                          row over every 32-column group -- summed in group order by snf_vit_row_stats (bit-reproducible).
 *                          (256-wide tiles; 128-wide ones where the last 256-wide column tile would be at most half used: n = 384.)
 *   domain of both: k % 32 == 0, k >= 96, n % 64 == 0, 16-byte aligned rows. */
int snf_gemm_bf16_lnfold(const void* a, int64_t lda, const void* w, int64_t ldw, const float* colsum, const float* bias,
                         const float* rowstats, int64_t m, int n, int k, int act, void* c, int64_t ldc, snf_stream_t stream);
int snf_gemm_bf16_resid(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, int64_t m, int n, int k, float* x,
                        int64_t ldx, void* x_bf16, int64_t ldxb, float* stats_part, snf_stream_t stream);
/* The same product in fp32-class arithmetic, ONE pass over the operands.  Both operands are INTERLEAVED split images ("hl"): a
 * row of 2 k bf16 in which every 32 true columns are stored as [hi(32) | lo(32)] (hi = bf16(v), lo = bf16(v - hi)) -- one
 * 128-byte line per K step and the same 4 bytes per element as the fp32 tensor.  C = act(A W^T + bias) with every product taken
 * as hi hi + hi lo + lo hi, fp32 accumulate (~1e-5 per product).  A K step stages ONE line per operand row by full-line LDS-DMA
 * and issues 96 MFMAs per wave out of 24 fragment reads.  256 x 256 tiles; k % 32 == 0; out_dtype SNF_DT_F32 / SNF_DT_BF16 /
 * SNF_DT_BF16_HL (the hl image of the result, [m, 2 n], n % 32 == 0: the A operand of the next such GEMM).
 *   snf_split_hl_f32            x [m, k] f32 (row pitch ldx) -> its hl image [m, 2 k]
 *   snf_layernorm_rows_hl_f32   snf_layernorm_rows_f32 whose output is written as the hl image (d % 32 == 0)
 * nn.Linear of snuffy.py:187-190,224-225 and of the ViT blocks in the reference's fp32 arithmetic. */
int snf_gemm_hl_bf16(const void* a_hl, int64_t lda, const void* w_hl, int64_t ldw, const float* bias, int64_t m, int n, int k,
                     int act, void* c, int64_t ldc, int out_dtype, snf_stream_t stream);
/* fp32 output with a residual: C = act(A W^T + bias) + resid [m, ldr] -- z = x + W2 act(W1 LN(y)) of snuffy.py:110 in one pass */
int snf_gemm_hl_resid_bf16(const void* a_hl, int64_t lda, const void* w_hl, int64_t ldw, const float* bias, const float* resid,
                           int64_t ldr, int64_t m, int n, int k, int act, void* c, int64_t ldc, int out_dtype, snf_stream_t stream);
/* The same with split-K of the LAST, partly filled round of 256 x 256 tiles (e.g. the FFN output projection of config B: 384 tiles on
 * 256 CUs): each tile of that round goes to 2..4 workgroups, a K range each; the last one to finish adds the others' fp32 partial tiles
 * in a fixed order and runs the epilogue (bit-reproducible; any activation / output type / residual).  workspace: at least
 * snf_gemm_hl_ws_bytes(m, n, k) bytes (0 = this shape does not split; workspace may then be NULL); plain scratch memory -- the call
 * zeroes its own tickets, nothing is expected in the buffer before a call or kept in it after one, so concurrent calls on different
 * streams only need different buffers.  NULL workspace = snf_gemm_hl_resid_bf16. */
size_t snf_gemm_hl_ws_bytes(int64_t m, int n, int k);
/* The FFN input gradient of the training step behind its ReLU (backward of snuffy.py:224-225: dhid = (dz W2) o [hid > 0]) in ONE pass:
 * c_hl [m, 2 n] = the hl image of (a w^T) with every element zeroed whose gate value -- the hi half of gate_hl [m, 2 n], the activation's own
 * hl image -- is not > 0.  Domain of snf_gemm_hl_bf16 with an hl output (n % 32 == 0). */
int snf_gemm_hl_gated_bf16(const void* a_hl, int64_t lda, const void* w_hl, int64_t ldw, const void* gate_hl, int64_t ldg, int64_t m, int n,
                           int k, void* c_hl, int64_t ldc, snf_stream_t stream);
/* The weight-gradient contraction of the training step (round 6; the autograd of nn.Linear under train.py:259,468-473: dW = dY^T X):
 *   c [p, ldc] f32 = A^T B over the n rows of two row-major bf16 images, on the matrix cores straight from those images (transposing
 *   LDS reads; no transposed copy).  A's operand is the p columns starting at column a_hi of a [n, lda], B's the q columns at b_hi of
 *   b [n, ldb].  a_lo, b_lo >= 0: the operands are split images with a lo plane at those column offsets (the [hi | hi | lo] images of
 *   snf_split3_colsum_f32 / snf_layernorm_rows_split3_f32 / a split3 GEMM output) and the product is fp32-class, hi hi + hi lo + lo hi;
 *   both -1: one bf16 product.  Tiles of 256 x 256 cut into row parts (one workgroup each), summed in part order (bit-reproducible).
 *   hl != 0: both operands are INTERLEAVED images instead ([hi(32) | lo(32)] per 32 columns; a_hi / b_hi = the image column where the
 *   operand starts, a multiple of 64; a_lo, b_lo ignored; p % 32 == 0, q % 32 == 0) -- full 128-byte lines all the way.
 *   workspace: snf_gemm_tn_ws_bytes(n, p, q) bytes of plain scratch memory.  Domain: n % 32 == 0, p % 8 == 0, q % 8 == 0, plane offsets
 *   % 8 == 0, 16-byte aligned rows. */
size_t snf_gemm_tn_ws_bytes(int64_t n, int p, int q);
int snf_gemm_tn_f32(const void* a, int64_t lda, int a_hi, int a_lo, const void* b, int64_t ldb, int b_hi, int b_lo, int hl, int64_t n, int p,
                    int q, float* c, int64_t ldc, void* workspace, size_t workspace_bytes, snf_stream_t stream);
int snf_gemm_hl_ws_bf16(const void* a_hl, int64_t lda, const void* w_hl, int64_t ldw, const float* bias, const float* resid,
                        int64_t ldr, int64_t m, int n, int k, int act, void* c, int64_t ldc, int out_dtype, void* workspace,
                        size_t workspace_bytes, snf_stream_t stream);
int snf_split_hl_f32(const float* x, int64_t ldx, int64_t m, int k, void* out_bf16, snf_stream_t stream);
int snf_layernorm_rows_hl_f32(const float* x, int64_t n, int d, const int32_t* slot_map, const float* patch_rows,
                              const float* gamma, const float* beta, float eps, void* out_bf16, snf_stream_t stream);
/* The K patched rows of a bag, re-normalised INTO an existing hl image (snuffy.py:108 + 110 for the selected rows only): row j of
 * (x [n, d] + addend [n, d] or NULL) is layer-normalised (gamma / beta NULL = no affine) and written as the hl row out_row_idx[j]
 * of out_hl [*, 2 d].  With one normalised image serving both sublayers of an encoder layer (LayerNorm affines folded into the
 * projections) this replaces the second full LayerNorm pass over the bag. */
int snf_layernorm_rows_hl_patch_f32(const float* x, const float* addend, int64_t n, int d, const int64_t* out_row_idx,
                                    const float* gamma, const float* beta, float eps, void* out_hl, snf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K12-K14  ViT patch-embedding extractor (compute_feats.py:239-247 -> IClassifier -> VisionTransformer.forward)
 *   reference model files: utils_ssls_cf/vision_transformer_with_adapter_dino_version.py (vd), vision_transformer_dino.py,
 *   adapter.py, models_adapter_mae.py.  The dense projections are snf_gemm_bf16 (above) or library GEMMs by a measured shape policy; these entry points are the rest.
 *
 *   snf_vit_patchify         im2col of PatchEmbed's Conv2d(3, D, p, p)                vd:141-146
 *       img [b, c, hgt, wid] f32 -> cols [b * (hgt/p) * (wid/p), c*p*p] (f32 or bf16), column order (c, i, j) = the
 *       flattening of the conv weight [D, c, p, p].
 *   snf_vit_assemble_tokens  tokens[b,0] = cls + pos[0]; tokens[b,1+i] = patch_emb[b*P+i] + pos[1+i]   vd:218-229
 *       (identical arithmetic for the MAE encoder, models_adapter_mae.py:176-186).  tokens [b, P+1, d] f32.
 *   snf_vit_residual_ln      x += add1 + scale2 * add2 (bf16 addends, nullable); ln_out = LayerNorm(x) (bf16, nullable);
 *       x_bf16 = x (nullable): the two residual adds of Block.forward fused with the following norm      vd:120-127
 *   snf_vit_attention_f32    exact multi-head self-attention  softmax(q k^T * scale) v                  vd:82-94
 *       qkv [b*t, 3*h*dk] f32 laid out as the reference's qkv Linear output (q | k | v, each [h][dk]);
 *       out [b*t, h*dk]; attn [b, h, t, t] nullable.  dk in {32, 64, 96, 128}.
 *   snf_vit_attention_mfma   same on the matrix cores, bf16 in / bf16 out, dk == 64, t <= SNF_VIT_MFMA_MAX_T (else
 *       SNF_EUNSUPPORTED).  t <= 256: the keys of an (image, head) sit in one LDS image; above (the reference's patch-8 recipe,
 *       README.md:552-565: t = 785) they are staged 256 at a time and a workgroup keeps its query tiles' softmax state across the chunks.
 *   snf_vit_attention_x3_f32 same program in the fp32-class arithmetic (every product hi hi + hi lo + lo hi on the bf16 matrix
 *       cores, fp32 accumulate): fp32 in, dk == 64, t <= SNF_VIT_MFMA_MAX_T; out [b*t, h*dk] f32 (out_dtype SNF_DT_F32) or its
 *       interleaved hl image [b*t, 2*h*dk] bf16 (SNF_DT_BF16_HL: the operand of the one-pass proj GEMM).  Does not materialise attn.
 * --------------------------------------------------------------------------------------------------------- */
int snf_vit_patchify(const float* img, int b, int c, int hgt, int wid, int patch, void* cols, int out_dtype,
                     snf_stream_t stream);
int snf_vit_assemble_tokens(const void* patch_emb, int pe_dtype, const float* cls_token, const float* pos_embed, int b,
                            int num_patches, int d, float* tokens, snf_stream_t stream);
int snf_vit_residual_ln(float* x, int64_t n, int d, const void* add1_bf16, const void* add2_bf16, float scale2,
                        const float* gamma, const float* beta, float eps, void* ln_out_bf16, void* x_bf16,
                        snf_stream_t stream);
/* (mean, rstd) [n][2] of LayerNorm(eps) for snf_gemm_bf16_lnfold: from the fp32 rows x [n, ldx] (then x_bf16, nullable, receives
 * their bf16 copy -- the GEMM's operand) or from the moment pairs part [n][slots][2] a snf_gemm_bf16_resid left.  Exactly one of x / part. */
int snf_vit_row_stats(const float* x, int64_t ldx, const float* part, int slots, int64_t n, int d, float eps, float* stats,
                      void* x_bf16, int64_t ldxb, snf_stream_t stream);
int snf_vit_attention_f32(const float* qkv, int b, int t, int h, int dk, float scale, float* out, float* attn,
                          snf_stream_t stream);
#define SNF_VIT_MFMA_MAX_T 4096
int snf_vit_attention_mfma(const void* qkv_bf16, int b, int t, int h, int dk, float scale, void* out_bf16,
                           snf_stream_t stream);
int snf_vit_attention_x3_f32(const float* qkv, int b, int t, int h, int dk, float scale, void* out, int out_dtype, snf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Tile preprocessing for the extractor, batched on the device     replaces the per-tile CPU transforms of
 *     compute_feats.py:104-152,173-177: Resize(224) (PIL bilinear + antialiasing), ToTensor (/ 255), NormalizeImage
 *   img_u8 [b, h, w, c] uint8 (decoded tiles, channels last, c <= 4).  The resize is Pillow's 8-bit resampler: horizontal
 *   then vertical pass with the integer coefficient tables computed by the caller (snuffy_amd/tiles.py restates Pillow's
 *   precompute_coeffs / normalize_coeffs_8bpc): hbounds [ow][2] = (first input column, count), hcoef [ow][hks]; vbounds /
 *   vcoef likewise for rows (device int32 arrays).  mean / std: HOST arrays of c floats (used when normalize != 0).
 *   Outputs, either nullable: out_f32 [b, c, oh, ow]; cols_bf16 [b * (oh/patch) * (ow/patch), c * patch * patch] = the
 *   patch-embedding GEMM operand (column order (c, i, j)).  Bit-exact against PIL + torch on the same tiles.
 * --------------------------------------------------------------------------------------------------------- */
int snf_tile_preprocess_u8(const void* img_u8, int b, int h, int w, int c, int oh, int ow, const int* hbounds, const int* hcoef,
                           int hks, const int* vbounds, const int* vcoef, int vks, int normalize, const float* mean,
                           const float* std_, float* out_f32, void* cols_bf16, int patch, snf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Varlen path: MANY bags in one launch (bags of <= 8 k patches are launch-latency bound).
 *     The reference runs one bag per forward (train.py:468-473 batch_size 1; snuffy.py:130-131 indexes with a 1-D tensor); this
 *     is the same per-bag arithmetic with the bags' rows PACKED into one [T, .] tensor, offsets[bags + 1] = first row of each bag.
 *     Row-wise kernels (critic, LayerNorm, the projections, the FFN) already take packed rows; the three per-bag reductions get
 *     segmented forms.  Every segmented launch is the CONCATENATION of per-bag grids (workgroup wg0 + i of bag b does what
 *     workgroup i of a launch of that bag alone does): a bag's result never depends on what it is packed with, bit for bit.
 *     top-k and the head are also bit-identical to the single-bag entry points above; the attention launches give small bags
 *     more rows per workgroup than a lone launch would (fewer partial tiles), so against snf_sparse_attn_fwd_mfma / _x3 the
 *     fp32 summation order of the [k, dk] partial tiles can differ (<= 1e-6 relative).
 *   snf_topk_segmented_f32       K2 of every bag (offsets in DEVICE memory; idx_out [bags, k] = indices inside the bag; a bag
 *                                with fewer than k rows fills its first n_b entries only).
 *   snf_sparse_attn_varlen_plan / snf_sparse_attn_x3_varlen_plan
 *                                host-side geometry of a varlen attention launch: offsets in HOST memory; call with
 *                                table = NULL for the sizes, then with a host buffer of *table_ints_needed int32 words, upload
 *                                it once per batch composition and pass the device copy to the launch.
 *   snf_sparse_attn_fwd_mfma_varlen   K7 bf16 form: q, v [T, ld] bf16, kp [bags * k, h * dk] (bag b's keys in rows b k ..),
 *                                out [bags * k, h * dk] f32, attn [h, T, k] / lse [h, T] nullable.  Single key chunk
 *                                (k <= 224 at dk = 128, 256 at dk = 64), inference only (no dropout).
 *   snf_sparse_attn_fwd_x3_varlen     K7 fp32-class form on f32 q, v, kp; same layout and limits.
 *   snf_ln_mean_head_varlen_plan / snf_ln_mean_head_varlen_f32
 *                                K11 for every bag: logits [bags, c_out], pooled [bags, d] nullable.
 * --------------------------------------------------------------------------------------------------------- */
int snf_topk_segmented_f32(const float* scores, const int64_t* offsets_dev, int bags, int64_t max_n, int k, int64_t* idx_out,
                           snf_stream_t stream);
int snf_sparse_attn_varlen_plan(const int64_t* offsets, int bags, int k, int h, int dk, int32_t* table, size_t table_ints,
                                size_t* table_ints_needed, size_t* workspace_bytes);
int snf_sparse_attn_fwd_mfma_varlen(const void* q, int64_t ldq, const void* v, int64_t ldv, const void* kp, int kp_dtype,
                                    const int64_t* offsets, int bags, int k, int h, int dk, float scale, float* out, float* attn,
                                    float* lse, const int32_t* table_dev, void* workspace, size_t workspace_bytes,
                                    snf_stream_t stream);
int snf_sparse_attn_x3_varlen_plan(const int64_t* offsets, int bags, int k, int h, int dk, int32_t* table, size_t table_ints,
                                   size_t* table_ints_needed, size_t* workspace_bytes);
int snf_sparse_attn_fwd_x3_varlen(const float* q, int64_t ldq, const float* v, int64_t ldv, const float* kp, const int64_t* offsets,
                                  int bags, int k, int h, int dk, float scale, float* out, float* attn, float* lse,
                                  const int32_t* table_dev, void* workspace, size_t workspace_bytes, snf_stream_t stream);
/* Ragged form for SMALL bags (exact fp32, any head width, every bag with its own key count: a bag shorter than Lambda selects
 * all of its rows -- snuffy.py:129 `min(ceil(...), n)` -- as the MIL benchmark sets do).  desc_dev [bags][4] int32 in DEVICE
 * memory = (first packed row, rows, first key row, keys) per bag; kp / out [sum of keys, h * dk]; attn [h, T, kmax] nullable
 * (bag b's A = attn[:, row0 : row0 + n, :k]).  kmax <= 256 and (16 + dk) * kmax floats of LDS <= 160 KiB. */
int snf_sparse_attn_fwd_ragged_f32(const float* q, int64_t ldq, const float* v, int64_t ldv, const float* kp, const int32_t* desc_dev,
                                   int bags, int64_t n_total, int kmax, int h, int dk, float scale, float* out, float* attn,
                                   float* lse, snf_stream_t stream);
int snf_ln_mean_head_varlen_plan(const int64_t* offsets, int bags, int d, int32_t* table, size_t table_ints,
                                 size_t* table_ints_needed, size_t* workspace_bytes);
int snf_ln_mean_head_varlen_f32(const float* z, const int64_t* offsets, int bags, int d, const void* add_bf16,
                                const float* add_bias, const int32_t* slot_map, const float* delta_rows, float* z_out,
                                const float* gamma, const float* beta, float eps, const float* w_head, const float* b_head,
                                int c_out, float* logits, float* pooled, const int32_t* table_dev, void* workspace,
                                size_t workspace_bytes, snf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K3  device-side random patch share (opt-in fast mode)       replaces snuffy.py:136-147 (np.random.choice on the host behind a
 *     device -> host copy of the selected rows) when the caller asks for it; the reference's own MT19937 draws stay the default.
 *   snf_random_share_keys_f32: keys[r] = the non-negative float whose bit pattern is a 30-bit Philox4x32-10 output for row r (key =
 *     state[0] = seed, counter = (r / 4, state[1] + (layer << 48)), element r & 3), keys[exclude_rows[i]] = -1.  The k2 largest keys
 *     (snf_topk_f32) are a uniform sample without replacement of the rows not excluded, in random order.
 *   state: 16-byte device record {u64 seed, u64 offset}, owned by the caller; snf_sampler_advance adds 1 to the offset ON THE DEVICE,
 *     so a captured graph (advance, keys, top-k) draws fresh rows on every replay.  Host twin: oracle/philox_ref.py.
 * --------------------------------------------------------------------------------------------------------- */
int snf_sampler_advance(void* state, snf_stream_t stream);
int snf_random_share_keys_f32(const void* state, int layer, int64_t n, const int64_t* exclude_rows, int n_exclude, float* keys,
                              snf_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Debug hooks -- PROCESS-WIDE state, outside the stateless / re-entrant contract at the top of this header (see there).  Callers:
 * tests/ (cross-checks of kernel variants against each other) and tools/ (A / B timing, traces); nothing in snuffy_amd/ calls them.
 * snf_debug_attn_trace(buf): device buffer of u64 that dev builds of the attention kernels (X3P_TRACE / SNF_ATTN_TRACE defines)
 * fill with s_memtime stamps of workgroup snf_debug_attn_trace_wg(wg); null switches the stamps off.  Shipped builds carry no
 * stamp code: the calls only set two host-side variables. */
void snf_debug_attn_trace(void* buf);
void snf_debug_attn_trace_wg(int wg);
/* snf_debug_x3p_kbw(kbw): key blocks per wave of the dk = 128 family of snf_sparse_attn_fwd_x3_hl for 129 .. 256 keys per launch --
 * 1 (default: one wave per key block, two per SIMD) or 2 (one wave per SIMD carrying two key blocks, round 5); same arithmetic,
 * results equal up to the fp32 order of the row sums.  For A / B timing (tools/x3p_dev.py) and the parity tests of the second form. */
void snf_debug_x3p_kbw(int kbw);
/* snf_debug_exact_attn_mfma(on): the exact-fp32 attention kernels (snf_sparse_attn_fwd_f32, the dKp contraction of the backward) on
 * v_mfma_f32_32x32x2_f32 (1, default; head widths % 8 == 0, <= 1024 keys for the scores) or on the vector ALUs for every shape (0);
 * 2 = the matrix-core forms with the dQ / dV kernel of the backward reading its operands straight from L2 instead of staging them through
 * LDS (the round-5 A / B partner; same fmaf chains, bit-identical results). */
void snf_debug_exact_attn_mfma(int on);

#ifdef __cplusplus
}
#endif
#endif /* SNUFFY_HIP_H */
