// Row-wise HBM-bound kernels of the Snuffy aggregator: critic GEMV (+ column max), LayerNorm with fused row patch,
// gather / scatter of the K selected rows, bias+activation epilogue, LayerNorm + mean-pool + head.
//
// Layout: one 64-lane wave owns one row at a time (4 rows per 256-thread workgroup, grid-stride over rows);
// a lane holds NV vectors of VEC floats of its row in registers (coalesced 16-B loads when d % 4 == 0), so every
// row is read from HBM exactly once per pass.
#include <math.h>

#include "common.h"
#include "philox.h"
#include "selector.h"

namespace {

constexpr int WG = 256;
constexpr int WAVES = WG / 64;

template <int VEC>
struct VecT;
template <>
struct VecT<4> {
    using type = float4;
};
template <>
struct VecT<1> {
    using type = float;
};

template <int VEC, int NV>
struct RowRegs {
    float v[NV * VEC];
};

// load a row (d floats) into registers: element e = (i*64 + lane)*VEC + t; out-of-range -> 0
template <int VEC, int NV>
__device__ __forceinline__ void load_row(const float* __restrict__ row, int d, int lane, float (&r)[NV * VEC]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int e = (i * 64 + lane) * VEC;
        if constexpr (VEC == 4) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < d) t = *reinterpret_cast<const float4*>(row + e);
            r[i * 4 + 0] = t.x;
            r[i * 4 + 1] = t.y;
            r[i * 4 + 2] = t.z;
            r[i * 4 + 3] = t.w;
        } else {
            r[i] = (e < d) ? row[e] : 0.f;
        }
    }
}

template <int VEC, int NV>
__device__ __forceinline__ void store_row_f32(float* __restrict__ row, int d, int lane, const float (&r)[NV * VEC]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int e = (i * 64 + lane) * VEC;
        if (e < d) {
            if constexpr (VEC == 4) {
                *reinterpret_cast<float4*>(row + e) = make_float4(r[i * 4], r[i * 4 + 1], r[i * 4 + 2], r[i * 4 + 3]);
            } else {
                row[e] = r[i];
            }
        }
    }
}

template <int VEC, int NV>
__device__ __forceinline__ void store_row_bf16(unsigned short* __restrict__ row, int d, int lane,
                                               const float (&r)[NV * VEC]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int e = (i * 64 + lane) * VEC;
        if (e < d) {
            if constexpr (VEC == 4) {
                uint2 p;
                p.x = pack_bf16x2(r[i * 4], r[i * 4 + 1]);
                p.y = pack_bf16x2(r[i * 4 + 2], r[i * 4 + 3]);
                *reinterpret_cast<uint2*>(row + e) = p;
            } else {
                row[e] = f32_to_bf16_bits(r[i]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K1 critic: scores[i, c] = x[i,:] . w[c,:] + b[c]          (FCLayer.forward, snuffy.py:39-41)
// ---------------------------------------------------------------------------------------------------------------
// LN = also emit the row-normalised bf16 copy (x - mean) * rstd that the first encoder layer needs (its LayerNorm affine is
// folded into the projection weights): the bag is read from HBM once for both, instead of once per kernel.
// LN == 2 (fp32-class path): emit LayerNorm(x) WITH its affine as the interleaved hi / lo image the one-pass GEMM reads
// (the output of snf_layernorm_rows_hl_f32 up to the last fp32 place of the normalised value) instead of the affine-free bf16 copy.
template <int VEC, int NV, int LN>
__global__ __launch_bounds__(WG) void critic_kernel(const float* __restrict__ x, int64_t n, int d,
                                                    const float* __restrict__ w, const float* __restrict__ b,
                                                    int c_out, float* __restrict__ scores, float eps,
                                                    unsigned short* __restrict__ xhat,
                                                    unsigned int* __restrict__ sel_hist,
                                                    const float* __restrict__ gamma = nullptr,
                                                    const float* __restrict__ beta = nullptr) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const float inv_d = 1.0f / (float)d;
    // fused selector (sel_hist != null, c_out == 1): the first radix digit of every score this workgroup writes is counted in
    // LDS and flushed once, at the end, into one of the global replicas -- integer atomics: exact and order-independent
    __shared__ unsigned int s_hist[snf::SEL_BINS];
    if (sel_hist) {
        for (int i = threadIdx.x; i < snf::SEL_BINS; i += WG) s_hist[i] = 0;
        __syncthreads();
    }
    // class-0 weights stay in registers (the usual critic has ONE output), and the next row of x is requested before the
    // current one is reduced: the pass is bound by one HBM round trip per row and wave, not by bytes
    float w0[NV * VEC];
    load_row<VEC, NV>(w, d, lane, w0);
    const float b0 = b ? b[0] : 0.f;
    const int64_t rstride = (int64_t)gridDim.x * WAVES;
    float nx[NV * VEC];
    int64_t row = (int64_t)blockIdx.x * WAVES + wave;
    if (row < n) load_row<VEC, NV>(x + row * d, d, lane, nx);
    for (; row < n; row += rstride) {
        float r[NV * VEC];
#pragma unroll
        for (int i = 0; i < NV * VEC; ++i) r[i] = nx[i];
        if (row + rstride < n) load_row<VEC, NV>(x + (row + rstride) * d, d, lane, nx);   // wave-uniform
        {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < NV * VEC; ++i) acc = fmaf(r[i], w0[i], acc);
            acc = wave_sum(acc);
            if (lane == 0) {
                scores[row * c_out] = acc + b0;
                if (sel_hist) atomicAdd(&s_hist[snf::orderable_desc(acc + b0) >> 21], 1u);
            }
        }
        for (int c = 1; c < c_out; ++c) {
            float wr[NV * VEC];
            load_row<VEC, NV>(w + (int64_t)c * d, d, lane, wr);
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < NV * VEC; ++i) acc = fmaf(r[i], wr[i], acc);
            acc = wave_sum(acc);
            if (lane == 0) scores[row * c_out + c] = acc + (b ? b[c] : 0.f);
        }
        if constexpr (LN != 0) {   // same arithmetic, in the same order, as layernorm_rows_kernel
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < NV * VEC; ++i) s1 += r[i];
            const float mean = wave_sum(s1) * inv_d;
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
#pragma unroll
                for (int t = 0; t < VEC; ++t) {
                    int e = (i * 64 + lane) * VEC + t;
                    float dv = (e < d) ? (r[i * VEC + t] - mean) : 0.f;
                    s2 = fmaf(dv, dv, s2);
                }
            }
            const float var = wave_sum(s2) * inv_d;
            const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
            for (int i = 0; i < NV * VEC; ++i) r[i] = (r[i] - mean) * rstd;
            if constexpr (LN == 1) store_row_bf16<VEC, NV>(xhat + row * d, d, lane, r);
            if constexpr (LN == 2 && VEC == 4) {
                // affine (the registers that held the critic weights' row are free again only after the loop: gamma / beta are
                // re-read per row from L1 / L2 -- 2 x d floats against the d floats of the row itself)
                // gamma == null: the affine-free image (one normalised image for both sublayers, affines folded into the projections)
                if (gamma) {
                    float g[NV * VEC], bt[NV * VEC];
                    load_row<VEC, NV>(gamma, d, lane, g);
                    load_row<VEC, NV>(beta, d, lane, bt);
#pragma unroll
                    for (int i = 0; i < NV * VEC; ++i) {
                        r[i] = fmaf(r[i], g[i], bt[i]);
                    }
                }
                unsigned short* o2 = xhat + row * 2 * d;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int e = (i * 64 + lane) * 4;
                    if (e < d) {
                        uint2 hi, lo;
                        hi.x = pack_bf16x2(r[i * 4], r[i * 4 + 1]);
                        hi.y = pack_bf16x2(r[i * 4 + 2], r[i * 4 + 3]);
                        lo.x = pack_bf16x2(r[i * 4] - __uint_as_float(hi.x << 16), r[i * 4 + 1] - __uint_as_float(hi.x & 0xffff0000u));
                        lo.y = pack_bf16x2(r[i * 4 + 2] - __uint_as_float(hi.y << 16), r[i * 4 + 3] - __uint_as_float(hi.y & 0xffff0000u));
                        unsigned short* o = o2 + 64 * (e >> 5) + (e & 31);
                        *reinterpret_cast<uint2*>(o) = hi;
                        *reinterpret_cast<uint2*>(o + 32) = lo;
                    }
                }
            }
        }
    }
    if (sel_hist) {
        __syncthreads();
        unsigned int* g = sel_hist + (blockIdx.x & (snf::SEL_REPL - 1)) * snf::SEL_BINS;
        for (int i = threadIdx.x; i < snf::SEL_BINS; i += WG) {
            const unsigned int c = s_hist[i];
            if (c) __hip_atomic_fetch_add(&g[i], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// column max + first index of the max (torch.max(ins_prediction, 1), train.py:831-834). One workgroup per class.
__global__ __launch_bounds__(1024) void colmax_kernel(const float* __restrict__ scores, int64_t n, int c_out,
                                                      float* __restrict__ out_val, int64_t* __restrict__ out_idx) {
    const int c = blockIdx.x;
    float best = -INFINITY;
    int64_t bi = INT64_MAX;
    bool has_nan = false;
    int64_t nan_i = INT64_MAX;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        float v = scores[i * c_out + c];
        if (v != v) {
            if (!has_nan) nan_i = i;
            has_nan = true;
        } else if (v > best || (v == best && i < bi)) {
            best = v;
            bi = i;
        }
    }
    __shared__ float sv[1024];
    __shared__ int64_t si[1024];
    __shared__ int64_t sn[1024];
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    sn[threadIdx.x] = has_nan ? nan_i : INT64_MAX;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            float ov = sv[threadIdx.x + s];
            int64_t oi = si[threadIdx.x + s];
            if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
            if (sn[threadIdx.x + s] < sn[threadIdx.x]) sn[threadIdx.x] = sn[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (sn[0] != INT64_MAX) {  // NaN propagates like torch.max
            if (out_val) out_val[c] = NAN;
            if (out_idx) out_idx[c] = sn[0];
        } else {
            if (out_val) out_val[c] = sv[0];
            if (out_idx) out_idx[c] = si[0];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K5 LayerNorm rows with fused patch-row read (snuffy.py:97,107,110 + 152-155)
// ---------------------------------------------------------------------------------------------------------------
template <int VEC, int NV>
__global__ __launch_bounds__(WG) void layernorm_rows_kernel(const float* __restrict__ x, int64_t n, int d,
                                                            const int32_t* __restrict__ slot_map,
                                                            const float* __restrict__ patch_rows,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            float* __restrict__ out_f32,
                                                            unsigned short* __restrict__ out_bf16,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            const int64_t* __restrict__ out_row_idx, int split3,
                                                            const float* __restrict__ addend = nullptr) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float g[NV * VEC], bt[NV * VEC];
    if (gamma) load_row<VEC, NV>(gamma, d, lane, g);
    if (beta) load_row<VEC, NV>(beta, d, lane, bt);
    const float inv_d = 1.0f / (float)d;
    for (int64_t row = (int64_t)blockIdx.x * WAVES + wave; row < n; row += (int64_t)gridDim.x * WAVES) {
        const float* src = x + row * d;
        if (slot_map) {
            int s = slot_map[row];
            if (s >= 0) src = patch_rows + (int64_t)s * d;
        }
        float r[NV * VEC];
        load_row<VEC, NV>(src, d, lane, r);
        if (addend) {   // row = x + addend (the sublayer's residual sum, snuffy.py:108), rounded once to fp32 like the tensor it replaces
            float ad[NV * VEC];
            load_row<VEC, NV>(addend + row * d, d, lane, ad);
#pragma unroll
            for (int i = 0; i < NV * VEC; ++i) r[i] += ad[i];
        }
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < NV * VEC; ++i) s1 += r[i];
        const float mean = wave_sum(s1) * inv_d;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
#pragma unroll
            for (int t = 0; t < VEC; ++t) {
                int e = (i * 64 + lane) * VEC + t;
                float dv = (e < d) ? (r[i * VEC + t] - mean) : 0.f;
                s2 = fmaf(dv, dv, s2);
            }
        }
        const float var = wave_sum(s2) * inv_d;
        const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int i = 0; i < NV * VEC; ++i) {
            float v = (r[i] - mean) * rstd;
            if (gamma) v *= g[i];
            if (beta) v += bt[i];
            r[i] = v;
        }
        const int64_t orow = out_row_idx ? out_row_idx[row] : row;
        if (out_f32) store_row_f32<VEC, NV>(out_f32 + orow * d, d, lane, r);
        if (out_bf16 && !split3) store_row_bf16<VEC, NV>(out_bf16 + orow * d, d, lane, r);
        if (out_bf16 && split3 == 1) {
            // [hi | hi | lo] image of the row (3 d bf16): the A operand of a split-bf16 x3 GEMM (gemm.hip, header comment)
            float lo[NV * VEC];
#pragma unroll
            for (int i = 0; i < NV * VEC; ++i) lo[i] = r[i] - bf16_bits_to_f32(f32_to_bf16_bits(r[i]));
            unsigned short* o3 = out_bf16 + orow * 3 * d;
            store_row_bf16<VEC, NV>(o3, d, lane, r);
            store_row_bf16<VEC, NV>(o3 + d, d, lane, r);
            store_row_bf16<VEC, NV>(o3 + 2 * d, d, lane, lo);
        }
        if constexpr (VEC == 4) {
            if (out_bf16 && split3 == 2) {
                // interleaved ("hl") image of the row (2 d bf16): every 32 columns as [hi(32) | lo(32)] -- one 128-byte line per
                // K step of the one-pass fp32-class GEMM (gemm.hip, gemm_hl_kernel).  A lane's 4 consecutive columns share a chunk.
                unsigned short* o2 = out_bf16 + orow * 2 * d;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int e = (i * 64 + lane) * 4;
                    if (e < d) {
                        uint2 hi, lo;
                        hi.x = pack_bf16x2(r[i * 4], r[i * 4 + 1]);
                        hi.y = pack_bf16x2(r[i * 4 + 2], r[i * 4 + 3]);
                        lo.x = pack_bf16x2(r[i * 4] - __uint_as_float(hi.x << 16), r[i * 4 + 1] - __uint_as_float(hi.x & 0xffff0000u));
                        lo.y = pack_bf16x2(r[i * 4 + 2] - __uint_as_float(hi.y << 16), r[i * 4 + 3] - __uint_as_float(hi.y & 0xffff0000u));
                        unsigned short* o = o2 + 64 * (e >> 5) + (e & 31);
                        *reinterpret_cast<uint2*>(o) = hi;
                        *reinterpret_cast<uint2*>(o + 32) = lo;
                    }
                }
            }
        }
        if (lane == 0) {
            if (mean_out) mean_out[orow] = mean;
            if (rstd_out) rstd_out[orow] = rstd;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm backward over rows (autograd through SublayerConnection.norm / Encoder.norm, snuffy.py:86,97,107,110)
//   xhat = (x - mean) * rstd (recomputed from x);  dxhat = dy * gamma
//   dx = res + rstd * (dxhat - mean_d(dxhat) - xhat * mean_d(dxhat * xhat))
//   partials[block] = (sum_rows dy * xhat, sum_rows dy): the caller sums them over blocks for dgamma / dbeta.
// dy_stride = 0 broadcasts ONE gradient row to every row (the mean-pooled head: d logits / d LN(z)_i is the same for all i).
// ---------------------------------------------------------------------------------------------------------------
template <int VEC, int NV>
__device__ __forceinline__ void load_row_bf16(const unsigned short* __restrict__ row, int d, int lane, float (&r)[NV * VEC]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int e = (i * 64 + lane) * VEC;
        if constexpr (VEC == 4) {
            uint2 t = make_uint2(0u, 0u);
            if (e < d) t = *reinterpret_cast<const uint2*>(row + e);
            r[i * 4 + 0] = __uint_as_float(t.x << 16);
            r[i * 4 + 1] = __uint_as_float(t.x & 0xffff0000u);
            r[i * 4 + 2] = __uint_as_float(t.y << 16);
            r[i * 4 + 3] = __uint_as_float(t.y & 0xffff0000u);
        } else {
            r[i] = (e < d) ? bf16_bits_to_f32(row[e]) : 0.f;
        }
    }
}

template <int VEC, int NV>
__global__ __launch_bounds__(WG) void ln_bwd_rows_kernel(const float* __restrict__ x, int64_t n, int d,
                                                         const void* __restrict__ dy, int dy_bf16, int64_t dy_stride,
                                                         const float* __restrict__ gamma, float eps,
                                                         const float* __restrict__ res, float* __restrict__ dx,
                                                         unsigned short* __restrict__ dx_bf16,
                                                         float* __restrict__ partials) {
    constexpr int R = NV * VEC;
    __shared__ float red[WAVES][2][R * 64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float g[R], gy[R], sg[R], sb[R];
    if (gamma) load_row<VEC, NV>(gamma, d, lane, g);
#pragma unroll
    for (int i = 0; i < R; ++i) sg[i] = 0.f, sb[i] = 0.f;
    const float inv_d = 1.0f / (float)d;
    for (int64_t row = (int64_t)blockIdx.x * WAVES + wave; row < n; row += (int64_t)gridDim.x * WAVES) {
        float r[R];
        load_row<VEC, NV>(x + row * d, d, lane, r);
        if (dy_bf16)
            load_row_bf16<VEC, NV>(reinterpret_cast<const unsigned short*>(dy) + row * dy_stride, d, lane, gy);
        else
            load_row<VEC, NV>(reinterpret_cast<const float*>(dy) + row * dy_stride, d, lane, gy);
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < R; ++i) s1 += r[i];
        const float mean = wave_sum(s1) * inv_d;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
#pragma unroll
            for (int t = 0; t < VEC; ++t) {
                int e = (i * 64 + lane) * VEC + t;
                float dv = (e < d) ? (r[i * VEC + t] - mean) : 0.f;
                r[i * VEC + t] = dv;
                s2 = fmaf(dv, dv, s2);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(s2) * inv_d + eps);
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            r[i] *= rstd;                       // xhat (0 past d)
            sg[i] = fmaf(gy[i], r[i], sg[i]);
            sb[i] += gy[i];
            if (gamma) gy[i] *= g[i];           // dxhat (0 past d: dy loads are zero-filled)
            m1 += gy[i];
            m2 = fmaf(gy[i], r[i], m2);
        }
        m1 = wave_sum(m1) * inv_d;
        m2 = wave_sum(m2) * inv_d;
        if (dx || dx_bf16) {
            float o[R];
            if (res) load_row<VEC, NV>(res + row * d, d, lane, o);
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const float v = rstd * (gy[i] - m1 - r[i] * m2);
                o[i] = res ? o[i] + v : v;
            }
            if (dx) store_row_f32<VEC, NV>(dx + row * d, d, lane, o);
            if (dx_bf16) store_row_bf16<VEC, NV>(dx_bf16 + row * d, d, lane, o);
        }
    }
    if (!partials) return;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int t = 0; t < VEC; ++t) {
            red[wave][0][(i * 64 + lane) * VEC + t] = sg[i * VEC + t];
            red[wave][1][(i * 64 + lane) * VEC + t] = sb[i * VEC + t];
        }
    __syncthreads();
    float* out = partials + (int64_t)blockIdx.x * 2 * d;
    for (int e = threadIdx.x; e < 2 * d; e += WG) {
        const int which = e >= d, c = which ? e - d : e;
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) a += red[w][which][c];
        out[e] = a;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Column sums of a [n, d] matrix fused with the elementwise work that the training step does on the same pass:
//   v = src[i, c]  (f32 or bf16)  ->  * wgt[i] (optional row weight)  ->  0 where gate[i, c] <= 0 (optional, bf16: the ReLU
//   mask taken from the activation's OUTPUT)  ->  dst[i, c] = bf16(v) (optional, may alias src)  ;  partial[b, c] += v
// Replaces  threshold_backward + sum(0),  to(bfloat16) + sum(0)  and the [1, n] x [n, d] GEMM of the critic's weight gradient
// (train.py:259 backward through snuffy.py:39-41, 224-225).  A workgroup owns a contiguous row range; thread t owns columns
// 8 t .. 8 t + 7 (+ 2048 j), so its sums need no reduction inside the workgroup; partial [gridDim.x, d] is summed by the caller
// in a fixed order.
// ---------------------------------------------------------------------------------------------------------------
template <bool SRC_BF16, int NCH>
__global__ __launch_bounds__(256) void colsum_fused_kernel(const void* __restrict__ src, int64_t n, int d,
                                                           const float* __restrict__ wgt, int64_t wgt_stride,
                                                           const unsigned short* __restrict__ gate,
                                                           unsigned short* __restrict__ dst, float* __restrict__ partial,
                                                           int rows_per_block) {
    float acc[NCH][8];
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > n) r1 = n;
    for (int64_t row = r0; row < r1; ++row) {
        const float wv = wgt ? wgt[row * wgt_stride] : 1.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = (j * 256 + threadIdx.x) * 8;
            if (c < d) {
                float v[8];
                if constexpr (SRC_BF16) {
                    const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(src) + row * d + c);
                    const unsigned w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] = __uint_as_float(w4[e] << 16);
                        v[2 * e + 1] = __uint_as_float(w4[e] & 0xffff0000u);
                    }
                } else {
                    const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + row * d + c);
                    const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + row * d + c + 4);
                    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
                }
                if (wgt) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= wv;
                }
                if (gate) {
                    const uint4 g = *reinterpret_cast<const uint4*>(gate + row * d + c);
                    const unsigned g4[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (!(__uint_as_float(g4[e] << 16) > 0.f)) v[2 * e] = 0.f;
                        if (!(__uint_as_float(g4[e] & 0xffff0000u) > 0.f)) v[2 * e + 1] = 0.f;
                    }
                }
                if (dst) {
                    uint4 o;
                    o.x = pack_bf16x2(v[0], v[1]), o.y = pack_bf16x2(v[2], v[3]);
                    o.z = pack_bf16x2(v[4], v[5]), o.w = pack_bf16x2(v[6], v[7]);
                    *reinterpret_cast<uint4*>(dst + row * d + c) = o;
                    // the sums are taken over the ROUNDED values that the following GEMMs consume
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned w = e == 0 ? o.x : e == 1 ? o.y : e == 2 ? o.z : o.w;
                        v[2 * e] = __uint_as_float(w << 16);
                        v[2 * e + 1] = __uint_as_float(w & 0xffff0000u);
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[j][e] += v[e];
            }
        }
    }
    float* out = partial + (int64_t)blockIdx.x * d;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = (j * 256 + threadIdx.x) * 8;
        if (c < d) {
            *reinterpret_cast<float4*>(out + c) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
            *reinterpret_cast<float4*>(out + c + 4) = make_float4(acc[j][4], acc[j][5], acc[j][6], acc[j][7]);
        }
    }
}

// x [m, k] f32 (row pitch ldx) -> out [m, 3 k] bf16 = [hi | hi | lo]: the A operand of an fp32-class (split-bf16 x3) GEMM.
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, int64_t ldx, int64_t m, int k,
                                                     unsigned short* __restrict__ out) {
    const int kq = k >> 3;                                   // 8-element groups per row
    const int64_t total = m * kq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / kq;
        const int c = (int)(i - row * kq) * 8;
        const float4 a = *reinterpret_cast<const float4*>(x + row * ldx + c);
        const float4 b = *reinterpret_cast<const float4*>(x + row * ldx + c + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        unsigned hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
            lo[e] = pack_bf16x2(v[2 * e] - __uint_as_float(hi[e] << 16), v[2 * e + 1] - __uint_as_float(hi[e] & 0xffff0000u));
        }
        unsigned short* o = out + row * 3 * k + c;
        const uint4 h4 = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(o) = h4;
        *reinterpret_cast<uint4*>(o + k) = h4;
        *reinterpret_cast<uint4*>(o + 2 * k) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
}

// w [m, k] f32 (row pitch ldw), optionally scaled per column (the LayerNorm gamma folded into the projection, W' = W diag(gamma), in fp32)
// -> out [m, 3 k] bf16 = [Wh | Wl | Wh]: the weight operand of the concatenated-K fp32-class GEMM, in ONE pass (the training step rebuilds
// its five weight images after every optimizer step: five launches instead of ~25 elementwise ones).
__global__ __launch_bounds__(256) void split3_weight_kernel(const float* __restrict__ w, int64_t ldw, int64_t m, int k,
                                                            const float* __restrict__ colscale, unsigned short* __restrict__ out) {
    const int kq = k >> 3;
    const int64_t total = m * kq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / kq;
        const int c = (int)(i - row * kq) * 8;
        const float4 a = *reinterpret_cast<const float4*>(w + row * ldw + c);
        const float4 b = *reinterpret_cast<const float4*>(w + row * ldw + c + 4);
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        if (colscale) {
            const float4 g0 = *reinterpret_cast<const float4*>(colscale + c), g1 = *reinterpret_cast<const float4*>(colscale + c + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= gg[e];
        }
        unsigned hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
            lo[e] = pack_bf16x2(v[2 * e] - __uint_as_float(hi[e] << 16), v[2 * e + 1] - __uint_as_float(hi[e] & 0xffff0000u));
        }
        unsigned short* o = out + row * 3 * k + c;
        const uint4 h4 = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(o) = h4;
        *reinterpret_cast<uint4*>(o + k) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        *reinterpret_cast<uint4*>(o + 2 * k) = h4;
    }
}

// split3_kernel + the ReLU gate + the column sums of the gated values in one pass (see snf_split3_colsum_f32); row / column ownership as
// colsum_fused_kernel: a workgroup owns a contiguous row range, thread t columns 8 t .. 8 t + 7 (+ 2048 j).
// HL: the image (and the gate's) is the interleaved one of gemm_hl / snf_gemm_tn_f32 -- [hi(32) | lo(32)] per 32 columns, 2 k columns wide
template <int NCH, bool HL = false>
__global__ __launch_bounds__(256) void split3_colsum_kernel(const float* __restrict__ x, int64_t ldx, int64_t m, int k,
                                                            const unsigned short* __restrict__ gate, int64_t ldg,
                                                            unsigned short* __restrict__ out, int64_t ldo, int64_t plane,
                                                            float* __restrict__ partial, int rows_per_block) {
    float acc[NCH][8];
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > m) r1 = m;
    for (int64_t row = r0; row < r1; ++row) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = (j * 256 + threadIdx.x) * 8;
            if (c < k) {
                const float4 a = *reinterpret_cast<const float4*>(x + row * ldx + c);
                const float4 b = *reinterpret_cast<const float4*>(x + row * ldx + c + 4);
                float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                const int ci = HL ? 64 * (c >> 5) + (c & 31) : c;   // position of the 8 columns' hi values inside an image row
                if (gate) {
                    const uint4 g = *reinterpret_cast<const uint4*>(gate + row * ldg + ci);
                    const unsigned g4[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (!(__uint_as_float(g4[e] << 16) > 0.f)) v[2 * e] = 0.f;
                        if (!(__uint_as_float(g4[e] & 0xffff0000u) > 0.f)) v[2 * e + 1] = 0.f;
                    }
                }
                unsigned hi[4], lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hi[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                    lo[e] = pack_bf16x2(v[2 * e] - __uint_as_float(hi[e] << 16), v[2 * e + 1] - __uint_as_float(hi[e] & 0xffff0000u));
                }
                unsigned short* o = out + row * ldo + ci;
                const uint4 h4 = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                *reinterpret_cast<uint4*>(o) = h4;
                if constexpr (HL) {
                    *reinterpret_cast<uint4*>(o + 32) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
                } else {
                    *reinterpret_cast<uint4*>(o + plane) = h4;
                    *reinterpret_cast<uint4*>(o + 2 * plane) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[j][e] += v[e];
            }
        }
    }
    if (!partial) return;
    float* po = partial + (int64_t)blockIdx.x * k;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = (j * 256 + threadIdx.x) * 8;
        if (c < k) {
            *reinterpret_cast<float4*>(po + c) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
            *reinterpret_cast<float4*>(po + c + 4) = make_float4(acc[j][4], acc[j][5], acc[j][6], acc[j][7]);
        }
    }
}

// x [m, k] f32 (row pitch ldx) -> out [m, 2 k] bf16, interleaved: columns 32 c .. 32 c + 31 as [hi(32) | lo(32)] (k % 32 == 0)
__global__ __launch_bounds__(256) void split_hl_kernel(const float* __restrict__ x, int64_t ldx, int64_t m, int k,
                                                       unsigned short* __restrict__ out) {
    const int kq = k >> 3;
    const int64_t total = m * kq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / kq;
        const int c = (int)(i - row * kq) * 8;
        const float4 a = *reinterpret_cast<const float4*>(x + row * ldx + c);
        const float4 b = *reinterpret_cast<const float4*>(x + row * ldx + c + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        unsigned hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
            lo[e] = pack_bf16x2(v[2 * e] - __uint_as_float(hi[e] << 16), v[2 * e + 1] - __uint_as_float(hi[e] & 0xffff0000u));
        }
        unsigned short* o = out + row * 2 * k + 64 * (c >> 5) + (c & 31);
        *reinterpret_cast<uint4*>(o) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(o + 32) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm affine folded into the following projection, and back (bf16 path, functional._folded / the training chain):
//   fold:    Wf[r, c] = bf16(W[r, c] * gamma[c]) ;  bf[r] = sum_c W[r, c] * beta[c] + bias[r]
//   unfold:  dW[r, c] = dWf[r, c] * gamma[c] + dbf[r] * beta[c]
//            partial[b] = (sum_r dWf[r, c] * W[r, c],  sum_r dbf[r] * W[r, c])   -> dgamma, dbeta after the caller's sum over b
// one wave per weight row (C <= 2048), the same row helpers as LayerNorm.
// ---------------------------------------------------------------------------------------------------------------
template <int VEC, int NV>
__global__ __launch_bounds__(WG) void fold_linear_kernel(const float* __restrict__ w, int r, int c,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ bias, unsigned short* __restrict__ wf,
                                                         int64_t ldwf, float* __restrict__ bf, unsigned short* __restrict__ bf16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[NV * VEC], bt[NV * VEC];
    load_row<VEC, NV>(gamma, c, lane, g);
    load_row<VEC, NV>(beta, c, lane, bt);
    for (int row = blockIdx.x * WAVES + wave; row < r; row += gridDim.x * WAVES) {
        float x[NV * VEC];
        load_row<VEC, NV>(w + (int64_t)row * c, c, lane, x);
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NV * VEC; ++i) {
            acc = fmaf(x[i], bt[i], acc);
            x[i] *= g[i];
        }
        acc = wave_sum(acc);
        store_row_bf16<VEC, NV>(wf + (int64_t)row * ldwf, c, lane, x);
        if (lane == 0) {
            const float b = acc + (bias ? bias[row] : 0.f);
            if (bf) bf[row] = b;
            if (bf16) bf16[row] = f32_to_bf16_bits(b);
        }
    }
}

template <int VEC, int NV>
__global__ __launch_bounds__(WG) void unfold_linear_kernel(const float* __restrict__ dwf, const float* __restrict__ w, int r, int c,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ dbf, float* __restrict__ dw,
                                                           float* __restrict__ partial) {
    constexpr int R = NV * VEC;
    __shared__ float red[WAVES][2][R * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[R], bt[R], sg[R], sb[R];
    load_row<VEC, NV>(gamma, c, lane, g);
    load_row<VEC, NV>(beta, c, lane, bt);
#pragma unroll
    for (int i = 0; i < R; ++i) sg[i] = 0.f, sb[i] = 0.f;
    for (int row = blockIdx.x * WAVES + wave; row < r; row += gridDim.x * WAVES) {
        float d[R], x[R];
        load_row<VEC, NV>(dwf + (int64_t)row * c, c, lane, d);
        load_row<VEC, NV>(w + (int64_t)row * c, c, lane, x);
        const float db = dbf[row];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            sg[i] = fmaf(d[i], x[i], sg[i]);
            sb[i] = fmaf(db, x[i], sb[i]);
            d[i] = fmaf(d[i], g[i], db * bt[i]);
        }
        store_row_f32<VEC, NV>(dw + (int64_t)row * c, c, lane, d);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int t = 0; t < VEC; ++t) {
            red[wave][0][(i * 64 + lane) * VEC + t] = sg[i * VEC + t];
            red[wave][1][(i * 64 + lane) * VEC + t] = sb[i * VEC + t];
        }
    __syncthreads();
    float* out = partial + (int64_t)blockIdx.x * 2 * c;
    for (int e = threadIdx.x; e < 2 * c; e += WG) {
        const int which = e >= c, col = which ? e - c : e;
        float a = 0.f;
#pragma unroll
        for (int wv = 0; wv < WAVES; ++wv) a += red[wv][which][col];
        out[e] = a;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// gather / scatter of the K selected rows
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void gather_rows_kernel(const float* __restrict__ x, int64_t n, int d,
                                                         const int64_t* __restrict__ idx, int k,
                                                         float* __restrict__ out) {
    const int j = blockIdx.x;
    if (j >= k) return;
    int64_t src = idx[j];
    if (src < 0 || src >= n) return;  // guarded on the host in debug paths; never write garbage
    const float* s = x + src * d;
    float* o = out + (int64_t)j * d;
    for (int e = threadIdx.x; e < d; e += blockDim.x) o[e] = s[e];
}

__global__ __launch_bounds__(WG) void copy_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t total) {
    int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (; i4 + 3 < total; i4 += stride) *reinterpret_cast<float4*>(y + i4) = *reinterpret_cast<const float4*>(x + i4);
    // tail (total % 4) handled by the last thread range
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int64_t i = total & ~(int64_t)3; i < total; ++i) y[i] = x[i];
    }
}

template <bool ADD>
__global__ __launch_bounds__(WG) void scatter_rows_kernel(float* __restrict__ y, int64_t n, int d,
                                                          const int64_t* __restrict__ idx, int k,
                                                          const float* __restrict__ rows) {
    const int j = blockIdx.x;
    if (j >= k) return;
    int64_t dst = idx[j];
    if (dst < 0 || dst >= n) return;
    float* o = y + dst * d;
    const float* s = rows + (int64_t)j * d;
    for (int e = threadIdx.x; e < d; e += blockDim.x) {
        if (ADD)
            o[e] += s[e];
        else
            o[e] = s[e];
    }
}

__global__ void fill_i32_kernel(int32_t* __restrict__ p, int64_t n, int32_t v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void slot_map_kernel(const int64_t* __restrict__ idx, int k, int64_t n, int32_t* __restrict__ map) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < k) {
        int64_t i = idx[j];
        if (i >= 0 && i < n) map[i] = j;
    }
}

// gather + slot map in ONE launch (three ~5 us launches otherwise: fill, scatter of the slots, gather).  Workgroups
// [0, k) copy the selected rows; the others build map[i] = j if idx[j] == i else -1 by searching the k indices held in LDS
// (k <= 2048 broadcast reads per thread: cheaper than a launch, and no fill-then-scatter ordering between kernels).
__global__ __launch_bounds__(WG) void gather_slot_map_kernel(const float* __restrict__ x, int64_t n, int d,
                                                             const int64_t* __restrict__ idx, int k,
                                                             float* __restrict__ xs, int32_t* __restrict__ map,
                                                             __bf16* __restrict__ xs16) {
    if ((int)blockIdx.x < k) {
        const int64_t src = idx[blockIdx.x];
        if (src < 0 || src >= n) return;
        const float* s = x + src * d;
        float* o = xs + (int64_t)blockIdx.x * d;
        for (int e = threadIdx.x; e < d; e += WG) {
            const float val = s[e];
            o[e] = val;
            if (xs16) xs16[(int64_t)blockIdx.x * d + e] = (__bf16)val;   // round to nearest even
        }
        return;
    }
    __shared__ int sel[2048];
    for (int j = threadIdx.x; j < k; j += WG) sel[j] = (int)idx[j];
    __syncthreads();
    const int64_t i = ((int64_t)blockIdx.x - k) * WG + threadIdx.x;
    if (i >= n) return;
    int slot = -1;
    const int me = (int)i;
    for (int j = 0; j < k; ++j) slot = (sel[j] == me) ? j : slot;
    map[i] = slot;
}

// ---------------------------------------------------------------------------------------------------------------
// K10 epilogue: h = act(h + bias)
// ---------------------------------------------------------------------------------------------------------------
// erf(x) by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 -- fp32 round-off class): 5 FMAs + one exp + one rcp, branch
// free, ~3x cheaper than libm's erff on the VALU (the GELU epilogue of a [N, 4D] hidden tensor is otherwise VALU-bound).
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
    const float r = fmaf(-p, e, 1.0f);
    return copysignf(r, x);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case SNF_ACT_RELU:
            return fmaxf(v, 0.f);
        case SNF_ACT_GELU:
            return 0.5f * v * (1.0f + fast_erf(v * 0.70710678118654752440f));
        case SNF_ACT_LEAKYRELU:
            return v > 0.f ? v : 0.01f * v;
        case SNF_ACT_SELU: {
            const float alpha = 1.6732632423543772848170429916717f, scale = 1.0507009873554804934193349852946f;
            return scale * (v > 0.f ? v : alpha * expm1f(v));
        }
        default:
            return v;
    }
}

__global__ __launch_bounds__(WG) void bias_act_f32_kernel(float* __restrict__ h, int64_t n, int f,
                                                          const float* __restrict__ bias, int act) {
    const int64_t total = n * (int64_t)f;
    if ((f & 3) == 0) {
        for (int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i4 < total;
             i4 += (int64_t)gridDim.x * blockDim.x * 4) {
            float4 v = *reinterpret_cast<float4*>(h + i4);
            int col = (int)(i4 % f);
            if (bias) {
                float4 bb = *reinterpret_cast<const float4*>(bias + col);
                v.x += bb.x;
                v.y += bb.y;
                v.z += bb.z;
                v.w += bb.w;
            }
            v.x = apply_act(v.x, act);
            v.y = apply_act(v.y, act);
            v.z = apply_act(v.z, act);
            v.w = apply_act(v.w, act);
            *reinterpret_cast<float4*>(h + i4) = v;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            float v = h[i] + (bias ? bias[i % f] : 0.f);
            h[i] = apply_act(v, act);
        }
    }
}

__global__ __launch_bounds__(WG) void bias_act_bf16_kernel(unsigned short* __restrict__ h, int64_t n, int f,
                                                           const float* __restrict__ bias, int act) {
    const int64_t total = n * (int64_t)f;
    if ((f & 7) == 0) {
        for (int64_t i8 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i8 < total;
             i8 += (int64_t)gridDim.x * blockDim.x * 8) {
            uint4 p = *reinterpret_cast<uint4*>(h + i8);
            unsigned int w[4] = {p.x, p.y, p.z, p.w};
            int col = (int)(i8 % f);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float lo = __uint_as_float(w[t] << 16), hi = __uint_as_float(w[t] & 0xffff0000u);
                if (bias) {
                    lo += bias[col + 2 * t];
                    hi += bias[col + 2 * t + 1];
                }
                w[t] = pack_bf16x2(apply_act(lo, act), apply_act(hi, act));
            }
            *reinterpret_cast<uint4*>(h + i8) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            float v = bf16_bits_to_f32(h[i]) + (bias ? bias[i % f] : 0.f);
            h[i] = f32_to_bf16_bits(apply_act(v, act));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K11 head: stage 1 -- per-workgroup column sums of the normalised rows; stage 2 -- fixed-order reduction, affine,
// mean, GEMV with the head.
// ---------------------------------------------------------------------------------------------------------------
template <int VEC, int NV>
__device__ __forceinline__ void add_row_bf16(const unsigned short* __restrict__ row, int d, int lane, float (&r)[NV * VEC]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int e = (i * 64 + lane) * VEC;
        if (e < d) {
            if constexpr (VEC == 4) {
                uint2 p = *reinterpret_cast<const uint2*>(row + e);
                r[i * 4 + 0] += __uint_as_float(p.x << 16);
                r[i * 4 + 1] += __uint_as_float(p.x & 0xffff0000u);
                r[i * 4 + 2] += __uint_as_float(p.y << 16);
                r[i * 4 + 3] += __uint_as_float(p.y & 0xffff0000u);
            } else {
                r[i] += bf16_bits_to_f32(row[e]);
            }
        }
    }
}

template <int VEC, int NV>
__global__ __launch_bounds__(WG) void ln_colsum_kernel(const float* __restrict__ z, int64_t n, int d, float eps,
                                                       const unsigned short* __restrict__ add_bf16,
                                                       const float* __restrict__ add_bias,
                                                       const int32_t* __restrict__ slot_map,
                                                       const float* __restrict__ delta_rows,
                                                       float* __restrict__ z_out,
                                                       float* __restrict__ partial /*[grid, d]*/,
                                                       const int* __restrict__ vl = nullptr, int vl_bags = 0) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int bid = blockIdx.x, gdim = gridDim.x;
    if (vl) {
        // varlen (many bags, one launch): this workgroup is workgroup (bid - wg0) of the bag's own launch of `parts` workgroups --
        // same rows, same order, so the bag's column sums are bit-identical to a per-bag launch.  Descriptor: wg0, row0, n, parts
        const int* __restrict__ dsc = vl + 4 * vl[4 * vl_bags + bid];
        const int64_t row0 = dsc[1];
        bid -= dsc[0];
        n = dsc[2];
        gdim = dsc[3];
        z += row0 * d;
        if (add_bf16) add_bf16 += row0 * d;
        if (slot_map) slot_map += row0;
        if (z_out) z_out += row0 * d;
    }
    float acc[NV * VEC];
#pragma unroll
    for (int i = 0; i < NV * VEC; ++i) acc[i] = 0.f;
    float bias_r[NV * VEC];
    if (add_bias) load_row<VEC, NV>(add_bias, d, lane, bias_r);
    const float inv_d = 1.0f / (float)d;
    // The pass is latency-bound (PMC: waves parked on vmcnt 79 % of their cycles, one row = one HBM round trip per trip):
    // the NEXT row's loads (f32 base + raw bf16 addend) are issued before the current row is reduced, so every wave keeps
    // two rows in flight.  Rows are consumed in the same order as before -> bit-identical sums.
    const int64_t rstride = (int64_t)gdim * WAVES;
    float nz[NV * VEC];     // prefetched base row
    uint2 nb[NV];           // prefetched raw bf16 addend (VEC == 4 only)
    auto issue = [&](int64_t row) __attribute__((always_inline)) {
        load_row<VEC, NV>(z + row * d, d, lane, nz);
        if constexpr (VEC == 4) {
            if (add_bf16) {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int e = (i * 64 + lane) * 4;
                    nb[i] = (e < d) ? *reinterpret_cast<const uint2*>(add_bf16 + row * d + e) : make_uint2(0u, 0u);
                }
            }
        }
    };
    int64_t row = (int64_t)bid * WAVES + wave;
    if (row < n) issue(row);
    for (; row < n; row += rstride) {
        float r[NV * VEC];
#pragma unroll
        for (int i = 0; i < NV * VEC; ++i) r[i] = nz[i];
        if constexpr (VEC == 4) {
            if (add_bf16) {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    r[i * 4 + 0] += __uint_as_float(nb[i].x << 16);
                    r[i * 4 + 1] += __uint_as_float(nb[i].x & 0xffff0000u);
                    r[i * 4 + 2] += __uint_as_float(nb[i].y << 16);
                    r[i * 4 + 3] += __uint_as_float(nb[i].y & 0xffff0000u);
                }
            }
        } else {
            if (add_bf16) add_row_bf16<VEC, NV>(add_bf16 + row * d, d, lane, r);
        }
        if (row + rstride < n) issue(row + rstride);   // wave-uniform
        if (add_bias) {
#pragma unroll
            for (int i = 0; i < NV * VEC; ++i) r[i] += bias_r[i];
        }
        if (slot_map) {
            int sl = slot_map[row];
            if (sl >= 0) {
                float dr[NV * VEC];
                load_row<VEC, NV>(delta_rows + (int64_t)sl * d, d, lane, dr);
#pragma unroll
                for (int i = 0; i < NV * VEC; ++i) r[i] += dr[i];
            }
        }
        if (z_out) store_row_f32<VEC, NV>(z_out + row * d, d, lane, r);
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < NV * VEC; ++i) s1 += r[i];
        const float mean = wave_sum(s1) * inv_d;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
#pragma unroll
            for (int t = 0; t < VEC; ++t) {
                int e = (i * 64 + lane) * VEC + t;
                float dv = (e < d) ? (r[i * VEC + t] - mean) : 0.f;
                s2 = fmaf(dv, dv, s2);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(s2) * inv_d + eps);
#pragma unroll
        for (int i = 0; i < NV * VEC; ++i) acc[i] += (r[i] - mean) * rstd;
    }
    // combine the 4 waves of the workgroup in a fixed order through LDS
    extern __shared__ float lds[];  // [WAVES][NV*VEC*64]
    float* mine = lds + wave * (NV * VEC * 64);
#pragma unroll
    for (int i = 0; i < NV * VEC; ++i) mine[i * 64 + lane] = acc[i];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
#pragma unroll
            for (int t = 0; t < VEC; ++t) {
                float s = 0.f;
                for (int w2 = 0; w2 < WAVES; ++w2) s += lds[w2 * (NV * VEC * 64) + (i * VEC + t) * 64 + lane];
                int e = (i * 64 + lane) * VEC + t;
                if (e < d) partial[(int64_t)blockIdx.x * d + e] = s;
            }
        }
    }
}

// stage 2a: second-level partial sums.  grid (ceil(d/64), NSLICE): workgroup (cx, sl) sums the partial rows
// p = sl, sl+NSLICE, ... of 64 columns; its 4 waves split those rows again and combine through LDS in a fixed order.
constexpr int HEAD_NSLICE = 16;
__global__ __launch_bounds__(256) void ln_colreduce_kernel(const float* __restrict__ partial, int nparts, int d,
                                                           float* __restrict__ partial2 /*[NSLICE, d]*/,
                                                           const int* __restrict__ vl = nullptr) {
    __shared__ float red[4][64];
    if (vl) {   // varlen: blockIdx.z = bag, its own partial rows and second-level slices
        const int* __restrict__ dsc = vl + 4 * blockIdx.z;
        partial += (int64_t)dsc[0] * d;
        nparts = dsc[3];
        partial2 += (int64_t)blockIdx.z * HEAD_NSLICE * d;
    }
    const int c = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + c;
    const int sl = blockIdx.y;
    float s = 0.f;
    if (e < d)
        for (int p = sl + HEAD_NSLICE * wv; p < nparts; p += HEAD_NSLICE * 4) s += partial[(int64_t)p * d + e];
    red[wv][c] = s;
    __syncthreads();
    if (wv == 0 && e < d) partial2[(int64_t)sl * d + e] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
}

// stage 2b: pooled[e] = gamma[e] * (sum_sl partial2[sl][e]) / n + beta[e];  logits[c] = w_head[c,:] . pooled + b_head[c]
__global__ __launch_bounds__(256) void head_gemv_kernel(const float* __restrict__ partial2, int64_t n, int d,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ w_head, const float* __restrict__ b_head,
                                                        float* __restrict__ pooled_out, float* __restrict__ logits,
                                                        const int* __restrict__ vl = nullptr) {
    __shared__ float red[256];
    if (vl) {   // varlen: blockIdx.y = bag
        partial2 += (int64_t)blockIdx.y * HEAD_NSLICE * d;
        n = vl[4 * blockIdx.y + 2];
        logits += (int64_t)blockIdx.y * gridDim.x;
        if (pooled_out) pooled_out += (int64_t)blockIdx.y * d;
    }
    const int c = blockIdx.x;
    const float inv_n = 1.0f / (float)n;
    float acc = 0.f;
    for (int e = threadIdx.x; e < d; e += 256) {
        float t = 0.f;
#pragma unroll
        for (int sl = 0; sl < HEAD_NSLICE; ++sl) t += partial2[(int64_t)sl * d + e];
        const float v = (gamma ? gamma[e] : 1.f) * (t * inv_n) + (beta ? beta[e] : 0.f);
        if (c == 0 && pooled_out) pooled_out[e] = v;
        acc = fmaf(w_head[(int64_t)c * d + e], v, acc);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) logits[c] = red[0] + (b_head ? b_head[c] : 0.f);
}

// ---------------------------------------------------------------------------------------------------------------
// host-side dispatch helpers
// ---------------------------------------------------------------------------------------------------------------
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// picks (VEC, NV) for a row width d; returns false if d is too wide for the register-resident row kernels
struct RowCfg {
    int vec, nv;
};
inline bool pick_row_cfg(int d, bool all_aligned, RowCfg* cfg) {
    if ((d & 3) == 0 && all_aligned) {
        int need = (d / 4 + 63) / 64;
        const int opts[] = {1, 2, 3, 4, 6, 8};
        for (int o : opts)
            if (o >= need) {
                *cfg = {4, o};
                return true;
            }
        return false;
    }
    int need = (d + 63) / 64;
    const int opts[] = {1, 2, 3, 4, 8, 16, 32};
    for (int o : opts)
        if (o >= need) {
            *cfg = {1, o};
            return true;
        }
    return false;
}

inline int row_grid(int64_t n) {
    int64_t want = (n + WAVES - 1) / WAVES;
    int64_t cap = (int64_t)snf::cu_count() * 8;
    return (int)(want < cap ? (want > 0 ? want : 1) : cap);
}

#define SNF_ROW_DISPATCH(cfg, ...)                                \
    do {                                                          \
        if (cfg.vec == 4) {                                       \
            switch (cfg.nv) {                                     \
                case 1: { constexpr int VEC = 4, NV = 1; __VA_ARGS__; } break; \
                case 2: { constexpr int VEC = 4, NV = 2; __VA_ARGS__; } break; \
                case 3: { constexpr int VEC = 4, NV = 3; __VA_ARGS__; } break; \
                case 4: { constexpr int VEC = 4, NV = 4; __VA_ARGS__; } break; \
                case 6: { constexpr int VEC = 4, NV = 6; __VA_ARGS__; } break; \
                default: { constexpr int VEC = 4, NV = 8; __VA_ARGS__; } break; \
            }                                                     \
        } else {                                                  \
            switch (cfg.nv) {                                     \
                case 1: { constexpr int VEC = 1, NV = 1; __VA_ARGS__; } break; \
                case 2: { constexpr int VEC = 1, NV = 2; __VA_ARGS__; } break; \
                case 3: { constexpr int VEC = 1, NV = 3; __VA_ARGS__; } break; \
                case 4: { constexpr int VEC = 1, NV = 4; __VA_ARGS__; } break; \
                case 8: { constexpr int VEC = 1, NV = 8; __VA_ARGS__; } break; \
                case 16: { constexpr int VEC = 1, NV = 16; __VA_ARGS__; } break; \
                default: { constexpr int VEC = 1, NV = 32; __VA_ARGS__; } break; \
            }                                                     \
        }                                                         \
    } while (0)

}  // namespace

namespace {
__global__ __launch_bounds__(256) void dropout_mask_kernel(snf::DropoutState st, int h, int64_t n, int k, float* __restrict__ mask,
                                                           int64_t groups) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    const int kg = (k + 3) >> 2;
    const int64_t rowi = g / kg;                      // a * n + row
    const int key0 = (int)(g - rowi * kg) * 4;
    snf::philox_f4 m = {1.f, 1.f, 1.f, 1.f};
    if (st.thresh) m = snf::dropout_mask4(st, (int)(rowi / n), n, rowi % n, k, key0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (key0 + e < k) mask[rowi * k + key0 + e] = m[e];
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// The loss head of a training step (reference train.py: SmallWeightTrainer._run_model -- max over the instance scores, two
// BCEWithLogits terms mixed by the single weight w, the bag prediction) as ONE launch forward and ONE backward instead of ~30
// scalar-sized ones:
//   m_c = max_n ins[n, c] (first index on ties)          bce(x, y) = weight_c ((1 - y) x + (1 + (pw_c - 1) y) (log1p(exp(-|x|)) + max(-x, 0)))
//   loss = w mean_c bce(logit_c, y_c) + (1 - w) mean_c bce(m_c, y_c)          bag_pred_c = (1 - w) sigmoid(m_c) + w sigmoid(logit_c)
// out = [loss, g_w, bag_pred[C], g_logit[C], g_max[C], m[C]] with g_* the gradients of loss (what the backward scales by grad_out).
// One workgroup: the scores are N x C floats (128 KiB at config B).  C <= 8.
constexpr int MIL_MAXC = 8;
__global__ __launch_bounds__(1024) void mil_loss_kernel(const float* __restrict__ ins, int64_t n, int c, const float* __restrict__ logits,
                                                        const float* __restrict__ label, const float* __restrict__ w,
                                                        const float* __restrict__ pos_weight, const float* __restrict__ weight,
                                                        float* __restrict__ out, int64_t* __restrict__ argmax) {
    __shared__ float s_val[MIL_MAXC][16];
    __shared__ long long s_idx[MIL_MAXC][16];
    float best[MIL_MAXC];
    long long bidx[MIL_MAXC];
#pragma unroll
    for (int j = 0; j < MIL_MAXC; ++j) best[j] = -INFINITY, bidx[j] = 0x7fffffffffffffffll;
    for (int64_t row = threadIdx.x; row < n; row += 1024) {
#pragma unroll
        for (int j = 0; j < MIL_MAXC; ++j)
            if (j < c) {
                const float v = ins[row * c + j];
                if (v > best[j] || (v == best[j] && row < bidx[j])) best[j] = v, bidx[j] = row;   // (NaN scores never win: documented)
            }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < MIL_MAXC; ++j) {
        if (j >= c) continue;
        float v = best[j];
        long long ix = bidx[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v, o, 64);
            const long long oi = __shfl_xor(ix, o, 64);
            if (ov > v || (ov == v && oi < ix)) v = ov, ix = oi;
        }
        if (lane == 0) s_val[j][wv] = v, s_idx[j][wv] = ix;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float wt = w[0];
    float lb = 0.f, lm = 0.f;
    for (int j = 0; j < c; ++j) {
        float v = s_val[j][0];
        long long ix = s_idx[j][0];
        for (int q = 1; q < 16; ++q)
            if (s_val[j][q] > v || (s_val[j][q] == v && s_idx[j][q] < ix)) v = s_val[j][q], ix = s_idx[j][q];
        if (n == 0) ix = 0;
        argmax[j] = ix;
        const float y = label[j], lw = 1.f + ((pos_weight ? pos_weight[j] : 1.f) - 1.f) * y, cw = weight ? weight[j] : 1.f;
        const float xs[2] = {logits[j], v};
        float l[2], g[2], sg[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float x = xs[t];
            const float sp = log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f);          // softplus(-x)
            l[t] = cw * ((1.f - y) * x + lw * sp);
            sg[t] = 1.f / (1.f + expf(-x));
            g[t] = cw * ((1.f - y) - lw * (1.f - sg[t]));                         // d bce / d x
        }
        lb += l[0], lm += l[1];
        out[2 + j] = (1.f - wt) * sg[1] + wt * sg[0];
        out[2 + c + j] = wt * g[0] / (float)c;
        out[2 + 2 * c + j] = (1.f - wt) * g[1] / (float)c;
        out[2 + 3 * c + j] = v;
    }
    lb /= (float)c, lm /= (float)c;
    out[0] = wt * lb + (1.f - wt) * lm;
    out[1] = lb - lm;
}

// backward: d_ins = 0 but grad_out g_max[c] at (argmax[c], c); small = [grad_out g_logit[C], grad_out g_w]
__global__ __launch_bounds__(256) void mil_loss_bwd_kernel(const float* __restrict__ grad_out, const float* __restrict__ fwd, const int64_t* __restrict__ argmax,
                                                           int64_t n, int c, float* __restrict__ d_ins, float* __restrict__ small) {
    const float go = grad_out[0];
    const int64_t total = n * c;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / c;
        const int j = (int)(i - row * c);
        d_ins[i] = row == argmax[j] ? go * fwd[2 + 2 * c + j] : 0.f;
    }
    if (blockIdx.x == 0 && threadIdx.x <= c) small[threadIdx.x] = go * (threadIdx.x < c ? fwd[2 + c + threadIdx.x] : fwd[1]);
}

extern "C" {

int snf_critic_f32(const float* x, int64_t n, int d, const float* w, const float* b, int c_out, float* scores,
                   float* colmax_val, int64_t* colmax_idx, snf_stream_t stream) {
    SNF_REQUIRE(x && w && scores, "snf_critic_f32: null pointer");
    SNF_REQUIRE(n >= 1 && d >= 1 && c_out >= 1, "snf_critic_f32: bad shape n=%lld d=%d c=%d", (long long)n, d, c_out);
    RowCfg cfg;
    SNF_REQUIRE(pick_row_cfg(d, aligned16(x) && aligned16(w), &cfg), "snf_critic_f32: d=%d too wide (max 2048)", d);
    hipStream_t s = snf::as_stream(stream);
    SNF_ROW_DISPATCH(cfg, hipLaunchKernelGGL((critic_kernel<VEC, NV, 0>), dim3(row_grid(n)), dim3(WG), 0, s, x, n, d, w,
                                              b, c_out, scores, 0.f, (unsigned short*)nullptr, (unsigned int*)nullptr));
    int rc = snf::check_launch("critic_kernel");
    if (rc) return rc;
    if (colmax_val || colmax_idx) {
        hipLaunchKernelGGL(colmax_kernel, dim3(c_out), dim3(1024), 0, s, scores, n, c_out, colmax_val, colmax_idx);
        rc = snf::check_launch("colmax_kernel");
    }
    return rc;
}

int snf_critic_ln_f32(const float* x, int64_t n, int d, const float* w, const float* b, int c_out, float* scores,
                      float eps, void* xhat_bf16, snf_stream_t stream) {
    SNF_REQUIRE(x && w && scores && xhat_bf16, "snf_critic_ln_f32: null pointer");
    SNF_REQUIRE(n >= 1 && d >= 1 && c_out >= 1, "snf_critic_ln_f32: bad shape n=%lld d=%d c=%d", (long long)n, d, c_out);
    RowCfg cfg;
    SNF_REQUIRE(pick_row_cfg(d, aligned16(x) && aligned16(w) && aligned16(xhat_bf16), &cfg),
                "snf_critic_ln_f32: d=%d too wide (max 2048)", d);
    hipStream_t s = snf::as_stream(stream);
    SNF_ROW_DISPATCH(cfg, hipLaunchKernelGGL((critic_kernel<VEC, NV, 1>), dim3(row_grid(n)), dim3(WG), 0, s, x, n, d, w,
                                              b, c_out, scores, eps, reinterpret_cast<unsigned short*>(xhat_bf16),
                                              (unsigned int*)nullptr));
    return snf::check_launch("critic_kernel<ln>");
}

int snf_critic_select_f32(const float* x, int64_t n, int d, const float* w, const float* b, float* scores, float eps,
                          void* xhat_bf16, void* selector_state, snf_stream_t stream) {
    SNF_REQUIRE(x && w && scores && selector_state, "snf_critic_select_f32: null pointer");
    SNF_REQUIRE(n >= 1 && d >= 1, "snf_critic_select_f32: bad shape n=%lld d=%d", (long long)n, d);
    SNF_REQUIRE(aligned16(selector_state), "snf_critic_select_f32: state must be 16-byte aligned");
    RowCfg cfg;
    SNF_REQUIRE(pick_row_cfg(d, aligned16(x) && aligned16(w) && (!xhat_bf16 || aligned16(xhat_bf16)), &cfg),
                "snf_critic_select_f32: d=%d too wide (max 2048)", d);
    hipStream_t s = snf::as_stream(stream);
    unsigned int* hist = reinterpret_cast<snf::SelectorState*>(selector_state)->hist;
    if (xhat_bf16) {
        SNF_ROW_DISPATCH(cfg, hipLaunchKernelGGL((critic_kernel<VEC, NV, 1>), dim3(row_grid(n)), dim3(WG), 0, s, x, n, d, w, b,
                                                  1, scores, eps, reinterpret_cast<unsigned short*>(xhat_bf16), hist));
    } else {
        SNF_ROW_DISPATCH(cfg, hipLaunchKernelGGL((critic_kernel<VEC, NV, 0>), dim3(row_grid(n)), dim3(WG), 0, s, x, n, d, w,
                                                  b, 1, scores, 0.f, (unsigned short*)nullptr, hist));
    }
    return snf::check_launch("critic_kernel<select>");
}

// critic + LayerNorm (with affine) of the same rows in ONE pass over the bag, the normalised rows leaving as the interleaved
// hi / lo image of the one-pass fp32-class GEMM (= snf_critic_f32 + snf_layernorm_rows_hl_f32, one HBM read of x instead of two).
// selector_state nullable: when given (one class), the first radix digit of the scores is counted as in snf_critic_select_f32.
int snf_critic_ln_hl_f32(const float* x, int64_t n, int d, const float* w, const float* b, int c_out, float* scores,
                         const float* gamma, const float* beta, float eps, void* out_hl, void* selector_state,
                         snf_stream_t stream) {
    SNF_REQUIRE(x && w && scores && out_hl && (!gamma == !beta), "snf_critic_ln_hl_f32: null pointer (gamma and beta: both or neither)");
    SNF_REQUIRE(n >= 1 && d >= 32 && d % 32 == 0 && c_out >= 1, "snf_critic_ln_hl_f32: bad shape n=%lld d=%d c=%d (d % 32 == 0)",
                (long long)n, d, c_out);
    SNF_REQUIRE(!selector_state || (c_out == 1 && aligned16(selector_state)), "snf_critic_ln_hl_f32: the selector needs one class "
                "and a 16-byte aligned state");
    SNF_REQUIRE(aligned16(x) && aligned16(w) && (!gamma || (aligned16(gamma) && aligned16(beta))) && aligned16(out_hl),
                "snf_critic_ln_hl_f32: buffers must be 16-byte aligned");
    RowCfg cfg;
    SNF_REQUIRE(pick_row_cfg(d, true, &cfg) && cfg.vec == 4, "snf_critic_ln_hl_f32: d=%d too wide (max 2048)", d);
    hipStream_t s = snf::as_stream(stream);
    unsigned int* hist = selector_state ? reinterpret_cast<snf::SelectorState*>(selector_state)->hist : nullptr;
    switch (cfg.nv) {
#define SNF_CRITIC_HL_CASE(N)                                                                                              \
    case N:                                                                                                                 \
        hipLaunchKernelGGL((critic_kernel<4, N, 2>), dim3(row_grid(n)), dim3(WG), 0, s, x, n, d, w, b, c_out, scores, eps,  \
                           reinterpret_cast<unsigned short*>(out_hl), hist, gamma, beta);                                   \
        break;
        SNF_CRITIC_HL_CASE(1)
        SNF_CRITIC_HL_CASE(2)
        SNF_CRITIC_HL_CASE(3)
        SNF_CRITIC_HL_CASE(4)
        SNF_CRITIC_HL_CASE(6)
        default:
            SNF_CRITIC_HL_CASE(8)
#undef SNF_CRITIC_HL_CASE
    }
    return snf::check_launch("critic_kernel<ln, hl>");
}

int snf_layernorm_rows_f32(const float* x, int64_t n, int d, const int32_t* slot_map, const float* patch_rows,
                           const float* gamma, const float* beta, float eps, float* out_f32, void* out_bf16,
                           float* mean, float* rstd, const int64_t* out_row_idx, snf_stream_t stream) {
    SNF_REQUIRE(x, "snf_layernorm_rows_f32: null x");
    SNF_REQUIRE(n >= 1 && d >= 1, "snf_layernorm_rows_f32: bad shape");
    SNF_REQUIRE(!slot_map || patch_rows, "snf_layernorm_rows_f32: slot_map without patch_rows");
    bool al = aligned16(x) && (!patch_rows || aligned16(patch_rows)) && (!gamma || aligned16(gamma)) &&
              (!beta || aligned16(beta)) && (!out_f32 || aligned16(out_f32)) && (!out_bf16 || aligned16(out_bf16));
    RowCfg cfg;
    SNF_REQUIRE(pick_row_cfg(d, al, &cfg), "snf_layernorm_rows_f32: d=%d too wide (max 2048)", d);
    hipStream_t s = snf::as_stream(stream);
    SNF_ROW_DISPATCH(cfg, hipLaunchKernelGGL((layernorm_rows_kernel<VEC, NV>), dim3(row_grid(n)), dim3(WG), 0, s, x, n, d,
                                              slot_map, patch_rows, gamma, beta, eps, out_f32,
                                              reinterpret_cast<unsigned short*>(out_bf16), mean, rstd, out_row_idx, 0));
    return snf::check_launch("layernorm_rows_kernel");
}

int snf_layernorm_rows_split3_f32(const float* x, int64_t n, int d, const int32_t* slot_map, const float* patch_rows,
                                  const float* gamma, const float* beta, float eps, void* out_bf16, snf_stream_t stream) {
    SNF_REQUIRE(x && out_bf16, "snf_layernorm_rows_split3_f32: null pointer");
    SNF_REQUIRE(n >= 1 && d >= 1, "snf_layernorm_rows_split3_f32: bad shape");
    SNF_REQUIRE(!slot_map || patch_rows, "snf_layernorm_rows_split3_f32: slot_map without patch_rows");
    bool al = aligned16(x) && (!patch_rows || aligned16(patch_rows)) && (!gamma || aligned16(gamma)) &&
              (!beta || aligned16(beta)) && aligned16(out_bf16);
    RowCfg cfg;
    SNF_REQUIRE(pick_row_cfg(d, al, &cfg), "snf_layernorm_rows_split3_f32: d=%d too wide (max 2048)", d);
    hipStream_t s = snf::as_stream(stream);
    SNF_ROW_DISPATCH(cfg, hipLaunchKernelGGL((layernorm_rows_kernel<VEC, NV>), dim3(row_grid(n)), dim3(WG), 0, s, x, n, d,
                                              slot_map, patch_rows, gamma, beta, eps, (float*)nullptr,
                                              reinterpret_cast<unsigned short*>(out_bf16), (float*)nullptr, (float*)nullptr,
                                              (const int64_t*)nullptr, 1));
    return snf::check_launch("layernorm_rows_kernel<split3>");
}

int snf_layernorm_bwd_blocks(int64_t n) { return row_grid(n); }

int snf_layernorm_rows_bwd_f32(const float* x, int64_t n, int d, const void* dy, int dy_dtype, int64_t dy_stride,
                               const float* gamma, float eps, const float* residual, float* dx, void* dx_bf16,
                               float* partials, snf_stream_t stream) {
    SNF_REQUIRE(x && dy, "snf_layernorm_rows_bwd_f32: null pointer");
    SNF_REQUIRE(n >= 1 && d >= 1, "snf_layernorm_rows_bwd_f32: bad shape");
    SNF_REQUIRE(dy_dtype == SNF_DT_F32 || dy_dtype == SNF_DT_BF16, "snf_layernorm_rows_bwd_f32: bad dy dtype %d", dy_dtype);
    SNF_REQUIRE(dy_stride == 0 || dy_stride >= d, "snf_layernorm_rows_bwd_f32: bad dy row pitch");
    bool al = aligned16(x) && aligned16(dy) && (!gamma || aligned16(gamma)) && (!residual || aligned16(residual)) &&
              (!dx || aligned16(dx)) && (!dx_bf16 || aligned16(dx_bf16)) && dy_stride % 4 == 0;
    RowCfg cfg;
    SNF_REQUIRE(pick_row_cfg(d, al, &cfg), "snf_layernorm_rows_bwd_f32: d=%d too wide (max 2048)", d);
    hipStream_t s = snf::as_stream(stream);
    SNF_ROW_DISPATCH(cfg, hipLaunchKernelGGL((ln_bwd_rows_kernel<VEC, NV>), dim3(row_grid(n)), dim3(WG), 0, s, x, n, d, dy,
                                              dy_dtype == SNF_DT_BF16 ? 1 : 0, dy_stride, gamma, eps, residual, dx,
                                              reinterpret_cast<unsigned short*>(dx_bf16), partials));
    return snf::check_launch("ln_bwd_rows_kernel");
}

int snf_colsum_blocks(int64_t n) {
    int64_t b = (n + 31) / 32;
    const int64_t cap = (int64_t)snf::cu_count() * 4;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

int snf_colsum_fused(const void* src, int src_dtype, int64_t n, int d, const float* row_weight, int64_t weight_stride,
                     const void* gate_bf16, void* dst_bf16, float* partial, snf_stream_t stream) {
    SNF_REQUIRE(src && partial, "snf_colsum_fused: null pointer");
    SNF_REQUIRE(n >= 1 && d >= 8 && d % 8 == 0 && d <= 8192, "snf_colsum_fused: bad shape n=%lld d=%d (d %% 8 == 0, d <= 8192)",
                (long long)n, d);
    SNF_REQUIRE(src_dtype == SNF_DT_F32 || src_dtype == SNF_DT_BF16, "snf_colsum_fused: bad dtype %d", src_dtype);
    SNF_REQUIRE(aligned16(src) && aligned16(partial) && (!gate_bf16 || aligned16(gate_bf16)) && (!dst_bf16 || aligned16(dst_bf16)),
                "snf_colsum_fused: buffers must be 16-byte aligned");
    const int blocks = snf_colsum_blocks(n);
    const int rpb = (int)((n + blocks - 1) / blocks);
    const int nch = (d + 2047) / 2048;
    hipStream_t s = snf::as_stream(stream);
    const unsigned short* g = reinterpret_cast<const unsigned short*>(gate_bf16);
    unsigned short* o = reinterpret_cast<unsigned short*>(dst_bf16);
#define SNF_COLSUM(BF, NC) \
    hipLaunchKernelGGL((colsum_fused_kernel<BF, NC>), dim3(blocks), dim3(256), 0, s, src, n, d, row_weight, weight_stride, g, o, \
                       partial, rpb)
    if (src_dtype == SNF_DT_BF16) {
        switch (nch) {
            case 1: SNF_COLSUM(true, 1); break;
            case 2: SNF_COLSUM(true, 2); break;
            default: SNF_COLSUM(true, 4); break;
        }
    } else {
        switch (nch) {
            case 1: SNF_COLSUM(false, 1); break;
            case 2: SNF_COLSUM(false, 2); break;
            default: SNF_COLSUM(false, 4); break;
        }
    }
#undef SNF_COLSUM
    return snf::check_launch("colsum_fused_kernel");
}

int snf_layernorm_rows_hl_f32(const float* x, int64_t n, int d, const int32_t* slot_map, const float* patch_rows,
                              const float* gamma, const float* beta, float eps, void* out_bf16, snf_stream_t stream) {
    SNF_REQUIRE(x && out_bf16, "snf_layernorm_rows_hl_f32: null pointer");
    SNF_REQUIRE(n >= 1 && d >= 32 && d % 32 == 0, "snf_layernorm_rows_hl_f32: d=%d must be a multiple of 32", d);
    SNF_REQUIRE(!slot_map || patch_rows, "snf_layernorm_rows_hl_f32: slot_map without patch_rows");
    SNF_REQUIRE(aligned16(x) && (!patch_rows || aligned16(patch_rows)) && (!gamma || aligned16(gamma)) && (!beta || aligned16(beta)) &&
                    aligned16(out_bf16), "snf_layernorm_rows_hl_f32: buffers must be 16-byte aligned");
    RowCfg cfg;
    SNF_REQUIRE(pick_row_cfg(d, true, &cfg) && cfg.vec == 4, "snf_layernorm_rows_hl_f32: d=%d too wide (max 2048)", d);
    hipStream_t s = snf::as_stream(stream);
    SNF_ROW_DISPATCH(cfg, hipLaunchKernelGGL((layernorm_rows_kernel<VEC, NV>), dim3(row_grid(n)), dim3(WG), 0, s, x, n, d,
                                              slot_map, patch_rows, gamma, beta, eps, (float*)nullptr,
                                              reinterpret_cast<unsigned short*>(out_bf16), (float*)nullptr, (float*)nullptr,
                                              (const int64_t*)nullptr, 2));
    return snf::check_launch("layernorm_rows_kernel<hl>");
}

int snf_layernorm_rows_hl_patch_f32(const float* x, const float* addend, int64_t n, int d, const int64_t* out_row_idx,
                                    const float* gamma, const float* beta, float eps, void* out_hl, snf_stream_t stream) {
    SNF_REQUIRE(x && out_hl && out_row_idx, "snf_layernorm_rows_hl_patch_f32: null pointer");
    SNF_REQUIRE(n >= 1 && d >= 32 && d % 32 == 0, "snf_layernorm_rows_hl_patch_f32: d=%d must be a multiple of 32", d);
    SNF_REQUIRE(aligned16(x) && (!addend || aligned16(addend)) && (!gamma || aligned16(gamma)) && (!beta || aligned16(beta)) &&
                    aligned16(out_hl), "snf_layernorm_rows_hl_patch_f32: buffers must be 16-byte aligned");
    RowCfg cfg;
    SNF_REQUIRE(pick_row_cfg(d, true, &cfg) && cfg.vec == 4, "snf_layernorm_rows_hl_patch_f32: d=%d too wide (max 2048)", d);
    hipStream_t s = snf::as_stream(stream);
    SNF_ROW_DISPATCH(cfg, hipLaunchKernelGGL((layernorm_rows_kernel<VEC, NV>), dim3(row_grid(n)), dim3(WG), 0, s, x, n, d,
                                              (const int32_t*)nullptr, (const float*)nullptr, gamma, beta, eps, (float*)nullptr,
                                              reinterpret_cast<unsigned short*>(out_hl), (float*)nullptr, (float*)nullptr,
                                              out_row_idx, 2, addend));
    return snf::check_launch("layernorm_rows_kernel<hl patch>");
}

int snf_split_hl_f32(const float* x, int64_t ldx, int64_t m, int k, void* out_bf16, snf_stream_t stream) {
    SNF_REQUIRE(x && out_bf16, "snf_split_hl_f32: null pointer");
    SNF_REQUIRE(m >= 1 && k >= 32 && k % 32 == 0 && ldx >= k && ldx % 4 == 0, "snf_split_hl_f32: bad shape m=%lld k=%d ldx=%lld",
                (long long)m, k, (long long)ldx);
    SNF_REQUIRE(aligned16(x) && aligned16(out_bf16), "snf_split_hl_f32: buffers must be 16-byte aligned");
    const int64_t total = m * (k >> 3);
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)snf::cu_count() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(split_hl_kernel, dim3((int)blocks), dim3(256), 0, snf::as_stream(stream), x, ldx, m, k,
                       reinterpret_cast<unsigned short*>(out_bf16));
    return snf::check_launch("split_hl_kernel");
}

int snf_split3_f32(const float* x, int64_t ldx, int64_t m, int k, void* out_bf16, snf_stream_t stream) {
    SNF_REQUIRE(x && out_bf16, "snf_split3_f32: null pointer");
    SNF_REQUIRE(m >= 1 && k >= 8 && k % 8 == 0 && ldx >= k && ldx % 4 == 0, "snf_split3_f32: bad shape m=%lld k=%d ldx=%lld",
                (long long)m, k, (long long)ldx);
    SNF_REQUIRE(aligned16(x) && aligned16(out_bf16), "snf_split3_f32: buffers must be 16-byte aligned");
    const int64_t total = m * (k >> 3);
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)snf::cu_count() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(split3_kernel, dim3((int)blocks), dim3(256), 0, snf::as_stream(stream), x, ldx, m, k,
                       reinterpret_cast<unsigned short*>(out_bf16));
    return snf::check_launch("split3_kernel");
}

int snf_split3_weight_f32(const float* w, int64_t ldw, int64_t m, int k, const float* colscale, void* out_bf16, snf_stream_t stream) {
    SNF_REQUIRE(w && out_bf16, "snf_split3_weight_f32: null pointer");
    SNF_REQUIRE(m >= 1 && k >= 8 && k % 8 == 0 && ldw >= k && ldw % 4 == 0, "snf_split3_weight_f32: bad shape m=%lld k=%d ldw=%lld",
                (long long)m, k, (long long)ldw);
    SNF_REQUIRE(aligned16(w) && aligned16(out_bf16) && (!colscale || aligned16(colscale)), "snf_split3_weight_f32: buffers must be 16-byte aligned");
    const int64_t total = m * (k >> 3);
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)snf::cu_count() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(split3_weight_kernel, dim3((int)blocks), dim3(256), 0, snf::as_stream(stream), w, ldw, m, k, colscale,
                       reinterpret_cast<unsigned short*>(out_bf16));
    return snf::check_launch("split3_weight_kernel");
}

int snf_split3_colsum_f32(const float* x, int64_t ldx, int64_t m, int k, const void* gate_bf16, int64_t ldg, void* out_bf16,
                          int64_t ldo, int64_t plane, float* partial, snf_stream_t stream) {
    SNF_REQUIRE(x && out_bf16, "snf_split3_colsum_f32: null pointer");
    SNF_REQUIRE(m >= 1 && k >= 8 && k % 8 == 0 && k <= 8192 && ldx >= k && ldx % 4 == 0 && plane >= k && plane % 8 == 0 &&
                    ldo >= 2 * plane + k && ldo % 8 == 0 && (!gate_bf16 || (ldg >= k && ldg % 8 == 0)),
                "snf_split3_colsum_f32: bad shape m=%lld k=%d ldx=%lld ldg=%lld ldo=%lld plane=%lld (k %% 8 == 0, k <= 8192, 16-byte rows)",
                (long long)m, k, (long long)ldx, (long long)ldg, (long long)ldo, (long long)plane);
    SNF_REQUIRE(aligned16(x) && aligned16(out_bf16) && (!gate_bf16 || aligned16(gate_bf16)) && (!partial || aligned16(partial)),
                "snf_split3_colsum_f32: buffers must be 16-byte aligned");
    const int blocks = snf_colsum_blocks(m);
    const int rpb = (int)((m + blocks - 1) / blocks);
    const int nch = (k + 2047) / 2048;
    hipStream_t s = snf::as_stream(stream);
    const unsigned short* g = reinterpret_cast<const unsigned short*>(gate_bf16);
    unsigned short* o = reinterpret_cast<unsigned short*>(out_bf16);
    switch (nch) {
        case 1: hipLaunchKernelGGL(split3_colsum_kernel<1>, dim3(blocks), dim3(256), 0, s, x, ldx, m, k, g, ldg, o, ldo, plane, partial, rpb); break;
        case 2: hipLaunchKernelGGL(split3_colsum_kernel<2>, dim3(blocks), dim3(256), 0, s, x, ldx, m, k, g, ldg, o, ldo, plane, partial, rpb); break;
        default: hipLaunchKernelGGL(split3_colsum_kernel<4>, dim3(blocks), dim3(256), 0, s, x, ldx, m, k, g, ldg, o, ldo, plane, partial, rpb); break;
    }
    return snf::check_launch("split3_colsum_kernel");
}

int snf_split_hl_colsum_f32(const float* x, int64_t ldx, int64_t m, int k, const void* gate_hl, int64_t ldg, void* out_hl, int64_t ldo,
                            float* partial, snf_stream_t stream) {
    SNF_REQUIRE(x && out_hl, "snf_split_hl_colsum_f32: null pointer");
    SNF_REQUIRE(m >= 1 && k >= 32 && k % 32 == 0 && k <= 8192 && ldx >= k && ldx % 4 == 0 && ldo >= 2 * (int64_t)k && ldo % 8 == 0 &&
                    (!gate_hl || (ldg >= 2 * (int64_t)k && ldg % 8 == 0)),
                "snf_split_hl_colsum_f32: bad shape m=%lld k=%d ldx=%lld ldg=%lld ldo=%lld (k %% 32 == 0, k <= 8192, images of 2 k columns, 16-byte rows)",
                (long long)m, k, (long long)ldx, (long long)ldg, (long long)ldo);
    SNF_REQUIRE(aligned16(x) && aligned16(out_hl) && (!gate_hl || aligned16(gate_hl)) && (!partial || aligned16(partial)),
                "snf_split_hl_colsum_f32: buffers must be 16-byte aligned");
    const int blocks = snf_colsum_blocks(m);
    const int rpb = (int)((m + blocks - 1) / blocks);
    const int nch = (k + 2047) / 2048;
    hipStream_t s = snf::as_stream(stream);
    const unsigned short* g = reinterpret_cast<const unsigned short*>(gate_hl);
    unsigned short* o = reinterpret_cast<unsigned short*>(out_hl);
    switch (nch) {
        case 1: hipLaunchKernelGGL((split3_colsum_kernel<1, true>), dim3(blocks), dim3(256), 0, s, x, ldx, m, k, g, ldg, o, ldo, (int64_t)0, partial, rpb); break;
        case 2: hipLaunchKernelGGL((split3_colsum_kernel<2, true>), dim3(blocks), dim3(256), 0, s, x, ldx, m, k, g, ldg, o, ldo, (int64_t)0, partial, rpb); break;
        default: hipLaunchKernelGGL((split3_colsum_kernel<4, true>), dim3(blocks), dim3(256), 0, s, x, ldx, m, k, g, ldg, o, ldo, (int64_t)0, partial, rpb); break;
    }
    return snf::check_launch("split3_colsum_kernel<hl>");
}

int snf_fold_blocks(int r) {
    int b = (r + WAVES - 1) / WAVES;
    const int cap = snf::cu_count() * 2;
    return b < 1 ? 1 : (b > cap ? cap : b);
}

int snf_fold_linear_f32(const float* w, int r, int c, const float* gamma, const float* beta, const float* bias, void* wf_bf16,
                        int64_t ldwf, float* bf_f32, void* bf_bf16, snf_stream_t stream) {
    SNF_REQUIRE(w && gamma && beta && wf_bf16, "snf_fold_linear_f32: null pointer");
    SNF_REQUIRE(r >= 1 && c >= 1 && ldwf >= c, "snf_fold_linear_f32: bad shape");
    const bool al = aligned16(w) && aligned16(gamma) && aligned16(beta) && aligned16(wf_bf16) && ldwf % 8 == 0;
    RowCfg cfg;
    SNF_REQUIRE(pick_row_cfg(c, al, &cfg), "snf_fold_linear_f32: c=%d too wide (max 2048)", c);
    hipStream_t s = snf::as_stream(stream);
    SNF_ROW_DISPATCH(cfg, hipLaunchKernelGGL((fold_linear_kernel<VEC, NV>), dim3(snf_fold_blocks(r)), dim3(WG), 0, s, w, r, c, gamma,
                                              beta, bias, reinterpret_cast<unsigned short*>(wf_bf16), ldwf, bf_f32,
                                              reinterpret_cast<unsigned short*>(bf_bf16)));
    return snf::check_launch("fold_linear_kernel");
}

int snf_unfold_linear_f32(const float* dwf, const float* w, int r, int c, const float* gamma, const float* beta, const float* dbf,
                          float* dw, float* partial, snf_stream_t stream) {
    SNF_REQUIRE(dwf && w && gamma && beta && dbf && dw && partial, "snf_unfold_linear_f32: null pointer");
    SNF_REQUIRE(r >= 1 && c >= 1, "snf_unfold_linear_f32: bad shape");
    const bool al = aligned16(dwf) && aligned16(w) && aligned16(gamma) && aligned16(beta) && aligned16(dw);
    RowCfg cfg;
    SNF_REQUIRE(pick_row_cfg(c, al, &cfg), "snf_unfold_linear_f32: c=%d too wide (max 2048)", c);
    hipStream_t s = snf::as_stream(stream);
    SNF_ROW_DISPATCH(cfg, hipLaunchKernelGGL((unfold_linear_kernel<VEC, NV>), dim3(snf_fold_blocks(r)), dim3(WG), 0, s, dwf, w, r, c,
                                              gamma, beta, dbf, dw, partial));
    return snf::check_launch("unfold_linear_kernel");
}

int snf_gather_rows_f32(const float* x, int64_t n, int d, const int64_t* idx, int k, float* out, snf_stream_t stream) {
    SNF_REQUIRE(x && idx && out, "snf_gather_rows_f32: null pointer");
    SNF_REQUIRE(n >= 1 && d >= 1 && k >= 0, "snf_gather_rows_f32: bad shape");
    if (k == 0) return SNF_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(k), dim3(WG), 0, snf::as_stream(stream), x, n, d, idx, k, out);
    return snf::check_launch("gather_rows_kernel");
}

int snf_scatter_rows_f32(const float* x, int64_t n, int d, const int64_t* idx, int k, const float* rows, float* y,
                         snf_stream_t stream) {
    SNF_REQUIRE(x && y && (k == 0 || (idx && rows)), "snf_scatter_rows_f32: null pointer");
    SNF_REQUIRE(n >= 1 && d >= 1 && k >= 0, "snf_scatter_rows_f32: bad shape");
    hipStream_t s = snf::as_stream(stream);
    if (x != y) {
        const int64_t total = n * (int64_t)d;
        if (aligned16(x) && aligned16(y)) {
            int64_t want = (total / 4 + WG - 1) / WG;
            int grid = (int)(want < (int64_t)snf::cu_count() * 8 ? (want > 0 ? want : 1) : (int64_t)snf::cu_count() * 8);
            hipLaunchKernelGGL(copy_f32_kernel, dim3(grid), dim3(WG), 0, s, x, y, total);
            int rc = snf::check_launch("copy_f32_kernel");
            if (rc) return rc;
        } else {
            if (hipMemcpyAsync(y, x, total * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
                snf::set_error("snf_scatter_rows_f32: hipMemcpyAsync failed");
                return SNF_ELAUNCH;
            }
        }
    }
    if (k == 0) return SNF_OK;
    hipLaunchKernelGGL(scatter_rows_kernel<false>, dim3(k), dim3(WG), 0, s, y, n, d, idx, k, rows);
    return snf::check_launch("scatter_rows_kernel");
}

int snf_scatter_add_rows_f32(float* z, int64_t n, int d, const int64_t* idx, int k, const float* delta,
                             snf_stream_t stream) {
    SNF_REQUIRE(z && (k == 0 || (idx && delta)), "snf_scatter_add_rows_f32: null pointer");
    SNF_REQUIRE(n >= 1 && d >= 1 && k >= 0, "snf_scatter_add_rows_f32: bad shape");
    if (k == 0) return SNF_OK;
    hipLaunchKernelGGL(scatter_rows_kernel<true>, dim3(k), dim3(WG), 0, snf::as_stream(stream), z, n, d, idx, k, delta);
    return snf::check_launch("scatter_add_rows_kernel");
}

int snf_slot_map_i32(const int64_t* idx, int k, int64_t n, int32_t* map, snf_stream_t stream) {
    SNF_REQUIRE(map && (k == 0 || idx), "snf_slot_map_i32: null pointer");
    SNF_REQUIRE(n >= 1 && k >= 0, "snf_slot_map_i32: bad shape");
    hipStream_t s = snf::as_stream(stream);
    int64_t want = (n + 255) / 256;
    int grid = (int)(want < 2048 ? want : 2048);
    hipLaunchKernelGGL(fill_i32_kernel, dim3(grid), dim3(256), 0, s, map, n, -1);
    int rc = snf::check_launch("fill_i32_kernel");
    if (rc || k == 0) return rc;
    hipLaunchKernelGGL(slot_map_kernel, dim3((k + 255) / 256), dim3(256), 0, s, idx, k, n, map);
    return snf::check_launch("slot_map_kernel");
}

int snf_gather_slot_map_f32(const float* x, int64_t n, int d, const int64_t* idx, int k, float* xs, int32_t* map,
                            void* xs_bf16, snf_stream_t stream) {
    SNF_REQUIRE(x && map && (k == 0 || (idx && xs)), "snf_gather_slot_map_f32: null pointer");
    SNF_REQUIRE(n >= 1 && n < 0x7fffffffll && d >= 1 && k >= 0 && k <= 2048,
                "snf_gather_slot_map_f32: bad shape n=%lld d=%d k=%d (k <= 2048)", (long long)n, d, k);
    const int64_t grid = k + (n + WG - 1) / WG;
    hipLaunchKernelGGL(gather_slot_map_kernel, dim3((unsigned)grid), dim3(WG), 0, snf::as_stream(stream), x, n, d, idx, k, xs,
                       map, reinterpret_cast<__bf16*>(xs_bf16));
    return snf::check_launch("gather_slot_map_kernel");
}

int snf_bias_act(void* h, int dtype, int64_t n, int f, const float* bias, int act, snf_stream_t stream) {
    SNF_REQUIRE(h, "snf_bias_act: null h");
    SNF_REQUIRE(n >= 1 && f >= 1, "snf_bias_act: bad shape");
    SNF_REQUIRE(act >= SNF_ACT_RELU && act <= SNF_ACT_NONE, "snf_bias_act: unknown activation %d", act);
    SNF_REQUIRE(dtype == SNF_DT_F32 || dtype == SNF_DT_BF16, "snf_bias_act: unknown dtype %d", dtype);
    SNF_REQUIRE(aligned16(h) && (!bias || aligned16(bias)), "snf_bias_act: buffers must be 16-byte aligned");
    const int64_t total = n * (int64_t)f;
    const int per = dtype == SNF_DT_F32 ? 4 : 8;
    int64_t want = (total / per + WG - 1) / WG;
    int64_t cap = (int64_t)snf::cu_count() * 8;
    int grid = (int)(want < cap ? (want > 0 ? want : 1) : cap);
    hipStream_t s = snf::as_stream(stream);
    if (dtype == SNF_DT_F32)
        hipLaunchKernelGGL(bias_act_f32_kernel, dim3(grid), dim3(WG), 0, s, reinterpret_cast<float*>(h), n, f, bias, act);
    else
        hipLaunchKernelGGL(bias_act_bf16_kernel, dim3(grid), dim3(WG), 0, s, reinterpret_cast<unsigned short*>(h), n, f,
                           bias, act);
    return snf::check_launch("bias_act_kernel");
}

static int ln_head_parts(int64_t n) {
    int64_t want = (n + WAVES - 1) / WAVES;
    int64_t cap = (int64_t)snf::cu_count() * 4;
    return (int)(want < cap ? (want > 0 ? want : 1) : cap);
}

size_t snf_ln_mean_head_workspace_bytes(int d) {
    if (d < 1) return 0;
    // worst-case number of partial rows (cu_count*4) + the second-level partial rows
    return ((size_t)snf::cu_count() * 4 + HEAD_NSLICE) * (size_t)d * sizeof(float);
}

int snf_ln_mean_head_f32(const float* z, int64_t n, int d, const void* add_bf16, const float* add_bias,
                         const int32_t* slot_map, const float* delta_rows, float* z_out, const float* gamma,
                         const float* beta, float eps, const float* w_head, const float* b_head, int c_out, float* logits,
                         float* pooled, void* workspace, size_t workspace_bytes, snf_stream_t stream) {
    SNF_REQUIRE(z && w_head && logits && workspace, "snf_ln_mean_head_f32: null pointer");
    SNF_REQUIRE(!slot_map || delta_rows, "snf_ln_mean_head_f32: slot_map without delta_rows");
    SNF_REQUIRE(n >= 1 && d >= 1 && c_out >= 1, "snf_ln_mean_head_f32: bad shape");
    const int parts = ln_head_parts(n);
    if (workspace_bytes < ((size_t)parts + HEAD_NSLICE) * d * sizeof(float)) {
        snf::set_error("snf_ln_mean_head_f32: workspace %zu < %zu", workspace_bytes,
                       ((size_t)parts + HEAD_NSLICE) * d * sizeof(float));
        return SNF_EWORKSPACE;
    }
    RowCfg cfg;
    const bool al = aligned16(z) && (!add_bf16 || aligned16(add_bf16)) && (!add_bias || aligned16(add_bias)) &&
                    (!delta_rows || aligned16(delta_rows)) && (!z_out || aligned16(z_out));
    SNF_REQUIRE(pick_row_cfg(d, al, &cfg), "snf_ln_mean_head_f32: d=%d too wide (max 2048)", d);
    hipStream_t s = snf::as_stream(stream);
    float* partial = reinterpret_cast<float*>(workspace);
    float* partial2 = partial + (size_t)parts * d;
    SNF_ROW_DISPATCH(cfg, hipLaunchKernelGGL((ln_colsum_kernel<VEC, NV>), dim3(parts), dim3(WG),
                                              WAVES * NV * VEC * 64 * sizeof(float), s, z, n, d, eps,
                                              reinterpret_cast<const unsigned short*>(add_bf16), add_bias, slot_map,
                                              delta_rows, z_out, partial));
    int rc = snf::check_launch("ln_colsum_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(ln_colreduce_kernel, dim3((d + 63) / 64, HEAD_NSLICE), dim3(256), 0, s, partial, parts, d, partial2);
    rc = snf::check_launch("ln_colreduce_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(head_gemv_kernel, dim3(c_out), dim3(256), 0, s, partial2, n, d, gamma, beta, w_head, b_head, pooled,
                       logits);
    return snf::check_launch("head_gemv_kernel");
}

// ---- varlen head: logits of every bag of a packed residual stream in three launches (instead of three per bag) --------------
// Plan: offsets [bags + 1] in HOST memory; table (host, may be null to size it) = [bags][4] (wg0, row0, n, parts) + the bag of
// every stage-1 workgroup.  Every bag keeps the partition of its own launch, so its logits are bit-identical to a per-bag call.
int snf_ln_mean_head_varlen_plan(const int64_t* offsets, int bags, int d, int32_t* table, size_t table_ints,
                                 size_t* table_ints_needed, size_t* workspace_bytes) {
    SNF_REQUIRE(offsets && bags >= 1 && d >= 1, "snf_ln_mean_head_varlen_plan: bad arguments");
    int64_t total = 0;
    for (int b = 0; b < bags; ++b) {
        const int64_t n = offsets[b + 1] - offsets[b];
        SNF_REQUIRE(n >= 1 && offsets[b + 1] < 0x7fffffffll, "snf_ln_mean_head_varlen_plan: bag %d has %lld rows", b, (long long)n);
        total += ln_head_parts(n);
    }
    SNF_REQUIRE(total < 0x3fffffff, "snf_ln_mean_head_varlen_plan: too many workgroups");
    const size_t need = (size_t)4 * bags + (size_t)total;
    if (table_ints_needed) *table_ints_needed = need;
    if (workspace_bytes) *workspace_bytes = ((size_t)total + (size_t)bags * HEAD_NSLICE) * d * sizeof(float);
    if (table) {
        SNF_REQUIRE(table_ints >= need, "snf_ln_mean_head_varlen_plan: table %zu < %zu ints", table_ints, need);
        int64_t wg = 0;
        for (int b = 0; b < bags; ++b) {
            const int64_t n = offsets[b + 1] - offsets[b];
            const int parts = ln_head_parts(n);
            table[4 * b + 0] = (int32_t)wg, table[4 * b + 1] = (int32_t)offsets[b], table[4 * b + 2] = (int32_t)n;
            table[4 * b + 3] = parts;
            for (int i = 0; i < parts; ++i) table[(size_t)4 * bags + wg + i] = b;
            wg += parts;
        }
    }
    return SNF_OK;
}

// z [T, d] packed rows (same optional addends as snf_ln_mean_head_f32, slot_map / delta_rows in packed coordinates);
// logits [bags, c_out], pooled [bags, d] or null
int snf_ln_mean_head_varlen_f32(const float* z, const int64_t* offsets, int bags, int d, const void* add_bf16,
                                const float* add_bias, const int32_t* slot_map, const float* delta_rows, float* z_out,
                                const float* gamma, const float* beta, float eps, const float* w_head, const float* b_head,
                                int c_out, float* logits, float* pooled, const int32_t* table_dev, void* workspace,
                                size_t workspace_bytes, snf_stream_t stream) {
    SNF_REQUIRE(z && offsets && w_head && logits && workspace && table_dev, "snf_ln_mean_head_varlen_f32: null pointer");
    SNF_REQUIRE(!slot_map || delta_rows, "snf_ln_mean_head_varlen_f32: slot_map without delta_rows");
    SNF_REQUIRE(bags >= 1 && d >= 1 && c_out >= 1, "snf_ln_mean_head_varlen_f32: bad shape");
    int64_t total = 0;
    for (int b = 0; b < bags; ++b) {
        SNF_REQUIRE(offsets[b + 1] > offsets[b], "snf_ln_mean_head_varlen_f32: empty bag %d", b);
        total += ln_head_parts(offsets[b + 1] - offsets[b]);
    }
    const size_t need = ((size_t)total + (size_t)bags * HEAD_NSLICE) * d * sizeof(float);
    if (workspace_bytes < need) {
        snf::set_error("snf_ln_mean_head_varlen_f32: workspace %zu < %zu", workspace_bytes, need);
        return SNF_EWORKSPACE;
    }
    RowCfg cfg;
    // packed rows start at row0 * d elements: 16-byte alignment of every bag's first row needs d % 4 == 0 (pick_row_cfg checks it)
    const bool al = aligned16(z) && (!add_bf16 || aligned16(add_bf16)) && (!add_bias || aligned16(add_bias)) &&
                    (!delta_rows || aligned16(delta_rows)) && (!z_out || aligned16(z_out));
    SNF_REQUIRE(pick_row_cfg(d, al, &cfg), "snf_ln_mean_head_varlen_f32: d=%d too wide (max 2048)", d);
    hipStream_t s = snf::as_stream(stream);
    float* partial = reinterpret_cast<float*>(workspace);
    float* partial2 = partial + (size_t)total * d;
    const int64_t n_all = offsets[bags];
    SNF_ROW_DISPATCH(cfg, hipLaunchKernelGGL((ln_colsum_kernel<VEC, NV>), dim3((unsigned)total), dim3(WG),
                                              WAVES * NV * VEC * 64 * sizeof(float), s, z, n_all, d, eps,
                                              reinterpret_cast<const unsigned short*>(add_bf16), add_bias, slot_map,
                                              delta_rows, z_out, partial, table_dev, bags));
    int rc = snf::check_launch("ln_colsum_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(ln_colreduce_kernel, dim3((d + 63) / 64, HEAD_NSLICE, bags), dim3(256), 0, s, partial, 0, d, partial2,
                       table_dev);
    rc = snf::check_launch("ln_colreduce_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(head_gemv_kernel, dim3(c_out, bags), dim3(256), 0, s, partial2, n_all, d, gamma, beta, w_head, b_head,
                       pooled, logits, table_dev);
    return snf::check_launch("head_gemv_kernel");
}

// The dropout keep-mask of the attention kernels as a tensor: mask[a, row, key] = 0 or 1 / (1 - p) (philox.h).  For the exact
// fp32 training path (which materialises P anyway) and for tests; the MFMA kernels never store it.
int snf_dropout_mask_f32(float dropout_p, uint64_t seed, uint64_t offset, int h, int64_t n, int k, float* mask,
                         snf_stream_t stream) {
    SNF_REQUIRE(mask && h >= 1 && n >= 1 && k >= 1, "snf_dropout_mask_f32: bad arguments");
    SNF_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "snf_dropout_mask_f32: dropout_p=%f outside [0, 1)", dropout_p);
    const snf::DropoutState st = snf::make_dropout(dropout_p, seed, offset);
    const int64_t groups = (int64_t)h * n * ((k + 3) >> 2);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, snf::as_stream(stream), st, h, n, k,
                       mask, groups);
    return snf::check_launch("dropout_mask_kernel");
}

int snf_mil_loss_f32(const float* ins, int64_t n, int c, const float* logits, const float* label, const float* w, const float* pos_weight,
                     const float* weight, float* out, int64_t* argmax, snf_stream_t stream) {
    SNF_REQUIRE(ins && logits && label && w && out && argmax, "snf_mil_loss_f32: null pointer");
    SNF_REQUIRE(n >= 1 && c >= 1 && c <= MIL_MAXC, "snf_mil_loss_f32: n=%lld c=%d (1 <= c <= %d)", (long long)n, c, MIL_MAXC);
    hipLaunchKernelGGL(mil_loss_kernel, dim3(1), dim3(1024), 0, snf::as_stream(stream), ins, n, c, logits, label, w, pos_weight, weight, out, argmax);
    return snf::check_launch("mil_loss_kernel");
}

int snf_mil_loss_bwd_f32(const float* grad_out, const float* fwd_out, const int64_t* argmax, int64_t n, int c, float* d_ins, float* d_small,
                         snf_stream_t stream) {
    SNF_REQUIRE(grad_out && fwd_out && argmax && d_ins && d_small, "snf_mil_loss_bwd_f32: null pointer");
    SNF_REQUIRE(n >= 1 && c >= 1 && c <= MIL_MAXC, "snf_mil_loss_bwd_f32: n=%lld c=%d (1 <= c <= %d)", (long long)n, c, MIL_MAXC);
    int64_t blocks = (n * c + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(mil_loss_bwd_kernel, dim3((int)blocks), dim3(256), 0, snf::as_stream(stream), grad_out, fwd_out, argmax, n, c, d_ins, d_small);
    return snf::check_launch("mil_loss_bwd_kernel");
}

}  // extern "C"
