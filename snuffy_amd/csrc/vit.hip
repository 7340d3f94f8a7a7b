// ViT patch-embedding extractor kernels (compute_feats.py path; DINO / MAE ViT with adapters).
//
//   snf_vit_patchify        16x16 (or any) patch unfold of the image batch into the im2col matrix of PatchEmbed's conv
//   snf_vit_assemble_tokens [cls ; patch embeddings] + pos_embed  -> fp32 token matrix
//   snf_vit_residual_ln     x += a1 + s2*a2 ; LayerNorm(x) -> bf16 ; optional bf16 copy of x   (Block residuals + next norm)
//   snf_vit_attention_f32   exact fp32 multi-head self-attention, any T / dk <= 128 (parity path, returns attn on request)
//   snf_vit_attention_mfma  bf16 MFMA self-attention for dk = 64 (ViT-S/B at 224/16: T = 197; keys in LDS chunks above T = 256):
//                           S^T = K Q^T  (v_mfma_f32_32x32x16_bf16, A = K fragment from LDS, B = Q fragment from HBM):
//                           queries land on lanes, keys in registers -> the row softmax is lane-local (+1 cross-half step),
//                           and P^T in C layout is directly the B operand of  O^T = V^T P^T  (A = V^T fragment from an LDS
//                           image transposed while staging).  One workgroup per (image, head); K and V^T stay in LDS.
//   snf_vit_attention_x3_f32 the same program in the fp32-class arithmetic (split-bf16 x3 products), fp32 in / fp32 out
#include <math.h>

#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// patchify: cols[(b*gh + gy)*gw + gx][(c*ps + i)*ps + j] = img[b][c][gy*ps + i][gx*ps + j]
// one thread per (patch row-segment): ps consecutive pixels of one image row
// ---------------------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, int B, int C, int H, int W, int ps,
                                                       void* __restrict__ cols) {
    const int gh = H / ps, gw = W / ps;
    const int64_t total = (int64_t)B * C * gh * ps * gw;  // number of ps-long segments
    const int kdim = C * ps * ps;
    // every segment starts on a 16-byte boundary in both tensors when ps and W are multiples of 4 and the bases are aligned
    const bool vec = (ps & 3) == 0 && (W & 3) == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(cols) & 15) == 0;
    if (vec) {
        // one thread per 4 pixels, consecutive threads on consecutive pixels of an image row: fully coalesced 16-byte loads, and the
        // ps / 4 threads of a segment write its 2 ps (bf16) or 4 ps (f32) bytes side by side
        const int qps = ps >> 2;
        const int64_t quads = total * qps;
        for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (int64_t)gridDim.x * blockDim.x) {
            const int j = (int)(q % qps) * 4;
            const int64_t s = q / qps;
            const int gx = (int)(s % gw);
            int64_t t = s / gw;
            const int i = (int)(t % ps);
            t /= ps;
            const int gy = (int)(t % gh);
            t /= gh;
            const int c = (int)(t % C);
            const int b = (int)(t / C);
            const float4 v = *reinterpret_cast<const float4*>(img + (((int64_t)b * C + c) * H + (gy * ps + i)) * W + gx * ps + j);
            const int64_t off = (((int64_t)b * gh + gy) * gw + gx) * kdim + (c * ps + i) * ps + j;
            if (BF16) {
                uint2 o;
                o.x = (unsigned)f32_to_bf16_bits(v.x) | ((unsigned)f32_to_bf16_bits(v.y) << 16);
                o.y = (unsigned)f32_to_bf16_bits(v.z) | ((unsigned)f32_to_bf16_bits(v.w) << 16);
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(cols) + off) = o;
            } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(cols) + off) = v;
            }
        }
        return;
    }
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += (int64_t)gridDim.x * blockDim.x) {
        // decompose with gx fastest: neighbouring threads read neighbouring segments of one image row (coalesced)
        int gx = (int)(s % gw);
        int64_t t = s / gw;
        int i = (int)(t % ps);
        t /= ps;
        int gy = (int)(t % gh);
        t /= gh;
        int c = (int)(t % C);
        int b = (int)(t / C);
        const float* src = img + (((int64_t)b * C + c) * H + (gy * ps + i)) * W + gx * ps;
        const int64_t row = ((int64_t)b * gh + gy) * gw + gx;
        const int64_t off = row * kdim + (c * ps + i) * ps;
        if (BF16) {
            unsigned short* dst = reinterpret_cast<unsigned short*>(cols) + off;
            for (int j = 0; j < ps; ++j) dst[j] = f32_to_bf16_bits(src[j]);
        } else {
            float* dst = reinterpret_cast<float*>(cols) + off;
            for (int j = 0; j < ps; ++j) dst[j] = src[j];
        }
    }
}

// tokens[b][0] = cls + pos[0]; tokens[b][1+p] = pe[b*P + p] + pos[1+p]
template <bool BF16>
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const void* __restrict__ pe, const float* __restrict__ cls,
                                                              const float* __restrict__ pos, int B, int P, int D,
                                                              float* __restrict__ tokens) {
    const int T = P + 1;
    const int64_t total = (int64_t)B * T * D;
    if ((D & 3) == 0 && ((reinterpret_cast<uintptr_t>(pe) | reinterpret_cast<uintptr_t>(cls) | reinterpret_cast<uintptr_t>(pos) |
                          reinterpret_cast<uintptr_t>(tokens)) & 15) == 0) {
        // four features per thread: 16-byte loads / stores (same values as the scalar form below)
        const int dq = D >> 2;
        const int64_t quads = total >> 2;
        for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (int64_t)gridDim.x * blockDim.x) {
            const int d = (int)(q % dq) * 4;
            const int64_t r = q / dq;
            const int t = (int)(r % T);
            const int b = (int)(r / T);
            float4 v;
            if (t == 0) {
                v = *reinterpret_cast<const float4*>(cls + d);
            } else {
                const int64_t src = ((int64_t)b * P + (t - 1)) * D + d;
                if (BF16) {
                    const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(pe) + src);
                    v = make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xffff0000u), __uint_as_float(h.y << 16),
                                    __uint_as_float(h.y & 0xffff0000u));
                } else {
                    v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(pe) + src);
                }
            }
            const float4 pz = *reinterpret_cast<const float4*>(pos + (int64_t)t * D + d);
            *reinterpret_cast<float4*>(tokens + q * 4) = make_float4(v.x + pz.x, v.y + pz.y, v.z + pz.z, v.w + pz.w);
        }
        return;
    }
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int d = (int)(e % D);
        int64_t r = e / D;
        int t = (int)(r % T);
        int b = (int)(r / T);
        float v;
        if (t == 0) {
            v = cls[d];
        } else {
            int64_t src = ((int64_t)b * P + (t - 1)) * D + d;
            v = BF16 ? bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(pe)[src]) : reinterpret_cast<const float*>(pe)[src];
        }
        tokens[e] = v + pos[(int64_t)t * D + d];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// x += a1 + s2 * a2 (bf16 addends, nullable); ln_out = LN(x) (bf16, nullable); x_bf16 = x (nullable).  Wave per row.
// ---------------------------------------------------------------------------------------------------------------
template <int NV>  // row width d <= 256 * NV, d % 4 == 0
__global__ __launch_bounds__(256) void residual_ln_kernel(float* __restrict__ x, int64_t n, int d,
                                                          const unsigned short* __restrict__ a1,
                                                          const unsigned short* __restrict__ a2, float s2,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, unsigned short* __restrict__ ln_out,
                                                          unsigned short* __restrict__ x_bf16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_d = 1.0f / (float)d;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < n; row += (int64_t)gridDim.x * 4) {
        float r[NV * 4];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = (i * 64 + lane) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < d) {
                v = *reinterpret_cast<const f32x4*>(x + row * d + e);
                if (a1) {
                    u32x2 p = *reinterpret_cast<const u32x2*>(a1 + row * d + e);
                    v[0] += __uint_as_float(p[0] << 16);
                    v[1] += __uint_as_float(p[0] & 0xffff0000u);
                    v[2] += __uint_as_float(p[1] << 16);
                    v[3] += __uint_as_float(p[1] & 0xffff0000u);
                }
                if (a2) {
                    u32x2 p = *reinterpret_cast<const u32x2*>(a2 + row * d + e);
                    v[0] = fmaf(s2, __uint_as_float(p[0] << 16), v[0]);
                    v[1] = fmaf(s2, __uint_as_float(p[0] & 0xffff0000u), v[1]);
                    v[2] = fmaf(s2, __uint_as_float(p[1] << 16), v[2]);
                    v[3] = fmaf(s2, __uint_as_float(p[1] & 0xffff0000u), v[3]);
                }
                if (a1 || a2) *reinterpret_cast<f32x4*>(x + row * d + e) = v;
                if (x_bf16) {
                    u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    *reinterpret_cast<u32x2*>(x_bf16 + row * d + e) = o;
                }
            }
            r[i * 4 + 0] = v[0];
            r[i * 4 + 1] = v[1];
            r[i * 4 + 2] = v[2];
            r[i * 4 + 3] = v[3];
        }
        if (!ln_out) continue;
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < NV * 4; ++i) s1 += r[i];
        const float mean = wave_sum(s1) * inv_d;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int e = (i * 64 + lane) * 4 + t;
                const float dv = e < d ? r[i * 4 + t] - mean : 0.f;
                sq = fmaf(dv, dv, sq);
            }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_d + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = (i * 64 + lane) * 4;
            if (e < d) {
                f32x4 g = *reinterpret_cast<const f32x4*>(gamma + e);
                f32x4 bb = *reinterpret_cast<const f32x4*>(beta + e);
                float o0 = (r[i * 4 + 0] - mean) * rstd * g[0] + bb[0];
                float o1 = (r[i * 4 + 1] - mean) * rstd * g[1] + bb[1];
                float o2 = (r[i * 4 + 2] - mean) * rstd * g[2] + bb[2];
                float o3 = (r[i * 4 + 3] - mean) * rstd * g[3] + bb[3];
                u32x2 o = {pack_bf16x2(o0, o1), pack_bf16x2(o2, o3)};
                *reinterpret_cast<u32x2*>(ln_out + row * d + e) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// (mean, rstd) per row for the LayerNorm folded into the next GEMM (snf_gemm_bf16_lnfold): from the fp32 rows themselves (x, a
// wave per row; also writes the bf16 copy the GEMM reads) or from the per-64-column moment pairs the producing GEMM left
// (snf_gemm_bf16_resid: part [n][slots][2], summed in slot order -- bit-reproducible).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ part, int slots,
                                                        int64_t n, int d, float eps, float* __restrict__ stats,
                                                        unsigned short* __restrict__ xb, int64_t ldxb) {
    if (part) {
        const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (row >= n) return;
        float s1 = 0.f, s2 = 0.f;
        for (int g = 0; g < slots; ++g) {
            const f32x2 p = *reinterpret_cast<const f32x2*>(part + 2 * (row * slots + g));
            s1 += p[0], s2 += p[1];
        }
        const float mean = s1 / d;
        const float var = fmaxf(s2 / d - mean * mean, 0.f);
        *reinterpret_cast<f32x2*>(stats + 2 * row) = f32x2{mean, rsqrtf(var + eps)};
        return;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < n; row += (int64_t)gridDim.x * 4) {
        const float* xr = x + row * ldx;
        float s1 = 0.f;
        for (int c = 4 * lane; c < d; c += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
            s1 += (v[0] + v[1]) + (v[2] + v[3]);
            if (xb) *reinterpret_cast<u32x2*>(xb + row * ldxb + c) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s1 += __shfl_xor(s1, o, 64);
        const float mean = s1 / d;
        float s2 = 0.f;
        for (int c = 4 * lane; c < d; c += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
            const float a0 = v[0] - mean, a1 = v[1] - mean, a2 = v[2] - mean, a3 = v[3] - mean;
            s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o, 64);
        if (lane == 0) *reinterpret_cast<f32x2*>(stats + 2 * row) = f32x2{mean, rsqrtf(s2 / d + eps)};
    }
}

// ---------------------------------------------------------------------------------------------------------------
// exact fp32 self-attention: qkv [B*T, 3*h*dk] (q | k | v, each [h][dk]); out [B*T, h*dk]; attn [B, h, T, T] nullable.
// workgroup = (head, image); a thread owns one query row; keys / values stream through LDS in chunks of 64.
// ---------------------------------------------------------------------------------------------------------------
template <int DK>
__global__ __launch_bounds__(256) void vit_attention_f32_kernel(const float* __restrict__ qkv, int B, int T, int h, float scale,
                                                                float* __restrict__ out, float* __restrict__ attn) {
    constexpr int KC = 64;
    __shared__ float lk[KC][DK + 1];
    __shared__ float lv[KC][DK + 1];
    const int a = blockIdx.x, b = blockIdx.y;
    const int D = h * DK;
    const int64_t base = (int64_t)b * T;
    for (int q0 = 0; q0 < T; q0 += 256) {
        const int qi = q0 + threadIdx.x;
        const bool qv = qi < T;
        float q[DK];
#pragma unroll
        for (int e = 0; e < DK; ++e) q[e] = qv ? qkv[(base + qi) * 3 * D + a * DK + e] * scale : 0.f;
        // pass 1: row max and sum (online)
        float m = -INFINITY, l = 0.f;
        for (int k0 = 0; k0 < T; k0 += KC) {
            __syncthreads();
            for (int e = threadIdx.x; e < KC * DK; e += 256) {
                int kk = e / DK, dd = e - kk * DK;
                lk[kk][dd] = (k0 + kk < T) ? qkv[(base + k0 + kk) * 3 * D + D + a * DK + dd] : 0.f;
            }
            __syncthreads();
            const int kn = (T - k0) < KC ? (T - k0) : KC;
            for (int kk = 0; kk < kn; ++kk) {
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < DK; ++e) s = fmaf(q[e], lk[kk][e], s);
                if (s > m) {
                    l = l * expf(m - s);
                    m = s;
                }
                l += expf(s - m);
            }
        }
        // pass 2: probabilities and P V
        float o[DK];
#pragma unroll
        for (int e = 0; e < DK; ++e) o[e] = 0.f;
        const float inv_l = 1.0f / l;
        for (int k0 = 0; k0 < T; k0 += KC) {
            __syncthreads();
            for (int e = threadIdx.x; e < KC * DK; e += 256) {
                int kk = e / DK, dd = e - kk * DK;
                const bool ok = k0 + kk < T;
                lk[kk][dd] = ok ? qkv[(base + k0 + kk) * 3 * D + D + a * DK + dd] : 0.f;
                lv[kk][dd] = ok ? qkv[(base + k0 + kk) * 3 * D + 2 * D + a * DK + dd] : 0.f;
            }
            __syncthreads();
            const int kn = (T - k0) < KC ? (T - k0) : KC;
            for (int kk = 0; kk < kn; ++kk) {
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < DK; ++e) s = fmaf(q[e], lk[kk][e], s);
                const float p = expf(s - m) * inv_l;
                if (attn && qv) attn[(((int64_t)b * h + a) * T + qi) * T + k0 + kk] = p;
#pragma unroll
                for (int e = 0; e < DK; ++e) o[e] = fmaf(p, lv[kk][e], o[e]);
            }
        }
        if (qv) {
#pragma unroll
            for (int e = 0; e < DK; ++e) out[(base + qi) * D + a * DK + e] = o[e];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// MFMA self-attention, dk = 64 (…dino_version.py:82-94).  qkv [B*T, 3*h*64]; out [B*T, h*64].
//   X3 = false: bf16 in / bf16 out, one product per term.
//   X3 = true (round 6): fp32 in / fp32 out in the fp32-class arithmetic of the aggregator -- every operand is hi + lo
//          (hi = bf16(x), lo = bf16(x - hi)), every product hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_bf16 with fp32
//          accumulation; Q is split in registers, K and V are staged as two LDS images each, P is split in registers.
//   CHUNK = false: all keys of the (image, head) fit one LDS image (T <= 32 NKB); a workgroup serves one (image, head), wave
//          w the query tiles w, w + 8, ..
//   CHUNK = true (round 6, the reference's patch-8 recipe: T = 785): keys are staged 32 NKB at a time; a workgroup serves `tpw`
//          query tiles of one (image, head) (blockIdx.z picks them, one per wave) and keeps each tile's running maximum / sum /
//          output accumulator in registers across the key chunks (that state lives across the staging: 256-register waves,
//          one workgroup per CU -- under the 128-register bound of the one-image form the chunked bf16 kernel spilled 85).
// ---------------------------------------------------------------------------------------------------------------
constexpr int VIT_ATTN_THREADS = 512;   // 8 waves: one 32-query tile each (T = 197 -> 7 tiles in one round)
__device__ __forceinline__ void vit_split8(const f32x8 x, bf16x8& hi, bf16x8& lo) {
    hi = __builtin_convertvector(x, bf16x8);
    lo = __builtin_convertvector(x - __builtin_convertvector(hi, f32x8), bf16x8);
}
template <int NKB, bool CHUNK, bool X3>
__global__ __launch_bounds__(VIT_ATTN_THREADS, (X3 || CHUNK) ? 2 : 4) void vit_attention_mfma_kernel(const void* __restrict__ qkv_, int B, int T,
                                                                                          int h, float scale, void* __restrict__ out_,
                                                                                          int tpw, int out_hl) {
    constexpr int DK = 64;
    constexpr int NW = VIT_ATTN_THREADS / 64;   // waves per workgroup = query tiles in flight
    constexpr int NP = X3 ? 2 : 1;            // operand planes (hi, lo)
    constexpr int KPITCH = DK + 8;            // bf16 elements; 144 B rows: conflict-free ds_read_b128 of a K fragment
    constexpr int KBYTES = 32 * NKB * KPITCH * 2, VBYTES = 32 * NKB * 128;
    typedef typename std::conditional<X3, float, unsigned short>::type elt_t;
    const elt_t* __restrict__ qkv = reinterpret_cast<const elt_t*>(qkv_);
    elt_t* __restrict__ out = reinterpret_cast<elt_t*>(out_);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // K planes [NP][32*NKB][KPITCH]; V planes [NP][32*NKB][128 B].
    // V stays ROW-major in LDS ([key][64], 128-byte rows, 16-byte chunk c of row r at chunk position (c + 4*((r>>1)&1)) & 7):
    // the A operand of O^T = V^T P^T (lane = d, 8 keys in registers) is read back with the hardware transpose-read
    // ds_read_b64_tr_b16, so staging is one 16-byte store per chunk instead of eight 2-byte transposing stores.  The
    // rotation spreads the 4 rows x 64 bytes a transpose-read group touches over all 64 banks.
    unsigned char* const lds_v0 = smem + NP * KBYTES;
    const int a = blockIdx.x, b = blockIdx.y;
    const int D = h * DK;
    const int64_t base = (int64_t)b * T;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, hf = lane >> 5;

    // B operand of S^T = K Q^T: lane = query, 8 consecutive dk per k-step
    const int ntile = (T + 31) / 32;
    bf16x8 qf[4], ql[X3 ? 4 : 1];
    auto load_q = [&](int tile) __attribute__((always_inline)) {
        int qrow = 32 * tile + j;
        if (qrow > T - 1) qrow = T - 1;
        const elt_t* qp = qkv + (base + qrow) * 3 * D + a * DK + 8 * hf;
        static_for<0, 4>([&](auto ks) __attribute__((always_inline)) {
            if constexpr (X3) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(qp + 16 * ks), x1 = *reinterpret_cast<const f32x4*>(qp + 16 * ks + 4);
                vit_split8(f32x8{x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]}, qf[ks], ql[ks]);
            } else {
                qf[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(qp + 16 * ks));
            }
        });
    };
    // stage the keys k0 .. k0 + 32 NKB - 1 of this (image, head): every global load of the thread is issued before the first
    // LDS store (one memory latency for the whole image, not one per chunk); rows >= T are zero
    constexpr int NI = (32 * NKB * 8 + VIT_ATTN_THREADS - 1) / VIT_ATTN_THREADS;
    auto stage = [&](int k0) __attribute__((always_inline)) {
        if constexpr (X3) {
            f32x8 kst[NI], vst[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int c = threadIdx.x + VIT_ATTN_THREADS * i;
                const int key = k0 + (c >> 3), part = c & 7;
                kst[i] = f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                vst[i] = kst[i];
                if (key < T && (c >> 3) < 32 * NKB) {
                    const float* rowp = qkv + (base + key) * 3 * D + a * DK + part * 8;
                    const f32x4 k0v = *reinterpret_cast<const f32x4*>(rowp + D), k1v = *reinterpret_cast<const f32x4*>(rowp + D + 4);
                    const f32x4 v0v = *reinterpret_cast<const f32x4*>(rowp + 2 * D), v1v = *reinterpret_cast<const f32x4*>(rowp + 2 * D + 4);
                    kst[i] = f32x8{k0v[0], k0v[1], k0v[2], k0v[3], k1v[0], k1v[1], k1v[2], k1v[3]};
                    vst[i] = f32x8{v0v[0], v0v[1], v0v[2], v0v[3], v1v[0], v1v[1], v1v[2], v1v[3]};
                }
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int c = threadIdx.x + VIT_ATTN_THREADS * i;
                const int key = c >> 3, part = c & 7;
                if (key < 32 * NKB) {
                    bf16x8 kh, kl, vh, vl;
                    vit_split8(kst[i], kh, kl);
                    vit_split8(vst[i], vh, vl);
                    unsigned char* kp = smem + (key * KPITCH + part * 8) * 2;
                    unsigned char* vp = lds_v0 + key * 128 + 16 * ((part + 4 * ((key >> 1) & 1)) & 7);
                    *reinterpret_cast<u32x4*>(kp) = __builtin_bit_cast(u32x4, kh);
                    *reinterpret_cast<u32x4*>(kp + KBYTES) = __builtin_bit_cast(u32x4, kl);
                    *reinterpret_cast<u32x4*>(vp) = __builtin_bit_cast(u32x4, vh);
                    *reinterpret_cast<u32x4*>(vp + VBYTES) = __builtin_bit_cast(u32x4, vl);
                }
            }
        } else {
            u32x4 kst[NI], vst[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int c = threadIdx.x + VIT_ATTN_THREADS * i;
                const int key = k0 + (c >> 3), part = c & 7;
                kst[i] = u32x4{0u, 0u, 0u, 0u};
                vst[i] = u32x4{0u, 0u, 0u, 0u};
                if (key < T && (c >> 3) < 32 * NKB) {
                    const unsigned short* rowp = qkv + (base + key) * 3 * D + a * DK + part * 8;
                    kst[i] = *reinterpret_cast<const u32x4*>(rowp + D);
                    vst[i] = *reinterpret_cast<const u32x4*>(rowp + 2 * D);
                }
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int c = threadIdx.x + VIT_ATTN_THREADS * i;
                const int key = c >> 3, part = c & 7;
                if (key < 32 * NKB) {
                    *reinterpret_cast<u32x4*>(smem + (key * KPITCH + part * 8) * 2) = kst[i];
                    *reinterpret_cast<u32x4*>(lds_v0 + key * 128 + 16 * ((part + 4 * ((key >> 1) & 1)) & 7)) = vst[i];
                }
            }
        }
    };
    // transpose-read addressing of a V^T fragment (d block db, keys k_base ..): this lane's 16-lane group g covers
    // d = 32 db + 16 (g & 1) + i, keys k_base + 4 (g >> 1) + {0..3} (first read) and the same + 8 (second)
    const int tg = lane >> 4, ti = lane & 15;
    const int vkey = 4 * (tg >> 1) + (ti >> 2);                 // the row this lane's chunk comes from
    const int vch = 2 * (tg & 1) + ((ti & 3) >> 1);             // 16-byte chunk inside the 64-byte d block, + 4 db
    const int vhalf = 8 * (ti & 1);
    auto v_frag = [&](int plane, int k_base, int db) __attribute__((always_inline)) -> bf16x8 {
        const int r0 = k_base + vkey, r1 = r0 + 8;
        const unsigned char* p0 = lds_v0 + plane * VBYTES + r0 * 128 + 16 * ((vch + 4 * db + 4 * ((r0 >> 1) & 1)) & 7) + vhalf;
        const unsigned char* p1 = lds_v0 + plane * VBYTES + r1 * 128 + 16 * ((vch + 4 * db + 4 * ((r1 >> 1) & 1)) & 7) + vhalf;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
        const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, vv);
    };

    const float c_exp = scale * 1.44269504088896340736f;
    // Flash-style pass over the key blocks: S^T block = K_jb Q^T (keys in registers, this lane's query on the lane), online
    // softmax with a running maximum, O^T += V^T P^T with P^T straight from the C registers.  ~100 registers per wave
    // instead of 16*NKB + ... for the whole score row: two workgroups of eight waves stay resident per CU (bf16 form).
    f32x16 o_acc[2];
    float m_run, l_lane;                      // running max (all-reduced over the two half-waves), lane-local sum
    auto reset = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o_acc[0][r] = 0.f;
            o_acc[1][r] = 0.f;
        }
        m_run = -INFINITY, l_lane = 0.f;
    };
    // key block jb of the staged image; its first key is key kg0 of the (image, head); first: no accumulator to rescale yet
    auto block = [&](int jb, int kg0, bool first) __attribute__((always_inline)) {
        f32x16 s_acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kg0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            s_acc[r] = (key >= T) ? -INFINITY : 0.f;
        }
        static_for<0, 4>([&](auto ks) __attribute__((always_inline)) {
            const unsigned char* kp = smem + ((32 * jb + j) * KPITCH + 16 * ks + 8 * hf) * 2;
            const bf16x8 kf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(kp));
            if constexpr (X3) {
                const bf16x8 kl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(kp + KBYTES));
                s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, ql[ks], s_acc, 0, 0, 0);
                s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qf[ks], s_acc, 0, 0, 0);
            }
            s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s_acc, 0, 0, 0);
        });
        float mb0 = fmaxf(s_acc[0], s_acc[1]), mb1 = fmaxf(s_acc[2], s_acc[3]);
#pragma unroll
        for (int r = 4; r < 16; r += 4) {
            mb0 = fmaxf(fmaxf(mb0, s_acc[r]), s_acc[r + 1]);
            mb1 = fmaxf(fmaxf(mb1, s_acc[r + 2]), s_acc[r + 3]);
        }
        float mb = fmaxf(mb0, mb1);
        mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
        // the first block always holds a valid key, so m_new is finite from the first block on
        const float m_new = fmaxf(m_run, mb);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_exp);   // exp2(-inf) = 0 on the first block
        const float mc = m_new * c_exp;
        m_run = m_new;
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(s_acc[r], c_exp, -mc));
            s_acc[r] = e;
            lsum += e;
        }
        l_lane = fmaf(l_lane, alpha, lsum);
        if (!first) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o_acc[0][r] *= alpha;
                o_acc[1][r] *= alpha;
            }
        }
        static_for<0, 2>([&](auto u_t) __attribute__((always_inline)) {
            constexpr int u = decltype(u_t)::value;
            f32x8 pv;
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = s_acc[8 * u + e];
            // this lane's 8 keys of the k-step: {16u + 4hf + 0..3} and {16u + 8 + 4hf + 0..3} of block jb
            if constexpr (X3) {
                bf16x8 ph, pl;
                vit_split8(pv, ph, pl);
                static_for<0, 2>([&](auto db_t) __attribute__((always_inline)) {
                    constexpr int db = decltype(db_t)::value;
                    const bf16x8 vh = v_frag(0, 32 * jb + 16 * u, db), vl = v_frag(1, 32 * jb + 16 * u, db);
                    o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, o_acc[db], 0, 0, 0);
                    o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, o_acc[db], 0, 0, 0);
                    o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, o_acc[db], 0, 0, 0);
                });
            } else {
                const bf16x8 pf = __builtin_convertvector(pv, bf16x8);
                static_for<0, 2>([&](auto db_t) __attribute__((always_inline)) {
                    constexpr int db = decltype(db_t)::value;
                    o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_frag(0, 32 * jb + 16 * u, db), pf, o_acc[db], 0, 0, 0);
                });
            }
        });
    };
    // store O of query tile `tile`: lane = query row, 4 consecutive d per register group
    auto finish = [&](int tile) __attribute__((always_inline)) {
        const float l = l_lane + __shfl_xor(l_lane, 32, 64);
        const float inv = __builtin_amdgcn_rcpf(l);
        const int q_out = 32 * tile + j;
        if (q_out < T) {
            elt_t* op = out + (base + q_out) * D + a * DK;
            static_for<0, 2>([&](auto db) __attribute__((always_inline)) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = 32 * db + 8 * g + 4 * hf;
                    if constexpr (X3) {
                        const f32x4 o4 = {o_acc[db][4 * g] * inv, o_acc[db][4 * g + 1] * inv, o_acc[db][4 * g + 2] * inv, o_acc[db][4 * g + 3] * inv};
                        if (out_hl) {
                            // the hl image of the row ([hi(32) | lo(32)] bf16 per 32 columns): what the proj GEMM reads -- O never exists in fp32
                            const u32x2 hi2 = {pack_bf16x2(o4[0], o4[1]), pack_bf16x2(o4[2], o4[3])};
                            const u32x2 lo2 = {pack_bf16x2(o4[0] - __uint_as_float(hi2[0] << 16), o4[1] - __uint_as_float(hi2[0] & 0xffff0000u)),
                                               pack_bf16x2(o4[2] - __uint_as_float(hi2[1] << 16), o4[3] - __uint_as_float(hi2[1] & 0xffff0000u))};
                            unsigned short* oh = reinterpret_cast<unsigned short*>(out_) + (base + q_out) * 2 * D + 2 * a * DK + 64 * db + 8 * g + 4 * hf;
                            *reinterpret_cast<u32x2*>(oh) = hi2;
                            *reinterpret_cast<u32x2*>(oh + 32) = lo2;
                        } else {
                            *reinterpret_cast<f32x4*>(op + d0) = o4;
                        }
                    } else {
                        u32x2 o = {pack_bf16x2(o_acc[db][4 * g] * inv, o_acc[db][4 * g + 1] * inv),
                                   pack_bf16x2(o_acc[db][4 * g + 2] * inv, o_acc[db][4 * g + 3] * inv)};
                        *reinterpret_cast<u32x2*>(op + d0) = o;
                    }
                }
            });
        }
    };

    if constexpr (CHUNK) {
        const int tile = tpw * blockIdx.z + w;
        const bool active = w < tpw && tile < ntile;   // wave-uniform
        load_q(active ? tile : ntile - 1);
        reset();
        for (int k0 = 0; k0 < T; k0 += 32 * NKB) {
            if (k0) __syncthreads();              // every wave is done with the previous chunk's image
            stage(k0);
            __syncthreads();
            if (active) {
#pragma unroll 1
                for (int jb = 0; jb < NKB; ++jb) {
                    if (k0 + 32 * jb >= T) break;
                    block(jb, k0 + 32 * jb, k0 == 0 && jb == 0);
                }
            }
        }
        if (active) finish(tile);
    } else {
        // this wave's first query tile is requested before anything else
        if (w < ntile) load_q(w);
        stage(0);
        __syncthreads();
        for (int tile = w; tile < ntile; tile += NW) {
            if (tile != w) load_q(tile);
            reset();
#pragma unroll 1   // a rolled loop: unrolled, the per-block LDS addresses alone cost ~70 registers (two workgroups per CU need <= 128)
            for (int jb = 0; jb < NKB; ++jb) block(jb, 32 * jb, jb == 0);
            finish(tile);
        }
    }
}

template <int NKB, bool CHUNK, bool X3>
int launch_vit_mfma(const void* qkv, int B, int T, int h, float scale, void* out, hipStream_t s, int out_hl = 0) {
    constexpr size_t lds = (size_t)(X3 ? 2 : 1) * ((size_t)(32 * NKB * (64 + 8)) * sizeof(unsigned short) + (size_t)32 * NKB * 128);
    static_assert(lds <= 160 * 1024, "vit_attention_mfma: LDS budget");
    static thread_local unsigned long long attr_set_mask = 0;   // devices (bit = device id) that have the opt-in
    const unsigned long long attr_set_bit = snf::device_bit();
    const bool attr_set = (attr_set_mask & attr_set_bit) != 0;
    auto kern = vit_attention_mfma_kernel<NKB, CHUNK, X3>;
    if (!attr_set && lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            snf::set_error("vit_attention_mfma: cannot reserve %zu bytes of LDS", lds);
            (void)hipGetLastError();
            return SNF_ELAUNCH;
        }
        attr_set_mask |= attr_set_bit;
    }
    // chunked keys: a workgroup serves tpw <= 8 query tiles (one per wave); the tiles are spread evenly over the workgroups
    const int ntile = (T + 31) / 32;
    const int nz = CHUNK ? (ntile + 7) / 8 : 1;
    const int tpw = CHUNK ? (ntile + nz - 1) / nz : 0;
    hipLaunchKernelGGL(kern, dim3(h, B, nz), dim3(VIT_ATTN_THREADS), lds, s, qkv, B, T, h, scale, out, tpw, out_hl);
    return snf::check_launch("vit_attention_mfma_kernel");
}

inline int grid_for(int64_t work_items, int per_block) {
    int64_t want = (work_items + per_block - 1) / per_block;
    int64_t cap = (int64_t)snf::cu_count() * 8;
    return (int)(want < 1 ? 1 : (want < cap ? want : cap));
}

}  // namespace

extern "C" {

int snf_vit_patchify(const float* img, int b, int c, int hgt, int wid, int patch, void* cols, int out_dtype,
                     snf_stream_t stream) {
    SNF_REQUIRE(img && cols, "snf_vit_patchify: null pointer");
    SNF_REQUIRE(b >= 1 && c >= 1 && patch >= 1 && hgt >= patch && wid >= patch, "snf_vit_patchify: bad shape");
    SNF_REQUIRE(hgt % patch == 0 && wid % patch == 0, "snf_vit_patchify: image %dx%d not a multiple of patch %d", hgt, wid, patch);
    SNF_REQUIRE(out_dtype == SNF_DT_F32 || out_dtype == SNF_DT_BF16, "snf_vit_patchify: bad dtype");
    const int64_t segs = (int64_t)b * c * hgt * (wid / patch);
    hipStream_t s = snf::as_stream(stream);
    if (out_dtype == SNF_DT_BF16)
        hipLaunchKernelGGL(patchify_kernel<true>, dim3(grid_for(segs, 256)), dim3(256), 0, s, img, b, c, hgt, wid, patch, cols);
    else
        hipLaunchKernelGGL(patchify_kernel<false>, dim3(grid_for(segs, 256)), dim3(256), 0, s, img, b, c, hgt, wid, patch, cols);
    return snf::check_launch("patchify_kernel");
}

int snf_vit_assemble_tokens(const void* patch_emb, int pe_dtype, const float* cls_token, const float* pos_embed, int b,
                            int num_patches, int d, float* tokens, snf_stream_t stream) {
    SNF_REQUIRE(patch_emb && cls_token && pos_embed && tokens, "snf_vit_assemble_tokens: null pointer");
    SNF_REQUIRE(b >= 1 && num_patches >= 1 && d >= 1, "snf_vit_assemble_tokens: bad shape");
    SNF_REQUIRE(pe_dtype == SNF_DT_F32 || pe_dtype == SNF_DT_BF16, "snf_vit_assemble_tokens: bad dtype");
    const int64_t total = (int64_t)b * (num_patches + 1) * d;
    hipStream_t s = snf::as_stream(stream);
    if (pe_dtype == SNF_DT_BF16)
        hipLaunchKernelGGL(assemble_tokens_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, s, patch_emb, cls_token,
                           pos_embed, b, num_patches, d, tokens);
    else
        hipLaunchKernelGGL(assemble_tokens_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, s, patch_emb, cls_token,
                           pos_embed, b, num_patches, d, tokens);
    return snf::check_launch("assemble_tokens_kernel");
}

int snf_vit_residual_ln(float* x, int64_t n, int d, const void* add1_bf16, const void* add2_bf16, float scale2,
                        const float* gamma, const float* beta, float eps, void* ln_out_bf16, void* x_bf16,
                        snf_stream_t stream) {
    SNF_REQUIRE(x, "snf_vit_residual_ln: null x");
    SNF_REQUIRE(n >= 1 && d >= 4 && (d % 4) == 0 && d <= 2048, "snf_vit_residual_ln: need d %% 4 == 0 and d <= 2048 (d=%d)", d);
    SNF_REQUIRE(!ln_out_bf16 || (gamma && beta), "snf_vit_residual_ln: LayerNorm output needs gamma and beta");
    hipStream_t s = snf::as_stream(stream);
    const int grid = grid_for(n, 4);
    const int nv = (d / 4 + 63) / 64;
    const unsigned short* a1 = reinterpret_cast<const unsigned short*>(add1_bf16);
    const unsigned short* a2 = reinterpret_cast<const unsigned short*>(add2_bf16);
    unsigned short* lo = reinterpret_cast<unsigned short*>(ln_out_bf16);
    unsigned short* xb = reinterpret_cast<unsigned short*>(x_bf16);
#define LAUNCH_RL(NV) \
    hipLaunchKernelGGL(residual_ln_kernel<NV>, dim3(grid), dim3(256), 0, s, x, n, d, a1, a2, scale2, gamma, beta, eps, lo, xb)
    if (nv <= 1) LAUNCH_RL(1);
    else if (nv <= 2) LAUNCH_RL(2);
    else if (nv <= 3) LAUNCH_RL(3);
    else if (nv <= 4) LAUNCH_RL(4);
    else LAUNCH_RL(8);
#undef LAUNCH_RL
    return snf::check_launch("residual_ln_kernel");
}

int snf_vit_row_stats(const float* x, int64_t ldx, const float* part, int slots, int64_t n, int d, float eps, float* stats,
                      void* x_bf16, int64_t ldxb, snf_stream_t stream) {
    SNF_REQUIRE(stats && (x || part) && !(x && part), "snf_vit_row_stats: give the fp32 rows OR the producer's moment pairs");
    SNF_REQUIRE(n >= 1 && d >= 4 && d % 4 == 0, "snf_vit_row_stats: bad shape n=%lld d=%d", (long long)n, d);
    SNF_REQUIRE(!part || (slots >= 1 && !x_bf16), "snf_vit_row_stats: moment pairs need slots >= 1 (and write no bf16 copy)");
    SNF_REQUIRE(!x || (ldx >= d && ldx % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0), "snf_vit_row_stats: x alignment");
    SNF_REQUIRE(!x_bf16 || (ldxb >= d && ldxb % 4 == 0 && reinterpret_cast<uintptr_t>(x_bf16) % 8 == 0), "snf_vit_row_stats: x_bf16 alignment");
    hipStream_t s = snf::as_stream(stream);
    const int grid = part ? (int)((n + 255) / 256) : grid_for(n, 4);
    hipLaunchKernelGGL(row_stats_kernel, dim3(grid), dim3(256), 0, s, x, ldx, part, slots, n, d, eps, stats,
                       reinterpret_cast<unsigned short*>(x_bf16), ldxb);
    return snf::check_launch("row_stats_kernel");
}

int snf_vit_attention_f32(const float* qkv, int b, int t, int h, int dk, float scale, float* out, float* attn,
                          snf_stream_t stream) {
    SNF_REQUIRE(qkv && out, "snf_vit_attention_f32: null pointer");
    SNF_REQUIRE(b >= 1 && t >= 1 && h >= 1, "snf_vit_attention_f32: bad shape");
    SNF_REQUIRE(b <= 65535, "snf_vit_attention_f32: batch too large for one launch");
    hipStream_t s = snf::as_stream(stream);
    dim3 grid(h, b);
    switch (dk) {
        case 32: hipLaunchKernelGGL(vit_attention_f32_kernel<32>, grid, dim3(256), 0, s, qkv, b, t, h, scale, out, attn); break;
        case 64: hipLaunchKernelGGL(vit_attention_f32_kernel<64>, grid, dim3(256), 0, s, qkv, b, t, h, scale, out, attn); break;
        case 96: hipLaunchKernelGGL(vit_attention_f32_kernel<96>, grid, dim3(256), 0, s, qkv, b, t, h, scale, out, attn); break;
        case 128: hipLaunchKernelGGL(vit_attention_f32_kernel<128>, grid, dim3(256), 0, s, qkv, b, t, h, scale, out, attn); break;
        default:
            snf::set_error("snf_vit_attention_f32: head dim %d not in {32, 64, 96, 128}", dk);
            return SNF_EUNSUPPORTED;
    }
    return snf::check_launch("vit_attention_f32_kernel");
}

int snf_vit_attention_mfma(const void* qkv_bf16, int b, int t, int h, int dk, float scale, void* out_bf16,
                           snf_stream_t stream) {
    SNF_REQUIRE(qkv_bf16 && out_bf16, "snf_vit_attention_mfma: null pointer");
    SNF_REQUIRE(b >= 1 && t >= 1 && h >= 1, "snf_vit_attention_mfma: bad shape");
    if (dk != 64 || t > SNF_VIT_MFMA_MAX_T || b > 65535) {
        snf::set_error("snf_vit_attention_mfma: unsupported shape dk=%d t=%d (need dk == 64, t <= %d)", dk, t, SNF_VIT_MFMA_MAX_T);
        return SNF_EUNSUPPORTED;
    }
    SNF_REQUIRE((reinterpret_cast<uintptr_t>(qkv_bf16) & 15) == 0, "snf_vit_attention_mfma: qkv must be 16-byte aligned");
    hipStream_t s = snf::as_stream(stream);
    const int nkb = (t + 31) / 32;
    if (nkb <= 2) return launch_vit_mfma<2, false, false>(qkv_bf16, b, t, h, scale, out_bf16, s);
    if (nkb <= 4) return launch_vit_mfma<4, false, false>(qkv_bf16, b, t, h, scale, out_bf16, s);
    if (nkb <= 7) return launch_vit_mfma<7, false, false>(qkv_bf16, b, t, h, scale, out_bf16, s);
    if (nkb <= 8) return launch_vit_mfma<8, false, false>(qkv_bf16, b, t, h, scale, out_bf16, s);
    return launch_vit_mfma<8, true, false>(qkv_bf16, b, t, h, scale, out_bf16, s);   // keys in chunks of 256 (patch 8: t = 785)
}

int snf_vit_attention_x3_f32(const float* qkv, int b, int t, int h, int dk, float scale, void* out, int out_dtype, snf_stream_t stream) {
    SNF_REQUIRE(qkv && out, "snf_vit_attention_x3_f32: null pointer");
    SNF_REQUIRE(out_dtype == SNF_DT_F32 || out_dtype == SNF_DT_BF16_HL, "snf_vit_attention_x3_f32: output is f32 or the hl image (dtype %d)", out_dtype);
    const int out_hl = out_dtype == SNF_DT_BF16_HL;
    SNF_REQUIRE(b >= 1 && t >= 1 && h >= 1, "snf_vit_attention_x3_f32: bad shape");
    if (dk != 64 || t > SNF_VIT_MFMA_MAX_T || b > 65535) {
        snf::set_error("snf_vit_attention_x3_f32: unsupported shape dk=%d t=%d (need dk == 64, t <= %d)", dk, t, SNF_VIT_MFMA_MAX_T);
        return SNF_EUNSUPPORTED;
    }
    SNF_REQUIRE((reinterpret_cast<uintptr_t>(qkv) & 15) == 0, "snf_vit_attention_x3_f32: qkv must be 16-byte aligned");
    hipStream_t s = snf::as_stream(stream);
    const int nkb = (t + 31) / 32;
    if (nkb <= 2) return launch_vit_mfma<2, false, true>(qkv, b, t, h, scale, out, s, out_hl);
    if (nkb <= 4) return launch_vit_mfma<4, false, true>(qkv, b, t, h, scale, out, s, out_hl);
    if (nkb <= 7) return launch_vit_mfma<7, false, true>(qkv, b, t, h, scale, out, s, out_hl);
    return launch_vit_mfma<7, true, true>(qkv, b, t, h, scale, out, s, out_hl);               // keys in chunks of 224
}

}  // extern "C"
