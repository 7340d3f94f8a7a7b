// K7 (exact-fp32, any-shape form): sparse attention of snuffy.py:160-168.
//
//   P_a = softmax_j(Q_a Kp_a^T * scale)   [n, k]       O_a = P_a^T V_a   [k, dk]
//
// Two kernels + a fixed-order reduction, all fp32 FMA (reference-class numerics, used by the fp32 parity path and
// as the fallback for shapes the MFMA kernel does not take):
//   1. scores_softmax: a workgroup owns 16 query rows of one head (a wave owns 4 of them, a lane owns keys
//      lane, lane+64, ...), Kp streamed through LDS in 64-key chunks, the whole softmax row lives in registers;
//      writes P (coalesced along keys) and the row log-sum-exp.
//   2. pt_v: O_a = P_a^T V_a as an LDS-tiled fp32 GEMM over a slice of the n rows -> partial [slice, h, k, dk].
//   3. reduce: out[k, d] = sum over slices in ascending order (deterministic).
#include <math.h>

#include <atomic>

#include "common.h"
#include "philox.h"

namespace {

constexpr int ROWS_PER_WG = 16;
constexpr int KCHUNK = 64;

template <int KPL>  // keys per lane: k <= 64 * KPL
__global__ __launch_bounds__(256) void scores_softmax_kernel(const float* __restrict__ q, const float* __restrict__ kp,
                                                             int64_t n, int k, int h, int dk, float scale,
                                                             float* __restrict__ p_out /*[h,n,k]*/,
                                                             float* __restrict__ lse /*[h,n] nullable*/) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int dkp = (dk + 3) & ~3;
    const int kpitch = dkp + 4;
    float* lq = lds;                      // [16][dkp]
    float* lk = lds + ROWS_PER_WG * dkp;  // [64][kpitch]
    const int a = blockIdx.y;
    const int d_model = h * dk;
    const int64_t row0 = (int64_t)blockIdx.x * ROWS_PER_WG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    for (int e = threadIdx.x; e < ROWS_PER_WG * dkp; e += 256) {
        int r = e / dkp, d = e - r * dkp;
        int64_t row = row0 + r;
        lq[e] = (row < n && d < dk) ? q[row * d_model + a * dk + d] : 0.f;
    }
    float s[4][KPL];
#pragma unroll
    for (int c = 0; c < KPL; ++c) {
        __syncthreads();  // previous chunk fully consumed (and lq visible on c == 0)
        for (int e = threadIdx.x; e < KCHUNK * dkp; e += 256) {
            int j = e / dkp, d = e - j * dkp;
            int key = c * KCHUNK + j;
            lk[j * kpitch + d] = (key < k && d < dk) ? kp[(int64_t)key * d_model + a * dk + d] : 0.f;
        }
        __syncthreads();
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float4* kv4 = reinterpret_cast<const float4*>(lk + lane * kpitch);
        for (int d4 = 0; d4 < dkp / 4; ++d4) {
            float4 kv = kv4[d4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float4 qv = reinterpret_cast<const float4*>(lq + (wave * 4 + r) * dkp)[d4];
                acc[r] = fmaf(qv.x, kv.x, acc[r]);
                acc[r] = fmaf(qv.y, kv.y, acc[r]);
                acc[r] = fmaf(qv.z, kv.z, acc[r]);
                acc[r] = fmaf(qv.w, kv.w, acc[r]);
            }
        }
        const bool valid = (c * KCHUNK + lane) < k;
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r][c] = valid ? acc[r] * scale : -INFINITY;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + wave * 4 + r;
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < KPL; ++c) m = fmaxf(m, s[r][c]);
        m = wave_max(m);
        float l = 0.f;
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            float e = expf(s[r][c] - m);  // exp(-inf) = 0 for padded keys
            s[r][c] = e;
            l += e;
        }
        l = wave_sum(l);
        const float inv = 1.0f / l;
        if (row < n) {
            float* prow = p_out + ((int64_t)a * n + row) * k;
#pragma unroll
            for (int c = 0; c < KPL; ++c) {
                int key = c * KCHUNK + lane;
                if (key < k) prow[key] = s[r][c] * inv;
            }
            if (lse && lane == 0) lse[(int64_t)a * n + row] = m + logf(l);
        }
    }
}

// O_a partial: tile 64 keys x 64 cols, rows [r_begin, r_end) of one slice; 256 threads, 4x4 micro-tiles.
__global__ __launch_bounds__(256) void pt_v_kernel(const float* __restrict__ p /*[h,n,k]*/, const float* __restrict__ v,
                                                   int64_t n, int k, int h, int dk, int64_t rows_per_slice,
                                                   float* __restrict__ partial /*[slices,h,k,dk]*/) {
    __shared__ float lp[16][64 + 4];
    __shared__ float lv[16][64 + 4];
    const int a = blockIdx.z % h;
    const int slice = blockIdx.z / h;
    const int j0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int d_model = h * dk;
    const int tj = threadIdx.x & 15, tc = threadIdx.x >> 4;
    const int64_t r_begin = (int64_t)slice * rows_per_slice;
    int64_t r_end = r_begin + rows_per_slice;
    if (r_end > n) r_end = n;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const int lr = threadIdx.x >> 4;        // 0..15: row within the 16-row step
    const int lc = (threadIdx.x & 15) * 4;  // 0..60
    for (int64_t r = r_begin; r < r_end; r += 16) {
        const int64_t row = r + lr;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int key = j0 + lc + t, col = c0 + lc + t;
            lp[lr][lc + t] = (row < r_end && key < k) ? p[((int64_t)a * n + row) * k + key] : 0.f;
            lv[lr][lc + t] = (row < r_end && col < dk) ? v[row * d_model + a * dk + col] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            float pv[4], vv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pv[i] = lp[rr][tj * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) vv[j] = lv[rr][tc * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(pv[i], vv[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* dst = partial + ((int64_t)slice * h + a) * (int64_t)k * dk;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int key = j0 + tj * 4 + i;
        if (key >= k) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int col = c0 + tc * 4 + j;
            if (col < dk) dst[(int64_t)key * dk + col] = acc[i][j];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same two kernels on the matrix cores in EXACT fp32 (round 5): v_mfma_f32_32x32x2_f32 takes f32 operands and is bit for bit
// a k-ordered fmaf chain (MI355X_MICROARCH.md), at the f32 vector peak -- but as one instruction per 4096 flops instead of 64 FMAs,
// fed by one 16-byte load per operand and four MFMAs.  Any head width that is a multiple of 8 (96, 192, ... -- the README recipes'
// h = 4 at D = 768, reference README.md:661-669, which no bf16 / split-bf16 kernel form fits), up to 1024 keys.
//   scores_softmax_mfma: a workgroup owns 32 query rows of one head (Q tile in LDS), wave w the key blocks w, w + 4, ..;
//     S^T[32 keys, 32 rows] per block (A = Kp rows straight from L2, B = Q from LDS; the contraction index is walked as
//     (8 T + 4 half + j) so that a lane's four k-steps are ONE float4), softmax in registers (lane = row), (max, sum) of the four
//     waves through LDS in wave order, P written once.
//   pt_v_mfma: a WAVE owns a [64 keys, 32 CT columns] tile of O over a slice of the rows; A = P^T and B = V are single coalesced
//     dword loads (the lane layouts of the instruction ARE the memory layouts), 2 CT MFMAs per row pair.
// ---------------------------------------------------------------------------------------------------------------
typedef float mf32x16 __attribute__((ext_vector_type(16)));
typedef float mf32x4 __attribute__((ext_vector_type(4)));

template <int KBW>   // key blocks per wave: k <= 128 * KBW
__global__ __launch_bounds__(256, KBW <= 4 ? 3 : 2) void scores_softmax_mfma_kernel(const float* __restrict__ q, const float* __restrict__ kp, int64_t n,
                                                                  int k, int h, int dk, float scale, float* __restrict__ p_out,
                                                                  float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int pitch = dk + 4;                  // floats: 16-byte aligned rows, 4 banks apart (conflict-free float4 reads down a column)
    float* lq = lds;                           // [32][pitch]
    float* lst = lds + 32 * pitch;             // [2][4][32]: per-wave row max, then row sum
    const int a = blockIdx.y;
    const int d_model = h * dk;
    const int64_t row0 = (int64_t)blockIdx.x * 32;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 31, hf = lane >> 5;
    const int dk4 = dk >> 2;
    for (int e = threadIdx.x; e < 32 * dk4; e += 256) {
        const int r = e / dk4, c4 = e - r * dk4;
        const int64_t row = row0 + r;
        mf32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (row < n) val = *reinterpret_cast<const mf32x4*>(q + row * d_model + a * dk + 4 * c4);
        *reinterpret_cast<mf32x4*>(lq + r * pitch + 4 * c4) = val;
    }
    __syncthreads();
    const int nkb = (k + 31) >> 5;
    const int nt = dk >> 3;
    mf32x16 S[KBW];
    const float* qr = lq + j * pitch + 4 * hf;
#pragma unroll
    for (int c = 0; c < KBW; ++c) {
#pragma unroll
        for (int i = 0; i < 16; ++i) S[c][i] = 0.f;
        const int kb = w + 4 * c;              // wave-uniform
        if (kb < nkb) {
            int key = 32 * kb + j;
            if (key > k - 1) key = k - 1;      // padded keys read the last row; their scores are masked below
            const float* kr = kp + (int64_t)key * d_model + a * dk + 4 * hf;
            int t = 0;
            for (; t + 4 <= nt; t += 4) {      // four steps' operands requested together (Kp comes from L2)
                mf32x4 a4[4], b4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a4[u] = *reinterpret_cast<const mf32x4*>(kr + 8 * (t + u));
                    b4[u] = *reinterpret_cast<const mf32x4*>(qr + 8 * (t + u));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) S[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u][e], b4[u][e], S[c], 0, 0, 0);
            }
            for (; t < nt; ++t) {
                const mf32x4 a4 = *reinterpret_cast<const mf32x4*>(kr + 8 * t);
                const mf32x4 b4 = *reinterpret_cast<const mf32x4*>(qr + 8 * t);
#pragma unroll
                for (int e = 0; e < 4; ++e) S[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], S[c], 0, 0, 0);
            }
        }
    }
    // lane (row j, half hf) holds keys 32 kb + (i & 3) + 8 (i >> 2) + 4 hf of ITS row
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < KBW; ++c) {
        const int kb = w + 4 * c;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * hf;
            S[c][i] = (kb < nkb && key < k) ? S[c][i] * scale : -INFINITY;
            m = fmaxf(m, S[c][i]);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (hf == 0) lst[w * 32 + j] = m;
    __syncthreads();
    m = fmaxf(fmaxf(lst[j], lst[32 + j]), fmaxf(lst[64 + j], lst[96 + j]));
    float l = 0.f;
#pragma unroll
    for (int c = 0; c < KBW; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float e = __builtin_amdgcn_exp2f((S[c][i] - m) * 1.44269504088896340736f);   // exp2(-inf) = 0 for padded keys
            S[c][i] = e;
            l += e;
        }
    l += __shfl_xor(l, 32, 64);
    if (hf == 0) lst[128 + w * 32 + j] = l;
    __syncthreads();
    l = ((lst[128 + j] + lst[160 + j]) + lst[192 + j]) + lst[224 + j];
    const float inv = 1.0f / l;
    const int64_t row = row0 + j;
    if (row < n) {
        float* prow = p_out + ((int64_t)a * n + row) * k;
        const bool vec = (k & 3) == 0 && (reinterpret_cast<uintptr_t>(p_out) & 15) == 0;
#pragma unroll
        for (int c = 0; c < KBW; ++c) {
            const int kb = w + 4 * c;
            if (kb < nkb) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int key0 = 32 * kb + 8 * q4 + 4 * hf;
                    const mf32x4 pv = {S[c][4 * q4] * inv, S[c][4 * q4 + 1] * inv, S[c][4 * q4 + 2] * inv, S[c][4 * q4 + 3] * inv};
                    if (vec && key0 + 4 <= k) {
                        *reinterpret_cast<mf32x4*>(prow + key0) = pv;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (key0 + e < k) prow[key0 + e] = pv[e];
                    }
                }
            }
        }
        if (lse && w == 0 && hf == 0) lse[(int64_t)a * n + row] = m + logf(l);
    }
}

template <int CT>    // 32-column blocks per wave tile
__global__ __launch_bounds__(256, 2) void pt_v_mfma_kernel(const float* __restrict__ p /*[h,n,k]*/, const float* __restrict__ v, int64_t n,
                                                           int k, int h, int dk, int64_t rows_per_slice, int slices,
                                                           float* __restrict__ partial /*[slices,h,k,dk]*/, int ldv /* row pitch of v */) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, hf = lane >> 5;
    const int a = blockIdx.z % h;
    const int slice = 4 * (blockIdx.z / h) + w;          // one wave = one slice of the rows
    if (slice >= slices) return;
    const int64_t r_begin = (int64_t)slice * rows_per_slice;
    int64_t r_end = r_begin + rows_per_slice;
    if (r_end > n) r_end = n;
    const int kb0 = 2 * blockIdx.x, cb0 = CT * blockIdx.y;
    // per-lane element offsets inside a row pair (32-bit), on top of a wave-uniform row pointer: lane (j, hf) reads row r + hf
    int offa[2], offb[CT];
    bool kok[2], cok[CT];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int key = 32 * (kb0 + t) + j;
        kok[t] = key < k;
        if (!kok[t]) key = k - 1;
        offa[t] = hf * k + key;
    }
    int col[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        col[c] = 32 * (cb0 + c) + j;
        cok[c] = col[c] < dk;
        if (!cok[c]) col[c] = dk - 1;
        offb[c] = hf * ldv + col[c];
    }
    mf32x16 acc[2][CT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][c][i] = 0.f;
    const float* pa = p + (int64_t)a * n * k;
    const float* va = v + a * dk;
    constexpr int U = CT == 4 ? 4 : 8;                   // row pairs per batch: all their loads in flight before the first MFMA
    int64_t r = r_begin;
    for (; r + 2 * U <= r_end; r += 2 * U) {             // full batches: wave-uniform row pointers, no row predicate
        const float* pr = pa + r * k;
        const float* vr = va + r * ldv;
        float av[U][2], bv[U][CT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < 2; ++t) av[u][t] = pr[(int64_t)(2 * u) * k + offa[t]];
#pragma unroll
            for (int c = 0; c < CT; ++c) bv[u][c] = vr[(int64_t)(2 * u) * ldv + offb[c]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < 2; ++t) av[u][t] = kok[t] ? av[u][t] : 0.f;
#pragma unroll
            for (int c = 0; c < CT; ++c) bv[u][c] = cok[c] ? bv[u][c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][t], bv[u][c], acc[t][c], 0, 0, 0);
    }
    for (; r < r_end; r += 2) {                          // the slice's last rows: per-lane row predicate
        int64_t row = r + hf;
        const bool rv = row < r_end;
        if (!rv) row = r_end - 1;
        float av[2], bv[CT];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float x = pa[row * k + (offa[t] - hf * k)];
            av[t] = (rv && kok[t]) ? x : 0.f;
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float x = va[row * ldv + col[c]];
            bv[c] = (rv && cok[c]) ? x : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[c], acc[t][c], 0, 0, 0);
    }
    float* dst = partial + ((int64_t)slice * h + a) * (int64_t)k * dk;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (!cok[c]) continue;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ky = 32 * (kb0 + t) + (i & 3) + 8 * (i >> 2) + 4 * hf;
                if (ky < k) dst[(int64_t)ky * dk + col[c]] = acc[t][c][i];
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// The scores + softmax kernel in the fp32-CLASS arithmetic of the pipelined kernels (split-bf16 x 3 on v_mfma_f32_32x32x16_bf16: every
// operand hi + lo, hi hi + hi lo + lo hi, fp32 accumulate; ~1e-5 per product) for the head widths those kernels do not take -- any
// dk % 16 == 0 up to 256 (the README recipe D = 768 / h = 4: dk = 192), up to 1024 keys.
//   scores_softmax_x3u: as scores_softmax_mfma; the Q tile is staged in LDS already split (hi plane | lo plane per row), the Kp rows are
//     split in registers (16 vector instructions per 16-deep step and key block, beside its three MFMAs): 214 us against 324 at
//     (30000, 500, 4, 192).
//   P^T V stays on pt_v_mfma (exact fp32): a split-in-registers P^T V was built and measured 2.2x SLOWER than the f32 form (684 us
//   against 313: five operand splits per six tile products, and 40 dword loads per 18 MFMAs) -- that product wants pre-split images,
//   which is what the pipelined kernels do.
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 xbf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 xbf16x4;
typedef __attribute__((ext_vector_type(8))) float mf32x8;
typedef __attribute__((ext_vector_type(4))) unsigned int xu32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int xu32x2;

__device__ __forceinline__ void x3u_split8(const mf32x8 x, xbf16x8& hi, xbf16x8& lo) {
    hi = __builtin_convertvector(x, xbf16x8);
    lo = __builtin_convertvector(x - __builtin_convertvector(hi, mf32x8), xbf16x8);
}

// RT = 32-row tiles per workgroup.  At RT = 1 every workgroup of 32 rows pulls the whole Kp of its head through L1 (16 key blocks x 24 KiB at
// k = 500, dk = 192: the texture path, not the matrix pipe, sets the pace); RT = 2 uses every Kp fragment -- and its split -- for two row tiles.
template <int KBW, int RT>
__global__ __launch_bounds__(256, (KBW * RT <= 4) ? 3 : 2) void scores_softmax_x3u_kernel(const float* __restrict__ q, const float* __restrict__ kp,
                                                                                          int64_t n, int k, int h, int dk, float scale,
                                                                                          float* __restrict__ p_out, float* __restrict__ lse, int64_t ldq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int pitch = 4 * dk + 16;             // bytes per row: hi plane (2 dk) | lo plane (2 dk) | pad; 4 banks apart row to row
    float* lst = reinterpret_cast<float*>(ldsb + 32 * RT * pitch);     // [RT][2][4][32]
    const int a = blockIdx.y;
    const int d_model = h * dk;
    const int64_t row0 = (int64_t)blockIdx.x * 32 * RT;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 31, hf = lane >> 5;
    const int dk4 = dk >> 2;
    for (int e = threadIdx.x; e < 32 * RT * dk4; e += 256) {
        const int r = e / dk4, c4 = e - r * dk4;
        const int64_t row = row0 + r;
        mf32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (row < n) val = *reinterpret_cast<const mf32x4*>(q + row * ldq + a * dk + 4 * c4);
        const xbf16x4 hi = __builtin_convertvector(val, xbf16x4);
        const xbf16x4 lo = __builtin_convertvector(val - __builtin_convertvector(hi, mf32x4), xbf16x4);
        *reinterpret_cast<xu32x2*>(ldsb + r * pitch + 8 * c4) = __builtin_bit_cast(xu32x2, hi);
        *reinterpret_cast<xu32x2*>(ldsb + r * pitch + 2 * dk + 8 * c4) = __builtin_bit_cast(xu32x2, lo);
    }
    __syncthreads();
    const int nkb = (k + 31) >> 5;
    const int nt = dk >> 4;                    // 16-deep steps
    mf32x16 S[KBW][RT];
    const unsigned char* qr = ldsb + j * pitch + 16 * hf;          // + 32 rt pitch: row tile rt
    auto step = [&](mf32x16 (&Sc)[RT], const mf32x4 x0, const mf32x4 x1, int t) __attribute__((always_inline)) {
        xbf16x8 kh, kl;
        x3u_split8(mf32x8{x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]}, kh, kl);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const xbf16x8 qh = __builtin_bit_cast(xbf16x8, *reinterpret_cast<const xu32x4*>(qr + 32 * rt * pitch + 32 * t));
            const xbf16x8 ql = __builtin_bit_cast(xbf16x8, *reinterpret_cast<const xu32x4*>(qr + 32 * rt * pitch + 2 * dk + 32 * t));
            Sc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql, Sc[rt], 0, 0, 0);
            Sc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh, Sc[rt], 0, 0, 0);
            Sc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh, Sc[rt], 0, 0, 0);
        }
    };
#pragma unroll
    for (int c = 0; c < KBW; ++c) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 16; ++i) S[c][rt][i] = 0.f;
        const int kb = w + 4 * c;
        if (kb < nkb) {
            int key = 32 * kb + j;
            if (key > k - 1) key = k - 1;
            const float* kr = kp + (int64_t)key * d_model + a * dk + 8 * hf;
            int t = 0;
            for (; t + 2 <= nt; t += 2) {      // two steps' Kp rows requested together (L2)
                mf32x4 x[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    x[u][0] = *reinterpret_cast<const mf32x4*>(kr + 16 * (t + u));
                    x[u][1] = *reinterpret_cast<const mf32x4*>(kr + 16 * (t + u) + 4);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) step(S[c], x[u][0], x[u][1], t + u);
            }
            for (; t < nt; ++t) {
                const mf32x4 x0 = *reinterpret_cast<const mf32x4*>(kr + 16 * t), x1 = *reinterpret_cast<const mf32x4*>(kr + 16 * t + 4);
                step(S[c], x0, x1, t);
            }
        }
    }
    float m[RT], l[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        m[rt] = -INFINITY;
#pragma unroll
        for (int c = 0; c < KBW; ++c) {
            const int kb = w + 4 * c;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * hf;
                S[c][rt][i] = (kb < nkb && key < k) ? S[c][rt][i] * scale : -INFINITY;
                m[rt] = fmaxf(m[rt], S[c][rt][i]);
            }
        }
        m[rt] = fmaxf(m[rt], __shfl_xor(m[rt], 32, 64));
        if (hf == 0) lst[rt * 256 + w * 32 + j] = m[rt];
    }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const float* ls = lst + rt * 256;
        m[rt] = fmaxf(fmaxf(ls[j], ls[32 + j]), fmaxf(ls[64 + j], ls[96 + j]));
        l[rt] = 0.f;
#pragma unroll
        for (int c = 0; c < KBW; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = __builtin_amdgcn_exp2f((S[c][rt][i] - m[rt]) * 1.44269504088896340736f);
                S[c][rt][i] = e;
                l[rt] += e;
            }
        l[rt] += __shfl_xor(l[rt], 32, 64);
        if (hf == 0) lst[rt * 256 + 128 + w * 32 + j] = l[rt];
    }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const float* ls = lst + rt * 256 + 128;
        const float lsum = ((ls[j] + ls[32 + j]) + ls[64 + j]) + ls[96 + j];
        const float inv = 1.0f / lsum;
        const int64_t row = row0 + 32 * rt + j;
        if (row < n) {
            float* prow = p_out + ((int64_t)a * n + row) * k;
            const bool vec = (k & 3) == 0 && (reinterpret_cast<uintptr_t>(p_out) & 15) == 0;
#pragma unroll
            for (int c = 0; c < KBW; ++c) {
                const int kb = w + 4 * c;
                if (kb < nkb) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int key0 = 32 * kb + 8 * q4 + 4 * hf;
                        const mf32x4 pv = {S[c][rt][4 * q4] * inv, S[c][rt][4 * q4 + 1] * inv, S[c][rt][4 * q4 + 2] * inv, S[c][rt][4 * q4 + 3] * inv};
                        if (vec && key0 + 4 <= k) {
                            *reinterpret_cast<mf32x4*>(prow + key0) = pv;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (key0 + e < k) prow[key0 + e] = pv[e];
                        }
                    }
                }
            }
            if (lse && w == 0 && hf == 0) lse[(int64_t)a * n + row] = m[rt] + logf(lsum);
        }
    }
}

__global__ __launch_bounds__(256) void reduce_slices_kernel(const float* __restrict__ partial, int slices, int k, int h,
                                                            int dk, float* __restrict__ out /*[k, h*dk]*/) {
    const int64_t total = (int64_t)h * k * dk;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int sl = 0; sl < slices; ++sl) s += partial[(int64_t)sl * total + e];
        int a = (int)(e / ((int64_t)k * dk));
        int64_t rem = e - (int64_t)a * k * dk;
        int key = (int)(rem / dk), col = (int)(rem - (int64_t)key * dk);
        out[(int64_t)key * (h * dk) + a * dk + col] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward (exact fp32).  With Pd = P o M (M = dropout keep-mask / (1 - p), or 1):
//   dV_a  = Pd_a dO_a                [n, dk]
//   dPd_a = V_a dO_a^T ; dP = dPd o M ; dS = P o (dP - rowsum(dP o P)) * scale      [n, k]
//   dQ_a  = dS_a Kp_a                [n, dk]
//   dKp_a = dS_a^T Q_a               [k, dk]   (contraction over the n rows: pt_v_kernel + reduce_slices on (dS, Q))
// bwd_ds mirrors scores_softmax (16 rows of one head per workgroup, dO streamed through LDS in 64-key chunks, a lane owns
// keys lane, lane+64, ...); bwd_dq_dv is an LDS-tiled fp32 GEMM pair over the key axis.
// ---------------------------------------------------------------------------------------------------------------
template <int KPL>
__global__ __launch_bounds__(256) void bwd_ds_kernel(const float* __restrict__ v, const float* __restrict__ dout,
                                                     const float* __restrict__ p /*[h,n,k]*/,
                                                     const float* __restrict__ mask /*[h,n,k] nullable*/, int64_t n, int k,
                                                     int h, int dk, float scale, float* __restrict__ ds /*[h,n,k]*/) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int dkp = (dk + 3) & ~3;
    const int kpitch = dkp + 4;
    float* lv = lds;                       // [16][dkp]   V rows of this workgroup
    float* ldo = lds + ROWS_PER_WG * dkp;  // [64][kpitch] dO rows of the current key chunk
    const int a = blockIdx.y;
    const int d_model = h * dk;
    const int64_t row0 = (int64_t)blockIdx.x * ROWS_PER_WG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < ROWS_PER_WG * dkp; e += 256) {
        int r = e / dkp, d = e - r * dkp;
        int64_t row = row0 + r;
        lv[e] = (row < n && d < dk) ? v[row * d_model + a * dk + d] : 0.f;
    }
    float dp[4][KPL];
#pragma unroll
    for (int c = 0; c < KPL; ++c) {
        __syncthreads();
        for (int e = threadIdx.x; e < KCHUNK * dkp; e += 256) {
            int j = e / dkp, d = e - j * dkp;
            int key = c * KCHUNK + j;
            ldo[j * kpitch + d] = (key < k && d < dk) ? dout[(int64_t)key * d_model + a * dk + d] : 0.f;
        }
        __syncthreads();
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float4* kv4 = reinterpret_cast<const float4*>(ldo + lane * kpitch);
        for (int d4 = 0; d4 < dkp / 4; ++d4) {
            float4 kv = kv4[d4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float4 qv = reinterpret_cast<const float4*>(lv + (wave * 4 + r) * dkp)[d4];
                acc[r] = fmaf(qv.x, kv.x, acc[r]);
                acc[r] = fmaf(qv.y, kv.y, acc[r]);
                acc[r] = fmaf(qv.z, kv.z, acc[r]);
                acc[r] = fmaf(qv.w, kv.w, acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) dp[r][c] = acc[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + wave * 4 + r;
        if (row >= n) continue;   // wave-uniform
        const int64_t base = ((int64_t)a * n + row) * k;
        float pr[KPL];
        float dsum = 0.f;
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int key = c * KCHUNK + lane;
            pr[c] = 0.f;
            if (key < k) {
                pr[c] = p[base + key];
                if (mask) dp[r][c] *= mask[base + key];
                dsum = fmaf(dp[r][c], pr[c], dsum);
            }
        }
        dsum = wave_sum(dsum);
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int key = c * KCHUNK + lane;
            if (key < k) ds[base + key] = pr[c] * (dp[r][c] - dsum) * scale;
        }
    }
}

// dV tile and dQ tile of 64 rows x 64 columns of one head: acc_v += (P o M)[rows, keys] dO[keys, cols],
// acc_q += dS[rows, keys] Kp[keys, cols]; keys in steps of 16 through LDS; 256 threads, 4x4 micro-tiles.
__global__ __launch_bounds__(256) void bwd_dq_dv_kernel(const float* __restrict__ p, const float* __restrict__ mask,
                                                        const float* __restrict__ ds, const float* __restrict__ dout,
                                                        const float* __restrict__ kp, int64_t n, int k, int h, int dk,
                                                        float* __restrict__ dq, float* __restrict__ dv) {
    __shared__ float lp[16][64 + 4];    // [key][row]  Pd^T tile
    __shared__ float lds_[16][64 + 4];  // [key][row]  dS^T tile
    __shared__ float ldo[16][64 + 4];   // [key][col]
    __shared__ float lkp[16][64 + 4];   // [key][col]
    const int a = blockIdx.z;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int d_model = h * dk;
    const int tr = threadIdx.x & 15, tc = threadIdx.x >> 4;
    float acc_v[4][4], acc_q[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc_v[i][j] = acc_q[i][j] = 0.f;
    // loaders: P / dS tile [64 rows][16 keys] read along keys (contiguous), stored transposed; dO / Kp tile [16 keys][64 cols]
    const int lrow = threadIdx.x >> 2;        // 0..63
    const int lkey = (threadIdx.x & 3) * 4;   // 0,4,8,12
    const int okey = threadIdx.x >> 4;        // 0..15
    const int ocol = (threadIdx.x & 15) * 4;  // 0..60
    for (int j0 = 0; j0 < k; j0 += 16) {
        const int64_t row = r0 + lrow;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int key = j0 + lkey + t;
            float pv = 0.f, sv = 0.f;
            if (row < n && key < k) {
                const int64_t e = ((int64_t)a * n + row) * k + key;
                pv = p[e];
                if (mask) pv *= mask[e];
                sv = ds[e];
            }
            lp[lkey + t][lrow] = pv;
            lds_[lkey + t][lrow] = sv;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int key = j0 + okey, col = c0 + ocol + t;
            const bool ok = key < k && col < dk;
            ldo[okey][ocol + t] = ok ? dout[(int64_t)key * d_model + a * dk + col] : 0.f;
            lkp[okey][ocol + t] = ok ? kp[(int64_t)key * d_model + a * dk + col] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float pv[4], sv[4], ov[4], kv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pv[i] = lp[kk][tr * 4 + i];
                sv[i] = lds_[kk][tr * 4 + i];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ov[j] = ldo[kk][tc * 4 + j];
                kv[j] = lkp[kk][tc * 4 + j];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc_v[i][j] = fmaf(pv[i], ov[j], acc_v[i][j]);
                    acc_q[i][j] = fmaf(sv[i], kv[j], acc_q[i][j]);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t row = r0 + tr * 4 + i;
        if (row >= n) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = c0 + tc * 4 + j;
            if (col < dk) {
                dv[row * d_model + a * dk + col] = acc_v[i][j];
                dq[row * d_model + a * dk + col] = acc_q[i][j];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The backward's two large kernels on the f32 matrix-core form (round 5; exact fp32, head widths % 8 == 0, k <= 1024 for dS).
//   bwd_ds_mfma: the twin of scores_softmax_mfma -- dP^T[32 keys, 32 rows] = dO V^T per key block (A = dO rows from L2, B = the V
//     tile in LDS), then per row dsum = sum_keys (dP o M) P over the four waves and dS = P (dP o M - dsum) scale, written once.
//   bwd_dq_dv_mfma: a wave owns 32 rows x 32 CT columns of dV AND of dQ: dV += (P o M)[rows, keys] dO[keys, cols], dQ += dS Kp; a
//     lane's four k-steps of P / dS are ONE float4 (the contraction index is walked as 8 T + 4 half + j), dO / Kp are coalesced
//     dword loads; 8 CT MFMAs per 2 + 8 CT loads.
// ---------------------------------------------------------------------------------------------------------------
template <int KBW>
__global__ __launch_bounds__(256, KBW <= 2 ? 3 : 2) void bwd_ds_mfma_kernel(const float* __restrict__ v, const float* __restrict__ dout,
                                                                            const float* __restrict__ p, const float* __restrict__ mask,
                                                                            int64_t n, int k, int h, int dk, float scale,
                                                                            float* __restrict__ ds, int64_t ldv /* row pitch of v */,
                                                                            const snf::DropoutState drop /* mask == null: regenerated here */) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int pitch = dk + 4;
    float* lv = lds;                           // [32][pitch]
    float* lst = lds + 32 * pitch;             // [4][32] per-wave partial dsum
    const int a = blockIdx.y;
    const int d_model = h * dk;
    const int64_t row0 = (int64_t)blockIdx.x * 32;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 31, hf = lane >> 5;
    const int dk4 = dk >> 2;
    for (int e = threadIdx.x; e < 32 * dk4; e += 256) {
        const int r = e / dk4, c4 = e - r * dk4;
        const int64_t row = row0 + r;
        mf32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (row < n) val = *reinterpret_cast<const mf32x4*>(v + row * ldv + a * dk + 4 * c4);
        *reinterpret_cast<mf32x4*>(lv + r * pitch + 4 * c4) = val;
    }
    __syncthreads();
    const int nkb = (k + 31) >> 5;
    const int nt = dk >> 3;
    mf32x16 D[KBW];
    const float* vr = lv + j * pitch + 4 * hf;
#pragma unroll
    for (int c = 0; c < KBW; ++c) {
#pragma unroll
        for (int i = 0; i < 16; ++i) D[c][i] = 0.f;
        const int kb = w + 4 * c;
        if (kb < nkb) {
            int key = 32 * kb + j;
            if (key > k - 1) key = k - 1;
            const float* orow = dout + (int64_t)key * d_model + a * dk + 4 * hf;
            int t = 0;
            for (; t + 4 <= nt; t += 4) {
                mf32x4 a4[4], b4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a4[u] = *reinterpret_cast<const mf32x4*>(orow + 8 * (t + u));
                    b4[u] = *reinterpret_cast<const mf32x4*>(vr + 8 * (t + u));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) D[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u][e], b4[u][e], D[c], 0, 0, 0);
            }
            for (; t < nt; ++t) {
                const mf32x4 a4 = *reinterpret_cast<const mf32x4*>(orow + 8 * t);
                const mf32x4 b4 = *reinterpret_cast<const mf32x4*>(vr + 8 * t);
#pragma unroll
                for (int e = 0; e < 4; ++e) D[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], D[c], 0, 0, 0);
            }
        }
    }
    // lane (row j, half hf): keys 32 kb + (i & 3) + 8 (i >> 2) + 4 hf of its row -- P (and the mask) in the same layout
    const int64_t row = row0 + j;
    const bool rvalid = row < n;
    const int64_t base = ((int64_t)a * n + (rvalid ? row : n - 1)) * k;
    const bool vec = (k & 3) == 0 && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(ds) |
                                       (mask ? reinterpret_cast<uintptr_t>(mask) : 0)) & 15) == 0;
    mf32x16 Pr[KBW];
    float dsum = 0.f;
#pragma unroll
    for (int c = 0; c < KBW; ++c) {
        const int kb = w + 4 * c;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int key0 = 32 * kb + 8 * q4 + 4 * hf;
            mf32x4 pv = {0.f, 0.f, 0.f, 0.f}, mv = {1.f, 1.f, 1.f, 1.f};
            if (kb < nkb) {
                // the dropout mask of the forward: a tensor, or (round 6) regenerated from the forward's Philox state -- the same
                // function of (head, row, key group) the forward kernel applied (philox.h), nothing to read
                if (!mask && drop.thresh) {
                    const snf::philox_f4 m4 = snf::dropout_mask4(drop, a, n, rvalid ? row : n - 1, k, key0);
                    mv = mf32x4{m4[0], m4[1], m4[2], m4[3]};
                }
                if (vec && key0 + 4 <= k) {
                    pv = *reinterpret_cast<const mf32x4*>(p + base + key0);
                    if (mask) mv = *reinterpret_cast<const mf32x4*>(mask + base + key0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (key0 + e < k) {
                            pv[e] = p[base + key0 + e];
                            if (mask) mv[e] = mask[base + key0 + e];
                        }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * q4 + e;
                D[c][i] *= mv[e];
                Pr[c][i] = pv[e];
                dsum = fmaf(D[c][i], pv[e], dsum);
            }
        }
    }
    dsum += __shfl_xor(dsum, 32, 64);
    if (hf == 0) lst[w * 32 + j] = dsum;
    __syncthreads();
    dsum = ((lst[j] + lst[32 + j]) + lst[64 + j]) + lst[96 + j];
    if (!rvalid) return;
#pragma unroll
    for (int c = 0; c < KBW; ++c) {
        const int kb = w + 4 * c;
        if (kb >= nkb) continue;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int key0 = 32 * kb + 8 * q4 + 4 * hf;
            mf32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = Pr[c][4 * q4 + e] * (D[c][4 * q4 + e] - dsum) * scale;
            if (vec && key0 + 4 <= k) {
                *reinterpret_cast<mf32x4*>(ds + base + key0) = o;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (key0 + e < k) ds[base + key0 + e] = o[e];
            }
        }
    }
}

template <int CT>
__global__ __launch_bounds__(256, 2) void bwd_dq_dv_mfma_kernel(const float* __restrict__ p, const float* __restrict__ mask,
                                                                const float* __restrict__ ds, const float* __restrict__ dout,
                                                                const float* __restrict__ kp, int64_t n, int k, int h, int dk,
                                                                float* __restrict__ dq, float* __restrict__ dv) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, hf = lane >> 5;
    const int a = blockIdx.z;
    const int d_model = h * dk;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + w) * 32;     // one wave = 32 rows
    if (row0 >= n) return;
    const int cb0 = CT * blockIdx.y;
    int64_t row = row0 + j;
    if (row > n - 1) row = n - 1;                                  // rows past the end: read the last row, never stored
    const float* prow = p + ((int64_t)a * n + row) * k + 4 * hf;   // lane's four k-steps of a group: keys 8 T + 4 hf .. + 3
    const float* srow = ds + ((int64_t)a * n + row) * k + 4 * hf;
    const float* mrow = mask ? mask + ((int64_t)a * n + row) * k + 4 * hf : nullptr;
    int col[CT];
    bool cok[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        col[c] = 32 * (cb0 + c) + j;
        cok[c] = col[c] < dk;
        if (!cok[c]) col[c] = dk - 1;
    }
    mf32x16 av[CT], aq[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) av[c][i] = 0.f, aq[c][i] = 0.f;
    const bool vec = (k & 3) == 0 && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(ds) |
                                       (mask ? reinterpret_cast<uintptr_t>(mask) : 0)) & 15) == 0;
    const float* ob = dout + a * dk;
    const float* kb_ = kp + a * dk;
    // operands of one group of 8 keys (4 MFMAs per accumulator); the NEXT group's are requested before the current group's MFMAs
    struct Grp {
        mf32x4 p4, s4;
        float od[4][CT], kd[4][CT];
    };
    auto load_grp = [&](int t8, Grp& g) __attribute__((always_inline)) {
        const int key0 = t8 + 4 * hf;
        g.p4 = mf32x4{0.f, 0.f, 0.f, 0.f}, g.s4 = mf32x4{0.f, 0.f, 0.f, 0.f};
        if (vec && key0 + 4 <= k) {
            g.p4 = *reinterpret_cast<const mf32x4*>(prow + t8);
            g.s4 = *reinterpret_cast<const mf32x4*>(srow + t8);
            if (mrow) g.p4 *= *reinterpret_cast<const mf32x4*>(mrow + t8);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (key0 + e < k) {
                    g.p4[e] = prow[t8 + e] * (mrow ? mrow[t8 + e] : 1.f);
                    g.s4[e] = srow[t8 + e];
                }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int key = key0 + e;
            const bool kok = key < k;
            if (!kok) key = k - 1;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float x = ob[(int64_t)key * d_model + col[c]], y = kb_[(int64_t)key * d_model + col[c]];
                g.od[e][c] = (kok && cok[c]) ? x : 0.f;
                g.kd[e][c] = (kok && cok[c]) ? y : 0.f;
            }
        }
    };
    Grp cur, nxt;
    load_grp(0, cur);
    for (int t8 = 0; t8 < k; t8 += 8) {
        if (t8 + 8 < k) load_grp(t8 + 8, nxt);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                av[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.p4[e], cur.od[e][c], av[c], 0, 0, 0);
                aq[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.s4[e], cur.kd[e][c], aq[c], 0, 0, 0);
            }
        cur = nxt;
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (!cok[c]) continue;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t r = row0 + (i & 3) + 8 * (i >> 2) + 4 * hf;
            if (r < n) {
                dv[r * d_model + a * dk + col[c]] = av[c][i];
                dq[r * d_model + a * dk + col[c]] = aq[c][i];
            }
        }
    }
}

// dQ / dV with the operands staged through LDS (round 5).  bwd_dq_dv_mfma_kernel above asks L2 for every operand of every MFMA group (34
// loads per 32 MFMAs, a quarter of each P / dS cache line used) and hides one group of latency: 489 us at config B for 128 us of
// matrix-pipe time.  Here a workgroup (4 waves x 32 rows of one head) walks the keys in chunks of 32: the chunk of dO and Kp
// ([32 keys, 32 CT columns] each) is staged once for the four waves, every wave stages its own [32 rows, 32 keys] tiles of P o M and
// dS -- all with coalesced 16-byte loads, requested a whole chunk (128 MFMAs per wave) ahead and parked in registers -- and the MFMA
// operands come out of LDS (A: one 16-byte read per four k-steps, row pitch 36 floats; B: conflict-free dword reads).
template <int CT, int MASK /* 0 none, 1 tensor, 2 regenerated from the Philox state */>
__global__ __launch_bounds__(256, 2) void bwd_dq_dv_lds_kernel(const float* __restrict__ p, const float* __restrict__ mask,
                                                               const float* __restrict__ ds, const float* __restrict__ dout,
                                                               const float* __restrict__ kp, int64_t n, int k, int h, int dk,
                                                               float* __restrict__ dq, float* __restrict__ dv, const snf::DropoutState drop) {
    constexpr int PA = 36, PB = 32 * CT;                      // row pitches (floats) of the A tiles and of the B chunks
    extern __shared__ __attribute__((aligned(16))) float dql[];
    float* lds_b = dql;                                        // [2][32][PB]: dO chunk | Kp chunk
    float* lds_a = dql + 2 * 32 * PB;                          // [4 waves][2][32][PA]: P o M | dS
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, hf = lane >> 5;
    const int a = blockIdx.z;
    const int d_model = h * dk;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + w) * 32;
    const int cb0 = CT * blockIdx.y;
    float* my_a = lds_a + w * 2 * 32 * PA;
    // staging roles.  A: lane (r8 = lane >> 3, q4 = lane & 7) fetches keys 4 q4 .. + 3 of rows 8 i + r8 (i = 0 .. 3) of its wave's tile.
    // B: thread t fetches float4 number t + 256 i (i < CT) of the chunk: key (t + 256 i) / (8 CT), columns 4 ((t + 256 i) % (8 CT)) ..
    const int r8 = lane >> 3, q4 = lane & 7;
    struct Stage {
        mf32x4 pa[4], sa[4], bo[CT], bk[CT];
    };
    auto load_stage = [&](int kc0, Stage& st) __attribute__((always_inline)) {
        // (opaque indices: hoisted out of the chunk loop the per-thread address terms did not fit the registers next to 128 accumulators -- hipcc
        // parked them in scratch and re-read each one behind a vmcnt(0) that also drained the previous operand load: recomputed per chunk instead)
        int tix = threadIdx.x, r8 = lane >> 3, q4 = lane & 7;
        asm volatile("" : "+v"(tix), "+v"(r8), "+v"(q4));
        const int key = kc0 + 4 * q4;
        const bool kok = key + 4 <= k;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int64_t row = row0 + 8 * i + r8;
            if (row > n - 1) row = n - 1;
            const int64_t off = ((int64_t)a * n + row) * k + key;
            st.pa[i] = mf32x4{0.f, 0.f, 0.f, 0.f}, st.sa[i] = mf32x4{0.f, 0.f, 0.f, 0.f};
            if (kok) {
                st.pa[i] = *reinterpret_cast<const mf32x4*>(p + off);
                st.sa[i] = *reinterpret_cast<const mf32x4*>(ds + off);
                if constexpr (MASK == 1) st.pa[i] *= *reinterpret_cast<const mf32x4*>(mask + off);
                if constexpr (MASK == 2) {
                    const snf::philox_f4 m4 = snf::dropout_mask4(drop, a, n, row, k, key);
                    st.pa[i] *= mf32x4{m4[0], m4[1], m4[2], m4[3]};
                }
            }
        }
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int idx = tix + 256 * i;
            const int bkey = kc0 + idx / (8 * CT), col = 32 * cb0 + 4 * (idx % (8 * CT));
            st.bo[i] = mf32x4{0.f, 0.f, 0.f, 0.f}, st.bk[i] = mf32x4{0.f, 0.f, 0.f, 0.f};
            if (bkey < k && col + 4 <= dk) {
                const int64_t off = (int64_t)bkey * d_model + a * dk + col;
                st.bo[i] = *reinterpret_cast<const mf32x4*>(dout + off);
                st.bk[i] = *reinterpret_cast<const mf32x4*>(kp + off);
            }
        }
    };
    auto park_stage = [&](const Stage& st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<mf32x4*>(my_a + (8 * i + r8) * PA + 4 * q4) = st.pa[i];
            *reinterpret_cast<mf32x4*>(my_a + 32 * PA + (8 * i + r8) * PA + 4 * q4) = st.sa[i];
        }
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int idx = threadIdx.x + 256 * i;
            const int o = (idx / (8 * CT)) * PB + 4 * (idx % (8 * CT));
            *reinterpret_cast<mf32x4*>(lds_b + o) = st.bo[i];
            *reinterpret_cast<mf32x4*>(lds_b + 32 * PB + o) = st.bk[i];
        }
    };
    mf32x16 av[CT], aq[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) av[c][i] = 0.f, aq[c][i] = 0.f;
    Stage st;
    load_stage(0, st);
    for (int kc0 = 0; kc0 < k; kc0 += 32) {
        __syncthreads();                                       // the previous chunk's fragment reads are done
        park_stage(st);
        __syncthreads();
        if (kc0 + 32 < k) load_stage(kc0 + 32, st);           // in flight under this chunk's 128 MFMAs
        const float* ar = my_a + j * PA + 4 * hf;              // lane (row j, half hf): keys 8 T + 4 hf .. + 3 of k-step group T
        const float* br = lds_b + (4 * hf) * PB + j;
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            if (kc0 + 8 * T >= k) break;                      // (wave-uniform) the last chunk's groups of 8 keys past k hold zeros
            const mf32x4 p4 = *reinterpret_cast<const mf32x4*>(ar + 8 * T);
            const mf32x4 s4 = *reinterpret_cast<const mf32x4*>(ar + 32 * PA + 8 * T);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const float od = br[(8 * T + e) * PB + 32 * c], kd = br[32 * PB + (8 * T + e) * PB + 32 * c];
                    av[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(p4[e], od, av[c], 0, 0, 0);
                    aq[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(s4[e], kd, aq[c], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int col = 32 * (cb0 + c) + j;
        if (col >= dk) continue;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t r = row0 + (i & 3) + 8 * (i >> 2) + 4 * hf;
            if (r < n) {
                dv[r * d_model + a * dk + col] = av[c][i];
                dq[r * d_model + a * dk + col] = aq[c][i];
            }
        }
    }
}

template <int CT>
int launch_dq_dv_lds(const float* p, const float* mask, const float* ds, const float* dout, const float* kp, int64_t n, int k, int h, int dk,
                     float* dq, float* dv, dim3 grid, hipStream_t s, const snf::DropoutState drop) {
    constexpr int lds = (2 * 32 * 32 * CT + 4 * 2 * 32 * 36) * (int)sizeof(float);
    static thread_local unsigned long long set_mask[3] = {0, 0, 0};    // devices that have the LDS opt-in, per instantiation
    const unsigned long long bit = snf::device_bit();
    const int which = mask ? 1 : drop.thresh ? 2 : 0;
    const void* fn = which == 1 ? reinterpret_cast<const void*>(bwd_dq_dv_lds_kernel<CT, 1>)
                                : which == 2 ? reinterpret_cast<const void*>(bwd_dq_dv_lds_kernel<CT, 2>)
                                             : reinterpret_cast<const void*>(bwd_dq_dv_lds_kernel<CT, 0>);
    if (!(set_mask[which] & bit)) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            snf::set_error("bwd_dq_dv_lds: cannot reserve %d bytes of LDS", lds);
            (void)hipGetLastError();
            return SNF_ELAUNCH;
        }
        set_mask[which] |= bit;
    }
    if (which == 1)
        hipLaunchKernelGGL((bwd_dq_dv_lds_kernel<CT, 1>), grid, dim3(256), lds, s, p, mask, ds, dout, kp, n, k, h, dk, dq, dv, drop);
    else if (which == 2)
        hipLaunchKernelGGL((bwd_dq_dv_lds_kernel<CT, 2>), grid, dim3(256), lds, s, p, mask, ds, dout, kp, n, k, h, dk, dq, dv, drop);
    else
        hipLaunchKernelGGL((bwd_dq_dv_lds_kernel<CT, 0>), grid, dim3(256), lds, s, p, mask, ds, dout, kp, n, k, h, dk, dq, dv, drop);
    return snf::check_launch("bwd_dq_dv_lds_kernel");
}

inline int generic_slices(int64_t n) {
    int64_t s = (n + 511) / 512;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    return (int)s;
}

// ---------------------------------------------------------------------------------------------------------------
// Ragged varlen form (exact fp32, any head width): many SMALL bags in one launch, every bag with its own number of keys
// (a bag shorter than Lambda selects all of its rows: K_b = N_b -- the MIL benchmark sets, MUSK / Elephant, live here).
// One workgroup per (bag, head); rows in chunks of 16: the chunk's probabilities go through LDS, every thread owns a fixed
// set of output elements O[key, col] (kept in LDS) and adds the chunk's rows in ascending order -> deterministic.
// desc[bag] = (row0, n, key0, k): first packed row / rows / first key row (= first output row) / keys of the bag.
// ---------------------------------------------------------------------------------------------------------------
constexpr int RG_ROWS = 16;
__global__ __launch_bounds__(256) void ragged_attn_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ v,
                                                          int64_t ldv, const float* __restrict__ kp, const int* __restrict__ desc,
                                                          int h, int dk, int kmax, float scale, float* __restrict__ out,
                                                          float* __restrict__ attn /*[h, T, kmax] nullable*/, int64_t n_total,
                                                          float* __restrict__ lse /*[h, T] nullable*/) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int bag = blockIdx.x, a = blockIdx.y;
    const int row0 = desc[4 * bag], n = desc[4 * bag + 1], key0 = desc[4 * bag + 2], k = desc[4 * bag + 3];
    const int64_t d = (int64_t)h * dk;
    float* lp = lds;                         // [RG_ROWS][kmax] scores, then probabilities
    float* lo = lds + RG_ROWS * kmax;        // [k][dk] output accumulators
    const int tid = threadIdx.x;
    const int nout = k * dk;
    for (int e = tid; e < nout; e += 256) lo[e] = 0.f;
    const int r = tid >> 4, l = tid & 15;    // phase 1: 16 threads per row, keys l, l + 16, ...
    const float* kpa = kp + (int64_t)key0 * d + a * dk;
    for (int c0 = 0; c0 < n; c0 += RG_ROWS) {
        const int row = c0 + r;
        const bool rvalid = row < n;
        const float* qr = q + (int64_t)(row0 + (rvalid ? row : n - 1)) * ldq + a * dk;
        float mx = -INFINITY;
        for (int j = l; j < k; j += 16) {
            const float* kj = kpa + (int64_t)j * d;
            float sc = 0.f;
            for (int c = 0; c < dk; ++c) sc = fmaf(qr[c], kj[c], sc);
            sc *= scale;
            lp[r * kmax + j] = sc;
            mx = fmaxf(mx, sc);
        }
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
        float sum = 0.f;
        for (int j = l; j < k; j += 16) {
            const float e = __expf(lp[r * kmax + j] - mx);
            lp[r * kmax + j] = e;
            sum += e;
        }
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
        const float inv = rvalid ? 1.f / sum : 0.f;
        for (int j = l; j < k; j += 16) {
            const float pj = lp[r * kmax + j] * inv;
            lp[r * kmax + j] = pj;
            if (attn && rvalid) attn[((int64_t)a * n_total + row0 + row) * kmax + j] = pj;
        }
        if (lse && rvalid && l == 0) lse[(int64_t)a * n_total + row0 + row] = mx + __logf(sum);
        __syncthreads();
        // phase 2: O[j, c] += sum over the chunk's rows of P[row, j] * V[row, c]
        const int rows = n - c0 < RG_ROWS ? n - c0 : RG_ROWS;
        for (int e = tid; e < nout; e += 256) {
            const int j = e / dk, c = e - j * dk;
            float acc = lo[e];
            for (int rr = 0; rr < rows; ++rr)
                acc = fmaf(lp[rr * kmax + j], v[(int64_t)(row0 + c0 + rr) * ldv + a * dk + c], acc);
            lo[e] = acc;
        }
        __syncthreads();
    }
    for (int e = tid; e < nout; e += 256) {
        const int j = e / dk, c = e - j * dk;
        out[(int64_t)(key0 + j) * d + a * dk + c] = lo[e];
    }
}

// P^T V on the f32 MFMA: one wave per (64 keys, 32 CT columns, row slice)
// P^T V (and dS^T Q) with the operands staged through LDS (round 5, the twin of bwd_dq_dv_lds_kernel): a workgroup owns ONE slice of the rows,
// 256 keys (wave w: keys 64 w .. + 63 of them) and 32 CT columns, and walks its rows in chunks of 32: the chunk of P ([32 rows, 256 keys]) and of V
// ([32 rows, 32 CT columns]) is fetched with coalesced 16-byte loads a whole chunk (128 MFMAs per wave) ahead, parked in registers, then in LDS,
// and the MFMA operands are conflict-free dword reads.  Against pt_v_mfma_kernel (a wave per slice, every operand straight from L2): V is read
// k / 256 times instead of k / 64, and nothing waits on L2 inside the MFMA stream.  Same row pairs in the same order: bit-identical partials.
template <int CT>
__global__ __launch_bounds__(256, 2) void pt_v_lds_kernel(const float* __restrict__ p /*[h,n,k]*/, const float* __restrict__ v, int64_t n, int k,
                                                          int h, int dk, int64_t rows_per_slice, float* __restrict__ partial /*[slices,h,k,dk]*/,
                                                          int ldv) {
    constexpr int PP = 256, PV = 32 * CT;                      // row pitches (floats) of the staged chunks
    __shared__ __attribute__((aligned(16))) float lp[32 * PP];
    __shared__ __attribute__((aligned(16))) float lv[32 * PV];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, hf = lane >> 5;
    const int a = blockIdx.z % h, slice = blockIdx.z / h;
    const int64_t r_begin = (int64_t)slice * rows_per_slice;
    int64_t r_end = r_begin + rows_per_slice;
    if (r_end > n) r_end = n;
    const int kt0 = 256 * blockIdx.x, cb0 = CT * blockIdx.y;
    const float* pa = p + (int64_t)a * n * k;
    const float* va = v + a * dk;
    struct Stage {
        mf32x4 pp[8], vv[CT];
    };
    auto load_stage = [&](int64_t r0, Stage& st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = threadIdx.x + 256 * i;
            const int64_t row = r0 + (idx >> 6);
            const int key = kt0 + 4 * (idx & 63);
            st.pp[i] = mf32x4{0.f, 0.f, 0.f, 0.f};
            if (row < r_end && key + 4 <= k) st.pp[i] = *reinterpret_cast<const mf32x4*>(pa + row * k + key);
        }
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int idx = threadIdx.x + 256 * i;
            const int64_t row = r0 + idx / (8 * CT);
            const int col = 32 * cb0 + 4 * (idx % (8 * CT));
            st.vv[i] = mf32x4{0.f, 0.f, 0.f, 0.f};
            if (row < r_end && col + 4 <= dk) st.vv[i] = *reinterpret_cast<const mf32x4*>(va + row * ldv + col);
        }
    };
    auto park_stage = [&](const Stage& st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = threadIdx.x + 256 * i;
            *reinterpret_cast<mf32x4*>(lp + (idx >> 6) * PP + 4 * (idx & 63)) = st.pp[i];
        }
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int idx = threadIdx.x + 256 * i;
            *reinterpret_cast<mf32x4*>(lv + (idx / (8 * CT)) * PV + 4 * (idx % (8 * CT))) = st.vv[i];
        }
    };
    mf32x16 acc[2][CT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][c][i] = 0.f;
    const bool active = kt0 + 64 * w < k;                      // (wave-uniform) this wave's 64 keys exist
    Stage st;
    if (r_begin < r_end) load_stage(r_begin, st);
    for (int64_t r0 = r_begin; r0 < r_end; r0 += 32) {
        __syncthreads();
        park_stage(st);
        __syncthreads();
        if (r0 + 32 < r_end) load_stage(r0 + 32, st);
        if (!active) continue;
        const float* ar = lp + hf * PP + 64 * w + j;            // lane (key j of block t, half hf): rows 2 T + hf
        const float* br = lv + hf * PV + j;
#pragma unroll
        for (int T = 0; T < 16; ++T) {
            if (r0 + 2 * T >= r_end) break;                    // (wave-uniform) the slice's last chunk
            const float a0 = ar[2 * T * PP], a1 = ar[2 * T * PP + 32];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float b = br[2 * T * PV + 32 * c];
                acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0][c], 0, 0, 0);
                acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1][c], 0, 0, 0);
            }
        }
    }
    if (!active) return;
    float* dst = partial + ((int64_t)slice * h + a) * (int64_t)k * dk;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int col = 32 * (cb0 + c) + j;
            if (col >= dk) continue;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ky = kt0 + 64 * w + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hf;
                if (ky < k) dst[(int64_t)ky * dk + col] = acc[t][c][i];
            }
        }
}

std::atomic<bool> g_pt_v_lds{true};      // snf_debug_exact_attn_mfma(2): the direct-from-L2 kernels (A / B partners of the LDS-staged ones)

int launch_pt_v_mfma(const float* p, const float* v, int64_t n, int k, int h, int dk, int64_t rows_per_slice, int slices, float* partial,
                     hipStream_t s, int ldv = 0) {
    if (ldv == 0) ldv = h * dk;
    const int ncb = (dk + 31) / 32;
    if (g_pt_v_lds && k % 4 == 0 && dk % 4 == 0 && ldv % 4 == 0 && (rows_per_slice & 1) == 0 &&
        ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(v)) & 15) == 0 && (int64_t)slices * h <= 65535) {
        const int ctl = ncb % 3 == 0 ? 3 : (ncb % 4 == 0 || ncb > 4) ? 4 : ncb;
        dim3 grid((unsigned)((k + 255) / 256), (unsigned)((ncb + ctl - 1) / ctl), (unsigned)(slices * h));
        switch (ctl) {
            case 1: hipLaunchKernelGGL((pt_v_lds_kernel<1>), grid, dim3(256), 0, s, p, v, n, k, h, dk, rows_per_slice, partial, ldv); break;
            case 2: hipLaunchKernelGGL((pt_v_lds_kernel<2>), grid, dim3(256), 0, s, p, v, n, k, h, dk, rows_per_slice, partial, ldv); break;
            case 3: hipLaunchKernelGGL((pt_v_lds_kernel<3>), grid, dim3(256), 0, s, p, v, n, k, h, dk, rows_per_slice, partial, ldv); break;
            default: hipLaunchKernelGGL((pt_v_lds_kernel<4>), grid, dim3(256), 0, s, p, v, n, k, h, dk, rows_per_slice, partial, ldv); break;
        }
        return snf::check_launch("pt_v_lds_kernel");
    }
    const int ct = ncb % 3 == 0 ? 3 : (ncb % 4 == 0 || ncb > 4) ? 4 : ncb;          // 96 / 192 -> 3, 128 / 256 -> 4, 64 -> 2, 32 -> 1
    dim3 grid((unsigned)(((k + 31) / 32 + 1) / 2), (unsigned)((ncb + ct - 1) / ct), (unsigned)(((slices + 3) / 4) * h));
    switch (ct) {
        case 1: hipLaunchKernelGGL((pt_v_mfma_kernel<1>), grid, dim3(256), 0, s, p, v, n, k, h, dk, rows_per_slice, slices, partial, ldv); break;
        case 2: hipLaunchKernelGGL((pt_v_mfma_kernel<2>), grid, dim3(256), 0, s, p, v, n, k, h, dk, rows_per_slice, slices, partial, ldv); break;
        case 3: hipLaunchKernelGGL((pt_v_mfma_kernel<3>), grid, dim3(256), 0, s, p, v, n, k, h, dk, rows_per_slice, slices, partial, ldv); break;
        default: hipLaunchKernelGGL((pt_v_mfma_kernel<4>), grid, dim3(256), 0, s, p, v, n, k, h, dk, rows_per_slice, slices, partial, ldv); break;
    }
    return snf::check_launch("pt_v_mfma_kernel");
}

// development / test switch (snf_debug_exact_attn_mfma): false = the round-1 vector-ALU kernels for every shape
std::atomic<bool> g_exact_mfma{true};
// ... and 2 = the matrix-core forms with the dQ / dV kernel that reads its operands straight from L2 (the A / B partner of the LDS-staged one)
std::atomic<bool> g_dq_dv_lds{true};

}  // namespace

namespace snf {
size_t generic_attn_workspace_bytes(int64_t n, int k, int h, int dk) {
    size_t pbytes = (size_t)h * (size_t)n * (size_t)k * sizeof(float);
    size_t part = (size_t)generic_slices(n) * h * (size_t)k * dk * sizeof(float);
    return ((pbytes + 255) & ~(size_t)255) + part;
}
}  // namespace snf

extern "C" {

size_t snf_sparse_attn_bwd_workspace_bytes(int64_t n, int k, int h, int dk) {
    if (n < 1 || k < 1 || h < 1 || dk < 1) return 0;
    return snf::generic_attn_workspace_bytes(n, k, h, dk);   // dS [h,n,k] + the dKp slice partials
}

int snf_sparse_attn_bwd_f32(const float* q, const float* kp, const float* v, const float* p, const float* mask,
                            const float* dout, int64_t n, int k, int h, int dk, float scale, float* dq, float* dkp,
                            float* dv, void* workspace, size_t workspace_bytes, snf_stream_t stream) {
    return snf_sparse_attn_bwd_ld_f32(q, (int64_t)h * dk, kp, v, (int64_t)h * dk, p, mask, dout, n, k, h, dk, scale, dq, dkp, dv, workspace,
                                      workspace_bytes, stream);
}

static int bwd_impl(const float* q, int64_t ldq, const float* kp, const float* v, int64_t ldv, const float* p, const float* mask,
                    const snf::DropoutState drop, const float* dout, int64_t n, int k, int h, int dk, float scale, float* dq, float* dkp,
                    float* dv, void* workspace, size_t workspace_bytes, snf_stream_t stream);

int snf_sparse_attn_bwd_ld_f32(const float* q, int64_t ldq, const float* kp, const float* v, int64_t ldv, const float* p, const float* mask,
                               const float* dout, int64_t n, int k, int h, int dk, float scale, float* dq, float* dkp,
                               float* dv, void* workspace, size_t workspace_bytes, snf_stream_t stream) {
    return bwd_impl(q, ldq, kp, v, ldv, p, mask, snf::make_dropout(0.f, 0, 0), dout, n, k, h, dk, scale, dq, dkp, dv, workspace, workspace_bytes,
                    stream);
}

// the same backward with the forward's dropout mask REGENERATED in the kernels from its Philox state (dropout_p, seed, offset: what
// snf_sparse_attn_fwd_x3_dropout was given) instead of read from a [h, n, k] tensor; only on the matrix-core kernels' route
int snf_sparse_attn_bwd_dropout_f32(const float* q, int64_t ldq, const float* kp, const float* v, int64_t ldv, const float* p, float dropout_p,
                                    uint64_t seed, uint64_t offset, const float* dout, int64_t n, int k, int h, int dk, float scale, float* dq,
                                    float* dkp, float* dv, void* workspace, size_t workspace_bytes, snf_stream_t stream) {
    SNF_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "snf_sparse_attn_bwd_dropout_f32: dropout_p = %f", (double)dropout_p);
    const bool route = g_exact_mfma && g_dq_dv_lds && k <= 1024 && k % 4 == 0 && dk % 8 == 0 && ldv % 4 == 0 && v && dout && p && kp &&
                       ((reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(p) |
                         reinterpret_cast<uintptr_t>(kp)) & 15) == 0;
    if (!route) {
        snf::set_error("snf_sparse_attn_bwd_dropout_f32: k=%d dk=%d outside the matrix-core route (k <= 1024, k %% 4 == 0, dk %% 8 == 0, 16-byte "
                       "aligned operands): hand the mask over as a tensor (snf_dropout_mask_f32 + snf_sparse_attn_bwd_ld_f32)", k, dk);
        return SNF_EUNSUPPORTED;
    }
    return bwd_impl(q, ldq, kp, v, ldv, p, nullptr, snf::make_dropout(dropout_p, seed, offset), dout, n, k, h, dk, scale, dq, dkp, dv, workspace,
                    workspace_bytes, stream);
}

static int bwd_impl(const float* q, int64_t ldq, const float* kp, const float* v, int64_t ldv, const float* p, const float* mask,
                    const snf::DropoutState drop, const float* dout, int64_t n, int k, int h, int dk, float scale, float* dq, float* dkp,
                    float* dv, void* workspace, size_t workspace_bytes, snf_stream_t stream) {
    SNF_REQUIRE(q && kp && v && p && dout && dq && dkp && dv, "snf_sparse_attn_bwd_f32: null pointer");
    SNF_REQUIRE(ldq >= (int64_t)h * dk && ldv >= (int64_t)h * dk && ldq < (1 << 24) && ldv < (1 << 24),
                "snf_sparse_attn_bwd_ld_f32: bad row pitch ldq=%lld ldv=%lld", (long long)ldq, (long long)ldv);
    SNF_REQUIRE(n >= 1 && k >= 1 && h >= 1 && dk >= 1, "snf_sparse_attn_bwd_f32: bad shape n=%lld k=%d h=%d dk=%d",
                (long long)n, k, h, dk);
    SNF_REQUIRE(k <= 2048 && dk <= 256 && h <= 65535, "snf_sparse_attn_bwd_f32: k=%d / dk=%d / h=%d out of range", k, dk, h);
    const size_t need = snf::generic_attn_workspace_bytes(n, k, h, dk);
    if (!workspace || workspace_bytes < need) {
        snf::set_error("snf_sparse_attn_bwd_f32: workspace %zu < %zu", workspace_bytes, need);
        return SNF_EWORKSPACE;
    }
    const size_t pbytes = ((size_t)h * (size_t)n * (size_t)k * sizeof(float) + 255) & ~(size_t)255;
    float* ds = reinterpret_cast<float*>(workspace);
    float* partial = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + pbytes);
    hipStream_t s = snf::as_stream(stream);
    const int dk4 = (dk + 3) & ~3;
    const size_t lds = (size_t)(ROWS_PER_WG * dk4 + KCHUNK * (dk4 + 4)) * sizeof(float);
    int rc;
    const bool mfma_ok = g_exact_mfma && dk % 8 == 0 && ldv % 4 == 0 &&
                         ((reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(dout)) & 15) == 0;
    if ((ldq != (int64_t)h * dk || ldv != (int64_t)h * dk) && !(mfma_ok && k <= 1024)) {
        snf::set_error("snf_sparse_attn_bwd_ld_f32: row pitches other than h dk need the matrix-core kernels (dk %% 8 == 0, k <= 1024, 16-byte rows)");
        return SNF_EUNSUPPORTED;
    }
    if (mfma_ok && k <= 1024) {
        const size_t lds_m = (size_t)(32 * (dk + 4) + 128) * sizeof(float);
        dim3 grid1((unsigned)((n + 31) / 32), (unsigned)h);
        const int kbw = ((k + 31) / 32 + 3) / 4;
#define LAUNCH_DSM(KBW) \
    hipLaunchKernelGGL((bwd_ds_mfma_kernel<KBW>), grid1, dim3(256), lds_m, s, v, dout, p, mask, n, k, h, dk, scale, ds, ldv, drop)
        if (kbw <= 1) LAUNCH_DSM(1);
        else if (kbw <= 2) LAUNCH_DSM(2);
        else if (kbw <= 4) LAUNCH_DSM(4);
        else LAUNCH_DSM(8);
#undef LAUNCH_DSM
        rc = snf::check_launch("bwd_ds_mfma_kernel");
    } else {
        dim3 grid1((unsigned)((n + ROWS_PER_WG - 1) / ROWS_PER_WG), (unsigned)h);
        const int kpl = (k + 63) / 64;
#define LAUNCH_DS(KPL) \
    hipLaunchKernelGGL((bwd_ds_kernel<KPL>), grid1, dim3(256), lds, s, v, dout, p, mask, n, k, h, dk, scale, ds)
        if (kpl <= 1) LAUNCH_DS(1);
        else if (kpl <= 2) LAUNCH_DS(2);
        else if (kpl <= 4) LAUNCH_DS(4);
        else if (kpl <= 8) LAUNCH_DS(8);
        else if (kpl <= 16) LAUNCH_DS(16);
        else LAUNCH_DS(32);
#undef LAUNCH_DS
        rc = snf::check_launch("bwd_ds_kernel");
    }
    if (rc) return rc;
    if (g_exact_mfma) {
        const int ncb = (dk + 31) / 32;
        const int ct = ncb % 3 == 0 ? 3 : (ncb >= 4 ? 4 : ncb);
        dim3 grid2((unsigned)((n + 127) / 128), (unsigned)((ncb + ct - 1) / ct), (unsigned)h);
        const bool lds_ok = g_dq_dv_lds && k % 4 == 0 && dk % 4 == 0 &&
                            ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(ds) | reinterpret_cast<uintptr_t>(dout) |
                              reinterpret_cast<uintptr_t>(kp) | (mask ? reinterpret_cast<uintptr_t>(mask) : 0)) & 15) == 0;
        if (lds_ok) {
            switch (ct) {
                case 1: rc = launch_dq_dv_lds<1>(p, mask, ds, dout, kp, n, k, h, dk, dq, dv, grid2, s, drop); break;
                case 2: rc = launch_dq_dv_lds<2>(p, mask, ds, dout, kp, n, k, h, dk, dq, dv, grid2, s, drop); break;
                case 3: rc = launch_dq_dv_lds<3>(p, mask, ds, dout, kp, n, k, h, dk, dq, dv, grid2, s, drop); break;
                default: rc = launch_dq_dv_lds<4>(p, mask, ds, dout, kp, n, k, h, dk, dq, dv, grid2, s, drop); break;
            }
            if (rc) return rc;
        } else
        switch (ct) {
            case 1: hipLaunchKernelGGL((bwd_dq_dv_mfma_kernel<1>), grid2, dim3(256), 0, s, p, mask, ds, dout, kp, n, k, h, dk, dq, dv); break;
            case 2: hipLaunchKernelGGL((bwd_dq_dv_mfma_kernel<2>), grid2, dim3(256), 0, s, p, mask, ds, dout, kp, n, k, h, dk, dq, dv); break;
            case 3: hipLaunchKernelGGL((bwd_dq_dv_mfma_kernel<3>), grid2, dim3(256), 0, s, p, mask, ds, dout, kp, n, k, h, dk, dq, dv); break;
            default: hipLaunchKernelGGL((bwd_dq_dv_mfma_kernel<4>), grid2, dim3(256), 0, s, p, mask, ds, dout, kp, n, k, h, dk, dq, dv); break;
        }
        rc = snf::check_launch("bwd_dq_dv_mfma_kernel");
    } else {
        dim3 grid2((unsigned)((n + 63) / 64), (unsigned)((dk + 63) / 64), (unsigned)h);
        hipLaunchKernelGGL(bwd_dq_dv_kernel, grid2, dim3(256), 0, s, p, mask, ds, dout, kp, n, k, h, dk, dq, dv);
        rc = snf::check_launch("bwd_dq_dv_kernel");
    }
    if (rc) return rc;
    // dKp = dS^T Q: the forward's P^T V machinery on (dS, Q)
    const int slices = generic_slices(n);
    const int64_t rows_per_slice = (((n + slices - 1) / slices) + 15) & ~(int64_t)15;
    if (g_exact_mfma)
        rc = launch_pt_v_mfma(ds, q, n, k, h, dk, rows_per_slice, slices, partial, s, (int)ldq);
    else {
        dim3 grid3((unsigned)((k + 63) / 64), (unsigned)((dk + 63) / 64), (unsigned)(slices * h));
        hipLaunchKernelGGL(pt_v_kernel, grid3, dim3(256), 0, s, ds, q, n, k, h, dk, rows_per_slice, partial);
        rc = snf::check_launch("pt_v_kernel(dS, Q)");
    }
    if (rc) return rc;
    const int64_t total = (int64_t)h * k * dk;
    int rgrid = (int)((total + 255) / 256);
    if (rgrid > 2048) rgrid = 2048;
    hipLaunchKernelGGL(reduce_slices_kernel, dim3(rgrid), dim3(256), 0, s, partial, slices, k, h, dk, dkp);
    return snf::check_launch("reduce_slices_kernel");
}

int snf_sparse_attn_dkp_f32(const float* ds, const float* q, int64_t n, int k, int h, int dk, float* dkp, void* workspace,
                            size_t workspace_bytes, snf_stream_t stream) {
    SNF_REQUIRE(ds && q && dkp, "snf_sparse_attn_dkp_f32: null pointer");
    SNF_REQUIRE(n >= 1 && k >= 1 && h >= 1 && dk >= 1 && h <= 65535, "snf_sparse_attn_dkp_f32: bad shape");
    const int slices = generic_slices(n);
    const size_t need = (size_t)slices * h * (size_t)k * dk * sizeof(float);
    if (!workspace || workspace_bytes < need) {
        snf::set_error("snf_sparse_attn_dkp_f32: workspace %zu < %zu", workspace_bytes, need);
        return SNF_EWORKSPACE;
    }
    float* partial = reinterpret_cast<float*>(workspace);
    hipStream_t s = snf::as_stream(stream);
    const int64_t rows_per_slice = (((n + slices - 1) / slices) + 15) & ~(int64_t)15;
    int rc;
    if (g_exact_mfma)
        rc = launch_pt_v_mfma(ds, q, n, k, h, dk, rows_per_slice, slices, partial, s);
    else {
        dim3 grid((unsigned)((k + 63) / 64), (unsigned)((dk + 63) / 64), (unsigned)(slices * h));
        hipLaunchKernelGGL(pt_v_kernel, grid, dim3(256), 0, s, ds, q, n, k, h, dk, rows_per_slice, partial);
        rc = snf::check_launch("pt_v_kernel(dS, Q)");
    }
    if (rc) return rc;
    const int64_t total = (int64_t)h * k * dk;
    int rgrid = (int)((total + 255) / 256);
    if (rgrid > 2048) rgrid = 2048;
    hipLaunchKernelGGL(reduce_slices_kernel, dim3(rgrid), dim3(256), 0, s, partial, slices, k, h, dk, dkp);
    return snf::check_launch("reduce_slices_kernel");
}

int snf_sparse_attn_fwd_f32(const float* q, const float* kp, const float* v, int64_t n, int k, int h, int dk, float scale,
                            float* out, float* attn, float* lse, void* workspace, size_t workspace_bytes,
                            snf_stream_t stream) {
    SNF_REQUIRE(q && kp && v && out, "snf_sparse_attn_fwd_f32: null pointer");
    SNF_REQUIRE(n >= 1 && k >= 1 && h >= 1 && dk >= 1, "snf_sparse_attn_fwd_f32: bad shape n=%lld k=%d h=%d dk=%d",
                (long long)n, k, h, dk);
    SNF_REQUIRE(k <= 2048, "snf_sparse_attn_fwd_f32: k=%d > 2048 not supported", k);
    SNF_REQUIRE(dk <= 256, "snf_sparse_attn_fwd_f32: dk=%d > 256 not supported", dk);
    SNF_REQUIRE(h <= 65535, "snf_sparse_attn_fwd_f32: too many heads");
    const size_t pbytes = ((size_t)h * (size_t)n * (size_t)k * sizeof(float) + 255) & ~(size_t)255;
    const int slices = generic_slices(n);
    const size_t part_bytes = (size_t)slices * h * (size_t)k * dk * sizeof(float);
    const size_t need = (attn ? 0 : pbytes) + part_bytes;
    if (!workspace || workspace_bytes < need) {
        snf::set_error("snf_sparse_attn_fwd_f32: workspace %zu < %zu", workspace_bytes, need);
        return SNF_EWORKSPACE;
    }
    float* p = attn ? attn : reinterpret_cast<float*>(workspace);
    float* partial = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + (attn ? 0 : pbytes));
    hipStream_t s = snf::as_stream(stream);

    int rc;
    const bool mfma_scores = g_exact_mfma && dk % 8 == 0 && k <= 1024 &&
                             ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(kp)) & 15) == 0;
    if (mfma_scores) {   // exact fp32 on the matrix cores
        const size_t lds = (size_t)(32 * (dk + 4) + 256) * sizeof(float);
        dim3 grid1((unsigned)((n + 31) / 32), (unsigned)h);
        const int kbw = ((k + 31) / 32 + 3) / 4;
#define LAUNCH_SM(KBW)                                                                                                                 \
    hipLaunchKernelGGL((scores_softmax_mfma_kernel<KBW>), grid1, dim3(256), lds, s, q, kp, n, k, h, dk, scale, p, lse)
        if (kbw <= 1) LAUNCH_SM(1);
        else if (kbw <= 2) LAUNCH_SM(2);
        else if (kbw <= 4) LAUNCH_SM(4);
        else LAUNCH_SM(8);
#undef LAUNCH_SM
        rc = snf::check_launch("scores_softmax_mfma_kernel");
    } else {
        const int dkp = (dk + 3) & ~3;
        const size_t lds = (size_t)(ROWS_PER_WG * dkp + KCHUNK * (dkp + 4)) * sizeof(float);
        dim3 grid1((unsigned)((n + ROWS_PER_WG - 1) / ROWS_PER_WG), (unsigned)h);
        const int kpl = (k + 63) / 64;
#define LAUNCH_SS(KPL)                                                                                              \
    hipLaunchKernelGGL((scores_softmax_kernel<KPL>), grid1, dim3(256), lds, s, q, kp, n, k, h, dk, scale, p, lse)
        if (kpl <= 1) LAUNCH_SS(1);
        else if (kpl <= 2) LAUNCH_SS(2);
        else if (kpl <= 4) LAUNCH_SS(4);
        else if (kpl <= 8) LAUNCH_SS(8);
        else if (kpl <= 16) LAUNCH_SS(16);
        else LAUNCH_SS(32);
#undef LAUNCH_SS
        rc = snf::check_launch("scores_softmax_kernel");
    }
    if (rc) return rc;

    const int64_t rows_per_slice = (((n + slices - 1) / slices) + 15) & ~(int64_t)15;
    if (g_exact_mfma)
        rc = launch_pt_v_mfma(p, v, n, k, h, dk, rows_per_slice, slices, partial, s);
    else {
        dim3 grid2((unsigned)((k + 63) / 64), (unsigned)((dk + 63) / 64), (unsigned)(slices * h));
        hipLaunchKernelGGL(pt_v_kernel, grid2, dim3(256), 0, s, p, v, n, k, h, dk, rows_per_slice, partial);
        rc = snf::check_launch("pt_v_kernel");
    }
    if (rc) return rc;
    const int64_t total = (int64_t)h * k * dk;
    int rgrid = (int)((total + 255) / 256);
    if (rgrid > 2048) rgrid = 2048;
    hipLaunchKernelGGL(reduce_slices_kernel, dim3(rgrid), dim3(256), 0, s, partial, slices, k, h, dk, out);
    return snf::check_launch("reduce_slices_kernel");
}

// fp32-CLASS attention for head widths outside the pipelined kernels: scores + softmax in split-bf16 x 3 on the bf16 matrix cores
// (operands split on the fly), P^T V exact on the f32 matrix cores.  dk % 16 == 0, dk <= 256, k <= 1024; same outputs / workspace as snf_sparse_attn_fwd_f32.
int snf_sparse_attn_fwd_x3u_f32(const float* q, int64_t ldq, const float* kp, const float* v, int64_t ldv, int64_t n, int k, int h, int dk,
                                float scale, float* out, float* attn, float* lse, void* workspace, size_t workspace_bytes,
                                snf_stream_t stream) {
    SNF_REQUIRE(q && kp && v && out, "snf_sparse_attn_fwd_x3u_f32: null pointer");
    SNF_REQUIRE(n >= 1 && k >= 1 && h >= 1 && h <= 65535, "snf_sparse_attn_fwd_x3u_f32: bad shape n=%lld k=%d h=%d", (long long)n, k, h);
    SNF_REQUIRE(ldq >= (int64_t)h * dk && ldv >= (int64_t)h * dk && ldq % 4 == 0 && ldv < (1 << 24),
                "snf_sparse_attn_fwd_x3u_f32: bad row pitch ldq=%lld ldv=%lld (>= h dk, ldq %% 4 == 0)", (long long)ldq, (long long)ldv);
    if (dk < 16 || dk % 16 || dk > 256 || k > 1024 || ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(kp)) & 15)) {
        snf::set_error("snf_sparse_attn_fwd_x3u_f32: dk=%d k=%d outside the kernel (dk %% 16 == 0, dk <= 256, k <= 1024, 16-byte aligned q / kp)", dk, k);
        return SNF_EUNSUPPORTED;
    }
    const size_t pbytes = ((size_t)h * (size_t)n * (size_t)k * sizeof(float) + 255) & ~(size_t)255;
    const int slices = generic_slices(n);
    const size_t need = (attn ? 0 : pbytes) + (size_t)slices * h * (size_t)k * dk * sizeof(float);
    if (!workspace || workspace_bytes < need) {
        snf::set_error("snf_sparse_attn_fwd_x3u_f32: workspace %zu < %zu", workspace_bytes, need);
        return SNF_EWORKSPACE;
    }
    float* p = attn ? attn : reinterpret_cast<float*>(workspace);
    float* partial = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + (attn ? 0 : pbytes));
    hipStream_t s = snf::as_stream(stream);
    // two row tiles per workgroup (every Kp fragment used twice) once the bag gives every CU a few such workgroups; up to 4 key blocks per wave
    const int kbw = ((k + 31) / 32 + 3) / 4;
    const int rt = (kbw <= 4 && n * h >= (int64_t)64 * 4 * snf::cu_count() && 64 * (4 * dk + 16) + 2048 <= 64 * 1024) ? 2 : 1;
    const size_t lds = (size_t)32 * rt * (4 * dk + 16) + (size_t)rt * 256 * sizeof(float);
    dim3 grid1((unsigned)((n + 32 * rt - 1) / (32 * rt)), (unsigned)h);
#define LAUNCH_SX(KBW, RT) hipLaunchKernelGGL((scores_softmax_x3u_kernel<KBW, RT>), grid1, dim3(256), lds, s, q, kp, n, k, h, dk, scale, p, lse, ldq)
    if (rt == 2) {
        if (kbw <= 1) LAUNCH_SX(1, 2);
        else if (kbw <= 2) LAUNCH_SX(2, 2);
        else LAUNCH_SX(4, 2);
    } else {
        if (kbw <= 1) LAUNCH_SX(1, 1);
        else if (kbw <= 2) LAUNCH_SX(2, 1);
        else if (kbw <= 4) LAUNCH_SX(4, 1);
        else LAUNCH_SX(8, 1);
    }
#undef LAUNCH_SX
    int rc = snf::check_launch("scores_softmax_x3u_kernel");
    if (rc) return rc;
    const int64_t rows_per_slice = (((n + slices - 1) / slices) + 15) & ~(int64_t)15;
    rc = launch_pt_v_mfma(p, v, n, k, h, dk, rows_per_slice, slices, partial, s, (int)ldv);
    if (rc) return rc;
    const int64_t total = (int64_t)h * k * dk;
    int rgrid = (int)((total + 255) / 256);
    if (rgrid > 2048) rgrid = 2048;
    hipLaunchKernelGGL(reduce_slices_kernel, dim3(rgrid), dim3(256), 0, s, partial, slices, k, h, dk, out);
    return snf::check_launch("reduce_slices_kernel");
}

// Ragged varlen attention, exact fp32 (see ragged_attn_kernel).  desc_dev [bags][4] int32 = (row0, n, key0, k) per bag in DEVICE
// memory; q, v [T, ld] f32, kp [sum k, h * dk] f32; out [sum k, h * dk]; attn [h, T, kmax] / lse [h, T] nullable (a bag's A is
// attn[:, row0 : row0 + n, :k]).  Limits: kmax <= 256, (16 kmax + kmax dk) floats of LDS <= 160 KiB, i.e. dk <= 144 at kmax = 256.
int snf_sparse_attn_fwd_ragged_f32(const float* q, int64_t ldq, const float* v, int64_t ldv, const float* kp, const int32_t* desc_dev,
                                   int bags, int64_t n_total, int kmax, int h, int dk, float scale, float* out, float* attn,
                                   float* lse, snf_stream_t stream) {
    SNF_REQUIRE(q && v && kp && out && desc_dev, "snf_sparse_attn_fwd_ragged_f32: null pointer");
    SNF_REQUIRE(bags >= 1 && bags <= 0x7fffffff && h >= 1 && h <= 65535 && dk >= 1 && kmax >= 1 && n_total >= 1,
                "snf_sparse_attn_fwd_ragged_f32: bad shape");
    const size_t lds = ((size_t)RG_ROWS * kmax + (size_t)kmax * dk) * sizeof(float);
    if (kmax > 256 || lds > 160 * 1024) {
        snf::set_error("snf_sparse_attn_fwd_ragged_f32: kmax=%d dk=%d needs %zu bytes of LDS (limit 160 KiB, kmax <= 256)", kmax, dk,
                       lds);
        return SNF_EUNSUPPORTED;
    }
    SNF_REQUIRE(ldq >= (int64_t)h * dk && ldv >= (int64_t)h * dk, "snf_sparse_attn_fwd_ragged_f32: row pitch below h * dk");
    static thread_local size_t lds_set = 0;
    static thread_local unsigned long long lds_dev = ~0ull;          // the opt-in is per device
    if (lds_dev != snf::device_bit()) lds_set = 0, lds_dev = snf::device_bit();
    if (lds > lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(ragged_attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            snf::set_error("snf_sparse_attn_fwd_ragged_f32: cannot reserve %zu bytes of LDS", lds);
            (void)hipGetLastError();
            return SNF_ELAUNCH;
        }
        lds_set = lds;
    }
    hipLaunchKernelGGL(ragged_attn_kernel, dim3((unsigned)bags, (unsigned)h), dim3(256), lds, snf::as_stream(stream), q, ldq, v, ldv,
                       kp, desc_dev, h, dk, kmax, scale, out, attn, n_total, lse);
    return snf::check_launch("ragged_attn_kernel");
}

// development / test hook: 1 (default) = the exact-fp32 attention runs on the f32 matrix-core forms where the shape allows, 0 = the
// vector-ALU kernels everywhere (the two agree to fp32 rounding: different summation orders of the same fmaf chains)
void snf_debug_exact_attn_mfma(int on) { g_exact_mfma = on != 0, g_dq_dv_lds = on == 1, g_pt_v_lds = on == 1; }

}  // extern "C"
