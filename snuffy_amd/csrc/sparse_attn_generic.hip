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

#include "common.h"

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
    SNF_REQUIRE(q && kp && v && p && dout && dq && dkp && dv, "snf_sparse_attn_bwd_f32: null pointer");
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
    int rc = snf::check_launch("bwd_ds_kernel");
    if (rc) return rc;
    dim3 grid2((unsigned)((n + 63) / 64), (unsigned)((dk + 63) / 64), (unsigned)h);
    hipLaunchKernelGGL(bwd_dq_dv_kernel, grid2, dim3(256), 0, s, p, mask, ds, dout, kp, n, k, h, dk, dq, dv);
    rc = snf::check_launch("bwd_dq_dv_kernel");
    if (rc) return rc;
    // dKp = dS^T Q: the forward's P^T V machinery on (dS, Q)
    const int slices = generic_slices(n);
    const int64_t rows_per_slice = (((n + slices - 1) / slices) + 15) & ~(int64_t)15;
    dim3 grid3((unsigned)((k + 63) / 64), (unsigned)((dk + 63) / 64), (unsigned)(slices * h));
    hipLaunchKernelGGL(pt_v_kernel, grid3, dim3(256), 0, s, ds, q, n, k, h, dk, rows_per_slice, partial);
    rc = snf::check_launch("pt_v_kernel(dS, Q)");
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
    dim3 grid((unsigned)((k + 63) / 64), (unsigned)((dk + 63) / 64), (unsigned)(slices * h));
    hipLaunchKernelGGL(pt_v_kernel, grid, dim3(256), 0, s, ds, q, n, k, h, dk, rows_per_slice, partial);
    int rc = snf::check_launch("pt_v_kernel(dS, Q)");
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
    int rc = snf::check_launch("scores_softmax_kernel");
    if (rc) return rc;

    const int64_t rows_per_slice = (((n + slices - 1) / slices) + 15) & ~(int64_t)15;
    dim3 grid2((unsigned)((k + 63) / 64), (unsigned)((dk + 63) / 64), (unsigned)(slices * h));
    hipLaunchKernelGGL(pt_v_kernel, grid2, dim3(256), 0, s, p, v, n, k, h, dk, rows_per_slice, partial);
    rc = snf::check_launch("pt_v_kernel");
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

}  // extern "C"
