// K6 / K10 / K12-K14: the dense projections of the hot path on the CDNA4 matrix cores.
//
//   C[M, N] = act(A[M, K] * W[N, K]^T + bias[N])       bf16 operands, fp32 accumulate, bf16 (or fp32) out
//
// replaces nn.Linear of snuffy.py:187-190 (Q | V and key projections), snuffy.py:224-225 (w_1 + activation, w_2) and the
// ViT Linear / Conv2d-as-GEMM layers (utils_ssls_cf/vision_transformer_with_adapter_dino_version.py:51-67, 82-94, 141-146).
// Both operands are K-major (A row-major, W as nn.Linear stores it), so one LDS image format serves both.
//
// Structure.  Persistent workgroups (one per CU, 8 waves = 2 (M) x 4 (N)) walk a stream of 256 x BN output tiles
// (BN = 256 or 128); a tile is a sequence of 32-deep K STEPS; the steps of consecutive tiles form one stream.
//   * HBM -> LDS by LDS-DMA (global_load_lds, 16 B per lane) into a ring of FOUR step buffers (A 256 x 32 + W BN x 32 bf16
//     each, 128 KiB in all).  The DMA runs two steps ahead of the reads -- across tile boundaries, so the first steps of the
//     next tile land while the current tile's epilogue stores drain; the only wait in the stream is one counted
//     s_waitcnt vmcnt(n) per step (never 0 inside a tile), placed one step before the first read of the data it retires.
//   * The destination of an LDS-DMA is lane-linear, so the bank swizzle is applied to the SOURCE address: a piece is
//     16 rows x 64 bytes, lane l lands at row l >> 2, slot l & 3 and fetches chunk slot ^ f(row).  f = (-(row >> 2)) & 3
//     for A (fragment rows consecutive) and (-(row >> 3)) & 3 for W (fragment rows 8 g + r, below): every ds_read_b128
//     lane group then covers the 16 slots of a 256-byte bank row exactly once (conflict-free; derivation in DESIGN.md).
//   * MFMA v_mfma_f32_16x16x32_bf16 with the operands SWAPPED (W fragment as srcA, A fragment as srcB): a lane's accumulator
//     holds 4 consecutive OUTPUT COLUMNS of one row.  W fragment ni, slot 4 g + r is column 8 g + 4 (ni & 1) + r + 32 (ni >> 1)
//     of the wave's 64: a lane owns 2 x 8 consecutive columns of a row, the 4 lanes of a row 2 x 64 contiguous bytes
//     -> 16-byte stores that fill 64-byte segments.
//   * Schedule: a step is {LDS reads of its 8 + NI fragments, LDS-DMA of the step two ahead, s_barrier, 8 NI MFMAs, s_barrier}.
//     The two wave groups (wr = 0 / 1, one wave of each per SIMD) run ONE BARRIER APART: one wave of a SIMD issues its MFMA
//     burst (32 MFMAs = 512 matrix-pipe cycles at BN = 256) while the other reads LDS, issues DMA and -- at a tile boundary
//     -- converts and stores its accumulators.  WAR safety of the ring is by construction: a buffer is re-staged two steps
//     after its last read was issued (two barrier pairs in between, the reads are consumed by MFMAs one pair earlier).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

struct GemmParams {
    const unsigned short* a;   // [M, lda] bf16
    const unsigned short* w;   // [N, ldw] bf16
    const float* bias;         // [N] or null
    void* c;                   // [M, ldc] bf16 or f32
    int64_t lda, ldw, ldc;
    int m, n, k;
    int act;
    int tiles_m, tiles_n;
    unsigned long long* trace;   // dev builds only (SNF_GEMM_TRACE), else null
    const float* resid = nullptr;   // gemm_hl_kernel, fp32 output: C += resid[m, ldr] (the residual stream of the FFN, snuffy.py:110)
    int64_t ldr = 0;
    // gemm_hl_kernel, split-K of the LAST, partly filled round of tiles (snf_gemm_hl_ws_bf16): an XCD whose q tiles leave r = q % W of
    // its W workgroups busy in the last round gives each of those tiles to S = min(W / r, split_cap) workgroups, a K range each.
    // Every part but the last to finish parks its accumulators in a slab; the last arriver (ticket) adds them in part order and runs the
    // epilogue.  ws: [8 * split_rcap] tickets (zeroed by the plain launch), then [8 * split_rcap][split_cap] slabs of 256 KiB.
    unsigned int* split_cnt = nullptr;
    float* split_slab = nullptr;
    int split_cap = 0, split_rcap = 0;
    // gemm_hl_kernel<NONE, 3, false, GATE = true> (the FFN input gradient of the training step, dhid = dz W2 behind the ReLU): elements whose
    // gate value (the hi half of the activation's own hl image, same position) is not > 0 leave as 0.  (Column sums in the same epilogue were
    // built and dropped: 16 more live registers there made hipcc spill 147 values around the store burst -- scratch traffic through the very
    // vector-memory pipeline the epilogue is bound by; the bias gradient is one pass of colsum_fused over the image instead.)
    const unsigned short* gate = nullptr;
    int64_t ldg = 0;
    // gemm_bf16_kernel, EPI = 1 (LayerNorm folded into the consumer, round 6): A holds the RAW rows x (bf16), W = W0 diag(gamma);
    //   C = act(rstd[m] (A W^T - mean[m] colsum[n]) + bias[n]),  colsum[n] = sum_k W[n, k] (of the rounded W), bias = W0 beta + b0
    const float* colsum = nullptr;   // [N]
    const float* rowstats = nullptr; // [M][2] = (mean, rstd) of the fp32 rows A was rounded from
    // gemm_bf16_kernel, EPI = 2 (residual stream in the producer, round 6): c = x [M, ldc] fp32, IN PLACE  x += A W^T + bias; a bf16
    // copy of the new x goes to c2 [M, ldc2] and, per row and 32-column group g (a 256-wide tile: 2 (4 tn + wave column) and the next one), the pair
    // (sum, sum of squares) of the new values to stats_part[(row * slots + g)] -- what snf_vit_row_stats turns into (mean, rstd)
    unsigned short* c2 = nullptr;
    int64_t ldc2 = 0;
    float* stats_part = nullptr;
    int slots = 0;
};
constexpr int HL_SPLIT_MIN_STEPS = 8;   // a K part is at least this many 32-column steps
constexpr int HL_SPLIT_MAX = 4;

constexpr int BM = 256, BKS = 32;
// ring of step buffers and LDS-DMA distance in steps.  A deeper ring (5 buffers = all 160 KiB, 3 steps ahead) measured the
// same on every shape: the stream is bound by the L2 -> LDS fill rate (~7 TB/s over the chip), not by its latency.
constexpr int NBUF = 4, AHEAD = 2;
constexpr int ROWB = BKS * 2;            // bytes of one row of a step image
constexpr int A_BYTES = BM * ROWB;       // 16 KiB

// two f32 -> packed bf16x2 (v_cvt_pk_bf16_f32: round to nearest even, NaN kept)
__device__ __forceinline__ unsigned int cvt_pk_bf16(float lo, float hi) {
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

// erf, Abramowitz & Stegun 7.1.26 (|err| <= 1.5e-7), branch-free
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
    return copysignf(fmaf(-p * t, e, 1.0f), x);
}

template <int ACT>
__device__ __forceinline__ float activate(float v) {
    if constexpr (ACT == SNF_ACT_RELU) return fmaxf(v, 0.f);
    if constexpr (ACT == SNF_ACT_GELU) return 0.5f * v * (1.0f + erf_as(v * 0.70710678118654752440f));
    if constexpr (ACT == SNF_ACT_LEAKYRELU) return v > 0.f ? v : 0.01f * v;
    if constexpr (ACT == SNF_ACT_SELU) {
        const float al = 1.6732632423543772848170429916717f, sc = 1.0507009873554804934193349852946f;
        return sc * (v > 0.f ? v : al * (__builtin_amdgcn_exp2f(v * 1.44269504088896340736f) - 1.0f));
    }
    return v;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// NI = 16-column W fragments per wave: 4 -> BN = 256, 2 -> BN = 128.  OUT: 0 = bf16 output, 1 = fp32, 2 = the bf16 image
// [hi | hi | lo] of the fp32 result (3 n columns, lo = bf16(v - hi)): the A operand of a following split-bf16 x3 GEMM; 3 = the
// interleaved hl image of the result (every 32 columns as [hi(32) | lo(32)], 2 n columns): what the pipelined attention streams.
// EPI (round 6, the ViT block without LayerNorm / residual passes; GemmParams has the formulas): 0 = plain; 1 = the LayerNorm of the
// consumer folded into the epilogue (A = raw rows, per-row (mean, rstd) and per-column weight sums); 2 = the residual stream updated
// in place by the producer (accumulators start from x instead of 0; fp32 x, its bf16 copy and the rows' partial moments leave together).
// MI = 16-row blocks per wave: 8 -> the 8 waves as 2 (rows) x 4 (columns), 128 rows x 16 NI columns each; 4 (with NI = 4: the 128-wide tile
// again) -> 4 x 2 waves of 64 x 64.  A step of the 128 x 32 wave tile reads 8 + 2 KiB of fragments per wave (80 KiB per workgroup), the 64 x 64
// one 4 + 4 KiB for the same 16 MFMAs: measured -4 % on config A's FFN output projection (85.6 -> 82.5 us), +1.6 % on its bag, level on the
// ViT shapes -- the 128-wide K loop is not LDS-read bound either; like the 256-wide one it runs at the workgroup's LDS-DMA rate.
template <int NI, int ACT, int OUT, int EPI = 0, int MI = 8>
__global__ __launch_bounds__(512, 2) void gemm_bf16_kernel(GemmParams P) {
    static_assert(MI == 8 || (MI == 4 && NI == 4 && EPI != 1), "gemm_bf16: wave tiles are 128 x 16 NI or 64 x 64");
    constexpr int WC = MI == 8 ? 4 : 2;        // wave columns (wave rows: 8 / WC)
    constexpr int RW = 16 * MI;                // rows per wave
    static_assert(EPI == 0 || (EPI == 1 ? (NI == 4 && OUT == 0) : ((NI == 4 || NI == 2) && OUT == 1 && ACT == SNF_ACT_NONE)),
                  "gemm_bf16: the LayerNorm-fold epilogue is 256-wide, the residual epilogue 256- or 128-wide");
    constexpr int BN = WC * 16 * NI;
    constexpr int W_BYTES = BN * ROWB;
    constexpr int STEP_BYTES = A_BYTES + W_BYTES;
    constexpr int WP = BN / 128;               // W pieces (16 rows each) staged by one wave per step
    constexpr int GL = 2 + WP;                 // LDS-DMA instructions per wave and step
    constexpr int NC = 4 * NI;                 // output columns per lane
    constexpr bool OUT_F32 = OUT == 1;
    // store instructions per lane and 16-row block (EPI 2: fp32 x + its bf16 copy + one moment pair)
    constexpr int NST = EPI == 2 ? NC / 4 + NC / 8 + 1 : (OUT == 1 || OUT == 3) ? NC / 4 : OUT == 2 ? 3 * NC / 8 : NC / 8;
    static_assert((AHEAD - 1) * GL + MI * NST <= 63, "gemm_bf16: counted waits are 6-bit");
    // bf16 rows leave as full 128-byte lines (epilogue; 4 KiB of LDS per wave) -- where the epilogue is not already bound by its own
    // arithmetic: behind the erf GELU the LDS round trip measured +3 % (ViT fc1 219 -> 226 us), behind none / ReLU -3 ... -5 %
    // (ViT qkv 144 -> 137.5 us, config-B FFN-in 193.6 -> 183.1 us; profiles/r06_gemm_line_stores.txt)
    constexpr bool LINE_STORES = OUT == 0 && NI == 4 && EPI != 2 && ACT != SNF_ACT_GELU && ACT != SNF_ACT_SELU;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [NBUF][A image | W image]

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = WC == 4 ? wid >> 2 : wid >> 1, wc = wid & (WC - 1);
    const int ns = P.k / BKS;                  // steps per tile (>= 2)
    unsigned char* const scr = smem + NBUF * STEP_BYTES + wid * 4096;   // LINE_STORES: this wave's two 2-KiB transpose buffers

    // ---- persistent tile stream, XCD-aware: workgroup b runs on XCD b % 8; every XCD owns a contiguous range of logical
    // tiles and its workgroups take consecutive tiles of it (column tiles of one A row panel first), so a panel is fetched
    // from HBM once per XCD and W stays in that XCD's L2
    const int ntiles = P.tiles_m * P.tiles_n;
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    int t_lo, t_hi;
    {
        const int q = ntiles >> 3, r = ntiles & 7;
        t_lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        t_hi = t_lo + q + (xcd < r ? 1 : 0);
    }
    int tile = t_lo + wg_in_xcd;
    if (tile >= t_hi) return;   // whole workgroup: no barrier has been executed yet

    // ---- LDS-DMA source offsets (elements) of a tile: piece p covers rows 16 p .. 16 p + 15 of an image; lane l lands at
    //      row 16 p + (l >> 2), slot l & 3.  Wave wid stages A pieces 2 wid, 2 wid + 1 and W pieces WP wid ..
    struct Src {
        int a[2], w[WP];
    };
    auto tile_src = [&](int tl) __attribute__((always_inline)) -> Src {
        Src s;
        const int tm = tl / P.tiles_n, tn = tl - tm * P.tiles_n;
        const int slot = lane & 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (2 * wid + j) * 16 + (lane >> 2);
            int grow = tm * BM + row;
            if (grow > P.m - 1) grow = P.m - 1;
            s.a[j] = grow * (int)P.lda + ((slot ^ ((-(row >> 2)) & 3)) << 3);
        }
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const int row = (WP * wid + j) * 16 + (lane >> 2);
            int grow = tn * BN + row;
            if (grow > P.n - 1) grow = P.n - 1;
            s.w[j] = grow * (int)P.ldw + ((slot ^ ((-(row >> 3)) & 3)) << 3);
        }
        return s;
    };
    auto stage = [&](const Src& s, int kstep, int buf) __attribute__((always_inline)) {
        unsigned char* base = smem + buf * STEP_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((glb_void*)(P.a + s.a[j] + kstep * BKS), (lds_void*)(base + (2 * wid + j) * 1024), 16, 0,
                                             0);
#pragma unroll
        for (int j = 0; j < WP; ++j)
            __builtin_amdgcn_global_load_lds((glb_void*)(P.w + s.w[j] + kstep * BKS),
                                             (lds_void*)(base + A_BYTES + (WP * wid + j) * 1024), 16, 0, 0);
    };

    // ---- fragment read offsets (bytes inside a step buffer)
    const int fi = lane & 15, fg = lane >> 4;
    const int fsw = (fg ^ ((-(fi >> 2)) & 3)) << 4;                           // same swizzle term for both operands
    const int xoff = (RW * wr + fi) * ROWB + fsw;                            // + mi * 1024
    const int woff = A_BYTES + ((NI == 4 ? 64 : 32) * wc + 8 * (fi >> 2) + (fi & 3)) * ROWB + fsw;   // + (4 (ni & 1) + 32 (ni >> 1)) * 64

    f32x4 acc[MI][NI];
    bf16x8 xf[MI], wf[NI];
    f32x4 bv4[NC / 4];   // bias of this lane's columns (hand-counted asm loads in the last step of a tile)
#pragma unroll
    for (int c4 = 0; c4 < NC / 4; ++c4) bv4[c4] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifdef SNF_GEMM_TRACE   // dev build (tools/gemm_trace.py): workgroup 0 stamps s_memtime around its barriers
    int trace_n = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if (P.trace && blockIdx.x == 0 && lane == 0 && (wid == 0 || wid == 4) && trace_n < 160)
            P.trace[(wid >> 2) * 160 + trace_n++] = __builtin_amdgcn_s_memtime();
    };
#else
    auto stamp = [&]() __attribute__((always_inline)) {};
#endif
    auto barrier = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        stamp();
        __builtin_amdgcn_s_barrier();
        stamp();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto read_frags = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* base = smem + buf * STEP_BYTES;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
            wf[ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + woff + (4 * (ni & 1) + 32 * (ni >> 1)) * ROWB));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) xf[mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + xoff + mi * 1024));
    };
    auto mma = [&](auto first_t) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_t)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#ifdef SNF_GEMM_NOMFMA   // timing ablation: operands kept alive, no matrix work
                if constexpr (FIRST) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
                asm volatile("" : "+v"(acc[mi][ni]) : "v"(wf[ni]), "v"(xf[mi]));
#else
                // first step of a tile: C is the constant 0 (no accumulator clearing pass between tiles)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], (FIRST && EPI != 2) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[mi][ni],
                                                                      0, 0, 0);
#endif
            }
        __builtin_amdgcn_s_setprio(0);
    };
    // hand-counted bias loads: older than every LDS-DMA issued after them, so the step's own counted wait retires them.
    // Column groups past N (partial column tile) are never stored; they read a valid address.
    auto load_bias = [&](int tl) __attribute__((always_inline)) {
        if (P.bias) {
            const int tn = tl % P.tiles_n;
            const int n0 = tn * BN + (NI == 4 ? 64 : 32) * wc + 8 * fg;
            const float* bp0 = P.bias + (n0 + 8 <= P.n ? n0 : 0);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bv4[0]) : "v"(bp0) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(bv4[1]) : "v"(bp0) : "memory");
            if constexpr (NI == 4) {
                const float* bp1 = P.bias + (n0 + 40 <= P.n ? n0 + 32 : 0);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bv4[2]) : "v"(bp1) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(bv4[3]) : "v"(bp1) : "memory");
            }
        }
    };
    // names the bias registers in a (free) asm statement placed behind the counted wait that retired their loads
    auto bias_landed = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int c4 = 0; c4 < NC / 4; ++c4) asm volatile("" : "+v"(bv4[c4]));
    };

    // EPI 2: the accumulators of a tile start as the residual stream x (plain loads at the tile's first step: the registers are
    // free since the last epilogue; the MFMAs of that step come behind the step's wait).  Rows / columns past the matrix read a valid
    // address and are never stored.
    auto preload_x = [&](int tl) __attribute__((always_inline)) {
        if constexpr (EPI == 2) {
            const int tm = tl / P.tiles_n, tn = tl - tm * P.tiles_n;
            const int n0 = tn * BN + (NI == 4 ? 64 : 32) * wc + 8 * fg;
            const int row0 = tm * BM + RW * wr + fi;
            const float* xb = reinterpret_cast<const float*>(P.c);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                int row = row0 + 16 * mi;
                if (row > P.m - 1) row = P.m - 1;
#pragma unroll
                for (int h = 0; h < NI / 2; ++h) {
                    const int col = n0 + 32 * h;
                    const float* src = xb + (int64_t)row * P.ldc + (col + 8 <= P.n ? col : 0);
                    acc[mi][2 * h] = *reinterpret_cast<const f32x4*>(src);
                    acc[mi][2 * h + 1] = *reinterpret_cast<const f32x4*>(src + 4);
                }
            }
        }
    };

    // epilogue of one tile: lane owns rows 16 mi + fi and columns n0 + {0..7} (+ 32 + {0..7} at BN = 256)
    auto epilogue = [&](int tl, auto full_t) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_t)::value;
        const int tm = tl / P.tiles_n, tn = tl - tm * P.tiles_n;
        const int n0 = tn * BN + (NI == 4 ? 64 : 32) * wc + 8 * fg;
        const int row0 = tm * BM + RW * wr + fi;
        f32x2 rst[EPI == 1 ? 8 : 1];          // EPI 1: (mean, rstd) of this lane's rows, weight column sums of its columns
        f32x4 cs4[EPI == 1 ? NC / 4 : 1];
        if constexpr (EPI == 1) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                int row = row0 + 16 * mi;
                if (row > P.m - 1) row = P.m - 1;
                rst[mi] = *reinterpret_cast<const f32x2*>(P.rowstats + 2 * (int64_t)row);
            }
#pragma unroll
            for (int h = 0; h < NI / 2; ++h) {
                const int col = n0 + 32 * h;
                const float* cp = P.colsum + (col + 8 <= P.n ? col : 0);
                cs4[2 * h] = *reinterpret_cast<const f32x4*>(cp);
                cs4[2 * h + 1] = *reinterpret_cast<const f32x4*>(cp + 4);
            }
        }
        int lane_o = lane;                      // (EPI 2 scratch addresses: recomputed per tile, see gemm_hl_kernel)
        asm volatile("" : "+v"(lane_o));
        const int fio = lane_o & 15, fgo = lane_o >> 4;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row = row0 + 16 * mi;
            float s1[NI / 2], s2[NI / 2];       // EPI 2: moments of the new x over this lane's columns of the row, per 32-column group
            u32x4 pk2[EPI == 2 ? NI / 2 : 1];
#pragma unroll
            for (int h = 0; h < NI / 2; ++h) {   // 8-column group: ni = 2 h, 2 h + 1
                float v[8];
                if constexpr (EPI == 1) {
                    const float rstd = rst[mi][1], nrm = -rst[mi][0] * rstd;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        v[e] = activate<ACT>(fmaf(rstd, acc[mi][2 * h + (e >> 2)][e & 3],
                                                  fmaf(nrm, cs4[2 * h + (e >> 2)][e & 3], bv4[2 * h + (e >> 2)][e & 3])));
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = activate<ACT>(acc[mi][2 * h + (e >> 2)][e & 3] + bv4[2 * h + (e >> 2)][e & 3]);
                }
                const int col = n0 + 32 * h;
#ifdef SNF_GEMM_NOSTORE   // timing ablation: no epilogue stores (and, with them, no epilogue arithmetic); results wrong
                const bool ok = P.k < 0 && row < P.m && col + 8 <= P.n;
#else
                const bool ok = FULL || (row < P.m && col + 8 <= P.n);
#endif
                if constexpr (EPI == 2) {
                    // fp32 rows and their bf16 copy leave as full lines (as LINE_STORES / gemm_hl_kernel): the fp32 chunks of this row
                    // block go to the wave's scratch now (chunk c = 8 h + 2 fg + half of the 256-byte span at position c ^ row), the
                    // bf16 chunks follow through the same scratch behind the fp32 reads (below)
                    // (128-wide tiles: the wave's span of a row is one 128-byte line, chunk c = 2 fg + half at position c ^ (row & 7))
                    (void)ok;
                    pk2[h] = u32x4{cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7])};
                    if constexpr (NI == 4) {
                        *reinterpret_cast<f32x4*>(scr + fio * 256 + (((8 * h + 2 * fgo) ^ fio) << 4)) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(scr + fio * 256 + (((8 * h + 2 * fgo + 1) ^ fio) << 4)) = f32x4{v[4], v[5], v[6], v[7]};
                    } else {
                        *reinterpret_cast<f32x4*>(scr + fio * 128 + (((2 * fgo) ^ (fio & 7)) << 4)) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(scr + fio * 128 + (((2 * fgo + 1) ^ (fio & 7)) << 4)) = f32x4{v[4], v[5], v[6], v[7]};
                    }
                    s1[h] = 0.f, s2[h] = 0.f;
                    if (FULL || col + 8 <= P.n) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) s1[h] += v[e], s2[h] = fmaf(v[e], v[e], s2[h]);
                    }
                } else if constexpr (OUT_F32) {
                    float* dst = reinterpret_cast<float*>(P.c) + (int64_t)row * P.ldc + col;
                    if (P.resid && ok) {   // residual added after the activation (round 6: z = x + W2 act(...) of small bags in one pass)
                        const float* rp = P.resid + (int64_t)row * P.ldr + col;
                        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] += r0[e];
                            v[4 + e] += r1[e];
                        }
                    }
                    if (ok) {
                        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    }
                } else if constexpr (OUT == 3) {   // hl image: the 8 columns sit inside one 32-column chunk: hi at 64 (col / 32) + col % 32, lo 32 further
                    const u32x4 pk = {cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7])};
                    float lo[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        lo[e] = v[e] - __uint_as_float((e & 1) ? (pk[e >> 1] & 0xffff0000u) : (pk[e >> 1] << 16));
                    const u32x4 pl = {cvt_pk_bf16(lo[0], lo[1]), cvt_pk_bf16(lo[2], lo[3]), cvt_pk_bf16(lo[4], lo[5]), cvt_pk_bf16(lo[6], lo[7])};
                    unsigned short* dst = reinterpret_cast<unsigned short*>(P.c) + (int64_t)row * P.ldc + 64 * (col >> 5) + (col & 31);
                    if (ok) {
                        *reinterpret_cast<u32x4*>(dst) = pk;
                        *reinterpret_cast<u32x4*>(dst + 32) = pl;
                    }
                } else if constexpr (LINE_STORES) {
                    // bf16 output, 256-wide tile: the wave's 16 rows x 128 bytes of this row block go through a wave-private LDS
                    // scratch (16-byte chunk c of row r at position c ^ (r & 7): conflict-free both ways) and leave as FULL 128-byte
                    // lines below -- straight from the accumulator layout a store instruction writes 16 rows x 64 bytes, and the write
                    // path handles one line REQUEST per ~10 cycles whatever its fill (tools/probes/store_probe.hip: the qkv output in
                    // bursts 35 us as half lines, 21 us as full lines)
                    const u32x4 pk = {cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7])};
                    *reinterpret_cast<u32x4*>(scr + (mi & 1) * 2048 + fi * 128 + (((fg + 4 * h) ^ (fi & 7)) << 4)) = pk;
                } else {
                    unsigned short* dst = reinterpret_cast<unsigned short*>(P.c) + (int64_t)row * P.ldc + col;
                    const u32x4 pk = {cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7])};
                    if (ok) *reinterpret_cast<u32x4*>(dst) = pk;
                    if constexpr (OUT == 2) {
                        float lo[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            lo[e] = v[e] - __uint_as_float((e & 1) ? (pk[e >> 1] & 0xffff0000u) : (pk[e >> 1] << 16));
                        const u32x4 pl = {cvt_pk_bf16(lo[0], lo[1]), cvt_pk_bf16(lo[2], lo[3]), cvt_pk_bf16(lo[4], lo[5]), cvt_pk_bf16(lo[6], lo[7])};
                        if (ok) {
                            *reinterpret_cast<u32x4*>(dst + P.n) = pk;
                            *reinterpret_cast<u32x4*>(dst + 2 * (int64_t)P.n) = pl;
                        }
                    }
                }
            }
            if constexpr (LINE_STORES) {
                const int lr = lane >> 3, lc = lane & 7;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int r = lr + 8 * jj;
                    const u32x4 val = *reinterpret_cast<const u32x4*>(scr + (mi & 1) * 2048 + r * 128 + ((lc ^ (r & 7)) << 4));
                    const int orow = tm * BM + RW * wr + 16 * mi + r, ocol = tn * BN + 64 * wc + 8 * lc;
#ifdef SNF_GEMM_NOSTORE
                    const bool ok2 = P.k < 0 && orow < P.m && ocol + 8 <= P.n;
#else
                    const bool ok2 = FULL || (orow < P.m && ocol + 8 <= P.n);
#endif
                    if (ok2) *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(P.c) + (int64_t)orow * P.ldc + ocol) = val;
                }
            }
            if constexpr (EPI == 2) {
                if constexpr (NI == 4) {
                    const int lr = lane_o >> 4, lc = lane_o & 15;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int r = lr + 4 * jj;
                        const u32x4 val = *reinterpret_cast<const u32x4*>(scr + r * 256 + ((lc ^ r) << 4));
                        const int orow = tm * BM + RW * wr + 16 * mi + r, ocol = tn * BN + 64 * wc + 4 * lc;
                        if (FULL || (orow < P.m && ocol + 4 <= P.n))
                            *reinterpret_cast<u32x4*>(reinterpret_cast<float*>(P.c) + (int64_t)orow * P.ldc + ocol) = val;
                    }
                    // the bf16 copy: 16 rows x 128 bytes, chunk c = fg + 4 h at position c ^ (row & 7), behind the reads above (LDS is in order)
#pragma unroll
                    for (int h = 0; h < NI / 2; ++h) *reinterpret_cast<u32x4*>(scr + fio * 128 + (((fgo + 4 * h) ^ (fio & 7)) << 4)) = pk2[h];
                    const int br = lane_o >> 3, bc = lane_o & 7;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int r = br + 8 * jj;
                        const u32x4 val = *reinterpret_cast<const u32x4*>(scr + r * 128 + ((bc ^ (r & 7)) << 4));
                        const int orow = tm * BM + RW * wr + 16 * mi + r, ocol = tn * BN + 64 * wc + 8 * bc;
                        if (FULL || (orow < P.m && ocol + 8 <= P.n)) *reinterpret_cast<u32x4*>(P.c2 + (int64_t)orow * P.ldc2 + ocol) = val;
                    }
                } else {
                    const int lr = lane_o >> 3, lc = lane_o & 7;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int r = lr + 8 * jj;
                        const u32x4 val = *reinterpret_cast<const u32x4*>(scr + r * 128 + ((lc ^ (r & 7)) << 4));
                        const int orow = tm * BM + RW * wr + 16 * mi + r, ocol = tn * BN + 32 * wc + 4 * lc;
                        if (FULL || (orow < P.m && ocol + 4 <= P.n))
                            *reinterpret_cast<u32x4*>(reinterpret_cast<float*>(P.c) + (int64_t)orow * P.ldc + ocol) = val;
                    }
                    // the bf16 copy of the wave's 32 columns is half a line per row: straight from the registers
                    if (FULL || (row < P.m && n0 + 8 <= P.n)) *reinterpret_cast<u32x4*>(P.c2 + (int64_t)row * P.ldc2 + n0) = pk2[0];
                }
                // a 32-column group of the row sits on the four lanes fi + 16 g: sum them (fixed order), lane g = 0 writes the pair(s) --
                // slot = 32-column group of the row; a 256-wide tile writes its wave's two adjacent slots in one store
#pragma unroll
                for (int h = 0; h < NI / 2; ++h) {
                    s1[h] += __shfl_xor(s1[h], 16, 64), s2[h] += __shfl_xor(s2[h], 16, 64);
                    s1[h] += __shfl_xor(s1[h], 32, 64), s2[h] += __shfl_xor(s2[h], 32, 64);
                }
                // (in a FULL tile every wave issues this store -- its 16 lanes g = 0 -- which the counted wait behind the epilogue relies on)
                if constexpr (NI == 4) {
                    const int g = 2 * (WC * tn + wc);
                    if (fg == 0 && (FULL || (g < P.slots && row < P.m)))
                        *reinterpret_cast<f32x4*>(P.stats_part + 2 * ((int64_t)row * P.slots + g)) = f32x4{s1[0], s2[0], s1[1], s2[1]};
                } else {
                    const int g = 4 * tn + wc;
                    if (fg == 0 && (FULL || (g < P.slots && row < P.m)))
                        *reinterpret_cast<f32x2*>(P.stats_part + 2 * ((int64_t)row * P.slots + g)) = f32x2{s1[0], s2[0]};
                }
            }
        }
    };

    // ---- stream prologue: steps 0 .. AHEAD - 1 of the first tile (ns >= AHEAD)
    Src cur = tile_src(tile);
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) stage(cur, i, i);
    wait_vmcnt<(AHEAD - 1) * GL>();   // step 0 has landed, the others stay in flight
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wid >= 4) __builtin_amdgcn_s_barrier();   // the second wave group (waves 4 .. 7) runs one barrier behind the first

    int rb = 0, sb = AHEAD;        // ring slots of the step being read and of the step being staged (global step mod NBUF)
    bool after_epilogue = false;   // stores of the previous tile may still be in flight (they count in vmcnt)
    while (true) {
        const int next = tile + wgs_per_xcd;
        const bool has_next = next < t_hi;
        const Src nxt = tile_src(has_next ? next : tile);
        const bool full = (tile / P.tiles_n + 1) * BM <= P.m && (tile % P.tiles_n + 1) * BN <= P.n;
        for (int s = 0; s < ns; ++s) {
            // ---- load part of step s
            if constexpr (EPI == 2)
                if (s == 0) preload_x(tile);
            read_frags(rb);
            if (s == ns - 1) load_bias(tile);
            // queue of this wave, oldest first: stage(+1) .. stage(+AHEAD-1) [, the previous tile's stores], then the
            // stage(+AHEAD) issued here.  The counted wait retires stage(+1) and everything older; at the end of the stream
            // the queue is shorter by the stages that no longer exist.
            const bool in_tile = s + AHEAD < ns;
            const int exist = has_next ? AHEAD : (ns - 1 - s < AHEAD ? ns - 1 - s : AHEAD);   // steps after this one, capped
            if (in_tile)
                stage(cur, s + AHEAD, sb);
            else if (has_next)
                stage(nxt, s + AHEAD - ns, sb);
            if (exist == AHEAD) {
                if (after_epilogue)
                    wait_vmcnt<(AHEAD - 1) * GL + MI * NST>();
                else
                    wait_vmcnt<(AHEAD - 1) * GL>();
            } else if (AHEAD == 3 && exist == 2) {
                wait_vmcnt<GL>();
            } else {
                wait_vmcnt<0>();
            }
            after_epilogue = false;
            rb = rb == NBUF - 1 ? 0 : rb + 1;
            sb = sb == NBUF - 1 ? 0 : sb + 1;
            barrier();
            // ---- MFMA part
            if (s == 0)
                mma(std::true_type{});
            else
                mma(std::false_type{});
            barrier();
        }
        bias_landed();
#ifdef SNF_GEMM_NOSTORE
        if (false) {
#else
        if (full) {
#endif
            epilogue(tile, std::true_type{});
            after_epilogue = true;     // exactly 8 NST stores issued by this wave
        } else {
            epilogue(tile, std::false_type{});
            wait_vmcnt<0>();           // predicated stores: their count is not known statically
        }
        if (!has_next) break;
        tile = next;
        cur = nxt;
    }
    if (wid < 4) __builtin_amdgcn_s_barrier();   // barrier counts of the two wave groups match again
}

// ---------------------------------------------------------------------------------------------------------------
// fp32-class GEMM in ONE pass over the operands: C = act(A W^T + bias) with both operands given as INTERLEAVED split images
// ("hl" format): a row of 2 k bf16 in which every 32 true columns c .. c + 31 are stored as [hi(32) | lo(32)], hi = bf16(v),
// lo = bf16(v - hi) -- 128 contiguous bytes, exactly one L2 line, and the same 4 bytes per element as the fp32 tensor.
// The concatenated form (gemm_bf16_kernel over [hi | hi | lo] x [Wh | Wl | Wh]) streams the hi halves twice, in 64-byte DMA
// segments, and pays three K steps of barriers / DMA issue / fragment reads per 32 true columns.  Here a step of 32 true columns
// stages ONE line per operand row (A 256 x 128 B + W 256 x 128 B = 64 KiB, full-line LDS-DMA) and issues the three products
// hi hi + hi lo + lo hi out of it: 96 MFMAs per wave and step behind 24 fragment reads.  256 x 256 tiles, two step buffers
// (128 KiB), all 8 waves in lockstep: the DMA of step s + 1 flies under the MFMA burst of step s.  LDS rows are 128 bytes, so
// the bank swizzle is the 8-chunk one (chunk ^ f(row) on the DMA source address; f = (row >> 1) & 7 for A, ((row >> 1) & 1) |
// ((row >> 3) & 3) << 1 for W: every ds_read_b128 lane group of 16 covers the 16 slots of a 256-byte bank row once).
// Register budget: the A-lo fragments are read into the A-hi registers while the second product (hi lo) is still issuing.
// OUT: 0 bf16, 1 fp32, 3 the hl image of the fp32 result (operand of the next one-pass GEMM).
// SPLIT (second launch of snf_gemm_hl_ws_bf16): the tiles of the last, partly filled round, one K part per workgroup (see GemmParams).
template <int ACT, int OUT, bool SPLIT = false, bool GATE = false>
__global__ __launch_bounds__(512, 2) void gemm_hl_kernel(GemmParams P) {
    static_assert(!GATE || (OUT == 3 && ACT == SNF_ACT_NONE && !SPLIT), "gemm_hl: the gated form writes an hl image, no activation, no split");
    constexpr int NI = 4, BN = 256;
    constexpr int ROWL = 128;                    // LDS row: hi(32) | lo(32) bf16
    constexpr int IMG = BM * ROWL;               // one operand's step image: 32 KiB
    constexpr int STEP_BYTES = 2 * IMG;          // A | W
    constexpr int NC = 4 * NI;
    // fp32 / hl-image rows leave as FULL 128-byte lines through a wave-private LDS transpose (round 6; as gemm_bf16_kernel's
    // LINE_STORES): straight from the accumulator layout a store instruction writes 16 rows x 64 (hl: hi or lo half) or 4 x 16
    // scattered bytes of 16 lines; transposed it writes 4 rows x 256 contiguous bytes.  Not behind the VALU-bound epilogues.
    constexpr bool LINE_STORES = (OUT == 1 || OUT == 3) && ACT != SNF_ACT_GELU && ACT != SNF_ACT_SELU;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][A | W] (+ 8 x 4 KiB transpose scratch)

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int ns_all = P.k / BKS;                // steps per tile (true K)
    unsigned char* const scr = smem + 2 * STEP_BYTES + wid * 4096;   // LINE_STORES: 16 rows x 256 B of this wave

    const int ntiles = P.tiles_m * P.tiles_n;
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    if constexpr (!SPLIT) {
        // The plain launch zeroes the tickets of the SPLIT launch that follows it on the stream (the kernel boundary orders the two):
        // the workspace is plain scratch memory of the caller, no state survives a call and none is expected before it.
        if (P.split_cap >= 2 && blockIdx.x == 0) {
            for (int i = threadIdx.x; i < 8 * P.split_rcap; i += 512) P.split_cnt[i] = 0u;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    int t_lo, t_hi;
    {
        const int q = ntiles >> 3, r = ntiles & 7;
        t_lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        t_hi = t_lo + q + (xcd < r ? 1 : 0);
    }
    // split-K of the last round (P.split_cap >= 2): the plain launch walks the full rounds only, the SPLIT launch gives workgroup
    // i of the XCD part i / r of remainder tile i % r
    int sp_part = 0, sp_parts = 1, sp_slot = 0, sp_tile = -1;
    if (P.split_cap >= 2) {
        const int q_x = t_hi - t_lo, full_rounds = q_x / wgs_per_xcd, r = q_x - full_rounds * wgs_per_xcd;
        int sn = r > 0 ? wgs_per_xcd / r : 0;
        if (sn > P.split_cap) sn = P.split_cap;
        if (sn > ns_all / HL_SPLIT_MIN_STEPS) sn = ns_all / HL_SPLIT_MIN_STEPS;
        if (sn >= 2 && r <= P.split_rcap) {
            if constexpr (SPLIT) {
                if (wg_in_xcd < r * sn) {
                    const int jr = wg_in_xcd % r;
                    sp_tile = t_lo + full_rounds * wgs_per_xcd + jr;
                    sp_part = wg_in_xcd / r, sp_parts = sn, sp_slot = xcd * P.split_rcap + jr;
                }
            } else {
                t_hi = t_lo + full_rounds * wgs_per_xcd;
            }
        }
    }
    int tile = t_lo + wg_in_xcd;
    if constexpr (SPLIT) {
        // (integer divisions run on the vector ALU: say that the results are wave-uniform)
        tile = __builtin_amdgcn_readfirstlane(sp_tile);
        sp_part = __builtin_amdgcn_readfirstlane(sp_part), sp_parts = __builtin_amdgcn_readfirstlane(sp_parts);
        sp_slot = __builtin_amdgcn_readfirstlane(sp_slot);
        if (tile < 0) return;
        t_hi = tile + 1;
        asm volatile("" : "+s"(t_hi));
    } else {
        if (tile >= t_hi) return;
    }
    // a K part is the same K loop over shifted operand bases: the loop keeps its shape (and its register allocation)
    const int s_begin = SPLIT ? __builtin_amdgcn_readfirstlane(sp_part * ns_all / sp_parts) : 0;
    const int ns = SPLIT ? __builtin_amdgcn_readfirstlane((sp_part + 1) * ns_all / sp_parts) - s_begin : ns_all;
    const unsigned short* const a_base = P.a + (SPLIT ? s_begin * 64 : 0);
    const unsigned short* const w_base = P.w + (SPLIT ? s_begin * 64 : 0);

    // LDS-DMA sources: an image is 32 pieces of 8 rows x 128 B; wave wid stages pieces 4 wid .. 4 wid + 3 of A and of W; lane l
    // lands at row 8 p + (l >> 3), chunk l & 7 and fetches chunk (l & 7) ^ f(row) of its row's line
    struct Src {
        int a[4], w[4];
    };
    auto tile_src_mn = [&](int tm, int tn) __attribute__((always_inline)) -> Src {
        Src s;
        // (opaque lane index: the per-lane row / swizzle terms below are a dozen integer instructions.  Hoisted out of the tile loop as
        // loop invariants they did not fit the K loop's registers: hipcc kept them in scratch and re-read them, one L2 round trip behind
        // the other, in front of the last step's MFMAs of every tile -- round 5, tools/scan_spills.py)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int c = ln & 7;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (4 * wid + j) * 8 + (ln >> 3);
            int grow = tm * BM + row;
            if (grow > P.m - 1) grow = P.m - 1;
            s.a[j] = grow * (int)P.lda + ((c ^ ((row >> 1) & 7)) << 3);
            int wrow = tn * BN + row;
            if (wrow > P.n - 1) wrow = P.n - 1;
            s.w[j] = wrow * (int)P.ldw + ((c ^ (((row >> 1) & 1) | (((row >> 3) & 3) << 1))) << 3);
        }
        return s;
    };
    auto tile_src = [&](int tl) __attribute__((always_inline)) -> Src {
        const int tm = tl / P.tiles_n;
        return tile_src_mn(tm, tl - tm * P.tiles_n);
    };
    auto stage = [&](const Src& s, int kstep, int buf) __attribute__((always_inline)) {
#ifdef X3_NOSTAGE   // timing ablations (dev builds, tools/gemm_x3_ablate.py): pieces of the step compiled out, results wrong
        return;
#endif
        unsigned char* base = smem + buf * STEP_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((glb_void*)(a_base + s.a[j] + kstep * 64), (lds_void*)(base + (4 * wid + j) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(w_base + s.w[j] + kstep * 64), (lds_void*)(base + IMG + (4 * wid + j) * 1024), 16, 0, 0);
        }
    };

    const int fi = lane & 15, fg = lane >> 4;
    const int fa = (fi >> 1) & 7, fw = ((fi >> 1) & 1) | ((fi >> 2) << 1);
    const int xoff_h = (128 * wr + fi) * ROWL + ((fg ^ fa) << 4), xoff_l = (128 * wr + fi) * ROWL + (((4 + fg) ^ fa) << 4);   // + mi * 2048
    const int wrow0 = IMG + (64 * wc + 8 * (fi >> 2) + (fi & 3)) * ROWL;
    const int woff_h = wrow0 + ((fg ^ fw) << 4), woff_l = wrow0 + (((4 + fg) ^ fw) << 4);   // + (4 (ni & 1) + 32 (ni >> 1)) * 128

    f32x4 acc[8][NI];
    bf16x8 xf[8], wh[NI], wl[NI];
    f32x4 bv4[NC / 4];

    auto frag = [&](const unsigned char* p) __attribute__((always_inline)) -> bf16x8 {
#ifdef X3_NOREAD
        u32x4 z = {1u, 2u, 3u, (unsigned)lane};
        asm volatile("" : "+v"(z));
        return __builtin_bit_cast(bf16x8, z);
#endif
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
    };
#ifdef X3_NOMFMA
#define X3_MFMA(a, b, c) ([&]() { f32x4 t_ = (c); asm volatile("" : "+v"(t_) : "v"(a), "v"(b)); return t_; }())
#else
#define X3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif
    auto load_bias = [&](int tl) __attribute__((always_inline)) {
#pragma unroll
        for (int c4 = 0; c4 < NC / 4; ++c4) bv4[c4] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (P.bias) {
            const int tn = tl % P.tiles_n;
            const int n0 = tn * BN + 64 * wc + 8 * fg;
            const float* bp0 = P.bias + (n0 + 8 <= P.n ? n0 : 0);
            const float* bp1 = P.bias + (n0 + 40 <= P.n ? n0 + 32 : 0);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bv4[0]) : "v"(bp0) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(bv4[1]) : "v"(bp0) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bv4[2]) : "v"(bp1) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(bv4[3]) : "v"(bp1) : "memory");
        }
    };
    auto bias_landed = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int c4 = 0; c4 < NC / 4; ++c4) asm volatile("" : "+v"(bv4[c4]));
    };
    auto epilogue = [&](int tl, auto full_t) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_t)::value;
        const int tm = tl / P.tiles_n, tn = tl - tm * P.tiles_n;
        const int n0 = tn * BN + 64 * wc + 8 * fg;
        const int row0 = tm * BM + 128 * wr + fi;
        // (LINE_STORES) the scratch addresses hang on an opaque copy of the lane index: a dozen integer instructions per tile that are
        // cheaper to redo here than to keep live through the K loop, where hipcc would park them in scratch (round 5's lesson)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int fio = lane_o & 15, fgo = lane_o >> 4;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int row = row0 + 16 * mi;
#pragma unroll
            for (int h = 0; h < NI / 2; ++h) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = activate<ACT>(acc[mi][2 * h + (e >> 2)][e & 3] + bv4[2 * h + (e >> 2)][e & 3]);
                const int col = n0 + 32 * h;
#ifdef X3_NOSTORE
                const bool ok = v[0] == 123.456f;
#else
                const bool ok = FULL || (row < P.m && col + 8 <= P.n);
#endif
                if constexpr (GATE) {
                    u32x4 gv = {0u, 0u, 0u, 0u};
                    if (ok) gv = *reinterpret_cast<const u32x4*>(P.gate + (int64_t)row * P.ldg + 64 * (col >> 5) + (col & 31));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (!(__uint_as_float(gv[e] << 16) > 0.f)) v[2 * e] = 0.f;
                        if (!(__uint_as_float(gv[e] & 0xffff0000u) > 0.f)) v[2 * e + 1] = 0.f;
                    }
                }
                if constexpr (OUT == 1) {
                    float* dst = reinterpret_cast<float*>(P.c) + (int64_t)row * P.ldc + col;
                    if (P.resid && ok) {   // residual added after the activation: z = x + W2 act(...) in one pass over z
                        const float* rp = P.resid + (int64_t)row * P.ldr + col;
                        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] += r0[e];
                            v[4 + e] += r1[e];
                        }
                    }
                    if constexpr (LINE_STORES) {
                        // 16-byte chunk c = 8 h + 2 fg + half of the wave's 256-byte row span, parked at position c ^ row
                        *reinterpret_cast<f32x4*>(scr + fio * 256 + (((8 * h + 2 * fgo) ^ fio) << 4)) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(scr + fio * 256 + (((8 * h + 2 * fgo + 1) ^ fio) << 4)) = f32x4{v[4], v[5], v[6], v[7]};
                    } else if (ok) {
                        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    }
                } else {
                    const u32x4 pk = {cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7])};
                    if constexpr (OUT == 0) {
                        unsigned short* dst = reinterpret_cast<unsigned short*>(P.c) + (int64_t)row * P.ldc + col;
                        if (ok) *reinterpret_cast<u32x4*>(dst) = pk;
                    } else {   // hl image: the 8 columns sit inside one 32-column chunk: hi at 64 (col / 32) + col % 32, lo 32 further
                        float lo[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            lo[e] = v[e] - __uint_as_float((e & 1) ? (pk[e >> 1] & 0xffff0000u) : (pk[e >> 1] << 16));
                        const u32x4 pl = {cvt_pk_bf16(lo[0], lo[1]), cvt_pk_bf16(lo[2], lo[3]), cvt_pk_bf16(lo[4], lo[5]), cvt_pk_bf16(lo[6], lo[7])};
                        unsigned short* dst = reinterpret_cast<unsigned short*>(P.c) + (int64_t)row * P.ldc + 64 * (col >> 5) + (col & 31);
                        if constexpr (LINE_STORES) {
                            // chunk c = 8 h + 4 plane + fg of the wave's 256-byte span of the image row ([hi(32) | lo(32)] per 32 columns)
                            *reinterpret_cast<u32x4*>(scr + fio * 256 + (((8 * h + fgo) ^ fio) << 4)) = pk;
                            *reinterpret_cast<u32x4*>(scr + fio * 256 + (((8 * h + 4 + fgo) ^ fio) << 4)) = pl;
                        } else if (ok) {
                            *reinterpret_cast<u32x4*>(dst) = pk;
                            *reinterpret_cast<u32x4*>(dst + 32) = pl;
                        }
                    }
                }
            }
            if constexpr (LINE_STORES) {
                const int lr = lane_o >> 4, lc = lane_o & 15;
                unsigned char* const crow = reinterpret_cast<unsigned char*>(P.c) +
                                            (OUT == 1 ? 4 * (int64_t)(tn * BN + 64 * wc) : 2 * 2 * (int64_t)(tn * BN + 64 * wc));
                const int64_t pitch = (OUT == 1 ? 4 : 2) * P.ldc;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int r = lr + 4 * jj;
                    const u32x4 val = *reinterpret_cast<const u32x4*>(scr + r * 256 + ((lc ^ r) << 4));
                    const int orow = tm * BM + 128 * wr + 16 * mi + r;
                    // true columns of chunk lc: fp32 4 lc .. 4 lc + 3; hl image 32 (lc >> 3) + 8 (lc & 3) .. + 7
                    const int ocol = tn * BN + 64 * wc + (OUT == 1 ? 4 * lc : 32 * (lc >> 3) + 8 * (lc & 3));
#ifdef X3_NOSTORE
                    const bool ok2 = val[0] == 0x12345678u && orow < 0;
#else
                    const bool ok2 = FULL || (orow < P.m && ocol + (OUT == 1 ? 4 : 8) <= P.n);
#endif
                    if (ok2) *reinterpret_cast<u32x4*>(crow + orow * pitch + 16 * lc) = val;
                }
            }
        }
    };

    Src cur = tile_src(tile);
    // SPLIT: one tile per workgroup.  The loop below keeps the shape of the tile walk all the same (has_next is false at run time,
    // not at compile time): with `cur` loop-invariant hipcc carries eight 64-bit DMA addresses through a K loop that has no register
    // to spare -- they spill, and their reloads sit between the barrier and the MFMA burst of every step.
    stage(cur, 0, 0);
    int buf = 0;
    bool after_epilogue = false;   // the previous tile's stores may still be in flight (they count in vmcnt)
    while (true) {
        const int next = tile + wgs_per_xcd;
        const bool has_next = next < t_hi;
        const bool full = (tile / P.tiles_n + 1) * BM <= P.m && (tile % P.tiles_n + 1) * BN <= P.n;
        for (int s = 0; s < ns; ++s) {
            // this step's two images have landed (every wave waits for its own pieces, then all meet); the other buffer is free:
            // its last fragment reads belong to the previous step's MFMAs, which every wave has issued before this barrier.
            // First step behind an epilogue: its images were waited for BEFORE the epilogue (the wait behind the bias loads), the
            // only thing in flight is the tile's store burst -- which must not be waited for here: every workgroup of the chip
            // stores its 256 KiB tile at the same moment, and draining that burst in front of the next tile's first MFMA was 40 us
            // of the Q | V projection (round-3 ablation).  The stores drain under the first step's MFMA burst instead.
            if (!(s == 0 && after_epilogue)) wait_vmcnt<0>();
            after_epilogue = false;
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* base = smem + buf * STEP_BYTES;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) wh[ni] = frag(base + woff_h + (4 * (ni & 1) + 32 * (ni >> 1)) * ROWL);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) xf[mi] = frag(base + xoff_h + mi * 16 * ROWL);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) wl[ni] = frag(base + woff_l + (4 * (ni & 1) + 32 * (ni >> 1)) * ROWL);
            __builtin_amdgcn_sched_barrier(0);
            // the next step's DMA behind the fragment reads (issued ahead of them it measured the same: +-3 % run to run)
            if (s + 1 < ns) {
                stage(cur, s + 1, buf ^ 1);
            } else if (has_next) {
                cur = tile_src(next);      // the next tile's first step flies under this tile's last burst and its epilogue
                stage(cur, 0, buf ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- hi hi
            __builtin_amdgcn_s_setprio(1);
            if (s == 0) {
#pragma unroll
                for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = X3_MFMA(wh[ni], xf[mi], (f32x4{0.f, 0.f, 0.f, 0.f}));
            } else {
#pragma unroll
                for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = X3_MFMA(wh[ni], xf[mi], acc[mi][ni]);
            }
            // ---- hi lo; the A-lo fragment of a row block replaces its A-hi fragment as soon as that block's products are issued
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = X3_MFMA(wl[ni], xf[mi], acc[mi][ni]);
                __builtin_amdgcn_sched_barrier(0);
                xf[mi] = frag(base + xoff_l + mi * 16 * ROWL);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- lo hi
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = X3_MFMA(wh[ni], xf[mi], acc[mi][ni]);
            __builtin_amdgcn_s_setprio(0);
            buf ^= 1;
        }
        if constexpr (SPLIT) {
            // Whoever draws the last ticket adds the other parts' slabs to its accumulators in part order and runs the epilogue;
            // everybody else parks its accumulators and leaves (cdna_hip_programming.md, in-launch split-K: plain slab stores ->
            // vmcnt(0) -> barrier -> one agent release -> ticket; reducer: one agent acquire -> barrier -> plain loads).
            // The ticket is broadcast through the step buffers, idle by now: a second __shared__ object would make hipcc drain the
            // LDS-DMA in front of every fragment read of the K loop.
            volatile unsigned int& s_ticket = *reinterpret_cast<volatile unsigned int*>(smem);
            float* slab0 = P.split_slab + (size_t)sp_slot * P.split_cap * (size_t)(BM * BN);
            // (opaque thread index: the addresses below must not be computed ahead of the K loop, whose registers are all taken)
            unsigned tix = threadIdx.x;
            asm volatile("" : "+v"(tix));
            f32x4* mine = reinterpret_cast<f32x4*>(slab0 + (size_t)sp_part * (BM * BN)) + tix;
            wait_vmcnt<0>();
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) mine[(mi * NI + ni) * 512] = acc[mi][ni];
            wait_vmcnt<0>();
            __syncthreads();
            if (tix == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                s_ticket = __hip_atomic_fetch_add(P.split_cnt + sp_slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            if (s_ticket != (unsigned)(sp_parts - 1)) return;
            __syncthreads();
            if (tix == 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(P.split_cnt + sp_slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
            }
            __syncthreads();
            // sum in part order (fixed: bit-reproducible whoever arrives last); this workgroup's own part comes from its registers
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
                    for (int pp = 0; pp < sp_parts; ++pp) {
                        f32x4 v = acc[mi][ni];
                        if (pp != sp_part) v = reinterpret_cast<const f32x4*>(slab0 + (size_t)pp * (BM * BN))[(mi * NI + ni) * 512 + tix];
                        sum = pp == 0 ? v : sum + v;
                    }
                    acc[mi][ni] = sum;
                }
        }
        // bias: fetched here (one L2 round trip per tile) instead of being carried in 16 registers through the K loop
        load_bias(tile);
        wait_vmcnt<0>();
        bias_landed();
        if (full)
            epilogue(tile, std::true_type{});
        else
            epilogue(tile, std::false_type{});
        after_epilogue = true;
        if (!has_next) break;
        tile = next;
    }
}

template <int ACT, int OUT>
int launch_hl(const GemmParams& P, hipStream_t s) {
    constexpr int lds = 2 * 2 * BM * 128 + (((OUT == 1 || OUT == 3) && ACT != SNF_ACT_GELU && ACT != SNF_ACT_SELU) ? 8 * 4096 : 0);
    static thread_local unsigned long long attr_set_mask = 0;   // devices (bit = device id) that have the opt-in
    const unsigned long long attr_set_bit = snf::device_bit();
    const bool attr_set = (attr_set_mask & attr_set_bit) != 0;
    auto kern = gemm_hl_kernel<ACT, OUT>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            snf::set_error("gemm_hl: cannot reserve %d bytes of LDS", lds);
            (void)hipGetLastError();
            return SNF_ELAUNCH;
        }
        attr_set_mask |= attr_set_bit;
    }
    const int ntiles = P.tiles_m * P.tiles_n;
    int grid = snf::cu_count() & ~7;
    if (grid < 8) grid = 8;
    {   // development probe (tools/gemm_hl_occupancy_probe.py): walk the tiles with fewer workgroups than CUs
        static const int dbg_grid = getenv("SNF_GEMM_HL_GRID") ? atoi(getenv("SNF_GEMM_HL_GRID")) & ~7 : 0;
        if (dbg_grid >= 8 && dbg_grid < grid && P.split_cap < 2) grid = dbg_grid;
    }
    const int per_xcd = (ntiles + 7) / 8;
    if (per_xcd * 8 < grid) grid = per_xcd * 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, P);
    int rc = snf::check_launch("gemm_hl_kernel");
    if (rc || P.split_cap < 2) return rc;
    // the last, partly filled round: every remainder tile on 2 .. 4 workgroups, a K range each (same grid: same tile -> XCD map)
    static thread_local unsigned long long attr_set2_mask = 0;   // devices (bit = device id) that have the opt-in
    const unsigned long long attr_set2_bit = snf::device_bit();
    const bool attr_set2 = (attr_set2_mask & attr_set2_bit) != 0;
    auto kern2 = gemm_hl_kernel<ACT, OUT, true>;
    if (!attr_set2) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern2), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            snf::set_error("gemm_hl: cannot reserve %d bytes of LDS", lds);
            (void)hipGetLastError();
            return SNF_ELAUNCH;
        }
        attr_set2_mask |= attr_set2_bit;
    }
    hipLaunchKernelGGL(kern2, dim3(grid), dim3(512), lds, s, P);
    return snf::check_launch("gemm_hl_kernel<split>");
}

int launch_hl_gated(const GemmParams& P, hipStream_t s) {
    constexpr int lds = 2 * 2 * BM * 128 + 8 * 4096;
    static thread_local unsigned long long attr_set_mask = 0;   // devices (bit = device id) that have the opt-in
    const unsigned long long bit = snf::device_bit();
    auto kern = gemm_hl_kernel<SNF_ACT_NONE, 3, false, true>;
    if (!(attr_set_mask & bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            snf::set_error("gemm_hl: cannot reserve %d bytes of LDS", lds);
            (void)hipGetLastError();
            return SNF_ELAUNCH;
        }
        attr_set_mask |= bit;
    }
    const int ntiles = P.tiles_m * P.tiles_n;
    int grid = snf::cu_count() & ~7;
    if (grid < 8) grid = 8;
    const int per_xcd = (ntiles + 7) / 8;
    if (per_xcd * 8 < grid) grid = per_xcd * 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, P);
    return snf::check_launch("gemm_hl_kernel<gated>");
}

template <int OUT>
int launch_hl_act(const GemmParams& P, hipStream_t s) {
    switch (P.act) {
        case SNF_ACT_RELU: return launch_hl<SNF_ACT_RELU, OUT>(P, s);
        case SNF_ACT_GELU: return launch_hl<SNF_ACT_GELU, OUT>(P, s);
        case SNF_ACT_LEAKYRELU: return launch_hl<SNF_ACT_LEAKYRELU, OUT>(P, s);
        case SNF_ACT_SELU: return launch_hl<SNF_ACT_SELU, OUT>(P, s);
        default: return launch_hl<SNF_ACT_NONE, OUT>(P, s);
    }
}

template <int NI, int ACT, int OUT, int EPI = 0, int MI = 8>
int launch(const GemmParams& P, hipStream_t s) {
    constexpr int lds = NBUF * (A_BYTES + (MI == 8 ? 64 : 32) * NI * ROWB) +
                        ((EPI == 2 || (OUT == 0 && NI == 4 && ACT != SNF_ACT_GELU && ACT != SNF_ACT_SELU)) ? 8 * 4096 : 0);
    static thread_local unsigned long long attr_set_mask = 0;   // devices (bit = device id) that have the opt-in
    const unsigned long long attr_set_bit = snf::device_bit();
    const bool attr_set = (attr_set_mask & attr_set_bit) != 0;
    auto kern = gemm_bf16_kernel<NI, ACT, OUT, EPI, MI>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
            hipSuccess) {
            snf::set_error("gemm_bf16: cannot reserve %d bytes of LDS", lds);
            (void)hipGetLastError();
            return SNF_ELAUNCH;
        }
        attr_set_mask |= attr_set_bit;
    }
    const int ntiles = P.tiles_m * P.tiles_n;
    int grid = snf::cu_count() & ~7;          // one persistent workgroup per CU, a multiple of the 8 XCDs
    if (grid < 8) grid = 8;
    const int per_xcd = (ntiles + 7) / 8;     // no XCD holds more tiles than this
    if (per_xcd * 8 < grid) grid = per_xcd * 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, P);
    return snf::check_launch("gemm_bf16_kernel");
}

#ifndef SNF_GEMM_128_MI4
#define SNF_GEMM_128_MI4 1   // 128-wide tiles: 1 = 4 x 2 waves of 64 x 64 (round 6), 0 = 2 x 4 waves of 128 x 32 (dev builds, A / B)
#endif
// NI == 2 names the 128-wide tile at the call sites; the instantiation behind it is the 64 x 64 wave layout
template <int NI, int ACT, int OUT, int EPI = 0>
int launch_tile(const GemmParams& P, hipStream_t s) {
    if constexpr (NI == 2 && SNF_GEMM_128_MI4)
        return launch<4, ACT, OUT, EPI, 4>(P, s);
    else
        return launch<NI, ACT, OUT, EPI>(P, s);
}

template <int NI, int OUT>
int launch_act(const GemmParams& P, hipStream_t s) {
    switch (P.act) {
        case SNF_ACT_RELU: return launch_tile<NI, SNF_ACT_RELU, OUT>(P, s);
        case SNF_ACT_GELU: return launch_tile<NI, SNF_ACT_GELU, OUT>(P, s);
        case SNF_ACT_LEAKYRELU: return launch_tile<NI, SNF_ACT_LEAKYRELU, OUT>(P, s);
        case SNF_ACT_SELU: return launch_tile<NI, SNF_ACT_SELU, OUT>(P, s);
        default: return launch_tile<NI, SNF_ACT_NONE, OUT>(P, s);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Skinny fp32-class projection: out[R, C] = x[R, K] w[C, K]^T + bias for a FEW HUNDRED rows (the K selected rows of a bag: key and
// output projections, snuffy.py:190, 205).  The tile kernels above need >= 256 rows per workgroup and thousands of tiles; the fp32
// library GEMM takes ~10 us for 200 x 768 x 768.  Here a workgroup owns a 32 x 32 output block, its 8 waves split the K axis, every
// wave splits its fp32 operands into bf16 hi / lo in registers (no images in memory) and issues hi hi + hi lo + lo hi on
// v_mfma_f32_32x32x16_bf16; the partial blocks are summed through LDS in a fixed order.  Rows / columns past R / C are clamped
// on load and masked on store.  k % 16 == 0, rows 16-byte aligned.
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16s;
typedef __attribute__((ext_vector_type(8))) float f32x8s;
__device__ __forceinline__ void skinny_split8(const f32x8s v, bf16x8& hi, bf16x8& lo) {
    hi = __builtin_convertvector(v, bf16x8);
    lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x8s), bf16x8);
}
__device__ __forceinline__ f32x8s skinny_load8(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    return f32x8s{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
// FRAG: the result leaves as the MFMA A-fragment image the pipelined attention kernel keeps in registers (sparse_attn_x3p.hip,
// x3p_prep_kp_kernel: [chunk][head][key block][dk / 16][hi | lo][64 lanes] x 16 bytes, value * fr.c_exp split into bf16 hi + lo,
// padded keys zero) instead of as a [r, c] matrix: the key projection then needs no fp32 Kp tensor and no prep launch.
struct SkinnyFrag {
    int dk, chunk_size;        // head width; keys per chunk (a multiple of 32, or r when there is one chunk)
    int64_t chunk_stride;      // u32x4 units between the chunks' images
    float c_exp;
    // round 6: the gather of the selected rows fused into their key projection (snuffy.py:131,145-147 + 190 in one launch):
    // idx != null: input row i of the projection is row idx[i] of x (the bag itself); xs (nullable) receives the gathered rows
    // [r, k] (written by the workgroups of the first column block from the registers they multiply); map (nullable, [n_rows] int32)
    // receives the row -> slot map (-1 = not selected), built by extra workgroups behind the projection's own
    const int64_t* idx = nullptr;
    int64_t n_rows = 0;
    float* xs = nullptr;
    int64_t ldxs = 0;
    int32_t* map = nullptr;
    // plain (non-fragment) f32 output, round 6: out2 [r, c] (nullable) = out + resid [r, c] -- x_sel = xs + delta of snuffy.py:108 in
    // the output projection's own store pass
    const float* resid = nullptr;
    float* out2 = nullptr;
    int64_t ldr = 0, ldo2 = 0;
};
template <bool OUT_BF16, bool FRAG = false>
__global__ __launch_bounds__(512) void skinny_linear_x3_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                                int64_t ldw, const float* __restrict__ bias, int r, int c, int k,
                                                                void* __restrict__ out, int64_t ldo, SkinnyFrag fr = SkinnyFrag{}) {
    constexpr int NW = 8, BATCH = 6;                        // waves splitting the K axis; MFMA steps (16 deep) requested at once
    __shared__ float red[NW - 1][32 * 32];
    if constexpr (FRAG) {
        const int nrb = (r + 31) >> 5;
        if ((int)blockIdx.y >= nrb) {                       // row -> slot map: the selected indices searched in LDS (as gather_slot_map_kernel)
            int* sel = reinterpret_cast<int*>(&red[0][0]);
            for (int q = threadIdx.x; q < r; q += 512) sel[q] = (int)fr.idx[q];
            __syncthreads();
            const int64_t i = ((int64_t)(blockIdx.y - nrb) * gridDim.x + blockIdx.x) * 512 + threadIdx.x;
            if (i >= fr.n_rows) return;
            int slot = -1;
            const int me = (int)i;
            for (int q = 0; q < r; ++q) slot = (sel[q] == me) ? q : slot;
            fr.map[i] = slot;
            return;
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 31, hf = lane >> 5;
    const int row0 = blockIdx.y * 32, col0 = blockIdx.x * 32;
    int xr = row0 + j, wr = col0 + j;
    if (xr > r - 1) xr = r - 1;
    if (wr > c - 1) wr = c - 1;
    const int xr_out = xr;                                  // row of the gathered copy
    bool xs_out = false;
    if constexpr (FRAG) {
        if (fr.idx) {
            int64_t src = fr.idx[xr];
            src = src < 0 ? 0 : (src > fr.n_rows - 1 ? fr.n_rows - 1 : src);
            xr = (int)src;
            xs_out = fr.xs != nullptr && blockIdx.x == 0 && row0 + j < r;
        }
    }
    // the K axis in 16-deep steps, split evenly over the waves: the whole pass is ONE memory round trip per wave when its share
    // fits a batch (k <= 768), so every load of the workgroup is in flight before the first MFMA
    const int steps = k >> 4, per = (steps + NW - 1) / NW;
    const int s_lo = wv * per, s_hi = s_lo + per < steps ? s_lo + per : steps;
    const float* xp = x + (int64_t)xr * ldx + 8 * hf;
    const float* wp = w + (int64_t)wr * ldw + 8 * hf;
    f32x16s acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // A = w fragment (output column on the lane), B = x fragment (row on the lane): C[col, row] -- lane (row j, half hf) then holds
    // columns (i & 3) + 8 (i >> 2) + 4 hf of ITS row in registers
    for (int s0 = s_lo; s0 < s_hi; s0 += BATCH) {
        // branch-free on purpose: a predicate around the loads makes the compiler wait for each pair before it requests the next
        // (six serial round trips: 12 us for 200 x 768 x 768).  Steps past the wave's share re-read its last step and contribute 0.
        f32x8s xv[BATCH], wvv[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int st = s0 + u < s_hi ? s0 + u : s_hi - 1;
            xv[u] = skinny_load8(xp + 16 * st);
            wvv[u] = skinny_load8(wp + 16 * st);
        }
        __builtin_amdgcn_sched_barrier(0);   // every load of the batch is requested before the first split / MFMA
        if constexpr (FRAG) {
            if (xs_out) {                        // the gathered rows leave from the registers they are multiplied from
#pragma unroll
                for (int u = 0; u < BATCH; ++u)
                    if (s0 + u < s_hi) {
                        float* d8 = fr.xs + (int64_t)xr_out * fr.ldxs + 8 * hf + 16 * (s0 + u);
                        *reinterpret_cast<f32x4*>(d8) = f32x4{xv[u][0], xv[u][1], xv[u][2], xv[u][3]};
                        *reinterpret_cast<f32x4*>(d8 + 4) = f32x4{xv[u][4], xv[u][5], xv[u][6], xv[u][7]};
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            bf16x8 xh, xl, wh, wl;
            const float live = s0 + u < s_hi ? 1.f : 0.f;
            skinny_split8(xv[u] * live, xh, xl);
            skinny_split8(wvv[u], wh, wl);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc, 0, 0, 0);
        }
    }
    // sum the waves' blocks in a fixed order (waves 1.. through LDS, wave 0 adds them to its own)
    if (wv > 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wv - 1][i * 64 + lane] = acc[i];
    }
    __syncthreads();
    // wave 0 sums, adds the bias and parks the block row-major in LDS; then every thread stores along a row (a lane of the MFMA
    // result holds one ROW's columns 4 apart in registers: stored from there, every 4-byte store of a wave hits another line)
    __shared__ float blk[32][33];
    if (wv == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = acc[i];
#pragma unroll
            for (int q = 0; q < NW - 1; ++q) v += red[q][i * 64 + lane];
            const int cl = (i & 3) + 8 * (i >> 2) + 4 * hf;
            blk[j][cl] = v;
        }
    }
    // the bias of this thread's column (threads 0..511 cover columns e & 31: the same column in both rounds), requested before the
    // barrier -- added in wave 0's registers it was 16 conditional loads, each a full round trip
    const int bcol = col0 + (threadIdx.x & 31);
    const float bv = (bias && bcol < c) ? bias[bcol] : 0.f;
    __syncthreads();
    if constexpr (FRAG) {
        // thread (key j, column group gq of 8): fragment lane 32 hf + j of k-step kb of head a, key block b of chunk ch
        if (threadIdx.x < 128) {
            const int gq = threadIdx.x >> 5;
            const int ch = row0 / fr.chunk_size, b = (row0 - ch * fr.chunk_size) >> 5;
            int kc = r - ch * fr.chunk_size;                   // keys of this chunk
            if (kc > fr.chunk_size) kc = fr.chunk_size;
            // key blocks per head in the image: the chunk's own, or (chunked launches) those of a FULL chunk -- every chunk of a
            // merged launch is laid out alike, a shorter last chunk leaves its trailing blocks unwritten (the kernel masks them)
            const int nkb = fr.chunk_size < r ? (fr.chunk_size + 31) >> 5 : (kc + 31) >> 5, nks = fr.dk >> 4;
            const int a = col0 / fr.dk, cin = col0 - a * fr.dk + 8 * gq;
            const int kb = cin >> 4, hf2 = (cin >> 3) & 1;
            f32x8s v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (blk[j][8 * gq + e] + (bias ? bias[col0 + 8 * gq + e] : 0.f)) * fr.c_exp;
            bf16x8 hi, lo;
            skinny_split8(v, hi, lo);
            u32x4 uh = __builtin_bit_cast(u32x4, hi), ul = __builtin_bit_cast(u32x4, lo);
            if (row0 + j >= r) uh = ul = u32x4{0u, 0u, 0u, 0u};
            u32x4* dst = reinterpret_cast<u32x4*>(out) + ch * fr.chunk_stride + ((int64_t)(a * nkb + b) * nks * 2 + 2 * kb) * 64 + 32 * hf2 + j;
            dst[0] = uh;
            dst[64] = ul;
        }
        return;
    }
    for (int e = threadIdx.x; e < 32 * 32; e += 512) {
        const int rl = e >> 5, cl = e & 31;
        const int row = row0 + rl, col = col0 + cl;
        if (row < r && col < c) {
            const float v = blk[rl][cl] + bv;
            if constexpr (OUT_BF16) {
                reinterpret_cast<unsigned short*>(out)[(int64_t)row * ldo + col] = (unsigned short)(cvt_pk_bf16(v, 0.f) & 0xffffu);
            } else {
                reinterpret_cast<float*>(out)[(int64_t)row * ldo + col] = v;
                if (fr.out2) fr.out2[(int64_t)row * fr.ldo2 + col] = v + fr.resid[(int64_t)row * fr.ldr + col];
            }
        }
    }
}

}  // namespace

// (internal) the key projection of the pipelined attention, written as its fragment image: sparse_attn_x3p.hip owns the layout
int snf::skinny_linear_x3_kpfrag(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int r, int c, int k, int dk,
                                 int chunk_size, int64_t chunk_stride, float c_exp, void* frag, hipStream_t s, const int64_t* idx,
                                 int64_t n_rows, float* xs, int64_t ldxs, int32_t* map) {
    dim3 grid((unsigned)((c + 31) / 32), (unsigned)((r + 31) / 32));
    SkinnyFrag fr;
    fr.dk = dk, fr.chunk_size = chunk_size, fr.chunk_stride = chunk_stride, fr.c_exp = c_exp;
    fr.idx = idx, fr.n_rows = n_rows, fr.xs = xs, fr.ldxs = ldxs, fr.map = idx ? map : nullptr;
    if (fr.map) grid.y += (unsigned)((n_rows + 512ll * grid.x - 1) / (512ll * grid.x));   // the map's workgroups, behind the projection's
    hipLaunchKernelGGL((skinny_linear_x3_kernel<false, true>), grid, dim3(512), 0, s, x, ldx, w, ldw, bias, r, c, k, frag, (int64_t)0, fr);
    return snf::check_launch("skinny_linear_x3_kernel<frag>");
}

static int gemm_bf16_impl(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, const float* resid, int64_t ldr,
                          int64_t m, int n, int k, int act, void* c, int64_t ldc, int out_dtype, int tile_n, snf_stream_t stream);
extern "C" int snf_gemm_bf16(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, int64_t m, int n,
                             int k, int act, void* c, int64_t ldc, int out_dtype, int tile_n, snf_stream_t stream) {
    return gemm_bf16_impl(a, lda, w, ldw, bias, nullptr, 0, m, n, k, act, c, ldc, out_dtype, tile_n, stream);
}
// fp32 output with a residual: c = act(a w^T + bias) + resid [m, ldr]
extern "C" int snf_gemm_bf16_resid_f32(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, const float* resid,
                                       int64_t ldr, int64_t m, int n, int k, int act, float* c, int64_t ldc, int tile_n, snf_stream_t stream) {
    SNF_REQUIRE(resid && ldr >= n && ldr % 4 == 0 && reinterpret_cast<uintptr_t>(resid) % 16 == 0, "snf_gemm_bf16_resid_f32: resid [m, ldr] f32, 16-byte aligned rows");
    return gemm_bf16_impl(a, lda, w, ldw, bias, resid, ldr, m, n, k, act, c, ldc, SNF_DT_F32, tile_n, stream);
}
static int gemm_bf16_impl(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, const float* resid, int64_t ldr,
                          int64_t m, int n, int k, int act, void* c, int64_t ldc, int out_dtype, int tile_n, snf_stream_t stream) {
    SNF_REQUIRE(a && w && c, "snf_gemm_bf16: null pointer");
    SNF_REQUIRE(m >= 1 && n >= 1 && k >= 1, "snf_gemm_bf16: bad shape m=%lld n=%d k=%d", (long long)m, n, k);
    SNF_REQUIRE(act >= SNF_ACT_RELU && act <= SNF_ACT_NONE, "snf_gemm_bf16: bad activation code %d", act);
    SNF_REQUIRE(out_dtype == SNF_DT_F32 || out_dtype == SNF_DT_BF16 || out_dtype == SNF_DT_BF16_SPLIT3 || out_dtype == SNF_DT_BF16_HL,
                "snf_gemm_bf16: bad output dtype %d", out_dtype);
    SNF_REQUIRE(out_dtype != SNF_DT_BF16_HL || n % 32 == 0, "snf_gemm_bf16: an hl-image output needs n %% 32 == 0 (n = %d)", n);
    if (k % BKS || k < AHEAD * BKS || n % 8 || lda % 8 || ldw % 8 || ldc % (out_dtype == SNF_DT_F32 ? 4 : 8) || lda < k || ldw < k ||
        ldc < (out_dtype == SNF_DT_BF16_SPLIT3 ? 3 * (int64_t)n : out_dtype == SNF_DT_BF16_HL ? 2 * (int64_t)n : n) || (reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(c)) % 16 ||
        (bias && reinterpret_cast<uintptr_t>(bias) % 16) || m * lda >= 0x7fffffffll || (int64_t)n * ldw >= 0x7fffffffll) {
        snf::set_error("snf_gemm_bf16: shape m=%lld n=%d k=%d (lda %lld ldw %lld ldc %lld) outside the kernel's domain "
                       "(k %% 32, k >= 64, n %% 8, 16-byte aligned rows, 31-bit element offsets)",
                       (long long)m, n, k, (long long)lda, (long long)ldw, (long long)ldc);
        return SNF_EUNSUPPORTED;
    }
    if (tile_n != 128 && tile_n != 256) {
        // 256-wide tiles once they give most of the chip a tile (>= 0.7 tiles per CU, partial column tiles counted); below
        // that 128-wide tiles spread the work over more CUs.  Measured on the config-A / config-B / ViT shapes
        // (profiles/r02_gemm_bench.txt): the rule picks the faster width on all of them but ViT fc2 (5 % off).
        const int64_t t256 = ((m + BM - 1) / BM) * ((n + 255) / 256);
        const int cus = snf::cu_count();
        tile_n = (n > 128 && t256 * 10 >= (int64_t)cus * 7) ? 256 : 128;
    }
    GemmParams P;
    P.a = reinterpret_cast<const unsigned short*>(a);
    P.w = reinterpret_cast<const unsigned short*>(w);
    P.bias = bias;
    P.c = c;
    P.lda = lda, P.ldw = ldw, P.ldc = ldc;
    P.m = (int)m, P.n = n, P.k = k, P.act = act;
    P.tiles_m = (int)((m + BM - 1) / BM);
    P.tiles_n = (n + tile_n - 1) / tile_n;
    P.trace = nullptr;
    P.resid = resid, P.ldr = ldr;
#ifdef SNF_GEMM_TRACE
    if (const char* e = getenv("SNF_GEMM_TRACE_PTR")) P.trace = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0));
#endif
    hipStream_t s = snf::as_stream(stream);
    if (out_dtype == SNF_DT_BF16_SPLIT3) return tile_n == 256 ? launch_act<4, 2>(P, s) : launch_act<2, 2>(P, s);
    if (out_dtype == SNF_DT_BF16_HL) return tile_n == 256 ? launch_act<4, 3>(P, s) : launch_act<2, 3>(P, s);
    if (tile_n == 256) return out_dtype == SNF_DT_F32 ? launch_act<4, 1>(P, s) : launch_act<4, 0>(P, s);
    return out_dtype == SNF_DT_F32 ? launch_act<2, 1>(P, s) : launch_act<2, 0>(P, s);
}

namespace {
// shared argument checks of the epilogue variants (256-wide tiles only)
int epi_domain(const char* who, const void* a, int64_t lda, const void* w, int64_t ldw, int64_t m, int n, int k) {
    if (k % BKS || k < (AHEAD + 1) * BKS || n % 64 || lda % 8 || ldw % 8 || lda < k || ldw < k ||
        (reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w)) % 16 || m * lda >= 0x7fffffffll || (int64_t)n * ldw >= 0x7fffffffll) {
        snf::set_error("%s: shape m=%lld n=%d k=%d (lda %lld ldw %lld) outside the kernel's domain (k %% 32, k >= 96, n %% 64, 16-byte "
                       "aligned rows, 31-bit element offsets)", who, (long long)m, n, k, (long long)lda, (long long)ldw);
        return SNF_EUNSUPPORTED;
    }
    return SNF_OK;
}
GemmParams epi_params(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, int64_t m, int n, int k, int act, void* c,
                      int64_t ldc) {
    GemmParams P;
    P.a = reinterpret_cast<const unsigned short*>(a);
    P.w = reinterpret_cast<const unsigned short*>(w);
    P.bias = bias;
    P.c = c;
    P.lda = lda, P.ldw = ldw, P.ldc = ldc;
    P.m = (int)m, P.n = n, P.k = k, P.act = act;
    P.tiles_m = (int)((m + BM - 1) / BM);
    P.tiles_n = (n + 255) / 256;
    P.trace = nullptr;
    return P;
}
}  // namespace

extern "C" int snf_gemm_bf16_lnfold(const void* a, int64_t lda, const void* w, int64_t ldw, const float* colsum, const float* bias,
                                    const float* rowstats, int64_t m, int n, int k, int act, void* c, int64_t ldc, snf_stream_t stream) {
    SNF_REQUIRE(a && w && c && colsum && bias && rowstats, "snf_gemm_bf16_lnfold: null pointer");
    SNF_REQUIRE(m >= 1 && n >= 1 && k >= 1, "snf_gemm_bf16_lnfold: bad shape m=%lld n=%d k=%d", (long long)m, n, k);
    SNF_REQUIRE(act == SNF_ACT_NONE || act == SNF_ACT_GELU, "snf_gemm_bf16_lnfold: activation %d not built (none, gelu)", act);
    int rc = epi_domain("snf_gemm_bf16_lnfold", a, lda, w, ldw, m, n, k);
    if (rc) return rc;
    SNF_REQUIRE(ldc >= n && ldc % 8 == 0 && reinterpret_cast<uintptr_t>(c) % 16 == 0 && reinterpret_cast<uintptr_t>(colsum) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(bias) % 16 == 0 && reinterpret_cast<uintptr_t>(rowstats) % 8 == 0,
                "snf_gemm_bf16_lnfold: output / vector alignment");
    GemmParams P = epi_params(a, lda, w, ldw, bias, m, n, k, act, c, ldc);
    P.colsum = colsum, P.rowstats = rowstats;
    hipStream_t s = snf::as_stream(stream);
    return act == SNF_ACT_GELU ? launch<4, SNF_ACT_GELU, 0, 1>(P, s) : launch<4, SNF_ACT_NONE, 0, 1>(P, s);
}

extern "C" int snf_gemm_bf16_resid(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, int64_t m, int n, int k,
                                   float* x, int64_t ldx, void* x_bf16, int64_t ldxb, float* stats_part, snf_stream_t stream) {
    SNF_REQUIRE(a && w && x && x_bf16 && stats_part && bias, "snf_gemm_bf16_resid: null pointer");
    SNF_REQUIRE(m >= 1 && n >= 1 && k >= 1, "snf_gemm_bf16_resid: bad shape m=%lld n=%d k=%d", (long long)m, n, k);
    int rc = epi_domain("snf_gemm_bf16_resid", a, lda, w, ldw, m, n, k);
    if (rc) return rc;
    SNF_REQUIRE(ldx >= n && ldx % 4 == 0 && ldxb >= n && ldxb % 8 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(x_bf16) % 16 == 0 && reinterpret_cast<uintptr_t>(bias) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(stats_part) % 16 == 0 && m * ldx < 0x7fffffffll,
                "snf_gemm_bf16_resid: output / vector alignment");
    GemmParams P = epi_params(a, lda, w, ldw, bias, m, n, k, SNF_ACT_NONE, x, ldx);
    P.c2 = reinterpret_cast<unsigned short*>(x_bf16), P.ldc2 = ldxb;
    P.stats_part = stats_part, P.slots = n / 32;
    // 128-wide tiles where the last 256-wide column tile would be at most half used (ViT-S, n = 384: 3 tiles of 128, not 2 of 256)
    const int rem = n % 256;
    if (rem >= 1 && rem <= 128) {
        P.tiles_n = (n + 127) / 128;
        return launch_tile<2, SNF_ACT_NONE, 1, 2>(P, snf::as_stream(stream));
    }
    return launch<4, SNF_ACT_NONE, 1, 2>(P, snf::as_stream(stream));
}

extern "C" int snf_gemm_hl_bf16(const void* a_hl, int64_t lda, const void* w_hl, int64_t ldw, const float* bias, int64_t m, int n,
                                int k, int act, void* c, int64_t ldc, int out_dtype, snf_stream_t stream) {
    return snf_gemm_hl_resid_bf16(a_hl, lda, w_hl, ldw, bias, nullptr, 0, m, n, k, act, c, ldc, out_dtype, stream);
}

namespace {
// split-K geometry of the last round (see GemmParams): the largest remainder r over the 8 XCDs and the largest part count
struct HlSplit {
    int cap, rcap;
};
HlSplit hl_split_geometry(int64_t m, int n, int k) {
    HlSplit g = {0, 0};
    if (m < 1 || n < 1 || k < BKS) return g;
    const int64_t ntiles64 = ((m + BM - 1) / BM) * ((n + 255) / 256);
    if (ntiles64 > 0x7fffffff) return g;
    const int ntiles = (int)ntiles64, ns = k / BKS;
    int grid = snf::cu_count() & ~7;
    if (grid < 8) grid = 8;
    const int per_xcd = (ntiles + 7) / 8;
    if (per_xcd * 8 < grid) grid = per_xcd * 8;
    const int w = grid >> 3;
    // Only where the partly filled round is a large share of the launch: measured (tools/gemm_hl_splitk_time.py) 423 -> 401 us on
    // config B's FFN output projection (1.5 rounds), but +1..2 % on config C's shapes (4.6 .. 18 rounds): behind many full rounds
    // the second launch and the slab round trip cost more than the idle workgroups of the last round did -- a half-filled chip
    // runs its tiles 1.25 .. 1.4x faster (clock 2.18 -> 2.40 GHz at the 1.4 kW cap, less L2 / fabric contention).
    if (ntiles > 2 * grid) return g;
    for (int x = 0; x < 8; ++x) {
        const int q = ntiles / 8 + (x < (ntiles & 7) ? 1 : 0), r = q % w;
        if (r == 0) continue;
        int sn = w / r;
        if (sn > HL_SPLIT_MAX) sn = HL_SPLIT_MAX;
        if (sn > ns / HL_SPLIT_MIN_STEPS) sn = ns / HL_SPLIT_MIN_STEPS;
        if (sn < 2) continue;
        if (sn > g.cap) g.cap = sn;
        if (r > g.rcap) g.rcap = r;
    }
    if (g.cap < 2) g.cap = g.rcap = 0;
    return g;
}
constexpr size_t HL_TICKET_BYTES = 4096;
}  // namespace

// Workspace of snf_gemm_hl_ws_bf16 for this shape; 0 = the shape has no partly filled last round worth splitting.  Plain scratch
// memory: the call zeroes its tickets (the first 4096 bytes) itself, so the buffer may come fresh from any allocator and carries no state
// between calls -- concurrent calls on different streams only need DIFFERENT buffers.
extern "C" size_t snf_gemm_hl_ws_bytes(int64_t m, int n, int k) {
    const HlSplit g = hl_split_geometry(m, n, k);
    if (g.cap < 2 || 8 * g.rcap * sizeof(unsigned int) > HL_TICKET_BYTES) return 0;
    return HL_TICKET_BYTES + (size_t)8 * g.rcap * g.cap * (size_t)(BM * 256) * sizeof(float);
}

extern "C" int snf_gemm_hl_resid_bf16(const void* a_hl, int64_t lda, const void* w_hl, int64_t ldw, const float* bias,
                                      const float* resid, int64_t ldr, int64_t m, int n, int k, int act, void* c, int64_t ldc,
                                      int out_dtype, snf_stream_t stream) {
    return snf_gemm_hl_ws_bf16(a_hl, lda, w_hl, ldw, bias, resid, ldr, m, n, k, act, c, ldc, out_dtype, nullptr, 0, stream);
}

extern "C" int snf_gemm_hl_ws_bf16(const void* a_hl, int64_t lda, const void* w_hl, int64_t ldw, const float* bias,
                                   const float* resid, int64_t ldr, int64_t m, int n, int k, int act, void* c, int64_t ldc,
                                   int out_dtype, void* workspace, size_t workspace_bytes, snf_stream_t stream) {
    SNF_REQUIRE(a_hl && w_hl && c, "snf_gemm_hl_bf16: null pointer");
    SNF_REQUIRE(m >= 1 && n >= 1 && k >= 1, "snf_gemm_hl_bf16: bad shape m=%lld n=%d k=%d", (long long)m, n, k);
    SNF_REQUIRE(act >= SNF_ACT_RELU && act <= SNF_ACT_NONE, "snf_gemm_hl_bf16: bad activation code %d", act);
    SNF_REQUIRE(out_dtype == SNF_DT_F32 || out_dtype == SNF_DT_BF16 || out_dtype == SNF_DT_BF16_HL,
                "snf_gemm_hl_bf16: bad output dtype %d", out_dtype);
    const bool hl_out = out_dtype == SNF_DT_BF16_HL;
    SNF_REQUIRE(!resid || (out_dtype == SNF_DT_F32 && ldr >= n && ldr % 4 == 0 && reinterpret_cast<uintptr_t>(resid) % 16 == 0),
                "snf_gemm_hl_bf16: a residual needs an fp32 output and 16-byte aligned rows (ldr=%lld)", (long long)ldr);
    if (k % BKS || k < BKS || n % 8 || (hl_out && n % 32) || lda % 8 || ldw % 8 || ldc % (out_dtype == SNF_DT_F32 ? 4 : 8) ||
        lda < 2 * (int64_t)k || ldw < 2 * (int64_t)k || ldc < (hl_out ? 2 * (int64_t)n : n) ||
        (reinterpret_cast<uintptr_t>(a_hl) | reinterpret_cast<uintptr_t>(w_hl) | reinterpret_cast<uintptr_t>(c)) % 16 ||
        (bias && reinterpret_cast<uintptr_t>(bias) % 16) || m * lda >= 0x7fffffffll || (int64_t)n * ldw >= 0x7fffffffll) {
        snf::set_error("snf_gemm_hl_bf16: shape m=%lld n=%d k=%d (lda %lld ldw %lld ldc %lld) outside the kernel's domain "
                       "(k %% 32, n %% 8 (%% 32 for an hl output), images of 2 k columns, 16-byte aligned rows, 31-bit element offsets)",
                       (long long)m, n, k, (long long)lda, (long long)ldw, (long long)ldc);
        return SNF_EUNSUPPORTED;
    }
    GemmParams P;
    P.a = reinterpret_cast<const unsigned short*>(a_hl);
    P.w = reinterpret_cast<const unsigned short*>(w_hl);
    P.bias = bias;
    P.c = c;
    P.lda = lda, P.ldw = ldw, P.ldc = ldc;
    P.m = (int)m, P.n = n, P.k = k, P.act = act;
    P.tiles_m = (int)((m + BM - 1) / BM);
    P.tiles_n = (n + 255) / 256;
    P.trace = nullptr;
    P.resid = resid;
    P.ldr = ldr;
    if (workspace) {
        const size_t need = snf_gemm_hl_ws_bytes(m, n, k);
        if (need) {
            if (workspace_bytes < need || reinterpret_cast<uintptr_t>(workspace) % 16) {
                snf::set_error("snf_gemm_hl_ws_bf16: workspace %zu < %zu (or not 16-byte aligned)", workspace_bytes, need);
                return SNF_EWORKSPACE;
            }
            const HlSplit g = hl_split_geometry(m, n, k);
            P.split_cnt = reinterpret_cast<unsigned int*>(workspace);
            P.split_slab = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + HL_TICKET_BYTES);
            P.split_cap = g.cap, P.split_rcap = g.rcap;
        }
    }
    hipStream_t s = snf::as_stream(stream);
    if (hl_out) return launch_hl_act<3>(P, s);
    return out_dtype == SNF_DT_F32 ? launch_hl_act<1>(P, s) : launch_hl_act<0>(P, s);
}

extern "C" int snf_gemm_hl_gated_bf16(const void* a_hl, int64_t lda, const void* w_hl, int64_t ldw, const void* gate_hl, int64_t ldg, int64_t m,
                                      int n, int k, void* c_hl, int64_t ldc, snf_stream_t stream) {
    SNF_REQUIRE(a_hl && w_hl && gate_hl && c_hl, "snf_gemm_hl_gated_bf16: null pointer");
    SNF_REQUIRE(m >= 1 && n >= 1 && k >= 1, "snf_gemm_hl_gated_bf16: bad shape m=%lld n=%d k=%d", (long long)m, n, k);
    if (k % BKS || k < BKS || n % 32 || lda % 8 || ldw % 8 || ldc % 8 || ldg % 8 || lda < 2 * (int64_t)k || ldw < 2 * (int64_t)k ||
        ldc < 2 * (int64_t)n || ldg < 2 * (int64_t)n ||
        (reinterpret_cast<uintptr_t>(a_hl) | reinterpret_cast<uintptr_t>(w_hl) | reinterpret_cast<uintptr_t>(c_hl) |
         reinterpret_cast<uintptr_t>(gate_hl)) % 16 ||
        m * lda >= 0x7fffffffll || (int64_t)n * ldw >= 0x7fffffffll) {
        snf::set_error("snf_gemm_hl_gated_bf16: shape m=%lld n=%d k=%d (lda %lld ldw %lld ldc %lld ldg %lld) outside the kernel's domain "
                       "(k %% 32, n %% 32, images of 2 k / 2 n columns, 16-byte aligned rows, 31-bit element offsets)",
                       (long long)m, n, k, (long long)lda, (long long)ldw, (long long)ldc, (long long)ldg);
        return SNF_EUNSUPPORTED;
    }
    GemmParams P;
    P.a = reinterpret_cast<const unsigned short*>(a_hl);
    P.w = reinterpret_cast<const unsigned short*>(w_hl);
    P.bias = nullptr;
    P.c = c_hl;
    P.lda = lda, P.ldw = ldw, P.ldc = ldc;
    P.m = (int)m, P.n = n, P.k = k, P.act = SNF_ACT_NONE;
    P.tiles_m = (int)((m + BM - 1) / BM);
    P.tiles_n = (n + 255) / 256;
    P.trace = nullptr;
    P.gate = reinterpret_cast<const unsigned short*>(gate_hl), P.ldg = ldg;
    return launch_hl_gated(P, snf::as_stream(stream));
}

// out [r, c] (f32 or bf16, row pitch ldo) = x [r, k] (f32, row pitch ldx) w [c, k]^T (f32, row pitch ldw) + bias [c] (nullable), fp32-class
// (split-bf16 x3 on the matrix cores, fp32 accumulate).  For the few hundred selected rows of a bag; r <= 8192, k % 64 == 0.
extern "C" int snf_linear_rows_x3_f32(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int r, int c, int k,
                                      void* out, int64_t ldo, int out_dtype, snf_stream_t stream) {
    SNF_REQUIRE(x && w && out, "snf_linear_rows_x3_f32: null pointer");
    SNF_REQUIRE(r >= 1 && r <= 8192 && c >= 1 && k >= 16 && k % 16 == 0, "snf_linear_rows_x3_f32: bad shape r=%d c=%d k=%d "
                "(r <= 8192, k %% 16 == 0)", r, c, k);
    SNF_REQUIRE(out_dtype == SNF_DT_F32 || out_dtype == SNF_DT_BF16, "snf_linear_rows_x3_f32: bad out dtype %d", out_dtype);
    SNF_REQUIRE(ldx >= k && ldw >= k && ldo >= c && (ldx % 4) == 0 && (ldw % 4) == 0 &&
                    ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) == 0,
                "snf_linear_rows_x3_f32: x / w rows must be 16-byte aligned");
    const dim3 grid((unsigned)((c + 31) / 32), (unsigned)((r + 31) / 32));
    hipStream_t s = snf::as_stream(stream);
    if (out_dtype == SNF_DT_BF16)
        hipLaunchKernelGGL(skinny_linear_x3_kernel<true>, grid, dim3(512), 0, s, x, ldx, w, ldw, bias, r, c, k, out, ldo);
    else
        hipLaunchKernelGGL(skinny_linear_x3_kernel<false>, grid, dim3(512), 0, s, x, ldx, w, ldw, bias, r, c, k, out, ldo);
    return snf::check_launch("skinny_linear_x3_kernel");
}

// the same with a second output out2 [r, c] = out + resid [r, c] (f32): delta = o Wo^T + bo AND x_sel = xs + delta (snuffy.py:205, 108)
extern "C" int snf_linear_rows_x3_resid_f32(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, const float* resid,
                                            int64_t ldr, int r, int c, int k, float* out, int64_t ldo, float* out2, int64_t ldo2,
                                            snf_stream_t stream) {
    SNF_REQUIRE(x && w && out && out2 && resid, "snf_linear_rows_x3_resid_f32: null pointer");
    SNF_REQUIRE(r >= 1 && r <= 8192 && c >= 1 && k >= 16 && k % 16 == 0, "snf_linear_rows_x3_resid_f32: bad shape r=%d c=%d k=%d "
                "(r <= 8192, k %% 16 == 0)", r, c, k);
    SNF_REQUIRE(ldx >= k && ldw >= k && ldo >= c && ldo2 >= c && ldr >= c && (ldx % 4) == 0 && (ldw % 4) == 0 &&
                    ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) == 0,
                "snf_linear_rows_x3_resid_f32: x / w rows must be 16-byte aligned");
    const dim3 grid((unsigned)((c + 31) / 32), (unsigned)((r + 31) / 32));
    SkinnyFrag fr;
    fr.resid = resid, fr.ldr = ldr, fr.out2 = out2, fr.ldo2 = ldo2;
    hipLaunchKernelGGL(skinny_linear_x3_kernel<false>, grid, dim3(512), 0, snf::as_stream(stream), x, ldx, w, ldw, bias, r, c, k, out, ldo, fr);
    return snf::check_launch("skinny_linear_x3_kernel");
}
