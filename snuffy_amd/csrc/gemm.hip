// K5 / K10 / K12-K14: the dense projections of the hot path on the CDNA4 matrix cores.
//
//   C[M, N] = act(A[M, K] * W[N, K]^T + bias[N])       bf16 operands, fp32 accumulate, bf16 (or fp32) out
//
// replaces nn.Linear of snuffy.py:187-190 (Q | V projection), snuffy.py:224-225 (w_1 + activation, w_2) and the ViT
// Linear / Conv2d-as-GEMM layers (utils_ssls_cf/vision_transformer_with_adapter_dino_version.py:51-67, 82-94, 141-146).
// Both operands are K-major (A row-major, W as nn.Linear stores it), so the same LDS image serves both.
//
// Structure (one 256 x BN output tile per workgroup, BN = 256 or 128, 8 waves = 2 (M) x 4 (N), K step 64):
//   * HBM -> LDS by LDS-DMA (global_load_lds, 16 B per lane) in HALF-TILES of 128 rows x 64 k = 16 KiB, one half-tile per
//     phase, two K tiles of LDS (128 KiB).  The destination of an LDS-DMA is lane-linear, so the bank swizzle is applied
//     to the SOURCE address: lane i of a piece fetches the 16-byte chunk that belongs at position i of the image.
//   * image: row r of a half-tile is 128 contiguous bytes (8 chunks of 8 k); chunk q sits at slot q ^ sw(r).  For the
//     operand whose fragment rows are consecutive (A) sw = (r >> 1) & 7; for W the wave reads rows 16 g + 4 ni + r'
//     (see the epilogue) and sw = ((r >> 1) & 1) | (((r >> 4) & 3) << 1): every ds_read_b128 lane group then covers the
//     16 slots of a 256-byte bank row exactly once (conflict-free, derivation in DESIGN.md).
//   * MFMA v_mfma_f32_16x16x32_bf16 with the operands SWAPPED (W fragment as srcA, A fragment as srcB): the accumulator
//     of a lane then holds 4 consecutive OUTPUT COLUMNS of one output row; with the W rows of fragment ni, slot 4 g + r'
//     chosen as column 16 g + 4 ni + r' a lane owns 16 consecutive columns of a row -> two 16-byte stores per row.
//   * schedule: the K tile is cut into 4 phases (one quadrant of the wave's 128 x 64 output each, 16 MFMAs); a phase is
//     {LDS reads of the quadrant's new fragments, LDS-DMA of one future half-tile, s_barrier, 16 MFMAs, s_barrier}.  The two
//     wave groups (wr = 0 / 1, one wave of each per SIMD) run ONE BARRIER APART, so one wave of a SIMD issues its MFMA
//     cluster while the other reads LDS and issues DMA.  The DMA runs two half-tiles ahead: the only wait in the loop is
//     one counted s_waitcnt vmcnt(4) per K tile (never 0), placed one phase before the first read of the tile it retires.
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
};

constexpr int BM = 256, BK = 64;
constexpr int HT_BYTES = 128 * BK * 2;   // one half-tile: 128 rows x 64 k bf16 = 16 KiB

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ int sw_a(int r) { return (r >> 1) & 7; }
// W rows of fragment ni, slot 4 g + r' are 16 g + 4 ni + r' (BN = 256, GS = 4) or 8 g + 4 ni + r' (BN = 128, GS = 3)
template <int GS>
__device__ __forceinline__ int sw_w(int r) { return ((r >> 1) & 1) | (((r >> GS) & 3) << 1); }

// erf, Abramowitz & Stegun 7.1.26 (|err| <= 1.5e-7), branch-free
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
    const float r = fmaf(-p * t, e, 1.0f);
    return copysignf(r, x);
}

// two f32 -> packed bf16x2 (v_cvt_pk_bf16_f32: round to nearest even, NaN kept)
__device__ __forceinline__ unsigned int cvt_pk_bf16(float lo, float hi) {
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
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

// NI = 16-column W fragments per wave: 4 -> BN = 256, 2 -> BN = 128.  OUT_F32: fp32 output instead of bf16.
template <int NI, int ACT, bool OUT_F32>
__global__ __launch_bounds__(512, 2) void gemm_bf16_kernel(GemmParams P) {
    constexpr int BN = 64 * NI;
    constexpr int NHT_W = BN / 128;               // W half-tiles per K tile (2 or 1)
    constexpr int NHT = 2 + NHT_W;                // half-tiles per K tile
    constexpr int NPH = (NI == 4) ? 4 : 2;        // phases per K tile
    constexpr int KT_BYTES = NHT * HT_BYTES;
    constexpr int GS = (NI == 4) ? 4 : 3;        // log2 of the column-group pitch of the W fragment rows
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2 K tiles][NHT half-tiles][16 KiB]

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 2, wc = wid & 3;

    // XCD-aware tile order: workgroup b runs on XCD b % 8; give every XCD a contiguous range of logical tiles, and walk
    // the tiles of one A row panel first, so the panel is fetched from HBM once per XCD and W stays in that XCD's L2
    const int nwg = P.tiles_m * P.tiles_n;
    int tile;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int tile_m = tile / P.tiles_n, tile_n = tile - tile_m * P.tiles_n;
    const int nt = P.k / BK;

    // ---- LDS-DMA source offsets (elements), two 1-KiB pieces per wave and half-tile: piece p = 2 wid + j covers rows
    //      8 p .. 8 p + 7; lane i lands at byte 16 i of the piece = row 8 p + (i >> 3), slot i & 7
    int goff[NHT][2];
    {
        const int slot = lane & 7;
#pragma unroll
        for (int ht = 0; ht < NHT; ++ht)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = (2 * wid + j) * 8 + (lane >> 3);
                if (ht < 2) {
                    int grow = tile_m * BM + ht * 128 + row;
                    if (grow > P.m - 1) grow = P.m - 1;
                    goff[ht][j] = grow * (int)P.lda + ((slot ^ sw_a(row)) << 3);
                } else {
                    int grow = tile_n * BN + (ht - 2) * 128 + row;
                    if (grow > P.n - 1) grow = P.n - 1;
                    goff[ht][j] = grow * (int)P.ldw + ((slot ^ sw_w<GS>(row)) << 3);
                }
            }
    }
    // this lane's output columns and their bias (requested first, used in the epilogue: the latency hides under the loop)
    const int n0 = tile_n * BN + (NI == 4 ? 64 : 32) * wc + (NI == 4 ? 16 : 8) * (lane >> 4);   // first column of this lane
    constexpr int NC = 4 * NI;                                                                 // columns per lane
    const bool col_ok = n0 + NC <= P.n;
    f32x4 bv4[NC / 4];
#pragma unroll
    for (int c4 = 0; c4 < NC / 4; ++c4) bv4[c4] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (P.bias && col_ok) {
#pragma unroll
        for (int c4 = 0; c4 < NC / 4; ++c4) bv4[c4] = *reinterpret_cast<const f32x4*>(P.bias + n0 + 4 * c4);
    }
    auto stage = [&](auto ht_t, int u, int buf) __attribute__((always_inline)) {
        constexpr int ht = decltype(ht_t)::value;
        const unsigned short* base = ht < 2 ? P.a : P.w;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned char* dst = smem + buf * KT_BYTES + ht * HT_BYTES + (2 * wid + j) * 1024;
            __builtin_amdgcn_global_load_lds((glb_void*)(base + goff[ht][j] + u * BK), (lds_void*)dst, 16, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes inside a half-tile), k half kk = 0 / 1
    const int fi = lane & 15, fg = lane >> 4;
    int xoff[2], woff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        xoff[kk] = fi * 128 + (((4 * kk + fg) ^ sw_a(fi)) << 4);                                   // + mi * 2048
        const int wrow = (NI == 4 ? 64 * (wc & 1) : 32 * wc) + (1 << GS) * (fi >> 2) + (fi & 3);   // + 4 ni
        woff[kk] = wrow * 128 + (((4 * kk + fg) ^ sw_w<GS>(wrow)) << 4);                           // + ni * 512
    }
    const int a_ht = wr;                               // this wave's A half-tile
    const int w_ht = 2 + (NI == 4 ? (wc >> 1) : 0);    // and W half-tile

    f32x4 acc[8][NI];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[4][2];          // A fragments of the current 64-row half of the wave's rows
    bf16x8 wf[NI][2];         // W fragments (all of the wave's columns stay in registers over the K tile)

    auto lds_frag = [&](const unsigned char* p) __attribute__((always_inline)) -> bf16x8 {
#ifdef SNF_GEMM_NOLDS   // timing ablation (tools/gemm_variants.sh): no LDS reads
        u32x4 z = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, (unsigned)(uintptr_t)p};
        asm volatile("" : "+v"(z));
        return __builtin_bit_cast(bf16x8, z);
#else
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
#endif
    };
    auto read_x = [&](int buf, auto ah_t) __attribute__((always_inline)) {
        constexpr int ah = decltype(ah_t)::value;
        const unsigned char* base = smem + buf * KT_BYTES + a_ht * HT_BYTES;
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) xf[m4][kk] = lds_frag(base + xoff[kk] + (4 * ah + m4) * 2048);
    };
    auto read_w = [&](int buf, auto lo_t, auto cnt_t) __attribute__((always_inline)) {
        constexpr int lo = decltype(lo_t)::value, cnt = decltype(cnt_t)::value;
        const unsigned char* base = smem + buf * KT_BYTES + w_ht * HT_BYTES;
#pragma unroll
        for (int n2 = 0; n2 < cnt; ++n2)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) wf[lo + n2][kk] = lds_frag(base + woff[kk] + (lo + n2) * 512);
    };
    auto mma = [&](auto ah_t, auto lo_t, auto cnt_t) __attribute__((always_inline)) {
        constexpr int ah = decltype(ah_t)::value, lo = decltype(lo_t)::value, cnt = decltype(cnt_t)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int m4 = 0; m4 < 4; ++m4)
#pragma unroll
                for (int n2 = 0; n2 < cnt; ++n2)
#ifdef SNF_GEMM_NOMFMA   // timing ablation: operands kept alive, no matrix work
                    asm volatile("" : "+v"(acc[4 * ah + m4][lo + n2]) : "v"(wf[lo + n2][kk]), "v"(xf[m4][kk]));
#else
                    acc[4 * ah + m4][lo + n2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[lo + n2][kk], xf[m4][kk],
                                                                                       acc[4 * ah + m4][lo + n2], 0, 0, 0);
#endif
        __builtin_amdgcn_s_setprio(0);
    };
    // end of a phase's load part: this wave's LDS reads have returned (so the slot may be re-staged one phase later),
    // then the workgroup barrier that hands the matrix pipe over
    auto load_done = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mma_done = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // ---- prologue: K tile 0 complete, the W half-tiles of K tile 1 in flight
    static_for<0, NHT_W>([&](auto h) __attribute__((always_inline)) { stage(std::integral_constant<int, 2 + decltype(h)::value>{}, 0, 0); });
    stage(I0{}, 0, 0);
    stage(I1{}, 0, 0);
    if (nt > 1) {
        static_for<0, NHT_W>([&](auto h) __attribute__((always_inline)) { stage(std::integral_constant<int, 2 + decltype(h)::value>{}, 1, 1); });
        if constexpr (NHT_W == 2)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) __builtin_amdgcn_s_barrier();   // the second wave group runs one barrier behind the first

    auto ktile = [&](int t, auto buf_t) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_t)::value;
#ifdef SNF_GEMM_NOSTAGE   // timing ablation: no LDS-DMA inside the loop
        const bool nxt = false, nxt2 = false;
#else
        const bool nxt = t + 1 < nt, nxt2 = t + 2 < nt;
#endif
        if constexpr (NI == 4) {
            // P0: quadrant (rows 0..63, columns 0..31)
            read_w(buf, I0{}, I2{});
            read_x(buf, I0{});
            if (nxt) stage(I0{}, t + 1, buf ^ 1);
            load_done();
            mma(I0{}, I0{}, I2{});
            mma_done();
            // P1: (rows 0..63, columns 32..63)
            read_w(buf, I2{}, I2{});
            if (nxt) stage(I1{}, t + 1, buf ^ 1);
            load_done();
            mma(I0{}, I2{}, I2{});
            mma_done();
            // P2: (rows 64..127, columns 32..63); the W half-tiles of this K tile are dead: re-stage them for t + 2
            read_x(buf, I1{});
            if (nxt2) stage(I2{}, t + 2, buf);
            load_done();
            mma(I1{}, I2{}, I2{});
            mma_done();
            // P3: (rows 64..127, columns 0..31), no LDS reads
            if (nxt2) {
                stage(I3{}, t + 2, buf);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // K tile t + 1 has landed; W(t + 2) stays in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            load_done();
            mma(I1{}, I0{}, I2{});
            mma_done();
        } else {
            // BN = 128: two phases per K tile (rows 0..63 / 64..127 x all 32 columns of the wave)
            read_w(buf, I0{}, I2{});
            read_x(buf, I0{});
            if (nxt) {
                stage(I0{}, t + 1, buf ^ 1);
                stage(I1{}, t + 1, buf ^ 1);
            }
            load_done();
            mma(I0{}, I0{}, I2{});
            mma_done();
            read_x(buf, I1{});
            if (nxt2) {
                stage(I2{}, t + 2, buf);
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // K tile t + 1 has landed; W(t + 2) stays in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            load_done();
            mma(I1{}, I0{}, I2{});
            mma_done();
        }
    };
    for (int t = 0; t < nt; t += 2) {
        ktile(t, I0{});
        if (t + 1 < nt) ktile(t + 1, I1{});
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();   // barrier counts of the two wave groups match again

    // ---- epilogue: lane owns rows 16 mi + fi and NC consecutive columns from n0: acc[mi][ni][r] = C[row][n0 + 4 ni + r]
    float bv[NC];
#pragma unroll
    for (int c4 = 0; c4 < NC / 4; ++c4) {
        asm volatile("" : "+v"(bv4[c4]));   // one unconditional wait for the bias here, none inside the row blocks below
        bv[4 * c4] = bv4[c4][0], bv[4 * c4 + 1] = bv4[c4][1], bv[4 * c4 + 2] = bv4[c4][2], bv[4 * c4 + 3] = bv4[c4][3];
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int row = tile_m * BM + 128 * wr + 16 * mi + fi;
        if (row < P.m && col_ok) {
            float v[NC];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * ni + r] = activate<ACT>(acc[mi][ni][r] + bv[4 * ni + r]);
            if constexpr (OUT_F32) {
                float* dst = reinterpret_cast<float*>(P.c) + (int64_t)row * P.ldc + n0;
#pragma unroll
                for (int c4 = 0; c4 < NC / 4; ++c4)
                    *reinterpret_cast<f32x4*>(dst + 4 * c4) = f32x4{v[4 * c4], v[4 * c4 + 1], v[4 * c4 + 2], v[4 * c4 + 3]};
            } else {
                unsigned short* dst = reinterpret_cast<unsigned short*>(P.c) + (int64_t)row * P.ldc + n0;
#pragma unroll
                for (int c8 = 0; c8 < NC / 8; ++c8) {
                    const u32x4 pk = {cvt_pk_bf16(v[8 * c8], v[8 * c8 + 1]), cvt_pk_bf16(v[8 * c8 + 2], v[8 * c8 + 3]),
                                      cvt_pk_bf16(v[8 * c8 + 4], v[8 * c8 + 5]), cvt_pk_bf16(v[8 * c8 + 6], v[8 * c8 + 7])};
#ifdef SNF_GEMM_NOSTORE   // timing ablation: epilogue arithmetic without the stores
                    asm volatile("" ::"v"(pk), "v"(dst));
#else
                    *reinterpret_cast<u32x4*>(dst + 8 * c8) = pk;
#endif
                }
            }
        }
    }
}

template <int NI, int ACT, bool OUT_F32>
int launch(const GemmParams& P, hipStream_t s) {
    constexpr int lds = 2 * (2 + NI / 2) * HT_BYTES;
    static thread_local bool attr_set = false;
    auto kern = gemm_bf16_kernel<NI, ACT, OUT_F32>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
            hipSuccess) {
            snf::set_error("gemm_bf16: cannot reserve %d bytes of LDS", lds);
            (void)hipGetLastError();
            return SNF_ELAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(P.tiles_m * P.tiles_n), dim3(512), lds, s, P);
    return snf::check_launch("gemm_bf16_kernel");
}

template <int NI, bool OUT_F32>
int launch_act(const GemmParams& P, hipStream_t s) {
    switch (P.act) {
        case SNF_ACT_RELU: return launch<NI, SNF_ACT_RELU, OUT_F32>(P, s);
        case SNF_ACT_GELU: return launch<NI, SNF_ACT_GELU, OUT_F32>(P, s);
        case SNF_ACT_LEAKYRELU: return launch<NI, SNF_ACT_LEAKYRELU, OUT_F32>(P, s);
        case SNF_ACT_SELU: return launch<NI, SNF_ACT_SELU, OUT_F32>(P, s);
        default: return launch<NI, SNF_ACT_NONE, OUT_F32>(P, s);
    }
}

}  // namespace

extern "C" int snf_gemm_bf16(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, int64_t m, int n,
                             int k, int act, void* c, int64_t ldc, int out_dtype, int tile_n, snf_stream_t stream) {
    SNF_REQUIRE(a && w && c, "snf_gemm_bf16: null pointer");
    SNF_REQUIRE(m >= 1 && n >= 1 && k >= 1, "snf_gemm_bf16: bad shape m=%lld n=%d k=%d", (long long)m, n, k);
    SNF_REQUIRE(act >= SNF_ACT_RELU && act <= SNF_ACT_NONE, "snf_gemm_bf16: bad activation code %d", act);
    SNF_REQUIRE(out_dtype == SNF_DT_F32 || out_dtype == SNF_DT_BF16, "snf_gemm_bf16: bad output dtype %d", out_dtype);
    if (k % BK || n % 16 || lda % 8 || ldw % 8 || ldc % (out_dtype == SNF_DT_F32 ? 4 : 8) || lda < k || ldw < k || ldc < n ||
        (reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(c)) % 16 ||
        (bias && reinterpret_cast<uintptr_t>(bias) % 16) || m * lda >= 0x7fffffffll || (int64_t)n * ldw >= 0x7fffffffll) {
        snf::set_error("snf_gemm_bf16: shape m=%lld n=%d k=%d (lda %lld ldw %lld ldc %lld) outside the kernel's domain "
                       "(k %% 64, n %% 16, 16-byte aligned rows, 31-bit element offsets)",
                       (long long)m, n, k, (long long)lda, (long long)ldw, (long long)ldc);
        return SNF_EUNSUPPORTED;
    }
    if (tile_n != 128 && tile_n != 256) {
        // 256-wide tiles when they fill the chip evenly (whole rounds of workgroups, or many rounds); else 128-wide
        const int64_t t256 = ((m + BM - 1) / BM) * ((n + 255) / 256);
        const int cus = snf::cu_count();
        tile_n = (n % 256 == 0 && (t256 % cus == 0 || t256 >= 3 * (int64_t)cus)) ? 256 : 128;
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
    hipStream_t s = snf::as_stream(stream);
    if (tile_n == 256) return out_dtype == SNF_DT_F32 ? launch_act<4, true>(P, s) : launch_act<4, false>(P, s);
    return out_dtype == SNF_DT_F32 ? launch_act<2, true>(P, s) : launch_act<2, false>(P, s);
}
