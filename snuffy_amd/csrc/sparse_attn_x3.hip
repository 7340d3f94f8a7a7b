// K7 (fp32-class form on the matrix cores): Snuffy's sparse attention with SPLIT-bf16 operands.
//
//   per head a:   P_a = softmax_j(Q_a Kp_a^T * scale)  [n, k]      O_a = P_a^T V_a  [k, dk]        (snuffy.py:160-168)
//
// The reference computes this in fp32.  gfx950 has no fast fp32 matrix path (v_mfma_f32_32x32x2_f32 runs at the vector
// rate, 1/16 of bf16), so every fp32 operand x is split into x = hi + lo with hi = bf16(x), lo = bf16(x - hi), and every
// product a b is taken as  ah bh + ah bl + al bh  -- three bf16 MFMAs with fp32 accumulate; the dropped term al bl is
// 2^-17 relative (fp32-class: measured <= 3.5e-6 on P and <= 1e-5 of its scale on O against the fp64 oracle on nine shapes,
// tests/test_gpu_kernels.py; the exact vector-ALU kernel gives 4e-7 / 9e-7, the bf16 kernel 5e-3).
// Softmax, the normalisation and all accumulation are fp32; P is split after the normalisation.
//
// Organisation (round 3): one workgroup (8 waves) per CU walks (head, 64-ROW tile) items and all 8 waves cooperate on a tile:
//   Kp      wave w owns key block w (32 keys) and keeps its hi / lo MFMA fragments in REGISTERS for the whole head (64 VGPRs at
//           dk = 128) -- a 112 KiB LDS image in the first version, which left room for 32-row tiles only
//   stage   Q and V rows: fp32 from HBM two tiles ahead (registers), split and written as MFMA-shaped images one tile ahead
//           (separate Q / P images, V double-buffered), while the current tile's P is being published
//   GEMM1   S^T[key, row] = Kp Q^T for key block w and both 32-row blocks, 3 MFMAs per 16-deep k-step, the two blocks'
//           accumulation chains interleaved (a dependent 32x32 MFMA waits for its predecessor)
//   softmax every lane holds 16 keys of ONE row (C layout of the swapped product) -> block-local max / exp / sum, one
//           (max, sum) pair per wave and row through LDS, combined exactly (as the key-chunked launches of the bf16 kernel)
//   GEMM2   O[key, col] += P^T V over the tile's 64 rows: 28 output tiles of 32 x 32 spread over the 8 waves, both operands
//           by hardware transpose-read (ds_read_b64_tr_b16) out of row-major images, 3 MFMAs per 16-row k-step
// Three workgroup barriers per 64 rows (four per 32 before).  Accumulators stay in registers until the head changes; partial
// tiles are written in fragment order and summed in ascending workgroup order by a second kernel (no float atomics).
// LDS at dk = 128, 224 keys: Q hi+lo 32 KiB | P hi+lo 56 KiB | V 2 x (hi+lo) 64 KiB | row statistics 4 KiB = 156 KiB.
#include <math.h>

#include <type_traits>

#include "common.h"
#include "philox.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct X3Params {
    const float* q;    // [n, ldq]
    const float* v;    // [n, ldv]
    const float* kp;   // [k, ldkp]
    int64_t n, ldq, ldv, ldkp;
    int k, h;
    float scale;
    float* attn;       // [h, n, attn_ld] (already offset to this launch's first key) or null
    int64_t attn_ld;
    float* lse;        // [h, n] or null
    // key-chunked launches (k above one LDS image): stats [nchunks][h][n] of (max * c, sum) pairs.  MODE 1 writes chunk
    // `chunk`'s pair per row; MODE 2 reads all chunks' pairs instead of combining its own (softmax exact over all keys)
    f32x2* stats;
    int nchunks, chunk;
    float* partial;    // [num_wg * seg_count][tiles][4][64][4]
    int tiles_per_head, tiles_per_wg, total_tiles, seg_count;
    int64_t n_stride;  // rows per head of attn / lse (= n; the packed row count of a varlen launch)
    const int* vl;     // varlen launch: [bags][VL_DESC] descriptors, then the bag of every workgroup (see below)
    int vl_bags;
    float* out_direct; // varlen: output [rows, h * dk] for bags with one workgroup per head (descriptor flag 10): stored straight from the
                       // accumulators, no partial tile and no reduction pass for that bag; null = always partials
    snf::DropoutState drop = {0u, 0u, 0u, 0u, 0u, 1.f};   // DROP instantiations (training, snuffy.py:166-167): O = (P o M)^T V, attn = P
};
// Varlen launch (many bags in one grid, single key chunk): the grid is the concatenation of per-bag grids -- every bag keeps a
// plan of its own (x3_plan with packed = true, a function of its length only), so a bag's result does not depend on what it is
// packed with, bit for bit; against snf_sparse_attn_fwd_x3 only the fp32 summation order of the partial tiles can differ.
// descriptor: wg0, row0, n, out_row0 (first Kp / output row), tiles_per_head, tiles_per_wg, total_tiles, seg_count, part0, num_wg,
// direct (tiles_per_wg == tiles_per_head: workgroup i of the bag is head i, whole)
constexpr int VL_DESC = 12;

constexpr int TROWS = 64;   // query rows per step (two 32-row blocks)
constexpr int p_row_bytes(int nkb) { return 64 * (nkb | 1); }   // odd multiple of 64 B (bank rule of the transpose-read)

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// x (8 fp32) -> hi = bf16(x), lo = bf16(x - hi)
__device__ __forceinline__ void split8(const f32x8 x, u32x4& hi, u32x4& lo) {
#ifdef X3_ABL_NOSPLIT   // timing ablation (tools/x3_ablate.sh): two cheap packs instead of the split, wrong numbers
    const u32x4 a = __builtin_bit_cast(u32x4, f32x4{x[0], x[2], x[4], x[6]}), b = __builtin_bit_cast(u32x4, f32x4{x[1], x[3], x[5], x[7]});
    hi = (a >> 16) | (b & 0xffff0000u);
    lo = (a & 0xffffu) | (b << 16);
    return;
#endif
    const bf16x8 h = __builtin_convertvector(x, bf16x8);
    const f32x8 r = x - __builtin_convertvector(h, f32x8);
    hi = __builtin_bit_cast(u32x4, h);
    lo = __builtin_bit_cast(u32x4, __builtin_convertvector(r, bf16x8));
}
__device__ __forceinline__ f32x8 load8(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p0, const unsigned char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ float xhalf_max(float v) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// MODE 0: one launch covers all keys.  MODE 1: statistics pass of one key chunk (GEMM1 + max / sum, nothing else).
// MODE 2: main pass of one key chunk with the row statistics of ALL chunks taken from P.stats.
//
// Round 3 organisation: wave w keeps the hi / lo fragments of ITS key block in registers for a whole head (2 NKS fragments = 64
// VGPRs at dk = 128) instead of re-reading them from a 112 KiB LDS image every tile.  The LDS that frees holds 64-ROW tiles with
// separate Q, P and (double-buffered) V images, so a tile costs three workgroup barriers instead of four per 32 rows, the next
// tile's rows are split and written while this tile's P is published, and their HBM loads have a whole tile of latency cover.
// DROP (round 5, single key chunk): the Philox keep-mask of csrc/philox.h (the one snf_dropout_mask_f32 writes out, bit for bit) is applied to
// P in registers before its split for GEMM2; the probabilities written to `attn` stay the undropped ones the backward wants.
template <int DK, int NKB, bool AUX, int MODE, bool VL = false, bool DROP = false>
__global__ __launch_bounds__(512, 2) void sparse_attn_x3_kernel(X3Params PA) {
    constexpr int NKS = DK / 16;               // k-steps of GEMM1
    X3Params P = PA;
    int bid = blockIdx.x;
    if constexpr (VL) {
        const int* __restrict__ tb = PA.vl;
        const int* __restrict__ dsc = tb + VL_DESC * tb[VL_DESC * PA.vl_bags + bid];
        const int row0 = dsc[1];
        bid -= dsc[0];
        P.n = dsc[2];
        P.q = PA.q + (int64_t)row0 * PA.ldq;
        P.v = PA.v + (int64_t)row0 * PA.ldv;
        P.kp = PA.kp + (int64_t)dsc[3] * PA.ldkp;
        if (PA.attn) P.attn = PA.attn + (int64_t)row0 * PA.attn_ld;
        if (PA.lse) P.lse = PA.lse + row0;
        P.tiles_per_head = dsc[4], P.tiles_per_wg = dsc[5], P.total_tiles = dsc[6], P.seg_count = dsc[7];
        P.partial = PA.partial + (int64_t)dsc[8] * (NKB * (DK / 32)) * 1024;
        P.out_direct = (PA.out_direct && dsc[10]) ? PA.out_direct + (int64_t)dsc[3] * ((int64_t)PA.h * DK) : nullptr;
    }
    constexpr int NCB = DK / 32;               // 32-wide column blocks of the output
    constexpr int TILES = NKB * NCB;
    constexpr int NT = (TILES + 7) / 8;        // output tiles owned by one wave
    constexpr int RB = TROWS / 32;             // 32-row blocks per tile
    constexpr int RS = p_row_bytes(NKB);       // row pitch of a P image
    constexpr int VRS = 2 * DK, NCH = DK / 8;  // row pitch of a V image, 16-byte chunks per row
    constexpr int Q_BYTES = RB * NKS * 1024, PI_BYTES = TROWS * RS, V_BYTES = TROWS * VRS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* lds_qh = reinterpret_cast<u32x4*>(smem);                        // [RB][NKS][64] B fragments of Q, hi
    u32x4* lds_ql = reinterpret_cast<u32x4*>(smem + Q_BYTES);              // lo
    unsigned char* lds_ph = smem + 2 * Q_BYTES;                            // [64 rows][RS] row-major P, hi
    unsigned char* lds_pl = lds_ph + PI_BYTES;                             // lo
    unsigned char* lds_v = lds_pl + PI_BYTES;                              // [2 buffers][hi | lo][64 rows][VRS], chunk-rotated
    f32x2* lds_st = reinterpret_cast<f32x2*>(lds_v + 4 * V_BYTES);         // [8 waves][64 rows] (max * c, sum)

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hf = lane >> 5;
    const int n32 = (int)P.n;
    const float c_exp = P.scale * 1.44269504088896340736f;

    const int f_begin = bid * P.tiles_per_wg;
    int f_end = f_begin + P.tiles_per_wg;
    if (f_end > P.total_tiles) f_end = P.total_tiles;
    if (f_begin >= f_end) return;
    const int first_head = f_begin / P.tiles_per_head;
    int a = first_head, t = f_begin - first_head * P.tiles_per_head;
    int cur_head = -1;

    // ---- staging of one tile's Q and V rows: piece p of Q = B fragment (rb, kb, lane': row 32 rb + (lane' & 31), 8 k from
    //      16 kb + 8 (lane' >> 5)); piece p of V = 8 columns (chunk p % NCH) of row p / NCH
    constexpr int QPT = RB * NKS * 64 / 512, VPT = TROWS * NCH / 512;     // pieces per thread: 2 + 2 at dk = 128, 1 + 1 at dk = 64
    f32x8 qpre[QPT], vpre[VPT];                // the rows of the tile after next, in flight for a whole tile
    int fa = a, ft = t;                        // fetch cursor
    auto fetch = [&]() __attribute__((always_inline)) {
        // (opaque thread index in the DROP instantiation: its per-thread row terms, hoisted out of the tile loop, were spilled next to the
        // Philox state and re-read behind vmcnt(0) waits that also drained the PREVIOUS row fetch -- the next tile's loads went out one
        // HBM round trip after the other)
        int tid = threadIdx.x;
        if constexpr (DROP) asm volatile("" : "+v"(tid));
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int p = tid + 512 * i;
            const int rb = p / (NKS * 64), kb = (p >> 6) & (NKS - 1), lp = p & 63;
            int row = ft * TROWS + 32 * rb + (lp & 31);
            if (row > n32 - 1) row = n32 - 1;
            qpre[i] = load8(P.q + (int64_t)row * P.ldq + fa * DK + 16 * kb + 8 * (lp >> 5));
        }
        if constexpr (MODE != 1) {
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const int p = tid + 512 * i;
                int row = ft * TROWS + p / NCH;
                if (row > n32 - 1) row = n32 - 1;
                vpre[i] = load8(P.v + (int64_t)row * P.ldv + fa * DK + 8 * (p % NCH));
            }
        }
        if (++ft == P.tiles_per_head) {
            ft = 0;
            ++fa;
        }
    };
    auto vrot = [](int r) __attribute__((always_inline)) -> int { return DK == 128 ? (r & 3) : ((r >> 1) & 1); };
    auto commit = [&](int vbuf) __attribute__((always_inline)) {
        u32x4 hi, lo;
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            split8(qpre[i], hi, lo);
            lds_qh[tid + 512 * i] = hi;
            lds_ql[tid + 512 * i] = lo;
        }
        if constexpr (MODE != 1) {
            unsigned char* vh = lds_v + vbuf * 2 * V_BYTES;
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                split8(vpre[i], hi, lo);
                const int p = tid + 512 * i;
                const int row = p / NCH, ch = p % NCH;
                const int off = row * VRS + 16 * ((ch + 4 * vrot(row)) & (NCH - 1));
                *reinterpret_cast<u32x4*>(vh + off) = hi;
                *reinterpret_cast<u32x4*>(vh + V_BYTES + off) = lo;
            }
        }
    };
    // this wave's key block (w < NKB) as MFMA A fragments, hi and lo, in registers for the whole head
    bf16x8 kph[NKS], kpl[NKS];
    auto load_kp = [&](int a_) __attribute__((always_inline)) {
        if (w < NKB) {
            int key = 32 * w + j;
            const bool pad = key >= P.k;
            if (pad) key = P.k - 1;
            f32x8 raw[NKS];
#pragma unroll
            for (int kb = 0; kb < NKS; ++kb) raw[kb] = load8(P.kp + (int64_t)key * P.ldkp + a_ * DK + 16 * kb + 8 * hf);
#pragma unroll
            for (int kb = 0; kb < NKS; ++kb) {
                u32x4 hi, lo;
                split8(raw[kb], hi, lo);
                if (pad) hi = lo = u32x4{0u, 0u, 0u, 0u};
                kph[kb] = __builtin_bit_cast(bf16x8, hi);
                kpl[kb] = __builtin_bit_cast(bf16x8, lo);
            }
        }
    };

    // ---- GEMM2 addressing (as in sparse_attn_mfma_impl.h): reader lane = group g (16 lanes) x i
    const int rg = lane >> 4, ri = lane & 15;
    const int rr0 = 8 * (rg >> 1) + (ri >> 2), rr1 = rr0 + 4;
    const int rch = 4 * (rg & 1) + (ri & 3);
    const int cb = w & (NCB - 1);                                     // column block of every tile of this wave
    const int kb0 = w / NCB;                                          // key block of tile ti: kb0 + ti * (8 / NCB)
    const int poff0 = rr0 * RS + 8 * (rch ^ ((rr0 >> 1) & 7)) + 64 * kb0;
    const int poff1 = rr1 * RS + 8 * (rch ^ ((rr1 >> 1) & 7)) + 64 * kb0;
    const int vrc = 4 * cb + 2 * (rg & 1) + ((ri & 3) >> 1);
    const int voff0 = rr0 * VRS + 16 * ((vrc + 4 * vrot(rr0)) & (NCH - 1)) + 8 * (ri & 1);
    const int voff1 = rr1 * VRS + 16 * ((vrc + 4 * vrot(rr1)) & (NCH - 1)) + 8 * (ri & 1);
    // P image writer (softmax): row 32 rb + j, 4 keys per 8-byte chunk; chunk (2 c4 + hf) of key block w at position ^ ((j >> 1) & 7)
    int waddr[4];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) waddr[c4] = j * RS + 64 * w + 8 * (((2 * c4) | hf) ^ ((j >> 1) & 7));

    f32x16 acc_o[NT];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[ti][r] = 0.f;
    };
    auto flush = [&](int head) __attribute__((always_inline)) {
        if constexpr (VL) {
            if (P.out_direct) {   // the whole head is in this workgroup: register 4 q4 + i of tile (kb, cb) = O[32 kb + i + 8 q4 + 4 hf, 32 cb + j]
                const int64_t ld = (int64_t)P.h * DK;
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) {
                    const int t_idx = w + 8 * ti;
                    if (t_idx < TILES) {
                        const int kb_ = t_idx / NCB, cb_ = t_idx - kb_ * NCB;
                        float* dcol = P.out_direct + head * DK + 32 * cb_ + j;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = 32 * kb_ + (r & 3) + 8 * (r >> 2) + 4 * hf;
                            if (key < P.k) dcol[(int64_t)key * ld] = acc_o[ti][r];
                        }
                    }
                }
                return;
            }
        }
        const int seg = head - first_head;
        float* dst = P.partial + ((int64_t)bid * P.seg_count + seg) * (int64_t)TILES * 1024;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            const int t_idx = w + 8 * ti;
            if (t_idx < TILES) {
                const int key0 = 32 * (t_idx / NCB);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    if (key0 + 8 * q4 < P.k) {
                        const f32x4 v4 = {acc_o[ti][q4 * 4], acc_o[ti][q4 * 4 + 1], acc_o[ti][q4 * 4 + 2], acc_o[ti][q4 * 4 + 3]};
                        *reinterpret_cast<f32x4*>(dst + ((int64_t)(t_idx * 4 + q4) * 64 + lane) * 4) = v4;
                    }
            }
        }
    };

    zero_acc();
    fetch();                                     // tile f_begin
    commit(0);
    if (f_begin + 1 < f_end) fetch();            // tile f_begin + 1 flies under the first tile
    const bool attn_vec = AUX && (P.attn_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(P.attn) & 15) == 0;
    for (int f = f_begin; f < f_end; ++f) {
        const int vb = (f - f_begin) & 1;
        int an = a, tn = t + 1;
        if (tn == P.tiles_per_head) {
            tn = 0;
            an = a + 1;
        }
        if (a != cur_head) {
            // new head: this wave's GEMM2 of the previous tile is behind it, so its accumulators can go
            if (MODE != 1 && cur_head >= 0) {
                flush(cur_head);
                zero_acc();
            }
            load_kp(a);
            cur_head = a;
        }
        __syncthreads();                         // B1: Q(f) / V(f) images complete, everybody is past GEMM2(f-1): the P images are free

        // ---- GEMM1 (swapped): S^T[key, row] for key block w, both row blocks; lane = (row j, half hf): keys 32 w + (r&3) + 8 (r>>2) + 4 hf
        f32x16 s[RB];
        float mw[RB], lw[RB];
        if (w < NKB) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[rb][r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < NKS; ++kb) {
                bf16x8 qh[RB], ql[RB];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    qh[rb] = __builtin_bit_cast(bf16x8, lds_qh[(rb * NKS + kb) * 64 + lane]);
                    ql[rb] = __builtin_bit_cast(bf16x8, lds_ql[(rb * NKS + kb) * 64 + lane]);
                }
                // the two row blocks' accumulation chains alternate: a dependent 32x32 MFMA waits 16 passes for its predecessor
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) s[rb] = mfma(kpl[kb], qh[rb], s[rb]);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) s[rb] = mfma(kph[kb], ql[rb], s[rb]);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) s[rb] = mfma(kph[kb], qh[rb], s[rb]);
            }
            // block-local softmax statistics (padded keys -> -inf)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                float mx = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * hf;
                    s[rb][r] = key < P.k ? s[rb][r] * c_exp : -INFINITY;
                    mx = fmaxf(mx, s[rb][r]);
                }
                mw[rb] = xhalf_max(mx);
                const float mref = mw[rb] == -INFINITY ? 0.f : mw[rb];   // a block of padding only: all zeros
                float l = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[rb][r] = __builtin_amdgcn_exp2f(s[rb][r] - mref);
                    l += s[rb][r];
                }
                lw[rb] = xhalf_sum(l);
                if (hf == 0) lds_st[w * TROWS + 32 * rb + j] = f32x2{mw[rb], lw[rb]};
            }
        }
        __syncthreads();                         // B2: statistics published, every wave is done with the Q images

        if constexpr (MODE == 1) {
            // statistics pass: this chunk's (max, sum) per row, then on to the next tile
            if (w < RB && hf == 0) {
                const int row = t * TROWS + 32 * w + j;
                if (row < n32) {
                    float m = -INFINITY;
#pragma unroll
                    for (int b = 0; b < NKB; ++b) m = fmaxf(m, lds_st[b * TROWS + 32 * w + j][0]);
                    float l = 0.f;
#pragma unroll
                    for (int b = 0; b < NKB; ++b) {
                        const f32x2 st = lds_st[b * TROWS + 32 * w + j];
                        if (st[0] != -INFINITY) l = fmaf(st[1], __builtin_amdgcn_exp2f(st[0] - m), l);
                    }
                    P.stats[((int64_t)P.chunk * P.h + a) * P.n + row] = f32x2{m, l};
                }
            }
            if (f + 1 < f_end) {
                commit(0);                       // Q images of the next tile
                if (f + 2 < f_end) fetch();
            }
            a = an;
            t = tn;
            continue;                            // B1 of the next tile orders the statistics slots
        }
        // ---- exact combination over the key blocks, normalisation, publish P = hi + lo
        if (w < NKB) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int row = t * TROWS + 32 * rb + j;
                const bool rvalid = row < n32;
                float m = -INFINITY, l = 0.f;
                if constexpr (MODE == 2) {
                    const int64_t so = (int64_t)a * P.n + (rvalid ? row : n32 - 1);
                    for (int c = 0; c < P.nchunks; ++c) m = fmaxf(m, P.stats[(int64_t)c * P.h * P.n + so][0]);
                    for (int c = 0; c < P.nchunks; ++c) {
                        const f32x2 st = P.stats[(int64_t)c * P.h * P.n + so];
                        if (st[0] != -INFINITY) l = fmaf(st[1], __builtin_amdgcn_exp2f(st[0] - m), l);
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < NKB; ++b) m = fmaxf(m, lds_st[b * TROWS + 32 * rb + j][0]);
#pragma unroll
                    for (int b = 0; b < NKB; ++b) {
                        const f32x2 st = lds_st[b * TROWS + 32 * rb + j];
                        l = fmaf(st[1], __builtin_amdgcn_exp2f(st[0] - m), l);
                    }
                }
                const float fscale = rvalid ? __builtin_amdgcn_exp2f(mw[rb] - m) / l : 0.f;
                if constexpr (AUX)
                    if (P.lse && rvalid && hf == 0 && w == 0) P.lse[(int64_t)a * P.n_stride + row] = (m + __log2f(l)) * 0.69314718055994530942f;
                float* arow = nullptr;
                if constexpr (AUX) arow = P.attn ? P.attn + ((int64_t)a * P.n_stride + row) * P.attn_ld + 32 * w + 4 * hf : nullptr;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    f32x4 p4 = {s[rb][4 * c4] * fscale, s[rb][4 * c4 + 1] * fscale, s[rb][4 * c4 + 2] * fscale, s[rb][4 * c4 + 3] * fscale};
                    // P is ROUNDED to fp32 here in every variant: without this the compiler contracts the product into the
                    // subtraction of the split below (fma) in the variants that do not store A, and their O differs in the last bits
                    asm volatile("" : "+v"(p4));
                    if constexpr (AUX) {
                        if (arow && rvalid) {
                            const int key0 = 32 * w + 8 * c4 + 4 * hf;
                            if (attn_vec && key0 + 4 <= P.k) {
                                *reinterpret_cast<f32x4*>(arow + 8 * c4) = p4;
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (key0 + e < P.k) arow[8 * c4 + e] = p4[e];
                            }
                        }
                    }
                    if constexpr (DROP) {
                        const snf::philox_f4 mk = snf::dropout_mask4(P.drop, a, P.n_stride, rvalid ? row : 0, P.k, 32 * w + 8 * c4 + 4 * hf);
                        p4 = f32x4{p4[0] * mk[0], p4[1] * mk[1], p4[2] * mk[2], p4[3] * mk[3]};
                        asm volatile("" : "+v"(p4));
                    }
                    const bf16x2 h01 = __builtin_convertvector(f32x2{p4[0], p4[1]}, bf16x2);
                    const bf16x2 h23 = __builtin_convertvector(f32x2{p4[2], p4[3]}, bf16x2);
                    const f32x2 r01 = f32x2{p4[0], p4[1]} - __builtin_convertvector(h01, f32x2);
                    const f32x2 r23 = f32x2{p4[2], p4[3]} - __builtin_convertvector(h23, f32x2);
                    *reinterpret_cast<u32x2*>(lds_ph + 32 * rb * RS + waddr[c4]) =
                        u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
                    *reinterpret_cast<u32x2*>(lds_pl + 32 * rb * RS + waddr[c4]) =
                        u32x2{__builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf16x2)),
                              __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf16x2))};
                }
            }
        }
        // the next tile's rows (fetched a tile ago): Q images are free since B2, the other V buffer since B1
        if (f + 1 < f_end) {
            commit(vb ^ 1);
            if (f + 2 < f_end) fetch();
        }
        __syncthreads();                         // B3: P images complete

        // ---- GEMM2: O[key, col] += P^T V over the 64 rows of the tile (four 16-row k-steps, 3 MFMAs each)
        const unsigned char* vh_img = lds_v + vb * 2 * V_BYTES;
#pragma unroll
        for (int sk = 0; sk < TROWS / 16; ++sk) {
            const bf16x8 vh = tr_frag(vh_img + voff0 + sk * 16 * VRS, vh_img + voff1 + sk * 16 * VRS);
            const bf16x8 vl = tr_frag(vh_img + V_BYTES + voff0 + sk * 16 * VRS, vh_img + V_BYTES + voff1 + sk * 16 * VRS);
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) {
                if (TILES % 8 == 0 || w + 8 * ti < TILES) {
                    const int off = sk * 16 * RS + ti * (8 / NCB) * 64;
                    const bf16x8 ph = tr_frag(lds_ph + poff0 + off, lds_ph + poff1 + off);
                    const bf16x8 pl = tr_frag(lds_pl + poff0 + off, lds_pl + poff1 + off);
                    acc_o[ti] = mfma(pl, vh, acc_o[ti]);
                    acc_o[ti] = mfma(ph, vl, acc_o[ti]);
                    acc_o[ti] = mfma(ph, vh, acc_o[ti]);
                }
            }
        }
        a = an;
        t = tn;
    }
    if constexpr (MODE != 1) flush(cur_head);
}

// out[key, a*DK + col] = sum over the (workgroup, segment) partials of head a, ascending workgroup order
template <int DK, int NKB>
__global__ __launch_bounds__(64) void x3_reduce_kernel(const float* __restrict__ partial, int num_wg, int seg_count,
                                                        int tiles_per_head, int tiles_per_wg, int k, int h, float* __restrict__ out,
                                                        const int* __restrict__ vl = nullptr, int direct_bags = 0) {
    constexpr int NCB = DK / 32, TILES = NKB * NCB;
    if (vl) {   // varlen: blockIdx.z = bag
        const int* __restrict__ dsc = vl + VL_DESC * blockIdx.z;
        if (dsc[10] && direct_bags) return;   // the main kernel stored this bag's heads itself
        tiles_per_head = dsc[4], tiles_per_wg = dsc[5], seg_count = dsc[7], num_wg = dsc[9];
        partial += (int64_t)dsc[8] * TILES * 1024;
        out += (int64_t)dsc[3] * (h * DK);
    }
    const int a = blockIdx.y;
    const int unit = blockIdx.x;   // (tile, q4): one wave per workgroup, so that the 4 TILES h units spread over all CUs
    const int lane = threadIdx.x;
    const int t_idx = unit >> 2, q4 = unit & 3;
    if (32 * (t_idx / NCB) + 8 * q4 >= k) return;
    const int f_lo = a * tiles_per_head, f_hi = (a + 1) * tiles_per_head - 1;
    const int b_lo = f_lo / tiles_per_wg;
    int b_hi = f_hi / tiles_per_wg;
    if (b_hi > num_wg - 1) b_hi = num_wg - 1;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const int64_t off = ((int64_t)(t_idx * 4 + q4) * 64 + lane) * 4;
    for (int b = b_lo; b <= b_hi; b += 16) {
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (b + u <= b_hi) {
                const int seg = a - ((b + u) * tiles_per_wg) / tiles_per_head;
                v[u] = *reinterpret_cast<const f32x4*>(partial + ((int64_t)(b + u) * seg_count + seg) * (int64_t)TILES * 1024 + off);
            }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    const int kb = t_idx / NCB, cbk = t_idx - kb * NCB;
    const int col = a * DK + 32 * cbk + (lane & 31);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = 32 * kb + i + 8 * q4 + 4 * (lane >> 5);
        if (key < k) out[(int64_t)key * (h * DK) + col] = s[i];
    }
}

struct X3Plan {
    int num_wg, tiles_per_head, tiles_per_wg, total_tiles, seg_count, nkb;
};
bool x3_plan(int64_t n, int k, int h, int dk, X3Plan* pl, bool packed = false) {
    if (!(dk == 64 || dk == 128) || k < 1 || k > (dk == 128 ? 224 : 256) || n < 1) return false;
    const int need = (k + 31) / 32;
    const int opts[] = {2, 4, 7, 8};
    int sel = 0;
    for (int o : opts)
        if (o >= need && !(o == 8 && dk == 128)) {
            sel = o;
            break;
        }
    if (!sel) return false;
    const int64_t tph = (n + TROWS - 1) / TROWS, total = tph * h;
    if (total > 0x7fffffff) return false;
    const int cus = snf::cu_count();
    int64_t num_wg = total < cus ? total : cus;
    int64_t tpw = (total + num_wg - 1) / num_wg;
    // a bag inside a packed (varlen) launch: at least 16 tiles (1024 rows) per workgroup -- see make_plan of the bf16 kernel
    constexpr int64_t SMALL_BAG_TILES = 16;
    if (packed && total <= cus) tpw = tph < SMALL_BAG_TILES ? tph : SMALL_BAG_TILES;
    num_wg = (total + tpw - 1) / tpw;
    pl->num_wg = (int)num_wg;
    pl->tiles_per_head = (int)tph;
    pl->tiles_per_wg = (int)tpw;
    pl->total_tiles = (int)total;
    pl->seg_count = (int)((tpw + tph - 1) / tph + 1);
    pl->nkb = sel;
    return true;
}
size_t x3_workspace(const X3Plan& pl, int dk) { return (size_t)pl.num_wg * pl.seg_count * (size_t)(pl.nkb * (dk / 32)) * 1024 * sizeof(float); }

template <int DK, int NKB, bool AUX, int MODE, bool VL = false, bool DROP = false>
int x3_launch(const X3Params& P, const X3Plan& pl, float* out, hipStream_t s) {
    constexpr int NKS = DK / 16;
    constexpr int q_bytes = (TROWS / 32) * NKS * 1024, p_bytes = TROWS * p_row_bytes(NKB), v_bytes = TROWS * 2 * DK;
    constexpr int lds = 2 * q_bytes + 2 * p_bytes + 4 * v_bytes + 8 * TROWS * 8;   // Q hi|lo, P hi|lo, V 2 x (hi|lo), statistics
    static thread_local unsigned long long attr_set_mask = 0;   // devices (bit = device id) that have the opt-in
    const unsigned long long attr_set_bit = snf::device_bit();
    const bool attr_set = (attr_set_mask & attr_set_bit) != 0;
    auto kern = sparse_attn_x3_kernel<DK, NKB, AUX, MODE, VL, DROP>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            snf::set_error("sparse_attn_x3: cannot reserve %d bytes of LDS", lds);
            (void)hipGetLastError();
            return SNF_ELAUNCH;
        }
        attr_set_mask |= attr_set_bit;
    }
    hipLaunchKernelGGL(kern, dim3(pl.num_wg), dim3(512), lds, s, P);
    int rc = snf::check_launch("sparse_attn_x3_kernel");
    if (rc || MODE == 1) return rc;
    constexpr int TILES = NKB * (DK / 32);
    if (VL && P.out_direct && pl.tiles_per_head == -1) return SNF_OK;   // every bag stored its heads itself (x3_varlen_plan)
    hipLaunchKernelGGL((x3_reduce_kernel<DK, NKB>), dim3(TILES * 4, P.h, VL ? P.vl_bags : 1), dim3(64), 0, s, P.partial, pl.num_wg,
                       pl.seg_count, pl.tiles_per_head, pl.tiles_per_wg, P.k, P.h, out, VL ? P.vl : nullptr,
                       (VL && P.out_direct) ? 1 : 0);
    return snf::check_launch("x3_reduce_kernel");
}
template <int DK>
int x3_dispatch_varlen(const X3Params& P, const X3Plan& pl, float* out, hipStream_t s) {
    const bool aux = P.attn != nullptr || P.lse != nullptr;
#define SNF_X3_VL_CASE(NB) \
    case NB: return aux ? x3_launch<DK, NB, true, 0, true>(P, pl, out, s) : x3_launch<DK, NB, false, 0, true>(P, pl, out, s);
    switch (pl.nkb) {
        SNF_X3_VL_CASE(2)
        SNF_X3_VL_CASE(4)
        SNF_X3_VL_CASE(7)
        case 8:
            if constexpr (DK == 64)
                return aux ? x3_launch<DK, 8, true, 0, true>(P, pl, out, s) : x3_launch<DK, 8, false, 0, true>(P, pl, out, s);
            break;
        default: break;
    }
#undef SNF_X3_VL_CASE
    snf::set_error("sparse_attn_x3 (varlen): key-block count %d not built", pl.nkb);
    return SNF_EUNSUPPORTED;
}
struct X3VarlenPlan {
    int64_t total_wg, partial_slots;
    int nkb;
    bool all_direct;   // every bag has one workgroup per head: no reduction pass
};
bool x3_varlen_plan(const int64_t* offsets, int bags, int k, int h, int dk, X3VarlenPlan* vp, int32_t* table, size_t table_ints) {
    vp->total_wg = 0, vp->partial_slots = 0, vp->nkb = 0, vp->all_direct = true;
    if (bags < 1 || k > (dk == 128 ? 224 : 256)) return false;   // single key chunk only
    for (int b = 0; b < bags; ++b) {
        const int64_t n = offsets[b + 1] - offsets[b];
        X3Plan pl;
        if (n < 1 || offsets[b] > 0x7fffffffll || !x3_plan(n, k, h, dk, &pl, true)) return false;
        const bool direct = pl.tiles_per_wg == pl.tiles_per_head;
        vp->all_direct = vp->all_direct && direct;
        if (table) {
            if ((size_t)(VL_DESC * bags) + (size_t)(vp->total_wg + pl.num_wg) > table_ints) return false;
            int32_t* d = table + (size_t)VL_DESC * b;
            d[0] = (int32_t)vp->total_wg, d[1] = (int32_t)offsets[b], d[2] = (int32_t)n, d[3] = b * k;
            d[4] = pl.tiles_per_head, d[5] = pl.tiles_per_wg, d[6] = pl.total_tiles, d[7] = pl.seg_count;
            d[8] = (int32_t)vp->partial_slots, d[9] = pl.num_wg, d[10] = direct ? 1 : 0, d[11] = 0;
            for (int i = 0; i < pl.num_wg; ++i) table[(size_t)VL_DESC * bags + vp->total_wg + i] = b;
        }
        vp->total_wg += pl.num_wg;
        vp->partial_slots += (int64_t)pl.num_wg * pl.seg_count;
        vp->nkb = pl.nkb;
        if (vp->total_wg > 0x3fffffff || vp->partial_slots > 0x3fffffff) return false;
    }
    return true;
}
template <int DK, int NB>
int x3_modes(const X3Params& P, const X3Plan& pl, float* out, hipStream_t s, int mode) {
    const bool aux = P.attn != nullptr || P.lse != nullptr;
    if (mode == 1) return x3_launch<DK, NB, false, 1>(P, pl, out, s);
    if (mode == 2) return aux ? x3_launch<DK, NB, true, 2>(P, pl, out, s) : x3_launch<DK, NB, false, 2>(P, pl, out, s);
    return aux ? x3_launch<DK, NB, true, 0>(P, pl, out, s) : x3_launch<DK, NB, false, 0>(P, pl, out, s);
}
template <int DK>
int x3_dispatch_dropout(const X3Params& P, const X3Plan& pl, float* out, hipStream_t s) {
    switch (pl.nkb) {
        case 2: return x3_launch<DK, 2, true, 0, false, true>(P, pl, out, s);
        case 4: return x3_launch<DK, 4, true, 0, false, true>(P, pl, out, s);
        case 7: return x3_launch<DK, 7, true, 0, false, true>(P, pl, out, s);
        case 8:
            if constexpr (DK == 64) return x3_launch<DK, 8, true, 0, false, true>(P, pl, out, s);
            break;
        default: break;
    }
    snf::set_error("sparse_attn_x3 (dropout): key-block count %d not built", pl.nkb);
    return SNF_EUNSUPPORTED;
}
template <int DK>
int x3_dispatch(const X3Params& P, const X3Plan& pl, float* out, hipStream_t s, int mode) {
    switch (pl.nkb) {
        case 2: return x3_modes<DK, 2>(P, pl, out, s, mode);
        case 4: return x3_modes<DK, 4>(P, pl, out, s, mode);
        case 7: return x3_modes<DK, 7>(P, pl, out, s, mode);
        case 8:
            if constexpr (DK == 64) return x3_modes<DK, 8>(P, pl, out, s, mode);
            break;
        default: break;
    }
    snf::set_error("sparse_attn_x3: key-block count %d not built", pl.nkb);
    return SNF_EUNSUPPORTED;
}

// keys per launch: one LDS image holds kmax keys; more keys run as up to 8 chunks of equal size (a multiple of 4)
struct X3Chunks {
    int count, size;
};
bool x3_chunks(int k, int dk, X3Chunks* c) {
    const int kmax = dk == 128 ? 224 : 256;
    if (!(dk == 64 || dk == 128) || k < 1 || k > 8 * kmax) return false;
    c->count = (k + kmax - 1) / kmax;
    c->size = c->count == 1 ? k : ((k + c->count - 1) / c->count + 3) & ~3;
    return c->size <= kmax;
}

}  // namespace

extern "C" {

size_t snf_sparse_attn_fwd_x3_workspace_bytes(int64_t n, int k, int h, int dk) {
    X3Plan pl;
    X3Chunks ch;
    if (h < 1 || !x3_chunks(k, dk, &ch) || !x3_plan(n, ch.size, h, dk, &pl)) return 0;
    const size_t stats = ch.count > 1 ? (size_t)ch.count * h * n * sizeof(f32x2) : 0;
    return x3_workspace(pl, dk) + stats;
}

static int x3_forward(const float* q, int64_t ldq, const float* v, int64_t ldv, const float* kp, int64_t n, int k, int h, int dk,
                      float scale, float* out, float* attn, float* lse, void* workspace, size_t workspace_bytes, snf_stream_t stream,
                      const snf::DropoutState* drop);

int snf_sparse_attn_fwd_x3(const float* q, int64_t ldq, const float* v, int64_t ldv, const float* kp, int64_t n, int k, int h,
                           int dk, float scale, float* out, float* attn, float* lse, void* workspace, size_t workspace_bytes,
                           snf_stream_t stream) {
    return x3_forward(q, ldq, v, ldv, kp, n, k, h, dk, scale, out, attn, lse, workspace, workspace_bytes, stream, nullptr);
}

int snf_sparse_attn_fwd_x3_dropout(const float* q, int64_t ldq, const float* v, int64_t ldv, const float* kp, int64_t n, int k, int h,
                                   int dk, float scale, float dropout_p, uint64_t seed, uint64_t offset, float* out, float* attn,
                                   float* lse, void* workspace, size_t workspace_bytes, snf_stream_t stream) {
    SNF_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "snf_sparse_attn_fwd_x3_dropout: dropout_p=%g outside [0, 1)", (double)dropout_p);
    const snf::DropoutState st = snf::make_dropout(dropout_p, seed, offset);
    return x3_forward(q, ldq, v, ldv, kp, n, k, h, dk, scale, out, attn, lse, workspace, workspace_bytes, stream, st.thresh ? &st : nullptr);
}

static int x3_forward(const float* q, int64_t ldq, const float* v, int64_t ldv, const float* kp, int64_t n, int k, int h, int dk,
                      float scale, float* out, float* attn, float* lse, void* workspace, size_t workspace_bytes, snf_stream_t stream,
                      const snf::DropoutState* drop) {
    SNF_REQUIRE(q && v && kp && out, "snf_sparse_attn_fwd_x3: null pointer");
    SNF_REQUIRE(n >= 1 && k >= 1 && h >= 1, "snf_sparse_attn_fwd_x3: bad shape");
    X3Plan pl;
    X3Chunks ch;
    if (!x3_chunks(k, dk, &ch) || !x3_plan(n, ch.size, h, dk, &pl)) {
        snf::set_error("snf_sparse_attn_fwd_x3: unsupported shape k=%d dk=%d (dk in {64, 128}, k <= 8 x 256 / 8 x 224)", k, dk);
        return SNF_EUNSUPPORTED;
    }
    const int64_t d = (int64_t)h * dk;
    SNF_REQUIRE(ldq >= d && ldv >= d && (ldq % 4) == 0 && (ldv % 4) == 0, "snf_sparse_attn_fwd_x3: ldq=%lld / ldv=%lld must be >= "
                "h*dk and keep rows 16-byte aligned", (long long)ldq, (long long)ldv);
    SNF_REQUIRE(((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(kp)) & 15) == 0,
                "snf_sparse_attn_fwd_x3: q / v / kp must be 16-byte aligned");
    const size_t part_bytes = x3_workspace(pl, dk);
    const size_t need = part_bytes + (ch.count > 1 ? (size_t)ch.count * h * n * sizeof(f32x2) : 0);
    if (!workspace || workspace_bytes < need) {
        snf::set_error("snf_sparse_attn_fwd_x3: workspace %zu < %zu", workspace_bytes, need);
        return SNF_EWORKSPACE;
    }
    X3Params P;
    P.q = q, P.v = v, P.kp = kp;
    P.n = n, P.ldq = ldq, P.ldv = ldv, P.ldkp = d;
    P.k = k, P.h = h, P.scale = scale;
    P.attn = attn, P.attn_ld = k, P.lse = lse;
    P.partial = reinterpret_cast<float*>(workspace);
    P.stats = reinterpret_cast<f32x2*>(reinterpret_cast<unsigned char*>(workspace) + part_bytes);
    P.nchunks = ch.count, P.chunk = 0;
    P.tiles_per_head = pl.tiles_per_head, P.tiles_per_wg = pl.tiles_per_wg, P.total_tiles = pl.total_tiles;
    P.seg_count = pl.seg_count;
    P.n_stride = n, P.vl = nullptr, P.vl_bags = 0, P.out_direct = nullptr;
    hipStream_t s = snf::as_stream(stream);
    if (drop) {
        if (ch.count != 1) {
            snf::set_error("snf_sparse_attn_fwd_x3_dropout: k=%d needs key chunks; the in-kernel mask covers one launch (k <= %d)", k,
                           dk == 128 ? 224 : 256);
            return SNF_EUNSUPPORTED;
        }
        P.drop = *drop;
        return dk == 128 ? x3_dispatch_dropout<128>(P, pl, out, s) : x3_dispatch_dropout<64>(P, pl, out, s);
    }
    if (ch.count == 1) return dk == 128 ? x3_dispatch<128>(P, pl, out, s, 0) : x3_dispatch<64>(P, pl, out, s, 0);
    // key chunks: statistics of every chunk first, then the chunks' main passes with the softmax exact over all keys
    for (int pass = 1; pass <= 2; ++pass)
        for (int c = 0; c < ch.count; ++c) {
            const int k0 = c * ch.size, kc = k - k0 < ch.size ? k - k0 : ch.size;
            X3Params C = P;
            C.kp = kp + (int64_t)k0 * d;
            C.k = kc, C.chunk = c;
            C.attn = (pass == 2 && attn) ? attn + k0 : nullptr;
            C.lse = (pass == 2 && c == 0) ? lse : nullptr;
            int rc = dk == 128 ? x3_dispatch<128>(C, pl, out + (int64_t)k0 * d, s, pass)
                               : x3_dispatch<64>(C, pl, out + (int64_t)k0 * d, s, pass);
            if (rc) return rc;
        }
    return SNF_OK;
}

// ---- varlen (see snf_sparse_attn_varlen_plan in sparse_attn_mfma.hip for the protocol; this is the fp32-class kernel's plan) ----
int snf_sparse_attn_x3_varlen_plan(const int64_t* offsets, int bags, int k, int h, int dk, int32_t* table, size_t table_ints,
                                   size_t* table_ints_needed, size_t* workspace_bytes) {
    SNF_REQUIRE(offsets && bags >= 1 && k >= 1 && h >= 1, "snf_sparse_attn_x3_varlen_plan: bad arguments");
    X3VarlenPlan vp;
    if (!x3_varlen_plan(offsets, bags, k, h, dk, &vp, nullptr, 0)) {
        snf::set_error("snf_sparse_attn_x3_varlen_plan: unsupported shape (bags=%d k=%d dk=%d: dk in {64, 128}, k <= %d, non-empty "
                       "bags)", bags, k, dk, dk == 128 ? 224 : 256);
        return SNF_EUNSUPPORTED;
    }
    const size_t need = (size_t)VL_DESC * bags + (size_t)vp.total_wg;
    if (table_ints_needed) *table_ints_needed = need;
    if (workspace_bytes) *workspace_bytes = (size_t)vp.partial_slots * (size_t)(vp.nkb * (dk / 32)) * 1024 * sizeof(float);
    if (table) {
        SNF_REQUIRE(table_ints >= need, "snf_sparse_attn_x3_varlen_plan: table %zu < %zu ints", table_ints, need);
        x3_varlen_plan(offsets, bags, k, h, dk, &vp, table, table_ints);
    }
    return SNF_OK;
}

// q, v [T, ld] f32 packed rows, kp [bags * k, h * dk] f32, out [bags * k, h * dk], attn [h, T, k] / lse [h, T] or null
int snf_sparse_attn_fwd_x3_varlen(const float* q, int64_t ldq, const float* v, int64_t ldv, const float* kp, const int64_t* offsets,
                                  int bags, int k, int h, int dk, float scale, float* out, float* attn, float* lse,
                                  const int32_t* table_dev, void* workspace, size_t workspace_bytes, snf_stream_t stream) {
    SNF_REQUIRE(q && v && kp && out && offsets && table_dev, "snf_sparse_attn_fwd_x3_varlen: null pointer");
    X3VarlenPlan vp;
    if (!x3_varlen_plan(offsets, bags, k, h, dk, &vp, nullptr, 0)) {
        snf::set_error("snf_sparse_attn_fwd_x3_varlen: unsupported shape (bags=%d k=%d dk=%d)", bags, k, dk);
        return SNF_EUNSUPPORTED;
    }
    const int64_t d = (int64_t)h * dk, total = offsets[bags];
    SNF_REQUIRE(ldq >= d && ldv >= d && (ldq % 4) == 0 && (ldv % 4) == 0, "snf_sparse_attn_fwd_x3_varlen: ldq=%lld / ldv=%lld must be "
                ">= h*dk and keep rows 16-byte aligned", (long long)ldq, (long long)ldv);
    SNF_REQUIRE(((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(kp)) & 15) == 0,
                "snf_sparse_attn_fwd_x3_varlen: q / v / kp must be 16-byte aligned");
    const size_t need = (size_t)vp.partial_slots * (size_t)(vp.nkb * (dk / 32)) * 1024 * sizeof(float);
    if (!workspace || workspace_bytes < need) {
        snf::set_error("snf_sparse_attn_fwd_x3_varlen: workspace %zu < %zu", workspace_bytes, need);
        return SNF_EWORKSPACE;
    }
    X3Params P;
    P.q = q, P.v = v, P.kp = kp;
    P.n = total, P.ldq = ldq, P.ldv = ldv, P.ldkp = d;
    P.k = k, P.h = h, P.scale = scale;
    P.attn = attn, P.attn_ld = k, P.lse = lse;
    P.partial = reinterpret_cast<float*>(workspace);
    P.stats = nullptr, P.nchunks = 1, P.chunk = 0;
    P.tiles_per_head = P.tiles_per_wg = P.total_tiles = P.seg_count = 0;   // per bag, from the table
    P.n_stride = total, P.vl = table_dev, P.vl_bags = bags, P.out_direct = out;
    X3Plan pl;
    pl.num_wg = (int)vp.total_wg, pl.nkb = vp.nkb;
    pl.tiles_per_head = pl.tiles_per_wg = pl.total_tiles = pl.seg_count = 0;
    if (vp.all_direct) pl.tiles_per_head = -1;   // x3_launch: no reduction pass
    hipStream_t s = snf::as_stream(stream);
    return dk == 128 ? x3_dispatch_varlen<128>(P, pl, out, s) : x3_dispatch_varlen<64>(P, pl, out, s);
}

}  // extern "C"
