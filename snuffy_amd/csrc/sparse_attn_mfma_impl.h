#pragma once
// K7 (fast form): Snuffy's sparse attention on the CDNA4 matrix cores.
// (implementation header: compiled once per head width by sparse_attn_mfma.hip (dk = 128 + the C entry points) and
// sparse_attn_mfma_dk64.hip, so the two halves of the ~80 kernel variants build in parallel)
//
//   per head a:   P_a = softmax_j(Q_a Kp_a^T * scale)  [n, k]      O_a = P_a^T V_a  [k, dk]        (snuffy.py:160-168)
//
// All n patches are queries, only the k selected rows are keys, and the probability matrix is used TRANSPOSED to pool
// the values of all n patches into k output rows.  The contraction of the second product runs over the QUERY axis, the
// opposite of flash attention:
//
//   GEMM1  S^T[key, q] = Kp Q^T   v_mfma_f32_32x32x16_bf16, A = Kp fragment (bf16 image in LDS), B = Q fragment (HBM -> regs).
//          Swapped on purpose: the C layout gives every lane ONE query row (16 keys per block in registers, the other 16
//          in lane ^ 32), so the softmax over keys is register-local plus one v_permlane32_swap, fp32.
//   publish  P (bf16) goes to LDS ROW-major, 4 keys per ds_write_b64, together with the wave's 32 rows of V (row-major from
//          HBM, parked in registers during the step).  Chunk rotation by row keeps stores and reads conflict-free.
//   GEMM2  O[key, col] += P^T V   needs both operands with 8 consecutive QUERY rows in a lane's registers, i.e. transposed
//          with respect to how they were written: ds_read_b64_tr_b16 (hardware transpose-read) delivers exactly that.
//
// Workgroup = 8 waves in two roles = 128 query rows per step of one head: the softmax waves 0..3 run GEMM1 + softmax for
// rows 32w..32w+31 against all keys and publish P, the pooling waves 4..7 (one on each softmax wave's SIMD) publish V and
// accumulate their share of the [k, dk] output tiles over all 128 rows (details at the kernel).  One workgroup per CU
// walks a contiguous range of (head, row-tile) work items; the accumulators stay in registers until the head changes;
// partial tiles are written in fragment order and summed in a fixed order by a second kernel (no float atomics ->
// bit-reproducible).
//
// HBM traffic per launch (algorithmic): read Q and V once (2*n*d*elt), Kp (bf16) once per workgroup (L2), write partials.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "philox.h"

namespace snf {
size_t generic_attn_workspace_bytes(int64_t n, int k, int h, int dk);
}

// launch descriptors shared by the two translation units (external linkage: they cross the TU boundary by reference)
namespace snf_attn {
struct AttnParams {
    const void* q;     // [n, ldq]   row-major, head a = columns a*dk .. (a+1)*dk
    const void* v;     // [n, ldv]   row-major, same column layout
    const unsigned short* kp;   // [k, ldkp] bf16 (the entry point converts an f32 Kp into the workspace first)
    int64_t n, ldq, ldv, ldkp;
    int k, h;
    float scale;
    float* attn;     // [h, n, attn_ld] or null (already offset to this key chunk's first column)
    int64_t attn_ld; // row pitch of attn (= total number of keys)
    // key-chunked launches (more keys than one LDS image holds): per-chunk row statistics [n_chunks][h][n][2] =
    // (row max * scale * log2 e, sum exp) written by sparse_attn_stats_kernel; null = single chunk, statistics computed here
    const float* stats;
    float* stats_out;
    int n_chunks;
    float* lse;      // [h, n] or null
    float* partial;  // [num_wg * seg_count][tiles][16][64]
    int tiles_per_head, tiles_per_wg, total_tiles, seg_count;
    unsigned long long* trace;  // debug: s_memtime stamps of workgroup trace_wg (tools/attn_trace.py), normally null
    int trace_wg;
    // attention dropout (training, snuffy.py:166-167): the keep-mask is regenerated from (seed, offset, a, row, key) in
    // registers (philox.h) and applied to the normalised probabilities before they are published -- O and the returned A
    // are those of the dropped P, as in the reference.  thresh == 0: off.  Only the AUX variants look at it.
    snf::DropoutState drop;
    int64_t n_stride;  // rows per head of attn / lse (= n for one bag; the packed row count of a varlen launch)
    // varlen launch (many bags in one grid, VL kernels): table of VL_DESC ints per bag, then the bag of every workgroup
    const int* vl;
    int vl_bags;
    // varlen: output [rows, h * dk] for bags whose heads fit ONE workgroup each (descriptor flag 10: tiles_per_wg == tiles_per_head):
    // such a workgroup holds the head's complete [k, dk] result and stores it straight to its place -- no partial tile, no
    // reduction pass for that bag (a 64 x 1000-patch batch spent 29 us copying single partial tiles).  Null: always partials.
    float* out_direct;
};
// One bag of a varlen launch.  The grid is the concatenation of per-bag grids: every bag keeps a plan of its own (make_plan with
// packed = true, a function of the bag's length only) and workgroup wg0 + i does what workgroup i of a launch of that bag alone
// would do (same tiles, same partial tiles, same summation order in the reduction) -- a bag's result does not depend on what it
// is packed with, bit for bit.  Against the single-bag entry points (latency plan: one tile per workgroup for small bags) only
// the fp32 summation order of the partial tiles can differ.
constexpr int VL_DESC = 12;   // wg0, row0, n, out_row0 (= first Kp / output row), tiles_per_head, tiles_per_wg, total_tiles,
                              // seg_count, part0 (first partial slot), num_wg, direct (one workgroup per head), 0
struct Plan {
    int num_wg, tiles_per_head, tiles_per_wg, total_tiles, seg_count, nkb;
};
}  // namespace snf_attn

namespace {
using snf_attn::AttnParams;
using snf_attn::Plan;
using snf_attn::VL_DESC;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int TILE_ROWS = 128;  // query rows per workgroup step (4 waves x 32)
// row pitch (bytes) of the bf16 P image in LDS: 64 bytes per key block, an ODD number of 64-byte units (bank rule below)
constexpr int p_row_bytes(int nkb) { return 64 * (nkb | 1); }



// compile-time loop: every index into the register-resident fragment arrays must be a constant, or the arrays go to
// scratch (a "#pragma unroll" is only a hint and gives up on the larger variants)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ bf16x8 zero_frag() {
    u32x4 z = {0u, 0u, 0u, 0u};
    return __builtin_bit_cast(bf16x8, z);
}

// 8 consecutive elements -> bf16x8 (f32 source converted with v_cvt_pk_bf16_f32, round-to-nearest-even)
__device__ __forceinline__ bf16x8 load_frag(const float* p) {
    f32x4 lo = *reinterpret_cast<const f32x4*>(p);
    f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
    f32x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_convertvector(v, bf16x8);
}
__device__ __forceinline__ bf16x8 load_frag(const unsigned short* p) {
    u32x4 v = *reinterpret_cast<const u32x4*>(p);
    return __builtin_bit_cast(bf16x8, v);
}
// S = Q Kp^T accumulates in ARCHITECTURAL VGPRs: the softmax reads S with the VALU, and the compiler's own choice for a
// builtin MFMA result is the AGPR half of the file, which costs a v_accvgpr_read per element and -- with the O
// accumulators already filling the AGPRs -- a storm of v_accvgpr_mov live-range splits (640 per tile, measured).
// Inline asm with a tied "+v" accumulator keeps S where the VALU wants it.  hipcc does not model the instruction inside
// an asm statement: the s_nop covers VALU-write -> MFMA-operand wait states, the caller parks before the first VALU read.
// NOP = the two wait states a 32x32 MFMA needs when (a) the previous instruction was an MFMA on the SAME accumulator
// (back-to-back dependent issue) or (b) an operand may have been written by the VALU just before (f32 -> bf16
// conversion, accumulator init).  "s_nop 1" is a full 8-cycle issue slot for a single wave, so it is only emitted then.
template <bool NOP>
__device__ __forceinline__ void mfma_vgpr(f32x16& acc, bf16x8 a, bf16x8 b) {
    const u32x4 au = __builtin_bit_cast(u32x4, a), bu = __builtin_bit_cast(u32x4, b);
    if constexpr (NOP)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(au), "v"(bu));
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(au), "v"(bu));
}

// first MFMA of an accumulation chain: C is the inline constant 0, D is write-only (early clobber: never overlaps A / B)
__device__ __forceinline__ void mfma_vgpr_zero_c(f32x16& acc, bf16x8 a, bf16x8 b) {
    const u32x4 au = __builtin_bit_cast(u32x4, a), bu = __builtin_bit_cast(u32x4, b);
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(au), "v"(bu));
}
__device__ __forceinline__ void park_after_mfma(f32x16& x) { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" : "+v"(x)); }
__device__ __forceinline__ void pin_vgpr(f32x16& x) { asm volatile("" : "+v"(x)); }

// One GEMM2 A operand (P^T fragment: key on the lane, 8 consecutive query rows in registers) out of the row-major P image:
// two hardware transpose-reads.  ds_read_b64_tr_b16 semantics (probed on gfx950, tools/probes/tr16_probe.hip): every lane
// supplies the address of its own 8-byte chunk; inside each group of 16 lanes the 16 chunks form a [4 rows][16 columns]
// bf16 matrix (lane i = row i>>2, columns 4(i&3)..+3) and lane i receives column i.
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__device__ __forceinline__ bf16x8 lds_read_p_frag(const unsigned char* p0, const unsigned char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// all-reduce across the two half-waves (lane l <-> lane l^32) on the VALU: v_permlane32_swap(x, x) = {x.lo, x.lo}, {x.hi, x.hi}
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

// AUX = the caller asked for the attention matrix and/or the row log-sum-exp (extra stores after the softmax).
// EXT = key-chunked launch: the row statistics come from sparse_attn_stats_kernel (all chunks), not from this chunk's scores.
//
// Workgroup = 8 waves in two ROLES, one wave of each role per SIMD (waves w and w + 4 share a SIMD):
//   softmax waves 0..3   Q loads, GEMM1 and the fp32 softmax of their 32 query rows, publish P (bf16) in LDS
//   pooling waves 4..7   V loads and the V image, GEMM2 for their share of the [k, dk] output tiles, the accumulators,
//                        the flush of the partial tiles
// Why two roles: a single wave issues at most one instruction every 4-5 cycles (measured, tools/probes), and the softmax
// is ~500 VALU instructions per 32 x 224 tile while GEMM2 is 56 MFMAs + 128 transpose-reads -- in one wave they only
// interleave, on two waves of the same SIMD the VALU stream and the MFMA / LDS stream issue side by side.  The register
// file splits accordingly: S (112 VGPRs) + Q fragments live only in the softmax waves, the output accumulators
// (112 VGPRs) + V rows only in the pooling waves, so both fit the 256 registers of a two-waves-per-SIMD launch.
//
// One step (tile f = 128 query rows of one head), barriers B (images complete) and A (images free) shared by all 8 waves:
//   softmax wave:  GEMM1(f)  B(f-1)  Q(f+1) loads, softmax(f)                A(f)  write P(f)
//   pooling wave:  V(f) loads B(f-1) GEMM2(f-1) out of the P / V images      A(f)  write V(f)
// B sits AFTER GEMM1 on purpose: the matrix pipe of the SIMD then runs GEMM1 and GEMM2 one after the other instead of
// both at half speed, and GEMM2 overlaps the VALU-only softmax.
// TAILP: register pairs of the LAST key block that can hold a real key (8 = all).  K = 200 (the reference's default Lambda)
// fills 8 keys of its 7th block -- pairs 0, 1 of every lane; with TAILP = 2 the softmax passes skip the other six pairs
// (12 of the lane's 112 score registers: they hold -inf, would give exp = 0, and their P columns feed output rows that are
// never stored), i.e. ~11 % of the softmax wave's VALU work, decided at compile time (the run-time form of the same skip
// measured slower in round 1: its wave-uniform branches cost more than they saved).
template <int DK, int NKB, typename QT, bool AUX, bool EXT, int TAILP = 8, bool VL = false>
__global__ __launch_bounds__(512) void sparse_attn_mfma_kernel(AttnParams PA) {
    constexpr int NKS = DK / 16;             // k-steps of GEMM1
    AttnParams P = PA;
    int bid = blockIdx.x;
    if constexpr (VL) {
        // varlen: become workgroup (bid - wg0) of this bag's own launch (descriptor: scalar loads, once per workgroup)
        const int* __restrict__ tb = PA.vl;
        const int* __restrict__ dsc = tb + VL_DESC * tb[VL_DESC * PA.vl_bags + bid];
        const int row0 = dsc[1];
        bid -= dsc[0];
        P.n = dsc[2];
        P.q = reinterpret_cast<const QT*>(PA.q) + (int64_t)row0 * PA.ldq;
        P.v = reinterpret_cast<const QT*>(PA.v) + (int64_t)row0 * PA.ldv;
        P.kp = PA.kp + (int64_t)dsc[3] * PA.ldkp;
        if (PA.attn) P.attn = PA.attn + (int64_t)row0 * PA.attn_ld;
        if (PA.lse) P.lse = PA.lse + row0;
        P.tiles_per_head = dsc[4];
        P.tiles_per_wg = dsc[5];
        P.total_tiles = dsc[6];
        P.seg_count = dsc[7];
        P.partial = PA.partial + (int64_t)dsc[8] * (NKB * (DK / 32)) * 1024;
        P.out_direct = (PA.out_direct && dsc[10]) ? PA.out_direct + (int64_t)dsc[3] * ((int64_t)PA.h * DK) : nullptr;
    }
    constexpr int NCB = DK / 32;             // 32-wide column blocks of the output
    constexpr int NT = (NKB * NCB + 3) / 4;  // output tiles owned by one pooling wave
    constexpr int M2 = 8 * NT;               // GEMM2 MFMAs per tile and wave
    constexpr int RS = p_row_bytes(NKB);     // row pitch of the P image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* lds_kp = reinterpret_cast<u32x4*>(smem);                    // [NKB][NKS][64] MFMA A fragments of Kp
    unsigned char* lds_p = smem + NKB * NKS * 1024;                    // [128 rows][RS] bf16 probabilities, swizzled
    unsigned char* lds_v = lds_p + TILE_ROWS * RS;                     // [128 rows][DK] bf16 values of the pending tile

    // Every kernel argument is requested in the FIRST scalar-load batch: the compiler otherwise fetches k / partial / scale
    // lazily, and each extra batch is one more cold round trip to the kernarg segment before the first HBM load can go out.
    asm volatile("" ::"s"(P.q), "s"(P.v), "s"(P.kp), "s"(P.n), "s"(P.ldq), "s"(P.ldv), "s"(P.k), "s"(P.scale),
                 "s"(P.partial), "s"(P.tiles_per_head), "s"(P.tiles_per_wg), "s"(P.total_tiles), "s"(P.seg_count),
                 "s"(P.trace), "s"(P.stats), "s"(P.n_chunks), "s"(P.attn_ld));
    const int lane = threadIdx.x & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = w8 & 3;   // position inside the role = SIMD
    const int j = lane & 31, hf = lane >> 5;
    const int n32 = (int)P.n;
    int trace_it = 0;
    // debug trace (tools/attn_trace.py): phases 0..4 are stamped by the softmax waves (step start, GEMM1 done, B passed,
    // softmax done, A passed), 5..7 by the pooling waves (B passed, GEMM2 done, A passed)
    auto stamp = [&](int phase) __attribute__((always_inline)) {
        if (P.trace && bid == P.trace_wg && lane == 0)
            P.trace[(trace_it * 8 + phase) * 4 + w] = __builtin_amdgcn_s_memtime();
    };
    auto stamp_abs = [&](int slot) __attribute__((always_inline)) {  // slots 56..63 of the trace: kernel milestones
        if (P.trace && bid == P.trace_wg && lane == 0) P.trace[(slot * 8) * 4 + w] = __builtin_amdgcn_s_memtime();
    };

    const int f_begin = bid * P.tiles_per_wg;
    int f_end = f_begin + P.tiles_per_wg;
    if (f_end > P.total_tiles) f_end = P.total_tiles;
    const int first_head = f_begin / P.tiles_per_head;
    int a = first_head, t = f_begin - first_head * P.tiles_per_head;   // (head, row tile) of the current work item
    int cur_head = f_begin < f_end ? first_head : -1;                  // head whose Kp image / accumulators are live
    bool published = false;   // P / V images written, their closing barrier B not passed yet (same value in both roles)

    // All loads are unconditional and in bounds: Q and V rows are clamped to n-1; the probabilities of those rows are
    // zeroed, so whatever (finite) row they re-read contributes nothing.  Element offsets are 32-bit (make_plan checks
    // n * ld < 2^31) and row * ld is one v_mad_u32_u24: the pointer arithmetic of a tile is ~3 VALU per load, not ~11.

    // Kp_a -> LDS as bf16 MFMA fragments: wave wi of NW participating waves owns fragments wi, wi + NW, ...  Phase 1 issues every
    // global load of the wave back to back (one latency, not one per fragment; padded keys re-read the last row), phase 2
    // stores (padded keys zeroed; their probabilities are forced to 0 through the -inf accumulator init anyway).
    constexpr int NF = (NKB * NKS + 3) / 4;   // fragments per wave with 4 waves (sizes the staging registers)
    auto kp_issue = [&](auto nw_t, int wi, int a_, u32x4(&raw)[NF]) __attribute__((always_inline)) {
        constexpr int NW = decltype(nw_t)::value;
        // opaque copy of the lane's key index: the fragment addresses are loop-invariant, and hoisting NF 64-bit
        // pointers out of the tile loop would hold 2 NF registers across the whole softmax for a once-per-head event
        int jo = j;
        asm volatile("" : "+v"(jo));
        static_for<0, (NKB * NKS + NW - 1) / NW>([&](auto i) __attribute__((always_inline)) {
            int fr = wi + NW * i;
            if (fr > NKB * NKS - 1) fr = NKB * NKS - 1;
            const int jb = fr / NKS, kb = fr - jb * NKS;
            int key = 32 * jb + jo;
            if (key > P.k - 1) key = P.k - 1;
            raw[i] = *reinterpret_cast<const u32x4*>(P.kp + (int64_t)key * P.ldkp + a_ * DK + 16 * kb + 8 * hf);
        });
    };
    auto kp_commit = [&](auto nw_t, int wi, u32x4(&raw)[NF]) __attribute__((always_inline)) {
        constexpr int NW = decltype(nw_t)::value;
        static_for<0, (NKB * NKS + NW - 1) / NW>([&](auto i) __attribute__((always_inline)) {
            const int fr = wi + NW * i;
            if (fr < NKB * NKS) {
                const int jb = fr / NKS;
                u32x4 v = raw[i];
                if (32 * jb + j >= P.k) v = u32x4{0u, 0u, 0u, 0u};
                lds_kp[fr * 64 + lane] = v;
            }
        });
    };
    using W4 = std::integral_constant<int, 4>;
    using W8 = std::integral_constant<int, 8>;

    if (w8 < 4) {
        // =========================================== softmax waves ===========================================
        stamp_abs(60);
        // the softmax waves are the critical path of every step: they win every issue arbitration against their SIMD partner
        __builtin_amdgcn_s_setprio(3);
        const QT* __restrict__ q = reinterpret_cast<const QT*>(P.q);
        const float c_exp = P.scale * 1.44269504088896340736f;
        const bool attn_vec = AUX && (P.attn_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(P.attn) & 15) == 0;
        // ---- P image addressing.  Row r of the tile lives at r * RS; the 8-byte chunk c (4 keys) of its 64-byte key-block
        // segment is stored at chunk position c ^ ((r >> 1) & 7).  RS is an odd multiple of 64 bytes, so
        //   * a ds_write_b64 group (16 lanes = 16 consecutive rows, same logical chunk) covers all 32 banks exactly once,
        //   * a transpose-read group (32 lanes = 4 consecutive rows x 64 bytes) covers all 64 banks exactly once.
        const int prow = 32 * w + j;                  // this lane's query row inside the tile
        const int pswz = (prow >> 1) & 7;
        int waddr[4];                                 // byte offset of chunk (2*c4 + hf) of key block 0 in row prow
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) waddr[c4] = prow * RS + 8 * (((2 * c4) | hf) ^ pswz);
        const int ldq32 = (int)P.ldq;
        auto q_off = [&](int a_, int t_) __attribute__((always_inline)) -> unsigned {
            int qrow = t_ * TILE_ROWS + prow;
            if (qrow > n32 - 1) qrow = n32 - 1;
            return __umul24((unsigned)qrow, (unsigned)ldq32) + (a_ * DK + 8 * hf);
        };
        bf16x8 qf[NKS];  // Q fragments of the tile about to enter GEMM1
        // Prologue.  The cold-start latencies (kernarg, TLB, first HBM touch) are paid ONCE: the first tile's Q fragments
        // and the first head's Kp rows are requested before anything else.
        if (f_begin < f_end) {
            u32x4 raw0[NF];
            const QT* qp0 = q + q_off(a, t);
            static_for<0, NKS>([&](auto kb) __attribute__((always_inline)) { qf[kb] = load_frag(qp0 + 16 * kb); });
            __builtin_amdgcn_sched_barrier(0);
            stamp_abs(56);
            kp_issue(W8{}, w8, a, raw0);
            __builtin_amdgcn_sched_barrier(0);
            stamp_abs(58);
            kp_commit(W8{}, w8, raw0);
        }
        __syncthreads();   // K: the Kp image is complete
        stamp_abs(61);
        for (int f = f_begin; f < f_end; ++f) {
            const int row = t * TILE_ROWS + prow;   // this lane's query row
            int an = a, tn = t + 1;   // next work item
            if (tn == P.tiles_per_head) {
                tn = 0;
                an = a + 1;
            }
            if (a != cur_head) {
                // new head: every softmax wave passed A of the previous tile, so nobody reads the old Kp image any more.
                // All 8 waves fetch fragments (the pooling waves request theirs before they drain the last tile)
                u32x4 raw[NF];
                kp_issue(W8{}, w8, a, raw);
                __builtin_amdgcn_sched_barrier(0);   // do not let the conversions pull the loads apart
                if (published) {
                    __syncthreads();   // B: lets the pooling waves drain the previous head's last tile meanwhile
                    published = false;
                }
                kp_commit(W8{}, w8, raw);
                __syncthreads();   // K
                cur_head = a;
            }

            stamp(0);
            // ---- GEMM1 (swapped): S^T[key, row] = Kp Q^T.  A = Kp fragment (LDS), B = Q fragment: the C layout puts this
            //      lane's ONE query row (column j) in registers -- 16 keys per block: key = 32 jb + (r&3) + 8 (r>>2) + 4 hf.
            f32x16 s_acc[NKB];
            {
                // k-step outer, key block inner (m = kb * NKB + jb): consecutive MFMAs hit different accumulators, so there
                // is no dependent-issue stall.  The asm MFMAs keep their program order, which lets a 4-deep ring of Kp
                // fragments (16 VGPRs) stay exactly 4 MFMAs (~128 cycles) ahead of its consumer -- enough for the LDS latency.
                constexpr int M1 = NKB * NKS;
                constexpr int RING = 4;
                bf16x8 kf[RING];
                static_for<0, (M1 < RING ? M1 : RING)>([&](auto m_t) __attribute__((always_inline)) {
                    constexpr int m = decltype(m_t)::value;
                    kf[m] = __builtin_bit_cast(bf16x8, lds_kp[((m % NKB) * NKS + m / NKB) * 64 + lane]);
                });
                static_for<0, M1>([&](auto m_t) __attribute__((always_inline)) {
                    constexpr int m = decltype(m_t)::value;
                    constexpr int kb = m / NKB, jb = m % NKB;
                    if constexpr (kb == 0) {
                        // padded keys (only possible in the last two blocks, see make_plan) start at -inf: exp() gives 0.
                        // The test is wave-uniform; a full block takes the zero-C form (no init moves at all).
                        if (jb >= NKB - 2 && P.k < 32 * jb + 32) {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                s_acc[jb][r] = (32 * jb + (r & 3) + 8 * (r >> 2) + 4 * hf >= P.k) ? -INFINITY : 0.f;
                            mfma_vgpr<true>(s_acc[jb], kf[m % RING], qf[kb]);
                        } else {
                            mfma_vgpr_zero_c(s_acc[jb], kf[m % RING], qf[kb]);   // C = inline constant 0
                        }
                    } else {
                        mfma_vgpr<(NKB == 1) || std::is_same<QT, float>::value>(s_acc[jb], kf[m % RING], qf[kb]);
                    }
                    if constexpr (m + RING < M1) {
                        constexpr int mn = m + RING;
                        kf[m % RING] = __builtin_bit_cast(bf16x8, lds_kp[((mn % NKB) * NKS + mn / NKB) * 64 + lane]);
                    }
                });
                // MFMA result -> VALU read hazard of the asm MFMAs above (the compiler does not see them as MFMAs): park
                // for the full pipeline depth once per tile; the "+v" operands pin every later read of S behind this point
                static_for<0, NKB>([&](auto jb) __attribute__((always_inline)) {
                    if constexpr (jb == 0)
                        park_after_mfma(s_acc[jb]);
                    else
                        pin_vgpr(s_acc[jb]);
                });
            }
            stamp(1);
            if (published) {
                __syncthreads();   // B: closes the previous publish -- from here the pooling waves run GEMM2(f-1)
                published = false;
            }
            stamp(2);
            // next tile's Q: the fragment registers are free, the loads fly under the whole softmax
            {
                const bool has_next = f + 1 < f_end;
                const QT* qnext = q + q_off(has_next ? an : a, has_next ? tn : t);
                static_for<0, NKS>([&](auto kb) __attribute__((always_inline)) { qf[kb] = load_frag(qnext + 16 * kb); });
            }

            // ---- softmax over the keys of this lane's row: 16*NKB values in registers + the partner lane l ^ 32, fp32, IN
            // PLACE in the S registers, two values per instruction wherever the ISA has a packed form (v_pk_fma / add / mul)
            const bool rvalid = row < n32;
            float mc, lrow = 0.f;
            if constexpr (EXT) {
                // key-chunked launch: the softmax runs over ALL chunks' keys -- combine their (max, sum) pairs
                const float* st0 = P.stats + ((int64_t)a * P.n + (rvalid ? row : 0)) * 2;
                const int64_t cs = (int64_t)P.h * P.n * 2;
                float m = st0[0];
                for (int c = 1; c < P.n_chunks; ++c) m = fmaxf(m, st0[c * cs]);
                float l = 0.f;
                for (int c = 0; c < P.n_chunks; ++c) l = fmaf(st0[c * cs + 1], __builtin_amdgcn_exp2f(st0[c * cs] - m), l);
                mc = m;
                lrow = l;
            } else {
                // four independent v_max3 chains
                float mx[4] = {fmaxf(s_acc[0][0], s_acc[0][1]), fmaxf(s_acc[0][2], s_acc[0][3]), fmaxf(s_acc[0][4], s_acc[0][5]),
                               fmaxf(s_acc[0][6], s_acc[0][7])};
                static_for<4, 8 * NKB>([&](auto st_t) __attribute__((always_inline)) {
                    constexpr int st = decltype(st_t)::value;
                    constexpr int jb = st / 8, r = 2 * (st % 8);
                    if constexpr (!(jb == NKB - 1 && (st % 8) >= TAILP))
                        mx[st % 4] = fmaxf(fmaxf(mx[st % 4], s_acc[jb][r]), s_acc[jb][r + 1]);
                });
                mc = xhalf_max(fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]))) * c_exp;
            }
            {
                const f32x2 c2 = {c_exp, c_exp}, nm2 = {-mc, -mc};
                f32x2 l2 = {0.f, 0.f};
                // software pipeline: the exponent of pair e + 2 is computed before the exps of pair e issue, and the running
                // sum takes pair e - 1 -- no instruction sits right behind the one that feeds it
                constexpr int AHEAD = 2;
                constexpr int NE = 8 * (NKB - 1) + TAILP;        // register pairs that can hold a real key
                f32x2 ar[AHEAD + 1];
                static_for<0, AHEAD>([&](auto e_t) __attribute__((always_inline)) {
                    constexpr int e = decltype(e_t)::value;
                    ar[e] = __builtin_elementwise_fma(f32x2{s_acc[e / 8][2 * (e % 8)], s_acc[e / 8][2 * (e % 8) + 1]}, c2, nm2);
                });
                static_for<0, NE>([&](auto e_t) __attribute__((always_inline)) {
                    constexpr int e = decltype(e_t)::value;
                    constexpr int jb = e / 8, pr = e % 8;
                    if constexpr (e + AHEAD < NE) {
                        constexpr int en = e + AHEAD;
                        ar[en % (AHEAD + 1)] = __builtin_elementwise_fma(
                            f32x2{s_acc[en / 8][2 * (en % 8)], s_acc[en / 8][2 * (en % 8) + 1]}, c2, nm2);
                    }
                    if constexpr (e > 0)
                        l2 += f32x2{s_acc[(e - 1) / 8][2 * ((e - 1) % 8)], s_acc[(e - 1) / 8][2 * ((e - 1) % 8) + 1]};
                    s_acc[jb][2 * pr] = __builtin_amdgcn_exp2f(ar[e % (AHEAD + 1)][0]);
                    s_acc[jb][2 * pr + 1] = __builtin_amdgcn_exp2f(ar[e % (AHEAD + 1)][1]);
                    __builtin_amdgcn_sched_barrier(0);
                });
                l2 += f32x2{s_acc[(NE - 1) / 8][2 * ((NE - 1) % 8)], s_acc[(NE - 1) / 8][2 * ((NE - 1) % 8) + 1]};
                if constexpr (!EXT) lrow = xhalf_sum(l2[0] + l2[1]);
            }
            const float inv = rvalid ? __builtin_amdgcn_rcpf(lrow) : 0.f;
            if constexpr (AUX)
                if (P.lse && rvalid && hf == 0) P.lse[(int64_t)a * P.n_stride + row] = mc * 0.69314718055994530942f + __logf(lrow);
            stamp(3);

            // ---- A: every pooling wave finished the GEMM2 reads of the previous images (they got there long ago: GEMM2 is
            // shorter than the passes above).  Normalise, convert and publish P block by block: the LDS stores drain under
            // the VALU work of the next block instead of in a store-only tail.
            __syncthreads();
            stamp(4);
            float* arow = nullptr;
            int klim = 0;
            if constexpr (AUX) {
                // attention matrix: this lane owns 4 consecutive keys per (block, c4) of ONE row -> 16-byte stores
                arow = P.attn + ((int64_t)a * P.n_stride + row) * P.attn_ld + 4 * hf;
                // keys this lane may store, relative to its first one.  Opaque to the optimiser on purpose: the bound is
                // loop-invariant, and hoisting the 28..112 compare masks out of the tile loop spills ~240 SGPRs.
                klim = (P.attn && rvalid) ? P.k - 4 * hf : 0;
                asm volatile("" : "+v"(klim));
            }
            const f32x2 inv2 = {inv, inv};
            static_for<0, NKB>([&](auto jb_t) __attribute__((always_inline)) {
                constexpr int jb = decltype(jb_t)::value;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    if (jb == NKB - 1 && 2 * c4 >= TAILP) continue;   // padding-only columns: nothing to publish
                    f32x2 p01 = f32x2{s_acc[jb][4 * c4], s_acc[jb][4 * c4 + 1]} * inv2;
                    f32x2 p23 = f32x2{s_acc[jb][4 * c4 + 2], s_acc[jb][4 * c4 + 3]} * inv2;
                    if constexpr (AUX) {
                        if (P.drop.thresh) {   // training: drop probabilities (wave-uniform branch, never taken in inference)
                            const snf::philox_f4 mk = snf::dropout_mask4(P.drop, a, P.n, row, (int)P.attn_ld, 32 * jb + 8 * c4 + 4 * hf);
                            p01 *= f32x2{mk[0], mk[1]};
                            p23 *= f32x2{mk[2], mk[3]};
                        }
                    }
                    if constexpr (AUX) {
                        // the normalised fp32 row is stored before it is converted
                        const int key0 = 32 * jb + 8 * c4;
                        float* dst = arow + key0;
                        if (attn_vec) {
                            if (key0 < klim) *reinterpret_cast<f32x4*>(dst) = f32x4{p01[0], p01[1], p23[0], p23[1]};
                        } else {
                            if (key0 < klim) dst[0] = p01[0];
                            if (key0 + 1 < klim) dst[1] = p01[1];
                            if (key0 + 2 < klim) dst[2] = p23[0];
                            if (key0 + 3 < klim) dst[3] = p23[1];
                        }
                    }
                    const u32x2 pk = {__builtin_bit_cast(unsigned, __builtin_convertvector(p01, bf16x2)),
                                      __builtin_bit_cast(unsigned, __builtin_convertvector(p23, bf16x2))};
                    *reinterpret_cast<u32x2*>(lds_p + waddr[c4] + jb * 64) = pk;
                }
            });
            published = true;
            ++trace_it;
            a = an;
            t = tn;
        }
        stamp_abs(62);
        if (published) __syncthreads();   // B of the last tile
    } else {
        // =========================================== pooling waves ===========================================
        const QT* __restrict__ vg = reinterpret_cast<const QT*>(P.v);
        const int cb = (NCB == 4) ? w : (w & (NCB - 1));
        // reader of the P image (GEMM2 A operand): lane = group g (16 lanes) x i; rows 8*(g>>1) + 4*s + (i>>2), chunk
        // 4*(g&1) + (i&3)
        const int rg = lane >> 4, ri = lane & 15;
        const int rr0 = 8 * (rg >> 1) + (ri >> 2), rr1 = rr0 + 4;
        const int rch = 4 * (rg & 1) + (ri & 3);
        // key block of output tile ti of this wave = ti * (4 / NCB) + w / NCB: the wave-dependent part goes into the base
        const unsigned char* rbase0 = lds_p + rr0 * RS + 8 * (rch ^ ((rr0 >> 1) & 7)) + 64 * (w / NCB);
        const unsigned char* rbase1 = lds_p + rr1 * RS + 8 * (rch ^ ((rr1 >> 1) & 7)) + 64 * (w / NCB);

        // ---- V image.  V arrives ROW-major (it comes out of the same GEMM as Q): each pooling wave fetches 32 rows of the
        // tile with fully coalesced 16-byte loads (one row = DK/8 chunks of 8 columns), parks them in registers until the
        // images are free and stores them next to the P image; GEMM2 reads its B fragments (column on the lane, 8
        // consecutive rows in registers) back with the same hardware transpose-read as P.  Row pitch = 2*DK bytes, no
        // padding: chunk c of row r sits at chunk position (c + 4 rot(r)) mod NCH, which spreads the 4 rows x 64 bytes of a
        // transpose-read group over all 64 banks (rot = r & 3 for DK = 128, (r >> 1) & 1 for DK = 64: rows of 128 bytes
        // already alternate halves).
        constexpr int VRS = 2 * DK, NCH = DK / 8;     // row pitch (bytes), 16-byte chunks per row
        constexpr int RPI = 64 / NCH, NVI = 32 / RPI; // rows per load instruction, load instructions per wave and tile
        auto vrot = [](int r) __attribute__((always_inline)) -> int { return DK == 128 ? (r & 3) : ((r >> 1) & 1); };
        const int vl_row = lane / NCH, vl_ch = lane % NCH;          // loader: row inside the instruction, chunk
        const int vwaddr = (32 * w + vl_row) * VRS + 16 * ((vl_ch + 4 * vrot(vl_row)) & (NCH - 1));   // + i * RPI * VRS
        const int vr0 = 8 * (rg >> 1) + (ri >> 2), vr1 = vr0 + 4;   // reader: rows of the two transpose-reads
        const int vrc = 4 * cb + 2 * (rg & 1) + ((ri & 3) >> 1);     // chunk of this lane's 4 columns, + 8 * (ri & 1) bytes
        const unsigned char* vbase0 = lds_v + vr0 * VRS + 16 * ((vrc + 4 * vrot(vr0)) & (NCH - 1)) + 8 * (ri & 1);
        const unsigned char* vbase1 = lds_v + vr1 * VRS + 16 * ((vrc + 4 * vrot(vr1)) & (NCH - 1)) + 8 * (ri & 1);
        const int ldv32 = (int)P.ldv;
        auto v_off = [&](int a_, int t_, int i) __attribute__((always_inline)) -> unsigned {
            int vrow = t_ * TILE_ROWS + 32 * w + RPI * i + vl_row;
            if (vrow > n32 - 1) vrow = n32 - 1;
            return __umul24((unsigned)vrow, (unsigned)ldv32) + (a_ * DK + 8 * vl_ch);
        };

        f32x16 acc_o[NT];
        bf16x8 vld[NVI];  // this wave's 32 rows of V(f), in flight / parked until the images are free
        bf16x8 vfr[2];    // B fragments of the current and the next GEMM2 k-step (read one k-step ahead)
        // GEMM2 of the pending tile, MFMA m = sk * NT + ti  (sk = 16-row k-step of the tile, ti = output tile of the wave):
        // A = P^T fragment (key on the lane, 8 rows in registers), B = V fragment, both by transpose-read.  The P fragment of
        // MFMA m + 3 is requested before MFMA m issues (4-slot ring), the V fragment one whole k-step ahead.
        bf16x8 pfr[4];
        auto p_read = [&](auto m_tag) __attribute__((always_inline)) {
            constexpr int m = decltype(m_tag)::value;
            if constexpr (m < M2) {
                constexpr int sk = m / NT, ti = m % NT;
                constexpr int off = sk * 16 * RS + ti * (4 / NCB) * 64;
                pfr[m % 4] = lds_read_p_frag(rbase0 + off, rbase1 + off);
            }
        };
        auto gemm2_one = [&](auto m_tag) __attribute__((always_inline)) {
            constexpr int m = decltype(m_tag)::value;
            constexpr int sk = m / NT, ti = m % NT;
            const int t_idx = w + 4 * ti;  // tile = kb * NCB + cb ; cb == t_idx % NCB is constant per wave
            if constexpr (m == 0) {
                vfr[0] = lds_read_p_frag(vbase0, vbase1);
                p_read(std::integral_constant<int, 0>{});
                p_read(std::integral_constant<int, 1>{});
                p_read(std::integral_constant<int, 2>{});
            }
            p_read(std::integral_constant<int, m + 3>{});
            if constexpr (ti == 0 && sk + 1 < 8)
                vfr[(sk + 1) & 1] = lds_read_p_frag(vbase0 + (sk + 1) * 16 * VRS, vbase1 + (sk + 1) * 16 * VRS);
            if (NT * 4 == NKB * NCB || t_idx < NKB * NCB)
                acc_o[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pfr[m % 4], vfr[sk & 1], acc_o[ti], 0, 0, 0);
        };
        auto gemm2_all = [&]() __attribute__((always_inline)) {
            static_for<0, M2>([&](auto m_tag) __attribute__((always_inline)) {
                gemm2_one(m_tag);
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        auto flush = [&](int head) __attribute__((always_inline)) {
            if constexpr (VL) {
                if (P.out_direct) {   // this workgroup owns the whole head: registers 4 q4 + i of tile (kb, cb) = O[32 kb + i + 8 q4 + 4 hf, 32 cb + j]
                    const int64_t ld = (int64_t)P.h * DK;
#pragma unroll
                    for (int ti = 0; ti < NT; ++ti) {
                        const int t_idx = w + 4 * ti;
                        if (t_idx < NKB * NCB) {
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
            float* dst = P.partial + ((int64_t)bid * P.seg_count + seg) * (int64_t)(NKB * NCB) * 1024;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) {
                const int t_idx = w + 4 * ti;
                if (t_idx < NKB * NCB) {
                    const int key0 = 32 * (t_idx / NCB);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        // registers 4 q4 .. 4 q4 + 3 hold keys key0 + 8 q4 + 4 hf + i: skip the quads that are padding only
                        if (key0 + 8 * q4 < P.k) {
                            f32x4 v = {acc_o[ti][q4 * 4], acc_o[ti][q4 * 4 + 1], acc_o[ti][q4 * 4 + 2], acc_o[ti][q4 * 4 + 3]};
                            *reinterpret_cast<f32x4*>(dst + ((int64_t)(t_idx * 4 + q4) * 64 + lane) * 4) = v;
                        }
                    }
                }
            }
        };
        auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[ti][r] = 0.f;
        };

        if (f_begin < f_end) {   // the pooling waves fetch their half of the first head's Kp fragments
            u32x4 raw0[NF];
            kp_issue(W8{}, w8, a, raw0);
            __builtin_amdgcn_sched_barrier(0);
            zero_acc();
            kp_commit(W8{}, w8, raw0);
        } else {
            zero_acc();
        }
        __syncthreads();   // K (prologue)
        for (int f = f_begin; f < f_end; ++f) {
            int an = a, tn = t + 1;
            if (tn == P.tiles_per_head) {
                tn = 0;
                an = a + 1;
            }
            // V(f): the registers are free (V(f-1) went into the image before B), the loads fly under GEMM2(f-1)
            static_for<0, NVI>([&](auto i_t) __attribute__((always_inline)) {
                constexpr int i = decltype(i_t)::value;
                vld[i] = load_frag(vg + v_off(a, t, i));
            });
            u32x4 kraw[NF];
            if (a != cur_head) kp_issue(W8{}, w8, a, kraw);   // new head: this wave's share of its Kp, in flight under the drain
            __builtin_amdgcn_sched_barrier(0);
            if (published) {
                __syncthreads();   // B: P(f-1) and V(f-1) are complete
                published = false;
                stamp(5);
                gemm2_all();
                stamp(6);
            }
            if (a != cur_head) {   // the tile just accumulated was the last one of its head
                kp_commit(W8{}, w8, kraw);
                // K first (the Kp image is replaced, the softmax waves go on with GEMM1 / softmax of the new head), THEN the
                // flush: the partial tiles drain to HBM beside the softmax waves' work instead of in front of it
                __syncthreads();
                flush(cur_head);
                zero_acc();
                cur_head = a;
            }
            __syncthreads();   // A: the images are free (all GEMM2 reads done, softmax(f) computed)
            stamp(7);
            static_for<0, NVI>([&](auto i_t) __attribute__((always_inline)) {
                constexpr int i = decltype(i_t)::value;
                *reinterpret_cast<u32x4*>(lds_v + vwaddr + i * RPI * VRS) = __builtin_bit_cast(u32x4, vld[i]);
            });
            published = true;
            ++trace_it;
            a = an;
            t = tn;
        }
        if (published) {
            __syncthreads();   // B of the last tile
            gemm2_all();
        }
        if (cur_head >= 0) flush(cur_head);
        stamp_abs(63);
    }
}


// Row statistics of one key chunk (key-chunked launches, k above what one LDS image holds): the GEMM1 + max / exp / sum half
// of the kernel above, nothing else.  stats_out[(a * n + row) * 2 + {0, 1}] = (row max * scale * log2 e, sum of exp).
template <int DK, int NKB, typename QT>
__global__ __launch_bounds__(256, 1) void sparse_attn_stats_kernel(AttnParams P) {
    constexpr int NKS = DK / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* lds_kp = reinterpret_cast<u32x4*>(smem);                    // [NKB][NKS][64] MFMA A fragments of Kp
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, hf = lane >> 5;
    const QT* __restrict__ q = reinterpret_cast<const QT*>(P.q);
    const float c_exp = P.scale * 1.44269504088896340736f;
    const int n32 = (int)P.n;
    const int prow = 32 * w + j;
    const int ldq32 = (int)P.ldq;
    const int f_begin = blockIdx.x * P.tiles_per_wg;
    int f_end = f_begin + P.tiles_per_wg;
    if (f_end > P.total_tiles) f_end = P.total_tiles;
    const int first_head = f_begin / P.tiles_per_head;
    int a = first_head, t = f_begin - first_head * P.tiles_per_head;
    int cur_head = -1;
    constexpr int NF = (NKB * NKS + 3) / 4;
    bf16x8 qf[NKS];
    auto load_q = [&](int a_, int t_) __attribute__((always_inline)) {
        int qrow = t_ * TILE_ROWS + prow;
        if (qrow > n32 - 1) qrow = n32 - 1;
        const QT* qp = q + __umul24((unsigned)qrow, (unsigned)ldq32) + (a_ * DK + 8 * hf);
        static_for<0, NKS>([&](auto kb) __attribute__((always_inline)) { qf[kb] = load_frag(qp + 16 * kb); });
    };
    if (f_begin < f_end) load_q(a, t);
    for (int f = f_begin; f < f_end; ++f) {
        const int row = t * TILE_ROWS + prow;
        int an = a, tn = t + 1;
        if (tn == P.tiles_per_head) {
            tn = 0;
            an = a + 1;
        }
        if (a != cur_head) {
            __syncthreads();  // everyone finished reading the previous head's Kp
            static_for<0, NF>([&](auto i) __attribute__((always_inline)) {
                const int fr = w + 4 * i;
                if (fr < NKB * NKS) {
                    const int jb = fr / NKS, kb = fr - jb * NKS;
                    int key = 32 * jb + j;
                    const bool pad = key >= P.k;
                    if (pad) key = P.k - 1;
                    u32x4 v = *reinterpret_cast<const u32x4*>(P.kp + (int64_t)key * P.ldkp + a * DK + 16 * kb + 8 * hf);
                    if (pad) v = u32x4{0u, 0u, 0u, 0u};
                    lds_kp[fr * 64 + lane] = v;
                }
            });
            __syncthreads();
            cur_head = a;
        }
        f32x16 s_acc[NKB];
        static_for<0, NKB * NKS>([&](auto m_t) __attribute__((always_inline)) {
            constexpr int m = decltype(m_t)::value;
            constexpr int kb = m / NKB, jb = m % NKB;
            const bf16x8 kf = __builtin_bit_cast(bf16x8, lds_kp[(jb * NKS + kb) * 64 + lane]);
            if constexpr (kb == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    s_acc[jb][r] = (32 * jb + (r & 3) + 8 * (r >> 2) + 4 * hf >= P.k) ? -INFINITY : 0.f;
            }
            s_acc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kb], s_acc[jb], 0, 0, 0);
        });
        const bool has_next = f + 1 < f_end;
        if (has_next) load_q(an, tn);   // next tile's Q under this tile's softmax
        float mx0 = s_acc[0][0], mx1 = s_acc[0][1];
        static_for<0, NKB>([&](auto jb_t) __attribute__((always_inline)) {
            constexpr int jb = decltype(jb_t)::value;
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                mx0 = fmaxf(fmaxf(mx0, s_acc[jb][r]), s_acc[jb][r + 1]);
                mx1 = fmaxf(fmaxf(mx1, s_acc[jb][r + 2]), s_acc[jb][r + 3]);
            }
        });
        const float mc = xhalf_max(fmaxf(mx0, mx1)) * c_exp;
        float l0 = 0.f, l1 = 0.f;
        static_for<0, NKB>([&](auto jb_t) __attribute__((always_inline)) {
            constexpr int jb = decltype(jb_t)::value;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                l0 += __builtin_amdgcn_exp2f(fmaf(s_acc[jb][r], c_exp, -mc));
                l1 += __builtin_amdgcn_exp2f(fmaf(s_acc[jb][r + 1], c_exp, -mc));
            }
        });
        const float lrow = xhalf_sum(l0 + l1);
        if (row < n32 && hf == 0) {
            float* dst = P.stats_out + ((int64_t)a * P.n + row) * 2;
            dst[0] = mc;
            dst[1] = lrow;
        }
        a = an;
        t = tn;
    }
}

// out[key, a*DK + col] = sum over the (workgroup, segment) partials of head a, ascending workgroup order
template <int DK, int NKB>
__global__ __launch_bounds__(64) void reduce_partials_kernel(const float* __restrict__ partial, int num_wg, int seg_count,
                                                              int tiles_per_head, int tiles_per_wg, int total_tiles,
                                                              int k, int h, float* __restrict__ out,
                                                              const int* __restrict__ vl = nullptr, int direct_bags = 0) {
    constexpr int NCB = DK / 32;
    constexpr int TILES = NKB * NCB;
    if (vl) {   // varlen: blockIdx.z = bag; the bag's own launch geometry, partial slots and output rows
        const int* __restrict__ dsc = vl + VL_DESC * blockIdx.z;
        if (dsc[10] && direct_bags) return;   // the main kernel stored this bag's heads itself
        tiles_per_head = dsc[4], tiles_per_wg = dsc[5], total_tiles = dsc[6], seg_count = dsc[7], num_wg = dsc[9];
        partial += (int64_t)dsc[8] * TILES * 1024;
        out += (int64_t)dsc[3] * (h * DK);
    }
    const int a = blockIdx.y;
    // one wave per workgroup: 4 TILES h small workgroups (672 at config B) spread over all CUs; as 168 workgroups of 4 waves
    // the pass left a third of the chip idle
    const int unit = blockIdx.x;  // (tile, q4)
    const int lane = threadIdx.x;
    const int t_idx = unit >> 2, q4 = unit & 3;
    if (32 * (t_idx / NCB) + 8 * q4 >= k) return;   // padding only: never written by the main kernel
    const int f_lo = a * tiles_per_head, f_hi = (a + 1) * tiles_per_head - 1;
    int b_lo = f_lo / tiles_per_wg, b_hi = f_hi / tiles_per_wg;
    if (b_hi > num_wg - 1) b_hi = num_wg - 1;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const int64_t off = ((int64_t)(t_idx * 4 + q4) * 64 + lane) * 4;
    auto src_of = [&](int b) -> const float* {
        const int seg = a - (b * tiles_per_wg) / tiles_per_head;
        return partial + ((int64_t)b * seg_count + seg) * (int64_t)TILES * 1024 + off;
    };
    // 48 loads in flight (a whole head at the usual 43 contributing workgroups), tail included (a serial tail costs one full memory latency per leftover partial); the summation
    // order stays ascending in b, so the result is bit-reproducible
    for (int b = b_lo; b <= b_hi; b += 48) {
        f32x4 v[48];
#pragma unroll
        for (int u = 0; u < 48; ++u) {
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (b + u <= b_hi) v[u] = *reinterpret_cast<const f32x4*>(src_of(b + u));
        }
#pragma unroll
        for (int u = 0; u < 48; ++u) s += v[u];
    }
    const int kb = t_idx / NCB, cb = t_idx - kb * NCB;
    const int col = a * DK + 32 * cb + (lane & 31);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = 32 * kb + i + 8 * q4 + 4 * (lane >> 5);
        if (key < k) out[(int64_t)key * (h * DK) + col] = s[i];
    }
}




inline bool make_plan(int64_t n, int k, int h, int dk, Plan* pl, bool packed = false) {
    if (!(dk == 64 || dk == 128) || k < 1 || k > (dk == 128 ? 224 : 256)) return false;   // LDS: Kp + P + V images
    int nkb = (k + 31) / 32;
    // instantiated key-block counts
    const int opts[] = {1, 2, 4, 6, 7, 8};
    int sel = 0;
    for (int o : opts)
        if (o >= nkb) {
            sel = o;
            break;
        }
    if (!sel) return false;
    if (n > 0xffff00ll) return false;   // 24-bit row x pitch products inside the kernel
    int64_t tph = (n + TILE_ROWS - 1) / TILE_ROWS;
    int64_t total = tph * h;
    if (total > 0x7fffffff) return false;
    int cus = snf::cu_count();
    int64_t num_wg = total < cus ? total : cus;
    int64_t tpw = (total + num_wg - 1) / num_wg;
    // A bag inside a PACKED (varlen) launch does not have the chip to itself: at least SMALL_BAG_TILES tiles (1024 rows) per
    // workgroup, so a head of a small bag is one or two partial tiles instead of one per 128 rows -- the partial tiles are the
    // bulk of such a launch's bytes (measured: 64 bags x 1000 rows, 87 -> 54 us + reduction 40 -> 29 us).  A bag launched ALONE
    // keeps one tile per workgroup: its tiles run side by side on idle CUs (8 in a row cost it ~25 us of latency, measured).
    constexpr int64_t SMALL_BAG_TILES = 8;
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

template <int DK, int NKB, typename QT, bool AUX, bool EXT = false, int TAILP = 8, bool VL = false>
int launch_variant(const AttnParams& P, const Plan& pl, float* out, hipStream_t s) {
    constexpr int NKS = DK / 16;
    const size_t lds = (size_t)(NKB * NKS) * 1024 + (size_t)TILE_ROWS * (p_row_bytes(NKB) + 2 * DK);
    static thread_local bool attr_set = false;
    auto kern = sparse_attn_mfma_kernel<DK, NKB, QT, AUX, EXT, TAILP, VL>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            snf::set_error("sparse_attn_mfma: cannot reserve %zu bytes of LDS", lds);
            (void)hipGetLastError();
            return SNF_ELAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(pl.num_wg), dim3(512), lds, s, P);
    int rc = snf::check_launch("sparse_attn_mfma_kernel");
    if (rc) return rc;
    constexpr int TILES = NKB * (DK / 32);
    if (VL && P.out_direct && pl.tiles_per_head == -1) return SNF_OK;   // every bag stored its heads itself (make_varlen_plan)
    // varlen: pl.num_wg is the whole grid (all bags), one reduction slice (blockIdx.z) per bag
    hipLaunchKernelGGL((reduce_partials_kernel<DK, NKB>), dim3(TILES * 4, P.h, VL ? P.vl_bags : 1), dim3(64), 0, s, P.partial,
                       pl.num_wg, pl.seg_count, pl.tiles_per_head, pl.tiles_per_wg, pl.total_tiles, P.k, P.h, out,
                       VL ? P.vl : nullptr, (VL && P.out_direct) ? 1 : 0);
    return snf::check_launch("reduce_partials_kernel");
}

// varlen launches: bf16 Q | V (what the packed pipeline produces), single key chunk
template <int DK>
int launch_nkb_varlen(const AttnParams& P, const Plan& pl, float* out, hipStream_t s) {
    using QT = unsigned short;
    const bool aux = P.attn != nullptr || P.lse != nullptr;
    if (pl.nkb == 7 && P.k > 192 && P.k <= 200 && !aux) return launch_variant<DK, 7, QT, false, false, 2, true>(P, pl, out, s);
#define SNF_ATTN_VL_CASE(NB)                                                                                              \
    case NB:                                                                                                              \
        return aux ? launch_variant<DK, NB, QT, true, false, 8, true>(P, pl, out, s)                                      \
                   : launch_variant<DK, NB, QT, false, false, 8, true>(P, pl, out, s);
    switch (pl.nkb) {
#ifndef SNF_ATTN_DEV
        SNF_ATTN_VL_CASE(1)
        SNF_ATTN_VL_CASE(2)
        SNF_ATTN_VL_CASE(4)
        SNF_ATTN_VL_CASE(6)
        case 8:
            if constexpr (DK == 64)
                return aux ? launch_variant<DK, 8, QT, true, false, 8, true>(P, pl, out, s)
                           : launch_variant<DK, 8, QT, false, false, 8, true>(P, pl, out, s);
            break;
#endif
        SNF_ATTN_VL_CASE(7)
        default: break;
    }
#undef SNF_ATTN_VL_CASE
    snf::set_error("sparse_attn_mfma (varlen): key-block count %d not built", pl.nkb);
    return SNF_EUNSUPPORTED;
}

// Geometry of a varlen launch: every bag keeps the plan of its own launch (make_plan), the grids are concatenated.
// table (host memory, may be null to size it) = [bags][VL_DESC] descriptors, then the bag index of every workgroup.
struct VarlenPlan {
    int64_t total_wg, partial_slots;   // workgroups of the whole launch; partial tiles-slots (num_wg * seg_count summed)
    int nkb;
    bool all_direct;                   // every bag has one workgroup per head: no reduction pass at all
};
inline bool make_varlen_plan(const int64_t* offsets, int bags, int k, int h, int dk, VarlenPlan* vp, int32_t* table,
                             size_t table_ints) {
    vp->total_wg = 0, vp->partial_slots = 0, vp->nkb = 0, vp->all_direct = true;
    for (int b = 0; b < bags; ++b) {
        const int64_t n = offsets[b + 1] - offsets[b];
        Plan pl;
        if (n < 1 || offsets[b] > 0x7fffffffll || !make_plan(n, k, h, dk, &pl, true)) return false;
        const bool direct = pl.tiles_per_wg == pl.tiles_per_head;   // workgroup i of the bag = head i, whole
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
    return bags >= 1;
}

#define SNF_ATTN_CASE(NB, EXT)                                                                     \
    case NB:                                                                                       \
        return aux ? launch_variant<DK, NB, QT, true, EXT>(P, pl, out, s) : launch_variant<DK, NB, QT, false, EXT>(P, pl, out, s);
// 8 key blocks (256 keys) only exist for dk = 64: with dk = 128 the three LDS images hold 224 keys at most
#define SNF_ATTN_CASE8(EXT)                                                                        \
    case 8:                                                                                        \
        if constexpr (DK == 64)                                                                    \
            return aux ? launch_variant<DK, 8, QT, true, EXT>(P, pl, out, s) : launch_variant<DK, 8, QT, false, EXT>(P, pl, out, s); \
        break;
template <int DK, typename QT>
int launch_nkb(const AttnParams& P, const Plan& pl, float* out, hipStream_t s) {
    const bool aux = P.attn != nullptr || P.lse != nullptr;
    if (P.stats) {   // key-chunked launch: chunk sizes are in (kmax/2, kmax] -> 4, 6, 7 or 8 key blocks
        switch (pl.nkb) {
#ifndef SNF_ATTN_DEV
            SNF_ATTN_CASE(4, true)
            SNF_ATTN_CASE(6, true)
            SNF_ATTN_CASE8(true)
#endif
            SNF_ATTN_CASE(7, true)
            default: break;
        }
        snf::set_error("sparse_attn_mfma: chunked key-block count %d not built", pl.nkb);
        return SNF_EUNSUPPORTED;
    }
    // the inference shape of the reference's default (Lambda = 200: 8 keys in the 7th block), bf16 operands, no A / lse output
    if (pl.nkb == 7 && P.k > 192 && P.k <= 200 && !aux && std::is_same<QT, unsigned short>::value)
        return launch_variant<DK, 7, QT, false, false, 2>(P, pl, out, s);
    switch (pl.nkb) {
#ifndef SNF_ATTN_DEV   // development builds instantiate the config-B shape only (the file takes minutes otherwise)
        SNF_ATTN_CASE(1, false)
        SNF_ATTN_CASE(2, false)
        SNF_ATTN_CASE(4, false)
        SNF_ATTN_CASE(6, false)
        SNF_ATTN_CASE8(false)
#endif
        SNF_ATTN_CASE(7, false)
        default: break;
    }
    snf::set_error("sparse_attn_mfma: key-block count %d not built", pl.nkb);
    return SNF_EUNSUPPORTED;
}
#undef SNF_ATTN_CASE
#undef SNF_ATTN_CASE8

inline size_t mfma_workspace_bytes(const Plan& pl, int dk) {
    return (size_t)pl.num_wg * pl.seg_count * (size_t)(pl.nkb * (dk / 32)) * 1024 * sizeof(float);
}

template <int DK, int NKB, typename QT>
int launch_stats_variant(const AttnParams& P, const Plan& pl, hipStream_t s) {
    constexpr int NKS = DK / 16;
    hipLaunchKernelGGL((sparse_attn_stats_kernel<DK, NKB, QT>), dim3(pl.num_wg), dim3(256), (size_t)(NKB * NKS) * 1024, s, P);
    return snf::check_launch("sparse_attn_stats_kernel");
}
template <int DK, typename QT>
int launch_stats(const AttnParams& P, const Plan& pl, hipStream_t s) {
    switch (pl.nkb) {
#ifndef SNF_ATTN_DEV
        case 4: return launch_stats_variant<DK, 4, QT>(P, pl, s);
        case 6: return launch_stats_variant<DK, 6, QT>(P, pl, s);
        case 8:
            if constexpr (DK == 64) return launch_stats_variant<DK, 8, QT>(P, pl, s);
            break;
#endif
        case 7: return launch_stats_variant<DK, 7, QT>(P, pl, s);
        default: break;
    }
    snf::set_error("sparse_attn_stats: key-block count %d not built", pl.nkb);
    return SNF_EUNSUPPORTED;
}

// Key chunking: one launch holds at most KMAX keys (Kp + P + V images in 160 KiB of LDS).  More keys are split into
// n_chunks near-equal chunks; every chunk gets a statistics launch (row max / sum over its keys) and then a full launch
// that normalises with the statistics of ALL chunks -- the softmax stays exact, Q is read 2 n_chunks times and V n_chunks
// times instead of once.
// f32 Kp -> bf16 (round to nearest even), 8 elements per thread
__global__ __launch_bounds__(256) void kp_to_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                        int64_t groups) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    *reinterpret_cast<u32x4*>(dst + g * 8) = __builtin_bit_cast(u32x4, load_frag(src + g * 8));
}
inline size_t kp_staging_bytes(int k, int h, int dk) { return ((size_t)k * h * dk * 2 + 255) / 256 * 256; }

constexpr int MAX_CHUNKS = 8;
struct ChunkPlan {
    int n_chunks, chunk_k;   // chunk c covers keys [c * chunk_k, min(k, (c + 1) * chunk_k))
};
inline bool make_chunks(int k, int dk, ChunkPlan* cp) {
    if (!(dk == 64 || dk == 128) || k < 1) return false;
    const int kmax = dk == 128 ? 224 : 256;
    const int nc = (k + kmax - 1) / kmax;
    if (nc > MAX_CHUNKS) return false;
    cp->n_chunks = nc;
    cp->chunk_k = (k + nc - 1) / nc;
    return true;
}
// bytes of the partial-accumulator area (largest chunk) and of the statistics area behind it
inline bool chunked_workspace(int64_t n, int k, int h, int dk, size_t* partial_bytes, size_t* stats_bytes) {
    ChunkPlan cp;
    Plan pl;
    if (!make_chunks(k, dk, &cp) || !make_plan(n, cp.chunk_k, h, dk, &pl)) return false;
    *partial_bytes = (mfma_workspace_bytes(pl, dk) + 255) / 256 * 256;
    *stats_bytes = cp.n_chunks > 1 ? (size_t)cp.n_chunks * h * n * 2 * sizeof(float) : 0;
    return true;
}

}  // namespace

namespace snf {
// one translation unit per head width instantiates the kernels; stats_pass = sparse_attn_stats_kernel (key-chunked launches)
int attn_launch_dk128(int qv_dtype, bool stats_pass, const snf_attn::AttnParams& P, const snf_attn::Plan& pl, float* out,
                      hipStream_t s);
int attn_launch_dk64(int qv_dtype, bool stats_pass, const snf_attn::AttnParams& P, const snf_attn::Plan& pl, float* out,
                     hipStream_t s);
int attn_launch_varlen_dk128(const snf_attn::AttnParams& P, const snf_attn::Plan& pl, float* out, hipStream_t s);
int attn_launch_varlen_dk64(const snf_attn::AttnParams& P, const snf_attn::Plan& pl, float* out, hipStream_t s);
}  // namespace snf
