#pragma once
// EXPERIMENT, NOT SHIPPED (round 2).  Parity-green (54 shapes) but slower than the two-role kernel: 58 us vs 45 us at config B.
// The premise -- a matrix phase of one wave hides a vector phase of its SIMD partner -- does not hold on gfx950: a foreign MFMA
// costs the partner's VALU stream ~24 issue cycles, so MFMA time and VALU time of a SIMD add up (trace and ablations in
// profiles/history/r02_attn_pingpong_trace.txt, discussion in DESIGN.md section 4).  To try it again: include this header from
// snuffy_amd/csrc/sparse_attn_mfma.hip after sparse_attn_mfma_impl.h and dispatch launch_pp<7> for bf16 operands without A / lse.
// K7 (fast form, inference): the "ping-pong" organisation of the bf16 sparse-attention kernel (dk = 128).
//
//   per head a:   P_a = softmax_j(Q_a Kp_a^T * scale)  [n, k]      O_a = P_a^T V_a  [k, dk]        (snuffy.py:160-168)
//
// Why another organisation.  In the two-role kernel (sparse_attn_mfma_impl.h) the softmax wave of a SIMD runs GEMM1 -> softmax
// -> publish serially (7.2 k cycles per 128-row step) while the matrix pipe is busy 3.6 k of them.  Here every wave has BOTH
// kinds of phases and the two waves of a SIMD are half a step apart, so a matrix phase of one always runs beside a vector
// phase of the other:
//   * workgroup = 8 waves = two GROUPS X (waves 0-3) and Y (waves 4-7) with one wave of each group per SIMD.  A group owns a
//     64-row tile; its wave (rb, kh) holds row block rb (32 rows) x key half kh (4 / 3 key blocks of 32): S = 64 registers.
//   * a step of a group = four phases separated by the workgroup barrier:
//       G1  S^T[key, row] = Kp Q^T for its (row block, key half)        32 / 24 MFMAs (Kp fragments from LDS, Q from registers)
//       S   max / exp / sum over its 64 / 48 values per lane, (max, sum) of the key half -> LDS              vector only
//       C   combine with the other key half's statistics, normalise, convert, publish P (bf16) in LDS        vector only
//       G2  O[key, col] += P^T V over the 64 rows of a tile -- done by ALL EIGHT waves for every tile (each wave owns 3 - 4 of
//           the 28 output tiles of the head: 64 accumulator registers, no exchange at the end)                 14 - 16 MFMAs
//     group Y runs TWO barriers behind X; the intervals between barriers pair
//         (G1x | Cy)   (Sx + G2(y) | G2(y))   (Cx | G1y)   (G2(x) | Sy + G2(x))
//     -- a matrix phase beside a vector phase in every interval, 60 MFMAs per wave and pair of tiles = the matrix pipe's
//     3.8 k cycles per 128 rows as the floor of a step.
//   * V rows arrive by LDS-DMA (16-byte pieces, chunk rotation applied to the SOURCE address) two phases before G2 needs them;
//     the next tile's Q fragments are loaded into registers under G2.
//   * when the head changes (or the walk ends) the pipeline drains and every wave writes its output tiles as ONE partial per
//     workgroup and head, in the fragment order the existing reduce_partials_kernel sums.
// Inference path only: bf16 q / v / kp, no attention-matrix / lse outputs, one key chunk.  Everything else (training outputs,
// f32 operands, dk = 64, key chunking) stays on the two-role kernel.
#include "sparse_attn_mfma_impl.h"

namespace {

constexpr int PP_ROWS = 64;   // query rows per group step

template <int NKB>
__global__ __launch_bounds__(512, 2) void sparse_attn_pp_kernel(AttnParams P) {
    constexpr int DK = 128, NKS = 8, NCB = 4;
    constexpr int NK0 = (NKB + 1) / 2;              // key blocks of key half 0 (the other half holds NKB - NK0)
    constexpr int RS = p_row_bytes(NKB);
    constexpr int VRS = 2 * DK;
    constexpr int KP_BYTES = NKB * NKS * 1024;
    constexpr int PI_BYTES = PP_ROWS * RS;
    constexpr int V_BYTES = PP_ROWS * VRS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* lds_kp = reinterpret_cast<u32x4*>(smem);                                  // [NKB][NKS][64] A fragments of Kp
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = w >> 2, u = w & 3, rb = u >> 1, kh = u & 1;
    unsigned char* lds_p = smem + KP_BYTES + g * PI_BYTES;                           // this group's P image [64][RS]
    unsigned char* lds_v = smem + KP_BYTES + 2 * PI_BYTES + g * V_BYTES;             // this group's V image [64][VRS]
    const unsigned char* lds_p_o = smem + KP_BYTES + (g ^ 1) * PI_BYTES;             // the other group's images (GEMM2 only)
    const unsigned char* lds_v_o = smem + KP_BYTES + 2 * PI_BYTES + (g ^ 1) * V_BYTES;
    f32x2* lds_st = reinterpret_cast<f32x2*>(smem + KP_BYTES + 2 * PI_BYTES + 2 * V_BYTES) + g * 2 * PP_ROWS;   // [2 kh][64 rows]
    const int j = lane & 31, hf = lane >> 5;
    const int n32 = (int)P.n;
    const int jb0 = kh * NK0;
    const float c_exp = P.scale * 1.44269504088896340736f;
    const unsigned short* __restrict__ q = reinterpret_cast<const unsigned short*>(P.q);
    const unsigned short* __restrict__ vg = reinterpret_cast<const unsigned short*>(P.v);
    const int ldq32 = (int)P.ldq, ldv32 = (int)P.ldv;

    const int f_begin = blockIdx.x * P.tiles_per_wg;
    int f_end = f_begin + P.tiles_per_wg;
    if (f_end > P.total_tiles) f_end = P.total_tiles;
    if (f_begin >= f_end) return;
    const int first_head = f_begin / P.tiles_per_head;

    // ---- addressing that does not depend on the tile
    const int prow = 32 * rb + j;                                                  // this lane's row inside the group's tile
    int waddr[4];                                                                  // P image: chunk (2 c4 + hf) of key block 0
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) waddr[c4] = prow * RS + 8 * (((2 * c4) | hf) ^ ((prow >> 1) & 7));
    const int rg = lane >> 4, ri = lane & 15;
    const int rr0 = 8 * (rg >> 1) + (ri >> 2), rr1 = rr0 + 4;
    const int rch = 4 * (rg & 1) + (ri & 3);
    // output tiles of this wave: t_idx = w + 8 ti -> column block u, key block (w >> 2) + 2 ti
    constexpr int NT = (NKB * NCB + 7) / 8;
    const int kbw = w >> 2;
    const int poff0 = rr0 * RS + 8 * (rch ^ ((rr0 >> 1) & 7)) + 64 * kbw, poff1 = rr1 * RS + 8 * (rch ^ ((rr1 >> 1) & 7)) + 64 * kbw;
    const int vrc = 4 * u + 2 * (rg & 1) + ((ri & 3) >> 1);                        // column block of this wave's output tiles = u
    const int voff0 = rr0 * VRS + 16 * ((vrc + 4 * (rr0 & 3)) & 15) + 8 * (ri & 1);
    const int voff1 = rr1 * VRS + 16 * ((vrc + 4 * (rr1 & 3)) & 15) + 8 * (ri & 1);
    // V image by LDS-DMA: piece = 4 rows x 16 chunks; lane l lands at row 4 piece + (l >> 4), slot l & 15 and must hold the
    // chunk that the rotation puts there: chunk = (slot - 4 (row & 3)) & 15
    const int vl_row = lane >> 4, vl_ch = ((lane & 15) - 4 * (vl_row & 3)) & 15;   // (piece rows are multiples of 4)

    f32x16 acc_o[NT];         // output tiles (key block kbw + 2 ti, column block u), accumulated over all tiles of the head
    u32x4 qf[NKS];            // Q fragments of the tile about to enter G1 (hand-counted asm loads, see load_q / q_landed)
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[ti][r] = 0.f;
    };
    // O[key, col] += P^T V over the 64 rows of one tile (images of either group), this wave's output tiles.  MFMA m = sk * NT + ti
    // (sk = 16-row k-step, ti = output tile).  Left alone the compiler serialises {LDS read, wait, MFMA} through ONE fragment
    // register (measured: 68 % of the wave cycles parked on lgkmcnt); the issue order is pinned instead: the P fragment of MFMA
    // m + 3 and the V fragment of the next k-step are requested before MFMA m issues (4-slot / 2-slot rings).
    auto tile_ok = [&](int ti) __attribute__((always_inline)) -> bool { return 2 * ti + 1 < NKB || (kbw == 0 && 2 * ti < NKB); };
    auto gemm2 = [&](const unsigned char* pimg, const unsigned char* vimg) __attribute__((always_inline)) {
        constexpr int NSK = PP_ROWS / 16, M2 = NSK * NT;
        bf16x8 pfr[4], vfr[2];
        auto p_read = [&](auto m_tag) __attribute__((always_inline)) {
            constexpr int m = decltype(m_tag)::value;
            if constexpr (m < M2) {
                constexpr int sk = m / NT, ti = m % NT;
                pfr[m % 4] = lds_read_p_frag(pimg + poff0 + sk * 16 * RS + ti * 128, pimg + poff1 + sk * 16 * RS + ti * 128);
            }
        };
        __builtin_amdgcn_s_setprio(1);
        vfr[0] = lds_read_p_frag(vimg + voff0, vimg + voff1);
        p_read(std::integral_constant<int, 0>{});
        p_read(std::integral_constant<int, 1>{});
        p_read(std::integral_constant<int, 2>{});
        static_for<0, M2>([&](auto m_tag) __attribute__((always_inline)) {
            constexpr int m = decltype(m_tag)::value;
            constexpr int sk = m / NT, ti = m % NT;
            p_read(std::integral_constant<int, m + 3>{});
            if constexpr (ti == 0 && sk + 1 < NSK)
                vfr[(sk + 1) & 1] = lds_read_p_frag(vimg + voff0 + (sk + 1) * 16 * VRS, vimg + voff1 + (sk + 1) * 16 * VRS);
            if (tile_ok(ti)) acc_o[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pfr[m % 4], vfr[sk & 1], acc_o[ti], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_s_setprio(0);
    };
    // Q fragments are loaded by inline asm: the compiler does not count them, so they stay in flight across three phases
    // (issued at the end of G1(i), consumed by G1(i + 1)) without being drained by the wait for the V LDS-DMA in between (the
    // wait at the end of C leaves the 8 Q loads in flight: they were issued after the 4 V pieces).
    auto load_q = [&](int a_, int t_) __attribute__((always_inline)) {
        int row = t_ * PP_ROWS + prow;
        if (row > n32 - 1) row = n32 - 1;
        const unsigned short* qp = q + __umul24((unsigned)row, (unsigned)ldq32) + a_ * DK + 8 * hf;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qf[0]) : "v"(qp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(qf[1]) : "v"(qp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(qf[2]) : "v"(qp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:96" : "=v"(qf[3]) : "v"(qp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:128" : "=v"(qf[4]) : "v"(qp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:160" : "=v"(qf[5]) : "v"(qp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:192" : "=v"(qf[6]) : "v"(qp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:224" : "=v"(qf[7]) : "v"(qp) : "memory");
    };
    auto q_landed = [&]() __attribute__((always_inline)) {   // nothing younger than the Q loads is in flight at this point
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]), "+v"(qf[4]), "+v"(qf[5]), "+v"(qf[6]), "+v"(qf[7])
                     :
                     : "memory");
    };
    auto stage_v = [&](int a_, int t_) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = 4 * u + i;
            int row = t_ * PP_ROWS + 4 * piece + vl_row;
            if (row > n32 - 1) row = n32 - 1;
            const unsigned short* src = vg + __umul24((unsigned)row, (unsigned)ldv32) + a_ * DK + 8 * vl_ch;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds_v + piece * 1024), 16, 0, 0);
        }
    };
    int trace_n = 0;   // dev trace (tools/pp_trace.py): workgroup 0, waves 0 and 4 stamp s_memtime at arrival / release of barriers
    auto stamp = [&]() __attribute__((always_inline)) {
        if (P.trace && blockIdx.x == 0 && lane == 0 && u == 0 && trace_n < 256) P.trace[g * 256 + trace_n++] = __builtin_amdgcn_s_memtime();
    };
    auto barrier = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        stamp();
        __builtin_amdgcn_s_barrier();
        stamp();
        __builtin_amdgcn_sched_barrier(0);
    };

    zero_acc();
    int f = f_begin;
    while (f < f_end) {
        // ---- one head segment [f, fe): pairs of tiles of head a (tiles_per_head and tiles_per_wg are even)
        const int a = f / P.tiles_per_head;
        int fe = (a + 1) * P.tiles_per_head;
        if (fe > f_end) fe = f_end;
        const int t0 = f - a * P.tiles_per_head;
        const int npairs = (fe - f) >> 1;
        // Kp_a -> LDS as bf16 MFMA A fragments, all 8 waves (padded keys zeroed; their scores are forced to -inf below)
        for (int fr = w; fr < NKB * NKS; fr += 8) {
            const int jb = fr / NKS, kb = fr - jb * NKS;
            int key = 32 * jb + j;
            const bool pad = key >= P.k;
            if (pad) key = P.k - 1;
            u32x4 v4 = *reinterpret_cast<const u32x4*>(P.kp + (int64_t)key * P.ldkp + a * DK + 16 * kb + 8 * hf);
            if (pad) v4 = u32x4{0u, 0u, 0u, 0u};
            lds_kp[fr * 64 + lane] = v4;
        }
        // the pipelined walk over the segment's tile pairs; NKW = key blocks of this wave's key half, a compile-time count (the
        // two key halves differ by one block: the branch on kh sits OUTSIDE the loop, both sides execute the same barriers)
        auto walk = [&](auto nkw_t, auto last_t) __attribute__((always_inline)) {
            constexpr int NKW = decltype(nkw_t)::value;
            constexpr bool LAST_HALF = decltype(last_t)::value;
            // G1 of one tile: S^T[key, row] = Kp Q^T (swapped: the lane's ONE row in registers, 16 keys per block).  Only the LAST
            // key block of the launch can hold padded keys (make_pp_plan builds exactly ceil(k / 32) blocks) and it belongs to
            // key half 1: it starts from -inf there (exp gives 0, no masking pass); its 16 start values depend on k through an
            // opaque copy -- loop-invariant start values would be hoisted out of the tile loop and held in registers.  Every
            // other block starts from the constant 0 of its first MFMA.
            f32x16 s[NKW];
            auto g1 = [&](int t_) __attribute__((always_inline)) {
                q_landed();               // Q(t) (issued three phases ago) is in the registers
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                // MFMA m = kb * NKW + jb (consecutive MFMAs hit different accumulators); the Kp fragment of MFMA m + 4 is
                // requested before MFMA m issues (4-slot ring, issue order pinned -- see gemm2)
                constexpr int M1 = NKS * NKW, RING = 4;
                bf16x8 kf[RING];
                auto k_read = [&](auto m_tag) __attribute__((always_inline)) {
                    constexpr int m = decltype(m_tag)::value;
                    if constexpr (m < M1) kf[m % RING] = __builtin_bit_cast(bf16x8, lds_kp[((jb0 + m % NKW) * NKS + m / NKW) * 64 + lane]);
                };
                static_for<0, RING>([&](auto m_tag) __attribute__((always_inline)) { k_read(m_tag); });
                static_for<0, M1>([&](auto m_tag) __attribute__((always_inline)) {
                    constexpr int m = decltype(m_tag)::value;
                    constexpr int kb = m / NKW, jb = m % NKW;
                    const bf16x8 qk = __builtin_bit_cast(bf16x8, qf[kb]);
                    if constexpr (kb == 0) {
                        f32x16 c0;
                        if (LAST_HALF && jb == NKW - 1) {
                            int kk = P.k - 32 * (NKB - 1) - 4 * hf;      // valid keys of the block from this lane's first one
                            asm volatile("" : "+v"(kk));
#pragma unroll
                            for (int r = 0; r < 16; ++r) c0[r] = ((r & 3) + 8 * (r >> 2) >= kk) ? -INFINITY : 0.f;
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r) c0[r] = 0.f;
                        }
                        s[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[m % RING], qk, c0, 0, 0, 0);
                    } else {
                        s[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[m % RING], qk, s[jb], 0, 0, 0);
                    }
                    k_read(std::integral_constant<int, m + RING>{});
                    __builtin_amdgcn_sched_barrier(0);
                });
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                // V(t) by LDS-DMA: its image was released by the barrier that closed the last GEMM2 on it.  Issued AFTER the
                // MFMA burst: with an LDS-DMA pending the compiler stops counting lgkmcnt for the Kp fragment ring (every
                // wait becomes lgkmcnt(0)); the pieces still have the S and C phases to land.
                if (!(P.trace_wg & 1) || t_ == t0 + g) stage_v(a, t_);     // (dev ablation bit 0: reuse the first V image)
            };
            // The loop is rotated: G1 of tile i + 1 closes iteration i.  The hand-counted Q loads are then issued and waited
            // for inside ONE iteration (a value defined by an asm load must not cross a loop phi: the compiler may copy the
            // register there before the data has landed); what crosses the back edge is S, an ordinary value.
            // (the first tile's Q loads are issued HERE, inside the variant that consumes them, for the same reason; they fly
            // under the barrier that publishes the Kp image and, for group Y, under its two idle intervals)
            load_q(a, t0 + g);
            __syncthreads();          // the Kp image is complete
            if (g == 1) {             // group Y runs two barriers behind group X
                barrier();
                barrier();
            }
            g1(t0 + g);
            for (int i = 0; i < npairs; ++i) {
                const int t = t0 + 2 * i + g;
                const int row = t * PP_ROWS + prow;
                const bool rvalid = row < n32;
                const bool has_next = i + 1 < npairs;
                if (has_next && !(P.trace_wg & 2)) load_q(a, t + 2);   // Q(t + 2) flies under S, C and G2 (dev ablation bit 1: skip)
                barrier();
                // ================= S: statistics of this key half -- max on the raw scores, exp2(s c - max c) as one fma + exp,
                //                   two values per instruction where the ISA has a packed form
                float mx0 = fmaxf(s[0][0], s[0][1]), mx1 = fmaxf(s[0][2], s[0][3]);
#pragma unroll
                for (int e = 1; e < 4 * NKW; ++e) {
                    mx0 = fmaxf(fmaxf(mx0, s[e >> 2][4 * (e & 3)]), s[e >> 2][4 * (e & 3) + 1]);
                    mx1 = fmaxf(fmaxf(mx1, s[e >> 2][4 * (e & 3) + 2]), s[e >> 2][4 * (e & 3) + 3]);
                }
                const float mraw = xhalf_max(fmaxf(mx0, mx1));
                const float mw = mraw * c_exp;                                        // -inf only if every key of the half is padding
                const f32x2 c2 = {c_exp, c_exp}, nm2 = {mraw == -INFINITY ? 0.f : -mw, mraw == -INFINITY ? 0.f : -mw};
                f32x2 l2 = {0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 8 * NKW; ++e) {
                    const f32x2 ar = __builtin_elementwise_fma(f32x2{s[e >> 3][2 * (e & 7)], s[e >> 3][2 * (e & 7) + 1]}, c2, nm2);
                    s[e >> 3][2 * (e & 7)] = __builtin_amdgcn_exp2f(ar[0]);
                    s[e >> 3][2 * (e & 7) + 1] = __builtin_amdgcn_exp2f(ar[1]);
                    l2 += f32x2{s[e >> 3][2 * (e & 7)], s[e >> 3][2 * (e & 7) + 1]};
                }
                const float lw = xhalf_sum(l2[0] + l2[1]);
                if (hf == 0) lds_st[kh * PP_ROWS + prow] = f32x2{mw, lw};
                // the other group's tile that was published one interval ago: X pools y(i - 1), Y pools x(i)
                if (g == 1 || i > 0) gemm2(lds_p_o, lds_v_o);
                barrier();
                // ================= C: combine the two key halves exactly, normalise, publish P
                {
                    const f32x2 so = lds_st[(kh ^ 1) * PP_ROWS + prow];
                    const float m = fmaxf(mw, so[0]);
                    const float l = fmaf(lw, __builtin_amdgcn_exp2f(mw - m), so[1] * __builtin_amdgcn_exp2f(so[0] - m));
                    const float fs = rvalid ? __builtin_amdgcn_exp2f(mw - m) * __builtin_amdgcn_rcpf(l) : 0.f;
                    const f32x2 fs2 = {fs, fs};
#pragma unroll
                    for (int jb = 0; jb < NKW; ++jb)
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4) {
                            const f32x2 p01 = f32x2{s[jb][4 * c4], s[jb][4 * c4 + 1]} * fs2;
                            const f32x2 p23 = f32x2{s[jb][4 * c4 + 2], s[jb][4 * c4 + 3]} * fs2;
                            const u32x2 pk = {__builtin_bit_cast(unsigned, __builtin_convertvector(p01, bf16x2)),
                                              __builtin_bit_cast(unsigned, __builtin_convertvector(p23, bf16x2))};
                            *reinterpret_cast<u32x2*>(lds_p + waddr[c4] + (jb0 + jb) * 64) = pk;
                        }
                }
                // this wave's pieces of V(t) have landed; the 8 Q loads issued after them may stay in flight
                if (has_next)
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                barrier();
                // ================= G2 of the own tile (all eight waves are in it)
                gemm2(lds_p, lds_v);
                barrier();
                // ================= G1 of the next tile
                if (has_next) g1(t + 2);
            }
        };
        if (kh == 0)
            walk(std::integral_constant<int, NK0>{}, std::false_type{});
        else
            walk(std::integral_constant<int, NKB - NK0>{}, std::true_type{});
        if (g == 0) {          // tail of group X: Y's last tile is pooled one interval after Y published it
            barrier();
            gemm2(lds_p_o, lds_v_o);
            barrier();
        }
        // ---- every wave is past the segment's last GEMM2: write this wave's output tiles as the workgroup's partial of head a
        {
            float* dst = P.partial + ((int64_t)blockIdx.x * P.seg_count + (a - first_head)) * (int64_t)(NKB * NCB) * 1024;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
                if (tile_ok(ti)) {
                    const int t_idx = w + 8 * ti;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        if (32 * (kbw + 2 * ti) + 8 * q4 < P.k)   // quads of padding only are never written (nor read by the reduce)
                            *reinterpret_cast<f32x4*>(dst + ((int64_t)(t_idx * 4 + q4) * 64 + lane) * 4) =
                                f32x4{acc_o[ti][4 * q4], acc_o[ti][4 * q4 + 1], acc_o[ti][4 * q4 + 2], acc_o[ti][4 * q4 + 3]};
                }
        }
        zero_acc();
        __syncthreads();       // nobody reads the Kp image any more: the next head's may be written
        f = fe;
    }
}

inline bool make_pp_plan(int64_t n, int k, int h, int dk, Plan* pl) {
    if (dk != 128 || k < 1 || k > 224) return false;
    const int nkb = (k + 31) / 32;
    if (nkb != 7 && nkb != 4) return false;                         // built key-block counts (193..224 and 97..128 keys)
    if (n > 0xffff00ll) return false;
    const int64_t tph = 2 * ((n + 2 * PP_ROWS - 1) / (2 * PP_ROWS));   // 64-row tiles per head, even (groups work in pairs)
    const int64_t total = tph * h;
    if (total > 0x7fffffff) return false;
    const int cus = snf::cu_count();
    int64_t num_wg = total / 2 < cus ? total / 2 : cus;
    int64_t tpw = (total + num_wg - 1) / num_wg;
    tpw += tpw & 1;
    num_wg = (total + tpw - 1) / tpw;
    pl->num_wg = (int)num_wg;
    pl->tiles_per_head = (int)tph;
    pl->tiles_per_wg = (int)tpw;
    pl->total_tiles = (int)total;
    pl->seg_count = (int)((tpw + tph - 1) / tph + 1);
    pl->nkb = nkb;
    return true;
}

template <int NKB>
int launch_pp(const AttnParams& P, const Plan& pl, float* out, hipStream_t s) {
    constexpr int lds = NKB * 8 * 1024 + 2 * PP_ROWS * p_row_bytes(NKB) + 2 * PP_ROWS * 256 + 2 * 2 * PP_ROWS * 8;
    static thread_local bool attr_set = false;
    auto kern = sparse_attn_pp_kernel<NKB>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            snf::set_error("sparse_attn_pp: cannot reserve %d bytes of LDS", lds);
            (void)hipGetLastError();
            return SNF_ELAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(pl.num_wg), dim3(512), lds, s, P);
    int rc = snf::check_launch("sparse_attn_pp_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL((reduce_partials_kernel<128, NKB>), dim3(NKB * 4, P.h), dim3(256), 0, s, P.partial, pl.num_wg, pl.seg_count,
                       pl.tiles_per_head, pl.tiles_per_wg, pl.total_tiles, P.k, P.h, out);
    return snf::check_launch("reduce_partials_kernel");
}

inline size_t pp_workspace_bytes(const Plan& pl) { return (size_t)pl.num_wg * pl.seg_count * (size_t)(pl.nkb * 4) * 1024 * sizeof(float); }

}  // namespace
