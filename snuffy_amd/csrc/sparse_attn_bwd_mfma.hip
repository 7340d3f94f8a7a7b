// K7-bwd (fast form): backward of Snuffy's sparse attention on the CDNA4 matrix cores (bf16 operands, fp32 accumulate).
//
//   forward (snuffy.py:160-168), per head a:   P = softmax_j(Q Kp^T * scale) [n, k],  Pd = P o M,  O = Pd^T V [k, dk]
//   backward, with the saved row log-sum-exp (P is recomputed, never read back):
//       dV = Pd dO              [n, dk]          dP = (V dO^T) o M           [n, k]
//       D  = rowsum(P o dP)     [n]              dS = P o (dP - D) * scale   [n, k]
//       dQ = dS Kp              [n, dk]          dKp = dS^T Q                [k, dk]   (not here: dS is written out and the
//                                                                                       forward's P^T V machinery contracts it)
//
// Dataflow of one wave = 32 query rows of one head, all products with v_mfma_f32_32x32x16_bf16 in the "swapped" form of the
// forward kernel (this lane's ONE query row on the lane axis, keys / columns in registers):
//   S^T  = Kp  Q^T     A = Kp fragment  (LDS image, 16-byte row reads),  B = Q fragment (HBM -> registers)
//   dP^T = dO  V^T     A = dO fragment  (LDS image, 16-byte row reads),  B = V fragment (HBM -> registers)
//   P, dP -> bf16 pairs in registers for all key blocks; D = sum_j P dP is register-local + one cross-half exchange
//   dV^T = dO^T Pd^T   A = dO^T fragment (SAME image, hardware transpose-read),  B = Pd^T straight from the registers
//   dQ^T = Kp^T dS^T   A = Kp^T fragment (SAME image, hardware transpose-read),  B = dS^T straight from the registers
// so P, dP and dS never touch LDS, the waves of a workgroup never synchronise inside a head, and dV / dQ rows are owned by
// one wave (no cross-workgroup reduction).  The Kp and dO images are row-major bf16 [key][dk]; the 16-byte chunks of a row
// are permuted by the row index (img_pos below), which makes BOTH access patterns bank-conflict-free: the 16-lane groups of a ds_read_b128 (16 different rows) and the 32-lane groups of a
// ds_read_b64_tr_b16 (4 consecutive rows x 64 bytes).
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "philox.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int TILE_ROWS = 128;  // query rows per workgroup step (4 waves x 32)

struct BwdParams {
    const void* q;      // [n, ldq]
    const void* v;      // [n, ldv]
    const float* kp;    // [k, d] f32
    const float* dout;  // [k, d] f32
    const float* lse;   // [h, n]   log-sum-exp of the scaled scores (forward)
    const float* mask;  // [h, n, k] dropout keep-mask / (1 - p), or null
    int64_t n, ldq, ldv, d;
    int k, h;
    float scale;
    void* dq;   // [n, ldd] f32 or bf16 (dqv_bf16)
    void* dv;   // [n, ldd]
    int64_t ldd;
    int dqv_bf16;
    void* ds;   // [h, n, k] f32, or bf16 when ds_bf16 (the caller then contracts it with a bf16 library GEMM)
    int ds_bf16;
    int tiles_per_head, tiles_per_wg, total_tiles;
    snf::DropoutState drop;   // mask regenerated in registers when thresh != 0 (and mask == null): same stream as the forward
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
__device__ __forceinline__ bf16x8 load_frag(const float* p) {
    f32x4 lo = *reinterpret_cast<const f32x4*>(p);
    f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
    f32x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_convertvector(v, bf16x8);
}
__device__ __forceinline__ bf16x8 load_frag(const unsigned short* p) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ bf16x8 lds_tr_frag(const unsigned char* p0, const unsigned char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_f(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// chunk position inside an image row of 2*DK bytes (DK/8 chunks of 16 bytes = DK/32 blocks of 64 bytes): the block index is
// rotated by the row (so the 4 consecutive rows x 64 bytes of a transpose-read group fall into 4 different 64-byte bank
// quarters: r & 3 for 256-byte rows, (r >> 1) & 1 for 128-byte rows, whose rows already alternate halves), the chunk inside
// the block is XOR-ed with (r >> 2) & 3 (so the 16 rows of a ds_read_b128 service group hit 16 different bank quads).
template <int DK>
__device__ __forceinline__ int img_pos(int row, int c) {
    const int rot = DK == 128 ? (row & 3) : ((row >> 1) & 1);
    return 4 * (((c >> 2) + rot) & (DK / 32 - 1)) + ((c & 3) ^ ((row >> 2) & 3));
}

// MODE 0: every option at run time (mask tensor, fp32 dS, K not a multiple of 4).  MODE 1 / 2: the training path's shape --
// bf16 dS, K % 4 == 0, no mask tensor -- without (1) / with (2) the in-register Philox mask.  The fast modes drop the per-store
// predicates of full key blocks, compare against K only in the last block, and (2) generate the mask ONCE: a dropped key is
// remembered in the sign bit of its bf16 probability (P >= 0), so the second use costs a max instead of ten Philox rounds.
template <int DK, int NKB, typename QT, int MODE>
__global__ __launch_bounds__(256, 1) void sparse_attn_bwd_mfma_kernel(BwdParams P) {
    constexpr bool FAST = MODE != 0;
    constexpr int NKS = DK / 16, NCB = DK / 32, RP = 2 * DK, NCH = DK / 8;   // k-steps, column blocks, image row pitch, chunks
    constexpr int IMG = 32 * NKB * RP;   // bytes of one image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* img_kp = smem;
    unsigned char* img_do = smem + IMG;

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, hf = lane >> 5;
    const QT* __restrict__ q = reinterpret_cast<const QT*>(P.q);
    const QT* __restrict__ vg = reinterpret_cast<const QT*>(P.v);
    const float c_exp = P.scale * 1.44269504088896340736f;
    const int n32 = (int)P.n;
    const int prow = 32 * w + j;

    // A fragments with the KEY on the lane (S^T, dP^T): row 32 jb + j, 16-byte chunk 2 kb + hf
    int ra[NKS];
#pragma unroll
    for (int kb = 0; kb < NKS; ++kb) ra[kb] = j * RP + 16 * img_pos<DK>(j, 2 * kb + hf);
    // A fragments with the COLUMN on the lane (dV^T, dQ^T), by transpose-read: 16-lane group tg, lane ti in it;
    // rows k_base + 4 (tg >> 1) + (ti >> 2) (+8 for the second read), chunk 4 db + 2 (tg & 1) + ((ti & 3) >> 1), half ti & 1
    const int tg = lane >> 4, ti = lane & 15;
    int rt0[NCB], rt1[NCB];
#pragma unroll
    for (int db = 0; db < NCB; ++db) {
        const int r0 = 4 * (tg >> 1) + (ti >> 2), r1 = r0 + 8;
        const int c = 4 * db + 2 * (tg & 1) + ((ti & 3) >> 1);
        rt0[db] = r0 * RP + 16 * img_pos<DK>(r0, c) + 8 * (ti & 1);
        rt1[db] = r1 * RP + 16 * img_pos<DK>(r1, c) + 8 * (ti & 1);
    }

    const int f_begin = blockIdx.x * P.tiles_per_wg;
    int f_end = f_begin + P.tiles_per_wg;
    if (f_end > P.total_tiles) f_end = P.total_tiles;
    int a = f_begin / P.tiles_per_head, t = f_begin - a * P.tiles_per_head;
    int cur_head = -1;

    for (int f = f_begin; f < f_end; ++f) {
        if (a != cur_head) {
            // Kp_a and dO_a -> LDS (bf16, row-major, swizzled): thread -> (row, chunk), a row's 512 bytes are read coalesced
            __syncthreads();
            for (int ci = threadIdx.x; ci < 32 * NKB * NCH; ci += 256) {
                const int row = ci / NCH, c = ci % NCH;
                u32x4 vk = {0u, 0u, 0u, 0u}, vo = {0u, 0u, 0u, 0u};
                if (row < P.k) {
                    vk = __builtin_bit_cast(u32x4, load_frag(P.kp + (int64_t)row * P.d + a * DK + 8 * c));
                    vo = __builtin_bit_cast(u32x4, load_frag(P.dout + (int64_t)row * P.d + a * DK + 8 * c));
                }
                const int off = row * RP + 16 * img_pos<DK>(row, c);
                *reinterpret_cast<u32x4*>(img_kp + off) = vk;
                *reinterpret_cast<u32x4*>(img_do + off) = vo;
            }
            __syncthreads();
            cur_head = a;
        }
        const int row = t * TILE_ROWS + prow;
        const bool rvalid = row < n32;
        const int lrow = rvalid ? row : n32 - 1;
        // B operands: this lane's row of Q and V, 8 consecutive columns per k-step
        bf16x8 qf[NKS], vf[NKS];
        {
            const QT* qp = q + (int64_t)lrow * P.ldq + a * DK + 8 * hf;
            const QT* vp = vg + (int64_t)lrow * P.ldv + a * DK + 8 * hf;
            static_for<0, NKS>([&](auto kb) __attribute__((always_inline)) {
                qf[kb] = load_frag(qp + 16 * kb);
                vf[kb] = load_frag(vp + 16 * kb);
            });
        }
        // rows past the bag end get P = 0 (exp2(-inf))
        const float lse2 = rvalid ? P.lse[(int64_t)a * P.n + row] * 1.44269504088896340736f : INFINITY;
        const float* mrow = P.mask ? P.mask + ((int64_t)a * P.n + lrow) * P.k : nullptr;
        int kk = P.k;
        asm volatile("" : "+s"(kk));

        unsigned ppk[NKB][8], dpk[NKB][8];   // Pd (= P o M) and dP (o M), later dS, as bf16 pairs: [block][pair of registers]
        float dsum = 0.f;
        static_for<0, NKB>([&](auto jb_t) __attribute__((always_inline)) {
            constexpr int jb = decltype(jb_t)::value;
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // padded keys: P = 0.  Only the last block can hold any; kk is opaque so that the compare masks are not hoisted
                // out of the tile loop (16 NKB SGPR pairs: that was 225 spilled SGPRs)
                if (!FAST || jb == NKB - 1)
                    s[r] = (32 * jb + (r & 3) + 8 * (r >> 2) + 4 * hf >= kk) ? -INFINITY : 0.f;
                else
                    s[r] = 0.f;
                dp[r] = 0.f;
            }
            static_for<0, NKS>([&](auto kb_t) __attribute__((always_inline)) {
                constexpr int kb = decltype(kb_t)::value;
                const bf16x8 ak = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(img_kp + ra[kb] + jb * 32 * RP));
                const bf16x8 ao = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(img_do + ra[kb] + jb * 32 * RP));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak, qf[kb], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ao, vf[kb], dp, 0, 0, 0);
            });
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                float pv[4], gv[4];
                f32x4 mk = {1.f, 1.f, 1.f, 1.f};
                if constexpr (MODE == 0) {
                    if (mrow) {
                        const int key0 = 32 * jb + 8 * c4 + 4 * hf;
#pragma unroll
                        for (int e = 0; e < 4; ++e) mk[e] = (key0 + e < P.k) ? mrow[key0 + e] : 0.f;
                    } else if (P.drop.thresh) {
                        const snf::philox_f4 m4 = snf::dropout_mask4(P.drop, a, P.n, lrow, P.k, 32 * jb + 8 * c4 + 4 * hf);
                        mk = f32x4{m4[0], m4[1], m4[2], m4[3]};
                    }
                } else if constexpr (MODE == 2) {
                    const snf::philox_f4 m4 = snf::dropout_mask4(P.drop, a, P.n, lrow, P.k, 32 * jb + 8 * c4 + 4 * hf);
                    mk = f32x4{m4[0], m4[1], m4[2], m4[3]};
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pv[e] = __builtin_amdgcn_exp2f(fmaf(s[4 * c4 + e], c_exp, -lse2));
                    gv[e] = dp[4 * c4 + e] * mk[e];          // dP = dPd o M
                    if constexpr (MODE == 2) pv[e] = mk[e] == 0.f ? -pv[e] : pv[e];   // dropped: remembered in the sign
                }
                // ppk holds P (un-masked: dS = P o (dP - D)); the masked Pd for dV is rebuilt from it below when a mask exists
                const unsigned p01 = pack2(pv[0], pv[1]), p23 = pack2(pv[2], pv[3]);
                const unsigned g01 = pack2(gv[0], gv[1]), g23 = pack2(gv[2], gv[3]);
                ppk[jb][2 * c4] = p01;
                ppk[jb][2 * c4 + 1] = p23;
                dpk[jb][2 * c4] = g01;
                dpk[jb][2 * c4 + 1] = g23;
                // D += P dP on the ROUNDED values, the ones dS is built from below: sum_j dS_j stays 0 to fp32 accuracy
                dsum = fmaf(lo_f(p01), lo_f(g01), dsum);
                dsum = fmaf(hi_f(p01), hi_f(g01), dsum);
                dsum = fmaf(lo_f(p23), lo_f(g23), dsum);
                dsum = fmaf(hi_f(p23), hi_f(g23), dsum);
            }
            // keep the key blocks apart: interleaved, the scheduler holds every block's 32 accumulator registers live at once
            // (512 registers + 190 spilled before this fence, profiles/history/r02_attn_bwd_resources.txt)
            __builtin_amdgcn_sched_barrier(0);
        });
        const float dtot = xhalf_sum(dsum);

        // dS = P o (dP - D) * scale: written out (fp32, for dKp) and kept as bf16 pairs for dQ; Pd = P o M for dV
        float* dsrow = reinterpret_cast<float*>(P.ds) + ((int64_t)a * P.n + lrow) * P.k;
        unsigned short* dsrow16 = reinterpret_cast<unsigned short*>(P.ds) + ((int64_t)a * P.n + lrow) * P.k;
        if constexpr (FAST) {
            // rows past the bag end skip the whole pass (their registers feed accumulators that are never stored)
            if (rvalid) {
                static_for<0, NKB>([&](auto jb_t) __attribute__((always_inline)) {
                    constexpr int jb = decltype(jb_t)::value;
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        const unsigned p01 = ppk[jb][2 * c4], p23 = ppk[jb][2 * c4 + 1];
                        const unsigned g01 = dpk[jb][2 * c4], g23 = dpk[jb][2 * c4 + 1];
                        const float pv[4] = {lo_f(p01), hi_f(p01), lo_f(p23), hi_f(p23)};
                        const float gv[4] = {lo_f(g01), hi_f(g01), lo_f(g23), hi_f(g23)};
                        f32x4 dsv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) dsv[e] = fabsf(pv[e]) * (gv[e] - dtot) * P.scale;
                        const int key0 = 32 * jb + 8 * c4 + 4 * hf;
                        const unsigned s01 = pack2(dsv[0], dsv[1]), s23 = pack2(dsv[2], dsv[3]);
                        if (jb < NKB - 1 || key0 < kk) *reinterpret_cast<uint2*>(dsrow16 + key0) = uint2{s01, s23};
                        dpk[jb][2 * c4] = s01;
                        dpk[jb][2 * c4 + 1] = s23;
                        if constexpr (MODE == 2) {   // Pd = P o M: kept keys scaled, dropped keys (negative sign) -> 0
                            ppk[jb][2 * c4] = pack2(fmaxf(pv[0], 0.f) * P.drop.scale, fmaxf(pv[1], 0.f) * P.drop.scale);
                            ppk[jb][2 * c4 + 1] = pack2(fmaxf(pv[2], 0.f) * P.drop.scale, fmaxf(pv[3], 0.f) * P.drop.scale);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        } else {
        const bool vec_ok = (P.k & 3) == 0;
        static_for<0, NKB>([&](auto jb_t) __attribute__((always_inline)) {
            constexpr int jb = decltype(jb_t)::value;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const unsigned p01 = ppk[jb][2 * c4], p23 = ppk[jb][2 * c4 + 1];
                const unsigned g01 = dpk[jb][2 * c4], g23 = dpk[jb][2 * c4 + 1];
                const float pv[4] = {lo_f(p01), hi_f(p01), lo_f(p23), hi_f(p23)};
                const float gv[4] = {lo_f(g01), hi_f(g01), lo_f(g23), hi_f(g23)};
                f32x4 dsv;
#pragma unroll
                for (int e = 0; e < 4; ++e) dsv[e] = pv[e] * (gv[e] - dtot) * P.scale;
                const int key0 = 32 * jb + 8 * c4 + 4 * hf;
                const unsigned s01 = pack2(dsv[0], dsv[1]), s23 = pack2(dsv[2], dsv[3]);
                if (rvalid) {
                    if (P.ds_bf16) {
                        if (vec_ok) {
                            if (key0 < P.k) *reinterpret_cast<uint2*>(dsrow16 + key0) = uint2{s01, s23};
                        } else {
                            const unsigned short hv[4] = {(unsigned short)(s01 & 0xffffu), (unsigned short)(s01 >> 16),
                                                          (unsigned short)(s23 & 0xffffu), (unsigned short)(s23 >> 16)};
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (key0 + e < P.k) dsrow16[key0 + e] = hv[e];
                        }
                    } else if (vec_ok) {
                        if (key0 < P.k) *reinterpret_cast<f32x4*>(dsrow + key0) = dsv;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (key0 + e < P.k) dsrow[key0 + e] = dsv[e];
                    }
                }
                dpk[jb][2 * c4] = s01;
                dpk[jb][2 * c4 + 1] = s23;
                if (mrow) {
                    float mk[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) mk[e] = (key0 + e < P.k) ? mrow[key0 + e] : 0.f;
                    ppk[jb][2 * c4] = pack2(pv[0] * mk[0], pv[1] * mk[1]);
                    ppk[jb][2 * c4 + 1] = pack2(pv[2] * mk[2], pv[3] * mk[3]);
                } else if (P.drop.thresh) {   // the mask is regenerated (second use): 4 keys per Philox call
                    const snf::philox_f4 m4 = snf::dropout_mask4(P.drop, a, P.n, lrow, P.k, key0);
                    ppk[jb][2 * c4] = pack2(pv[0] * m4[0], pv[1] * m4[1]);
                    ppk[jb][2 * c4 + 1] = pack2(pv[2] * m4[2], pv[3] * m4[3]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        }

        // dV^T[col, row] += dO^T Pd^T ;  dQ^T[col, row] += Kp^T dS^T : k-step = 16 keys (registers 8u .. 8u+7 of block jb)
        f32x16 acc_v[NCB], acc_q[NCB];
#pragma unroll
        for (int db = 0; db < NCB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_v[db][r] = acc_q[db][r] = 0.f;
        static_for<0, NKB>([&](auto jb_t) __attribute__((always_inline)) {
            constexpr int jb = decltype(jb_t)::value;
            static_for<0, 2>([&](auto u_t) __attribute__((always_inline)) {
                constexpr int u = decltype(u_t)::value;
                const u32x4 pb = {ppk[jb][4 * u], ppk[jb][4 * u + 1], ppk[jb][4 * u + 2], ppk[jb][4 * u + 3]};
                const u32x4 sb = {dpk[jb][4 * u], dpk[jb][4 * u + 1], dpk[jb][4 * u + 2], dpk[jb][4 * u + 3]};
                constexpr int koff = (32 * jb + 16 * u) * RP;
                static_for<0, NCB>([&](auto db_t) __attribute__((always_inline)) {
                    constexpr int db = decltype(db_t)::value;
                    const bf16x8 ao = lds_tr_frag(img_do + rt0[db] + koff, img_do + rt1[db] + koff);
                    const bf16x8 ak = lds_tr_frag(img_kp + rt0[db] + koff, img_kp + rt1[db] + koff);
                    acc_v[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ao, __builtin_bit_cast(bf16x8, pb), acc_v[db], 0, 0, 0);
                    acc_q[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak, __builtin_bit_cast(bf16x8, sb), acc_q[db], 0, 0, 0);
                });
            });
        });
        // C layout of X^T[col, row]: lane = row, registers = columns 32 db + (r & 3) + 8 (r >> 2) + 4 hf -> 16-byte stores
        if (rvalid && !P.dqv_bf16) {
            float* dvp = reinterpret_cast<float*>(P.dv) + (int64_t)row * P.ldd + a * DK + 4 * hf;
            float* dqp = reinterpret_cast<float*>(P.dq) + (int64_t)row * P.ldd + a * DK + 4 * hf;
#pragma unroll
            for (int db = 0; db < NCB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    *reinterpret_cast<f32x4*>(dvp + 32 * db + 8 * g) =
                        f32x4{acc_v[db][4 * g], acc_v[db][4 * g + 1], acc_v[db][4 * g + 2], acc_v[db][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(dqp + 32 * db + 8 * g) =
                        f32x4{acc_q[db][4 * g], acc_q[db][4 * g + 1], acc_q[db][4 * g + 2], acc_q[db][4 * g + 3]};
                }
        }
        if (rvalid && P.dqv_bf16) {   // gradients leave as bf16 (operands of the weight-gradient GEMMs), 8-byte stores
            unsigned short* dvp = reinterpret_cast<unsigned short*>(P.dv) + (int64_t)row * P.ldd + a * DK + 4 * hf;
            unsigned short* dqp = reinterpret_cast<unsigned short*>(P.dq) + (int64_t)row * P.ldd + a * DK + 4 * hf;
#pragma unroll
            for (int db = 0; db < NCB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 pv, pq;
                    pv.x = pack_bf16x2(acc_v[db][4 * g], acc_v[db][4 * g + 1]);
                    pv.y = pack_bf16x2(acc_v[db][4 * g + 2], acc_v[db][4 * g + 3]);
                    pq.x = pack_bf16x2(acc_q[db][4 * g], acc_q[db][4 * g + 1]);
                    pq.y = pack_bf16x2(acc_q[db][4 * g + 2], acc_q[db][4 * g + 3]);
                    *reinterpret_cast<uint2*>(dvp + 32 * db + 8 * g) = pv;
                    *reinterpret_cast<uint2*>(dqp + 32 * db + 8 * g) = pq;
                }
        }
        if (++t == P.tiles_per_head) {
            t = 0;
            ++a;
        }
    }
}

struct BwdPlan {
    int num_wg, tiles_per_head, tiles_per_wg, total_tiles, nkb;
};
inline bool make_bwd_plan(int64_t n, int k, int h, int dk, BwdPlan* pl) {
    if (!((dk == 128 && k <= 224) || (dk == 64 && k <= 256)) || k < 1 || n < 1 || n > 0x7fffff00ll) return false;
    const int nkb = (k + 31) / 32;
    const int64_t tph = (n + TILE_ROWS - 1) / TILE_ROWS, total = tph * h;
    if (total > 0x7fffffff) return false;
    const int cus = snf::cu_count();
    int64_t num_wg = total < cus ? total : cus;
    const int64_t tpw = (total + num_wg - 1) / num_wg;
    num_wg = (total + tpw - 1) / tpw;
    *pl = {(int)num_wg, (int)tph, (int)tpw, (int)total, nkb};
    return true;
}

template <int DK, int NKB, typename QT, int MODE>
int launch_bwd_mode(const BwdParams& P, const BwdPlan& pl, hipStream_t s) {
    const size_t lds = (size_t)2 * 32 * NKB * 2 * DK;
    static thread_local unsigned long long attr_set_mask = 0;   // devices (bit = device id) that have the opt-in
    const unsigned long long attr_set_bit = snf::device_bit();
    const bool attr_set = (attr_set_mask & attr_set_bit) != 0;
    auto kern = sparse_attn_bwd_mfma_kernel<DK, NKB, QT, MODE>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            snf::set_error("sparse_attn_bwd_mfma: cannot reserve %zu bytes of LDS", lds);
            (void)hipGetLastError();
            return SNF_ELAUNCH;
        }
        attr_set_mask |= attr_set_bit;
    }
    hipLaunchKernelGGL(kern, dim3(pl.num_wg), dim3(256), lds, s, P);
    return snf::check_launch("sparse_attn_bwd_mfma_kernel");
}
template <int DK, int NKB, typename QT>
int launch_bwd(const BwdParams& P, const BwdPlan& pl, hipStream_t s) {
    if constexpr (std::is_same<QT, unsigned short>::value) {   // the training path: bf16 operands, bf16 dS, K % 4 == 0
        if (P.ds_bf16 && !P.mask && (P.k & 3) == 0)
            return P.drop.thresh ? launch_bwd_mode<DK, NKB, QT, 2>(P, pl, s) : launch_bwd_mode<DK, NKB, QT, 1>(P, pl, s);
    }
    return launch_bwd_mode<DK, NKB, QT, 0>(P, pl, s);
}
template <int DK, typename QT>
int launch_bwd_nkb(const BwdParams& P, const BwdPlan& pl, hipStream_t s) {
    switch (pl.nkb) {
        case 1: return launch_bwd<DK, 1, QT>(P, pl, s);
        case 2: return launch_bwd<DK, 2, QT>(P, pl, s);
        case 3: return launch_bwd<DK, 3, QT>(P, pl, s);
        case 4: return launch_bwd<DK, 4, QT>(P, pl, s);
        case 5: return launch_bwd<DK, 5, QT>(P, pl, s);
        case 6: return launch_bwd<DK, 6, QT>(P, pl, s);
        case 7: return launch_bwd<DK, 7, QT>(P, pl, s);
        default:
            if constexpr (DK == 64) return launch_bwd<DK, 8, QT>(P, pl, s);
            snf::set_error("sparse_attn_bwd_mfma: key-block count %d not built", pl.nkb);
            return SNF_EUNSUPPORTED;
    }
}

}  // namespace

extern "C" {

int snf_sparse_attn_bwd_mfma(const void* q, int64_t ldq, const void* v, int64_t ldv, int qv_dtype, const float* kp,
                             const float* dout, const float* lse, const float* mask, int64_t n, int k, int h, int dk,
                             float scale, float* dq, float* dv, void* ds, int ds_dtype, snf_stream_t stream) {
    return snf_sparse_attn_bwd_mfma_dropout(q, ldq, v, ldv, qv_dtype, kp, dout, lse, mask, 0.f, 0, 0, n, k, h, dk, scale, dq, dv, ds,
                                            ds_dtype, stream);
}

int snf_sparse_attn_bwd_mfma_dropout(const void* q, int64_t ldq, const void* v, int64_t ldv, int qv_dtype, const float* kp,
                                     const float* dout, const float* lse, const float* mask, float dropout_p, uint64_t seed,
                                     uint64_t offset, int64_t n, int k, int h, int dk, float scale, float* dq, float* dv,
                                     void* ds, int ds_dtype, snf_stream_t stream) {
    return snf_sparse_attn_bwd_mfma_ex(q, ldq, v, ldv, qv_dtype, kp, dout, lse, mask, dropout_p, seed, offset, n, k, h, dk, scale,
                                       dq, dv, (int64_t)h * dk, SNF_DT_F32, ds, ds_dtype, stream);
}

int snf_sparse_attn_bwd_mfma_ex(const void* q, int64_t ldq, const void* v, int64_t ldv, int qv_dtype, const float* kp,
                                const float* dout, const float* lse, const float* mask, float dropout_p, uint64_t seed,
                                uint64_t offset, int64_t n, int k, int h, int dk, float scale, void* dq, void* dv, int64_t ldd,
                                int dqv_dtype, void* ds, int ds_dtype, snf_stream_t stream) {
    SNF_REQUIRE(q && v && kp && dout && lse && dq && dv && ds, "snf_sparse_attn_bwd_mfma: null pointer");
    SNF_REQUIRE(dqv_dtype == SNF_DT_F32 || dqv_dtype == SNF_DT_BF16, "snf_sparse_attn_bwd_mfma: bad dq / dv dtype %d", dqv_dtype);
    SNF_REQUIRE(ldd >= (int64_t)h * dk && ldd % (dqv_dtype == SNF_DT_BF16 ? 8 : 4) == 0,
                "snf_sparse_attn_bwd_mfma: ldd=%lld must be >= h*dk and keep rows 16-byte aligned", (long long)ldd);
    SNF_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "snf_sparse_attn_bwd_mfma: dropout_p=%f outside [0, 1)", dropout_p);
    SNF_REQUIRE(!(mask && dropout_p > 0.f), "snf_sparse_attn_bwd_mfma: pass a mask tensor OR (dropout_p, seed, offset), not both");
    SNF_REQUIRE(qv_dtype == SNF_DT_F32 || qv_dtype == SNF_DT_BF16, "snf_sparse_attn_bwd_mfma: bad dtype %d", qv_dtype);
    SNF_REQUIRE(ds_dtype == SNF_DT_F32 || ds_dtype == SNF_DT_BF16, "snf_sparse_attn_bwd_mfma: bad ds dtype %d", ds_dtype);
    BwdPlan pl;
    if (!make_bwd_plan(n, k, h, dk, &pl)) {
        snf::set_error("snf_sparse_attn_bwd_mfma: unsupported shape k=%d dk=%d (need dk == 128 with k <= 224 or dk == 64 "
                       "with k <= 256)", k, dk);
        return SNF_EUNSUPPORTED;
    }
    const int64_t d = (int64_t)h * dk;
    const int align = qv_dtype == SNF_DT_BF16 ? 8 : 4;
    SNF_REQUIRE(ldq >= d && ldv >= d && (ldq % align) == 0 && (ldv % align) == 0,
                "snf_sparse_attn_bwd_mfma: ldq=%lld / ldv=%lld must be >= h*dk and keep rows 16-byte aligned", (long long)ldq,
                (long long)ldv);
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    SNF_REQUIRE(al16(q) && al16(v) && al16(kp) && al16(dout) && al16(dq) && al16(dv) && al16(ds),
                "snf_sparse_attn_bwd_mfma: buffers must be 16-byte aligned");
    BwdParams P;
    P.q = q;
    P.v = v;
    P.kp = kp;
    P.dout = dout;
    P.lse = lse;
    P.mask = mask;
    P.drop = snf::make_dropout(dropout_p, seed, offset);
    P.n = n;
    P.ldq = ldq;
    P.ldv = ldv;
    P.d = d;
    P.k = k;
    P.h = h;
    P.scale = scale;
    P.dq = dq;
    P.dv = dv;
    P.ldd = ldd;
    P.dqv_bf16 = dqv_dtype == SNF_DT_BF16;
    P.ds = ds;
    P.ds_bf16 = ds_dtype == SNF_DT_BF16;
    P.tiles_per_head = pl.tiles_per_head;
    P.tiles_per_wg = pl.tiles_per_wg;
    P.total_tiles = pl.total_tiles;
    hipStream_t s = snf::as_stream(stream);
    if (dk == 128)
        return qv_dtype == SNF_DT_F32 ? launch_bwd_nkb<128, float>(P, pl, s) : launch_bwd_nkb<128, unsigned short>(P, pl, s);
    return qv_dtype == SNF_DT_F32 ? launch_bwd_nkb<64, float>(P, pl, s) : launch_bwd_nkb<64, unsigned short>(P, pl, s);
}

}  // extern "C"
