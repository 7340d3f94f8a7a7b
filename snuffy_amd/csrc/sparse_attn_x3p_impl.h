// K7, fp32-class: the pipelined split-bf16 x3 sparse attention on PRE-SPLIT operands (rounds 4-5).  Kernel templates; the translation
// units sparse_attn_x3p.hip (dk = 128, one key block per wave; host API), sparse_attn_x3p_k2.hip (dk = 128, two key blocks per wave)
// and sparse_attn_x3p_dk64.hip (dk = 64) instantiate them.
//
//   per head a:   P_a = softmax_j(Q_a Kp_a^T * scale)  [n, k]      O_a = P_a^T V_a  [k, dk]        (snuffy.py:160-168)
//
// Organisation (round 4; DESIGN.md section 4 has the measurements):
//   * operands arrive PRE-SPLIT: Q and V as the interleaved "hl" images the projection GEMM writes in its epilogue (every 32 true
//     columns as [hi(32) | lo(32)] bf16, gemm.hip) -- the same 4 bytes per element as the fp32 tensors.  Rows go HBM -> LDS by
//     LDS-DMA in full lines per row and head (no staging registers, no split, no ds_write), three tiles ahead; Kp arrives as the
//     MFMA fragment image (key projection epilogue or x3p_prep_kp_kernel), scaled by scale * log2(e).
//   * a wave OWNS KBW key blocks (32 keys each) for everything -- their Kp fragments (registers, whole head), the scores
//     S^T[32 keys, 32 rows] of every tile, their softmax, and the 32 x dk slices of the output accumulator.  P^T of a key block is
//     produced and consumed by the same wave: it crosses a wave-private LDS region only to change from the accumulator layout
//     (lane = row) to the A-operand layout (lane = key), with no barrier.  The only cross-wave exchange per tile is one
//     (max, sum) pair per row and wave.
//   * 32-row tiles, ONE workgroup barrier per tile, two-stage software pipeline.  Iteration i issues, in one instruction stream,
//         first half:   GEMM1(i + 1) on the matrix pipe  |  combine + normalise + split + publish P(i) on the vector ALU
//         second half:  GEMM2(i)     on the matrix pipe  |  max / exp2 / sum of S(i + 1), its statistics, the LDS-DMA issue
//     with the vector work cut into small units that are dropped, by hand, into the gaps behind the MFMAs.
//   * KBW = 1 (round 4): ceil(k / 32) waves of <= 256 registers, two per SIMD.  Every wave reads the whole Q tile and the whole V
//     tile out of LDS for 48 MFMAs.
//     KBW = 2 (round 5): ceil(k / 64) waves, ONE per SIMD, 512 registers (accumulators and Kp fragments in the AGPR half): a Q or V
//     fragment read feeds the MFMAs of two key blocks, i.e. half the LDS reads, half the per-tile fixed work (statistics combine,
//     barrier, addressing) per MFMA, and no issue-port sharing between two waves of a SIMD.  The MFMAs of the two key blocks
//     alternate, so that no MFMA waits for its predecessor's accumulator.
//   * dk = 64 (round 5): the same program on 256-byte rows (config A, the D = 384 column of the sweep).
//
// LDS (dk = 128): Q ring 3 x 17 KiB (rows padded to 528 B: conflict-free B reads at immediate offsets) | V ring 3 x 16 KiB |
// P 4 KiB per key block (hi 2 KiB + lo 2 KiB) | statistics 2 x waves x 256 B | DMA offset table.
// Partial accumulators [workgroup][segment][key block][column block][4][64][4] fp32, summed in fixed order by x3p_reduce_kernel.
#pragma once
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace snf {
extern unsigned long long* g_attn_trace;   // debug hook of sparse_attn_mfma.hip (snf_debug_attn_trace)
extern int g_attn_trace_wg;

namespace x3p {
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

struct X3PParams {
    const unsigned short* q;   // hl image [n, ldq] bf16: head a, true column c at 2 a dk + 64 (c / 32) + c % 32 (hi), + 32 (lo)
    const unsigned short* v;   // hl image [n, ldv]
    const float* kp;           // [k, ldkp] f32
    const u32x4* kp_frag;      // workspace: the fragment-ordered split image of kp (x3p_prep_kp_kernel)
    int64_t n, ldq, ldv, ldkp;
    int k, h;
    float scale;
    float* attn;               // [h, n, attn_ld] (already offset to this launch's first key) or null
    int64_t attn_ld;
    float* lse;                // [h, n] or null
    f32x2* stats;              // key-chunked launches: [nchunks][h][n] (row max in scaled log2 units, row sum) per chunk.  MODE 1 writes
                               // chunk `chunk`'s pair per row, MODE 2 reads all chunks' pairs (softmax exact over all keys)
    int nchunks, chunk;
    // merged key-chunk launches (K > 256, every chunk the same number of key blocks): ONE grid holds the workgroups of all `merged`
    // chunks, co-resident; workgroup b serves chunk (b / 8) % merged of row range (b / (8 merged)) * 8 + b % 8, so that the workgroups
    // that stream the same Q / V rows sit on the same XCD (b % 8), start together and share those rows through its L2.  k is then
    // the TOTAL key count, kp_frag / partial / attn the first chunk's; chunk c starts at key c * chunk_size.
    int merged, chunk_size;
    int64_t kpfrag_stride, partial_stride;   // per chunk, in u32x4 / float units
    float* partial;            // [num_wg * seg_count][NKB * dk / 32 tiles][4][64][4]
    int tiles_per_head, tiles_per_wg, total_tiles, seg_count;
    unsigned long long* trace;   // dev builds (X3P_TRACE): s_memtime stamps of workgroup trace_wg, [wave][64 iterations][8]
    int trace_wg;
};

struct X3PPlan {
    int num_wg, tiles_per_head, tiles_per_wg, total_tiles, seg_count, nkb;
};

// per translation unit: launch the kernel set of one (dk, key blocks per wave) family.  mode 0: all keys in one launch; 1: the
// statistics pass of a key chunk; 2: its main pass (see the kernel).  Returns SNF_EUNSUPPORTED for a key-block count not built.
int run_dk128_k1(const X3PParams& P, const X3PPlan& pl, float* out, hipStream_t s, int mode);
int run_dk128_k2(const X3PParams& P, const X3PPlan& pl, float* out, hipStream_t s, int mode);
int run_dk64_k1(const X3PParams& P, const X3PPlan& pl, float* out, hipStream_t s, int mode);
}  // namespace x3p
}  // namespace snf

namespace {

using snf::x3p::X3PParams;
using snf::x3p::X3PPlan;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int TR = 32;       // query rows per tile

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ f32x8 load8(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ void split8(const f32x8 x, u32x4& hi, u32x4& lo) {
    const bf16x8 h = __builtin_convertvector(x, bf16x8);
    const f32x8 r = x - __builtin_convertvector(h, f32x8);
    hi = __builtin_bit_cast(u32x4, h);
    lo = __builtin_bit_cast(u32x4, __builtin_convertvector(r, bf16x8));
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
// timing ablations (dev builds, tools/x3p_abl.sh; results wrong): a stage's MFMAs replaced by an opaque use of their operands
__device__ __forceinline__ f32x16 mfma_off(bf16x8 a, bf16x8 b, f32x16 c) {
    asm volatile("" : "+v"(c) : "v"(a), "v"(b));
    return c;
}
#ifdef X3P_ABL_NO_G1
#define X3P_MFMA1 mfma_off
#else
#define X3P_MFMA1 mfma
#endif
#ifdef X3P_ABL_NO_G2
#define X3P_MFMA2 mfma_off
#else
#define X3P_MFMA2 mfma
#endif
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
#define X3P_FENCE() __builtin_amdgcn_sched_barrier(0)
// s_waitcnt vmcnt(0) in the form hipcc's wait-count pass understands (vmcnt = 0, expcnt / lgkmcnt untouched): the LDS-DMAs are
// inline asm and invisible to that pass, so it must also be told when ITS OWN loads (spill reloads of the slow paths) are done --
// otherwise it parks a vmcnt(0) for them inside the fast loop, where it drains the prefetch every iteration
#define X3P_WAIT_VM0()                          \
    do {                                        \
        __builtin_amdgcn_s_waitcnt(0x0F70);     \
        asm volatile("" ::: "memory");          \
    } while (0)

#ifdef X3P_TRACE
#define X3P_STAMP(it, k)                                                                                          \
    do {                                                                                                          \
        if (P.trace && bid == P.trace_wg && lane == 0 && (it) >= 0 && (it) < 64)                                  \
            P.trace[(w * 64 + (it)) * 8 + (k)] = __builtin_amdgcn_s_memtime();                                    \
    } while (0)
// finer: stamp idx (0 .. 15) inside iteration `it`, second table behind the first
#define X3P_STAMP2(it, idx)                                                                                       \
    do {                                                                                                          \
        if (P.trace && bid == P.trace_wg && lane == 0 && (it) >= 0 && (it) < 64)                                  \
            P.trace[8 * 64 * 8 + (w * 64 + (it)) * 16 + (idx)] = __builtin_amdgcn_s_memtime();                    \
    } while (0)
#else
#define X3P_STAMP(it, k) do { } while (0)
#define X3P_STAMP2(it, idx) do { } while (0)
#endif

// (head, tile) cursor over the flattened item space of one workgroup
struct Cur {
    int a, t;
};

constexpr int x3p_qslot(int dk) { return ((TR * (4 * dk + 16) + 1023) / 1024) * 1024; }   // padded Q slot, whole DMA instructions
constexpr int x3p_dma_u(int n, int nw) { return (n + nw - 1) / nw; }
// Waves.  KBW key blocks per wave -> NW = ceil(nkb / KBW) computing waves.
// Who issues the LDS-DMA.  Waves w and w + 4 share a SIMD.
//   KBW = 1: with 5 .. 7 key blocks some SIMDs carry two waves and set the iteration time while wave slot nkb is free on a SIMD that
//            carries one: a LOADER wave sits there and issues all the DMA of the workgroup (nothing else: no MFMA, a few dozen
//            registers), and the key-block waves carry no DMA code at all.  With 4 or 8 key blocks every wave issues its share.
//   KBW = 2: one wave per SIMD.  3 computing waves (5, 6 key blocks): the LOADER wave takes the free SIMD.  7 key blocks: the last
//            wave carries ONE key block -- half the work of the others -- and issues all the DMA (the LIGHT wave).  8: every wave its share.
constexpr int x3p_nw(int nkb, int kbw) { return (nkb + kbw - 1) / kbw; }
constexpr bool x3p_loader(int nkb, int kbw) { return kbw == 1 ? (nkb > 4 && nkb < 8) : x3p_nw(nkb, kbw) < 4; }
constexpr int x3p_light(int nkb, int kbw) { return (kbw == 2 && (nkb & 1) && !x3p_loader(nkb, kbw)) ? x3p_nw(nkb, kbw) - 1 : -1; }
constexpr int x3p_issuers(int nkb, int kbw) { return (x3p_loader(nkb, kbw) || x3p_light(nkb, kbw) >= 0) ? 1 : x3p_nw(nkb, kbw); }
constexpr int x3p_waves(int nkb, int kbw) { return x3p_nw(nkb, kbw) + (x3p_loader(nkb, kbw) ? 1 : 0); }
constexpr int x3p_lds_bytes(int dk, int nkb, int kbw) {
    const int nq = x3p_qslot(dk) / 1024, nv = TR * 4 * dk / 1024, nl = x3p_issuers(nkb, kbw);
    // (offset table of the issuing waves; the statistics pass of an all-issue configuration has one issuer fewer -- see the kernel)
    const int t_all = nl * (x3p_dma_u(nq, nl) + x3p_dma_u(nv, nl)) * 256;
    const int t_m1 = nl > 1 ? (nl - 1) * (x3p_dma_u(nq, nl - 1) + x3p_dma_u(nv, nl - 1)) * 256 : 0;
    return 3 * x3p_qslot(dk) + 3 * TR * 4 * dk + nkb * (TR * 64 * 2) + 2 * x3p_nw(nkb, kbw) * TR * 8 +
           (x3p_loader(nkb, kbw) ? 0 : (t_all > t_m1 ? t_all : t_m1));
}

// MODE 0: all keys in this launch.  MODE 1: statistics pass of one key chunk (GEMM1 + max / sum per row, written to P.stats; no
// GEMM2, V is not touched).  MODE 2: main pass of one key chunk with the row statistics of ALL chunks taken from P.stats.
template <int DK, int NKB, int KBW, bool AUX, int MODE>
__global__ __launch_bounds__(64 * x3p_waves(NKB, KBW), KBW == 2 ? 1 : 2) void sparse_attn_x3p_kernel(const X3PParams P) {
    static_assert((DK == 128 || DK == 64) && (KBW == 1 || KBW == 2) && MODE >= 0 && MODE <= 2, "sparse_attn_x3p: dk = 64 / 128, 1 or 2 key blocks per wave");
    // Counted DMA waits need every other vector-memory operation of the wave to be ordered with the DMA loads.  Global STORES are not
    // (AUX: the attention matrix; MODE 1: the statistics pairs): those builds drain.  The statistics LOADS of MODE 2 are loads like
    // the DMA -- returned in order, and waited for by the compiler before their use in the first half of the iteration, long before
    // the counted wait at its end -- so the main pass of a key-chunked launch keeps its prefetch depth.
    // A statistics pass stores from wave 0 only: in the all-issue configurations that wave issues no DMA (the others share its part),
    // and where a loader / light wave exists the other waves issue none anyway.
    constexpr bool DRAIN = AUX;
    constexpr int NW = x3p_nw(NKB, KBW);     // computing waves
    constexpr int RLAST = NKB - KBW * (NW - 1);   // key blocks of the last computing wave (1 .. KBW)
    constexpr int NKS = DK / 16;             // k-steps of GEMM1
    constexpr int NCB = DK / 32;             // 32-column blocks of the output
    constexpr int ROWB = 4 * DK;             // bytes of one row of one head in an hl image
    constexpr int QP = ROWB + 16;            // row pitch of a Q slot: one 16-byte pad chunk per row (bank rotation of the B reads)
    constexpr int QCH = QP / 16;             // 16-byte positions per Q row (33 / 17)
    constexpr int QSLOT = x3p_qslot(DK);     // 17 KiB / 9 KiB
    constexpr int NQDMA = QSLOT / 1024;      // LDS-DMA instructions of a Q tile (17 / 9)
    constexpr int VSLOT = TR * ROWB;         // 16 KiB / 8 KiB, rows of [hi plane | lo plane], 64-byte segments XOR-rotated by row
    constexpr int NVDMA = VSLOT / 1024;      // 16 / 8
    constexpr bool LOADER = x3p_loader(NKB, KBW);                           // wave NW issues all the LDS-DMA and nothing else
    constexpr int LIGHT = x3p_light(NKB, KBW);                              // >= 0: this computing wave issues all the LDS-DMA
    constexpr int L0 = LOADER ? NW : LIGHT >= 0 ? LIGHT : (MODE == 1 ? 1 : 0);   // first issuing wave
    constexpr int NL = (LOADER || LIGHT >= 0) ? 1 : NW - L0;               // issuing waves L0 .. L0 + NL - 1
    static_assert(NL >= 1, "sparse_attn_x3p: no wave left to issue the DMA");
    constexpr int UQ = x3p_dma_u(NQDMA, NL), UV = x3p_dma_u(NVDMA, NL);   // DMA instructions per issuing wave and tile, at most
    static_assert(UQ + UV <= 40, "DMA instructions per wave: 64-bit masks, counted waits up to 40");
    constexpr int Q_OFF = 0, V_OFF = 3 * QSLOT, P_OFF = V_OFF + 3 * VSLOT;
    constexpr int PBUF = TR * 64 * 2;        // the P image of one key block: hi plane 2 KiB | lo plane 2 KiB
    constexpr int ST_OFF = P_OFF + NKB * PBUF;
    constexpr int TB_OFF = ST_OFF + 2 * NW * TR * 8;        // per-lane DMA source offsets, [NL][UQ + UV][64] ints
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, hf = lane >> 5;
    const int n32 = (int)P.n;
    int bid = blockIdx.x;
    // key chunk of this workgroup (wave-uniform, from the block index only)
    int chunk = P.chunk, kk = P.k;
    const u32x4* kp_frag = P.kp_frag;
    float* partial = P.partial;
    float* attn_base = P.attn;
    if constexpr (MODE != 0) {
        if (P.merged > 1) {
            const int slot = bid >> 3;
            chunk = slot % P.merged;
            bid = (slot / P.merged) * 8 + (bid & 7);
            const int k0 = chunk * P.chunk_size;
            kk = P.k - k0 < P.chunk_size ? P.k - k0 : P.chunk_size;
            kp_frag += chunk * P.kpfrag_stride;
            partial += chunk * P.partial_stride;
            if (attn_base) attn_base += k0;
        }
    }

    const int f_begin = bid * P.tiles_per_wg;
    int f_end = f_begin + P.tiles_per_wg;
    if (f_end > P.total_tiles) f_end = P.total_tiles;
    if (f_begin >= f_end) return;
    const int T = f_end - f_begin;            // items of this workgroup
    const int first_head = f_begin / P.tiles_per_head;
    // cursors of the items of the pipeline (head, tile), advanced by one item per iteration: no division in the loop
    auto cur_next = [&](Cur c) __attribute__((always_inline)) -> Cur {
        ++c.t;
        if (c.t == P.tiles_per_head) c.t = 0, ++c.a;
        return c;
    };

    // ---------------------------------------------------------------- LDS-DMA of one tile's Q or V rows
    // An instruction fills 1 KiB = 64 lanes x 16 B of its slot, lane-linear; which 16 bytes of the tile a lane FETCHES is free.
    //   Q instruction e: position p = 64 e + lane -> row p / QCH, chunk p % QCH of the row's line (the last chunk = pad)
    //   V instruction e: 1024 / ROWB rows; a row is 2 NCB segments of 64 bytes, logical segment NCB * plane + column block; position
    //                    segment ps of row r holds logical segment ps ^ (r & 3)
    //                    (keeps the four rows of a transpose-read in four different 64-byte bank segments)
    // Every instruction reads whole lines of consecutive rows.  Issuing wave li (= w - L0) issues Q instructions li, li + NL, ..
    // and V instructions li, li + NL, ..
    const int ldq_b = (int)(P.ldq * 2), ldv_b = (int)(P.ldv * 2);
    auto q_rc = [&](int e, int& row, int& chunk) __attribute__((always_inline)) {
        const int p = 64 * e + lane;
        row = p / QCH;
        chunk = p - row * QCH;
        if (chunk > ROWB / 16 - 1) chunk = ROWB / 16 - 1;
        if (row > TR - 1) row = TR - 1;
    };
    auto v_rc = [&](int e, int& row, int& chunk) __attribute__((always_inline)) {
        constexpr int LPR = ROWB / 16;                       // lanes (16-byte positions) per row
        const int s = lane % LPR;
        row = (64 / LPR) * e + lane / LPR;
        const int seg = (s >> 2) ^ (row & 3);                // position segment s >> 2 holds logical segment seg
        chunk = 8 * (seg % NCB) + 4 * (seg / NCB) + (s & 3);
    };
    // The per-lane source offsets of a FULL tile (row_in_tile * ld_bytes + 16 * chunk) are parked in LDS: registers are the scarce
    // resource of this kernel, and an offset is needed once per iteration.
    const bool issuer = w >= L0 && w < L0 + NL;
    const int li = issuer ? w - L0 : 0;
    int* const dma_tab = reinterpret_cast<int*>(smem + TB_OFF) + li * ((UQ + UV) * 64) + lane;
    if (!LOADER && issuer) {
#pragma unroll
        for (int u = 0; u < UQ; ++u) {
            int row, chunk;
            q_rc(li + NL * u, row, chunk);
            dma_tab[u * 64] = row * ldq_b + 16 * chunk;
        }
#pragma unroll
        for (int u = 0; u < UV; ++u) {
            int row, chunk;
            v_rc(li + NL * u, row, chunk);
            dma_tab[(UQ + u) * 64] = row * ldv_b + 16 * chunk;
        }
    }
    // one instruction: 32-bit per-lane offset + 64-bit wave-uniform base (SGPR pair) -> 1 KiB at the wave-uniform LDS address dst.
    // Hand-written: behind the builtin hipcc puts s_waitcnt vmcnt(0) in front of the first LDS read it cannot prove disjoint from
    // the DMA's destination (the transpose-reads of the OTHER ring slots), which drains the prefetch every iteration.  The asm form
    // is invisible to that pass; the kernel's own counted s_waitcnt vmcnt + s_barrier at the end of an iteration order the data
    // (cdna_hip_programming.md 5.7: M0 is written in the statement that uses it, and restored).
    auto dma_1k = [&](const unsigned char* base, int off, int dst) __attribute__((always_inline)) {
#ifdef X3P_ABL_NODMA   // timing ablation: no operand traffic at all (results wrong)
        asm volatile("" ::"v"(off), "s"(base), "s"(dst));
#else
        unsigned keep;
        // (readfirstlane: the operands ARE wave-uniform, but the asm "s" constraint needs the compiler to know it in every instantiation)
        const uint64_t b64 = reinterpret_cast<uint64_t>(base);
        const uint64_t bu = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(b64 >> 32)) << 32) |
                            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(off), "s"(bu), "s"(__builtin_amdgcn_readfirstlane(dst))
                     : "memory");
#endif
    };
    struct DmaCtx {                           // wave-uniform, per tile pair to fetch
        const unsigned char *bq, *bv;
        uint64_t mask;                        // bit u: instruction u of the wave's list (Q: u < UQ, V: UQ + ..) is to be issued
        int dstq, dstv;
    };
    uint64_t full_mask = 0;                   // the instructions this wave owns
#pragma unroll
    for (int u = 0; u < UQ; ++u) full_mask |= (issuer && li + NL * u < NQDMA) ? 1ull << u : 0ull;
#pragma unroll
    for (int u = 0; u < UV; ++u) full_mask |= (issuer && li + NL * u < NVDMA) ? 1ull << (UQ + u) : 0ull;
    constexpr uint64_t QBITS = (1ull << UQ) - 1ull, VBITS = ((1ull << UV) - 1ull) << UQ;
    const unsigned tile_q_b = (unsigned)(TR * ldq_b), tile_v_b = (unsigned)(TR * ldv_b);
    // instruction u of the wave's list, full tile (the common case: nothing but the table lookup and the instruction)
    auto dma_unit = [&](auto u_t, const DmaCtx& dc, int off) __attribute__((always_inline)) {
        constexpr int u = decltype(u_t)::value;
        constexpr bool isq = u < UQ;
        constexpr int uu = isq ? u : u - UQ;
        if (dc.mask & (1ull << u)) dma_1k(isq ? dc.bq : dc.bv, off, (isq ? dc.dstq : dc.dstv) + NL * uu * 1024);
    };
    // the same for the last tile of a bag: rows past the end re-read the last row (their P is forced to 0).  Rare: not interleaved.
    auto dma_partial = [&](const DmaCtx& dc, uint64_t mask, int rmaxq, int rmaxv) __attribute__((always_inline)) {
        static_for<0, UQ + UV>([&](auto u_t) __attribute__((always_inline)) {
            constexpr int u = decltype(u_t)::value;
            constexpr bool isq = u < UQ;
            constexpr int uu = isq ? u : u - UQ;
            if (mask & (1ull << u)) {
                int row, chunk;
                if constexpr (isq) q_rc(li + NL * uu, row, chunk); else v_rc(li + NL * uu, row, chunk);
                const int rmax = isq ? rmaxq : rmaxv;
                if (row > rmax) row = rmax;
                dma_1k(isq ? dc.bq : dc.bv, row * (isq ? ldq_b : ldv_b) + 16 * chunk, (isq ? dc.dstq : dc.dstv) + NL * uu * 1024);
            }
        });
    };
    // Q rows of item cq -> Q slot sq (if doq), V rows of item cv -> V slot sv (if dov).  Partial tiles are issued here, at once.
    auto dma_ctx = [&](bool doq, Cur cq, int sq, bool dov, Cur cv, int sv) __attribute__((always_inline)) -> DmaCtx {
        DmaCtx dc;
        dc.bq = reinterpret_cast<const unsigned char*>(P.q) + (uint64_t)((unsigned)cq.t * (uint64_t)tile_q_b) + (unsigned)(cq.a * ROWB);
        dc.bv = reinterpret_cast<const unsigned char*>(P.v) + (uint64_t)((unsigned)cv.t * (uint64_t)tile_v_b) + (unsigned)(cv.a * ROWB);
        dc.dstq = Q_OFF + sq * QSLOT + li * 1024, dc.dstv = V_OFF + sv * VSLOT + li * 1024;
        dc.mask = full_mask & ((doq ? QBITS : 0ull) | (dov ? VBITS : 0ull));
        const int rmaxq = n32 - 1 - cq.t * TR, rmaxv = n32 - 1 - cv.t * TR;   // last existing row, tile-relative
        uint64_t part = 0;
        if (rmaxq < TR - 1) part |= QBITS;
        if (rmaxv < TR - 1) part |= VBITS;
        part &= dc.mask;
        if (part) {
            dma_partial(dc, part, rmaxq, rmaxv);
            dc.mask &= ~part;
        }
        return dc;
    };
    // NOTE: instructions issued by dma_ctx itself (partial tiles) are counted by the caller through popcount(issued) below

    // ---------------------------------------------------------------- addressing (one register per stream where possible)
    // GEMM1 B fragment (kb, lo) of lane (row j, half hf): chunk 8 (kb >> 1) + 4 lo + 2 (kb & 1) + hf of row j -- an immediate offset
    const int q_lane = Q_OFF + j * QP + 16 * hf;
    // GEMM2 transpose-reads (lane = 16-lane group g x index i, as sparse_attn_x3.hip): rows rr0 / rr0 + 4 of a 16-row k-step
    const int rg = lane >> 4, ri = lane & 15;
    const int rr0 = 8 * (rg >> 1) + (ri >> 2), rr1 = rr0 + 4;
    const int rch = 4 * (rg & 1) + (ri & 3);
    const int p_wave = P_OFF + (KBW * w) * PBUF;                         // this wave's P images (one per key block, consecutive)
    const int poff0 = p_wave + rr0 * 64 + 8 * (rch ^ ((rr0 >> 1) & 7));
    const int poff1 = p_wave + rr1 * 64 + 8 * (rch ^ ((rr1 >> 1) & 7));
    // V fragment of column block cb, plane lo, row rr0 (rr1: + 4 rows): 64-byte segment (NCB lo + cb) ^ (rr0 & 3) -> voff0 ^ (64 (NCB lo + cb))
    const int voff0 = V_OFF + rr0 * ROWB + 64 * (rr0 & 3) + 32 * (rg & 1) + 8 * (ri & 3);
    // P image writer: row j, 8-byte chunk (2 c4 + hf) of the 64-byte row at position ^ ((j >> 1) & 7) -> waddr0 ^ (16 c4)
    const int waddr0 = p_wave + j * 64 + 8 * (hf ^ ((j >> 1) & 7));
    const int st_lane = ST_OFF + 8 * j;                                  // statistics [2][NW][TR] of (max, sum)
    const bool attn_vec = AUX && (P.attn_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(attn_base) & 15) == 0;

    // fragment reads
    auto q_frag = [&](int qa, int kb, int lo) __attribute__((always_inline)) -> bf16x8 {
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smem + qa + 16 * (8 * (kb >> 1) + 4 * lo + 2 * (kb & 1))));
    };
    // (the plane bit 64 NCB lo is XORed like the column block: at dk = 128 it lies above the rotated bits and the XOR is an addition
    //  that folds into the read's immediate offset; at dk = 64 it is one of the two rotated bits)
    auto v_frag = [&](int va, int sk, int lo) __attribute__((always_inline)) -> bf16x8 {
        const unsigned char* vp = DK == 128 ? smem + va + sk * 16 * ROWB + 64 * NCB * lo : smem + (va ^ (64 * NCB * lo)) + sk * 16 * ROWB;
        return tr_frag(vp, vp + 4 * ROWB);
    };
    auto p_frag = [&](int r, int sk, int lo) __attribute__((always_inline)) -> bf16x8 {
        const int o = r * PBUF + lo * (PBUF / 2) + sk * 16 * 64;
        return tr_frag(smem + poff0 + o, smem + poff1 + o);
    };

    // ================================================================ the wave program
    // R: key blocks of this wave (KBW, the last wave: RLAST).  LASTW: its last key block is the launch's last (padded keys are masked).
    // ISS: this wave issues LDS-DMA (the others carry no DMA code at all)
    auto run = [&](auto r_t, auto lastw_t, auto iss_t) __attribute__((always_inline)) {
        constexpr int R = decltype(r_t)::value;
        constexpr bool LASTW = decltype(lastw_t)::value, ISS = decltype(iss_t)::value;
        const int gkb0 = KBW * w;                                       // the wave's first key block

        // ---- Kp fragments of the wave's key blocks: MFMA A operands, hi and lo, for the whole head -- 16-byte loads out of the
        // fragment-ordered image (scaled and split by its producer: nothing but the loads happens here, so a head change inside a
        // workgroup's range costs one L2 round trip)
        bf16x8 kph[R][NKS], kpl[R][NKS];
        auto load_kp = [&](int a_) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const u32x4* src = kp_frag + ((int64_t)(a_ * NKB + gkb0 + r) * NKS * 2) * 64 + lane;
#pragma unroll
                for (int kb = 0; kb < NKS; ++kb) {
                    kph[r][kb] = __builtin_bit_cast(bf16x8, src[(2 * kb) * 64]);
                    kpl[r][kb] = __builtin_bit_cast(bf16x8, src[(2 * kb + 1) * 64]);
                }
            }
        };
        // ---- state
#ifndef X3P_NACC
#define X3P_NACC 1
#endif
        // score accumulators of GEMM1 (tile i + 1): NACC per key block, the k-steps' MFMAs alternating between them (summed in the
        // statistics units).  Dev knob: 2 (no MFMA behind its predecessor's result) costs 16 registers + 16 adds per key block and
        // measured nothing, at one and at two key blocks per wave (round 4 and round 5)
        constexpr int NACC = X3P_NACC;
        f32x16 Tacc[R][NACC];
        f32x16 Ta[R];                         // the scores (sum of the accumulators), from the statistics units on
        // exp2(s - max) of tile i, as register pairs.  The arithmetic on them is issued as SINGLE fp32 instructions: packed forms
        // (v_pk_add_f32 / v_pk_mul_f32, -DX3P_PK) save 31 of 356 instructions per tile and measured 2.5 us SLOWER per launch (79.2 ->
        // 76.7 us under rocprofv3, profiles/r05_attn_x3p_kbw.txt) -- a packed fp32 op beside MFMAs costs more than the two it replaces
        f32x2 E2[R][8];
        constexpr int NPF = 4;                // MODE 2, up to NPF chunks: the chunks' (max, sum) of the NEXT tile's rows, one iteration ahead
        f32x2 pf[NPF];
#pragma unroll
        for (int c = 0; c < NPF; ++c) pf[c] = f32x2{c == 0 ? 0.f : -INFINITY, c == 0 ? 1.f : 0.f};
        float mw = 0.f;                       // the wave's row maximum (over all its key blocks) that belongs to E
        f32x16 acc_o[R][NCB];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                E2[r][i >> 1][i & 1] = 0.f, Ta[r][i] = 0.f;
#pragma unroll
                for (int c = 0; c < NACC; ++c) Tacc[r][c][i] = 0.f;
            }
        auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc_o[r][cb][i] = 0.f;
        };
        zero_acc();
        auto flush = [&](int head) __attribute__((always_inline)) {
            const int seg = head - first_head;
            float* dst = partial + ((int64_t)bid * P.seg_count + seg) * (int64_t)(NKB * NCB) * 1024;
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    const int t_idx = (gkb0 + r) * NCB + cb;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        if (32 * (gkb0 + r) + 8 * q4 < kk) {
                            const f32x4 v4 = {acc_o[r][cb][q4 * 4], acc_o[r][cb][q4 * 4 + 1], acc_o[r][cb][q4 * 4 + 2], acc_o[r][cb][q4 * 4 + 3]};
                            *reinterpret_cast<f32x4*>(dst + ((int64_t)(t_idx * 4 + q4) * 64 + lane) * 4) = v4;
                        }
                }
        };

        // ---- the vector (and scalar) work of one iteration, cut into UNITS (executed in this order).  A unit is a handful of
        // instructions; the three-instruction chains (subtract -> exp2 -> accumulate) are software-pipelined ACROSS units.
        //   first half, tile i (E -> P images of this wave's key blocks):
        //     1               fetch the (max, sum) pairs of all waves (published before the last barrier)
        //     1               max over the waves
        //     NH + 2          wave b: d_b = max_b - max | e_(b-1) = exp2(d_(b-1)) | l += sum_(b-2) e_(b-2)
        //     1               the row's factor exp2(max_wave - max) / l (0 for rows that do not exist)
        //     12 R            per key block and 4-key chunk: {scale 4 values, pack hi, store | residuals | pack lo, store}
        //   second half, tile i + 1 (score tuples -> E) and the DMA of the tiles to come:
        //     1 + 2 R + 1     DMA offset | max over 8 scores each (LASTW: padded keys -> -inf first) | across the lane halves
        //     8 R + 2         score pair p: x_p = s_p - max | e_(p-1) = exp2(x_(p-1)) | sum += e_(p-2)
        //     1               sum across the halves, publish (max, sum) of the wave's 32 rows
        //     UQ + UV         one LDS-DMA instruction each: rows of Q(i + 3) / V(i + 2)
        // (the combine of the waves' (max, sum) pairs is split over the two lane halves, which hold the same rows: half hf takes the
        //  waves 2 i + hf and the halves meet through v_permlane32_swap -- NH = ceil(NW / 2) pairs per lane instead of NW)
        constexpr int NH = (NW + 1) / 2;
        constexpr int U_CMAX = 1, U_CW = 2, U_FS = U_CW + NH + 2, U_NORM = U_FS + 1, N_FIRST = U_NORM + 12 * R;
        // second-half sequence: offsets | 2 R max units, halves | NXS exp stages with the NDU DMA instructions spread between them | publish
        // exp stages work on PAIRS of scores (two single instructions per operation, see E2)
        constexpr int NPAIR = 8 * R, NXS = NPAIR + 2;
        constexpr int NDU = ISS ? UQ + UV : 0, N_SECOND = 2 + 2 * R + NXS + NDU + 1, NUNITS = N_FIRST + N_SECOND;
        struct SecondMap {
            int kind[N_SECOND], arg[N_SECOND];   // kind 0: offset fetch, 1: max, 2: halves, 3: exp stage, 4: DMA, 5: publish
            constexpr SecondMap() : kind{}, arg{} {
                int v = 0, d = 0;
                kind[v] = 0, arg[v++] = 0;
                for (int m = 0; m < 2 * R; ++m) kind[v] = 1, arg[v++] = m;
                kind[v] = 2, arg[v++] = 0;
                for (int q = 0; q < NXS; ++q) {
                    kind[v] = 3, arg[v++] = q;
                    while (d < NDU && (d + 1) * NXS <= (q + 1) * NDU) kind[v] = 4, arg[v++] = d++;
                }
                kind[v] = 5, arg[v++] = 0;
            }
        };
        constexpr SecondMap smap{};
        struct VS {
            float mx, m, l, fscale, d0, d1, e0, e1;
            f32x2 l2, x0, x1;
            int doff;                         // per-lane source offset of the wave's NEXT DMA instruction (fetched from LDS one unit ahead)
            f32x2 sv[(NW + 1) / 2];
            f32x4 p4;
            unsigned h01, h23;
            float r0, r1, r2, r3;
        };
        // cno / rows_ok: the (head, tile) of tile i and the number of its rows that exist (0 in the fill iteration: P := 0)
#ifdef X3P_NORM_SKEW
        struct NormMap {
            int ch[12 * R], part[12 * R];
            constexpr NormMap() : ch{}, part{} {
                int v = 0;
                for (int t = 0; t < 4 * R + 2; ++t)
                    for (int pp = 0; pp < 3; ++pp)
                        if (t - pp >= 0 && t - pp < 4 * R) ch[v] = t - pp, part[v++] = pp;
            }
        };
        constexpr NormMap nmap{};
        struct NS {
            float fscale;
            f32x4 p4;
            unsigned h01, h23;
            float r0, r1, r2, r3;
        };
        NS sv_[4];
#endif
        auto unit = [&](auto u_t, VS& s0, int par_n, const Cur cno, const Cur cnx, int rows_ok, const DmaCtx& dc) __attribute__((always_inline)) {
            constexpr int u = decltype(u_t)::value;
            VS& s = s0;
#ifdef X3P_ABL_NO_U1
            if constexpr (u < N_FIRST) return;
#endif
#ifdef X3P_ABL_NO_U2
            if constexpr (u >= N_FIRST) { if constexpr (smap.kind[u - N_FIRST] != 4 && smap.kind[u - N_FIRST] != 0) return; }
#endif
            if constexpr (u == 0) {
                if constexpr (MODE == 2) {   // every chunk's (max, sum) of this lane's row, from the statistics passes
                    if (P.nchunks == 2) {   // config C: the two-chunk form without the loop's predicates (485 -> 440 us for its main pass)
                        s.m = fmaxf(pf[0][0], pf[1][0]);
                        s.l = fmaf(pf[1][1], __builtin_amdgcn_exp2f(pf[1][0] - s.m), pf[0][1] * __builtin_amdgcn_exp2f(pf[0][0] - s.m));
                        int row = cnx.t * TR + j;
                        if (row > n32 - 1) row = n32 - 1;
                        const f32x2* st = P.stats + (int64_t)cnx.a * P.n + row;
                        pf[0] = st[0], pf[1] = st[(int64_t)P.h * P.n];
                    } else if (P.nchunks <= NPF) {
                        // up to four chunks (config C: two): the pairs of tile i were requested a whole iteration ago (a global round
                        // trip in the first half of every iteration was a stall of its own: 533 -> 440 us for config C's main
                        // pass); request tile i + 1's now
                        s.m = pf[0][0];
#pragma unroll
                        for (int c = 1; c < NPF; ++c) s.m = fmaxf(s.m, pf[c][0]);
                        s.l = 0.f;
#pragma unroll
                        for (int c = 0; c < NPF; ++c) s.l = fmaf(pf[c][1], __builtin_amdgcn_exp2f(pf[c][0] - s.m), s.l);
                        int row = cnx.t * TR + j;
                        if (row > n32 - 1) row = n32 - 1;
                        const f32x2* st = P.stats + (int64_t)cnx.a * P.n + row;
#pragma unroll
                        for (int c = 0; c < NPF; ++c)
                            if (c < P.nchunks) pf[c] = st[(int64_t)c * P.h * P.n];   // (absent chunks keep (-inf, 0): no contribution)
                    } else {
                        int row = cno.t * TR + j;
                        if (row > n32 - 1) row = n32 - 1;
                        const f32x2* st = P.stats + (int64_t)cno.a * P.n + row;
                        s.m = -INFINITY;
                        for (int c = 0; c < P.nchunks; ++c) s.m = fmaxf(s.m, st[(int64_t)c * P.h * P.n][0]);
                        s.l = 0.f;
                        for (int c = 0; c < P.nchunks; ++c) {
                            const f32x2 pr = st[(int64_t)c * P.h * P.n];
                            s.l = fmaf(pr[1], __builtin_amdgcn_exp2f(pr[0] - s.m), s.l);
                        }
                    }
                } else {
                    const int st_h = st_lane + hf * (TR * 8);
#pragma unroll
                    for (int i = 0; i < NH; ++i) {
                        constexpr int last = NW - 1;
                        // wave 2 i + hf; past the last wave (odd NW, upper half, last pair): re-read the last wave's pair and void it
                        const bool over = 2 * i + 1 > last;         // compile-time per i: only the upper half can be over
                        const int b_off = over ? (par_n * NW + last) * (TR * 8) - hf * (TR * 8) : (par_n * NW + 2 * i) * (TR * 8);
                        f32x2 pr = *reinterpret_cast<const f32x2*>(smem + st_h + b_off);
                        if (over) pr = hf ? f32x2{-INFINITY, 0.f} : pr;
                        s.sv[i] = pr;
                    }
                }
            } else if constexpr (u == U_CMAX) {
                if constexpr (MODE != 2) {
                    float m = s.sv[0][0];
#pragma unroll
                    for (int b = 1; b < NH; ++b) m = fmaxf(m, s.sv[b][0]);
                    s.m = xhalf_max(m), s.l = 0.f;
                }
            } else if constexpr (u < U_FS) {
                if constexpr (MODE == 2) return;
                constexpr int b = u - U_CW;       // stage b: sub of wave b, exp of wave b - 1, fma of wave b - 2
                if constexpr (b >= 2) s.l = fmaf(s.sv[b - 2][1], (b & 1) ? s.e1 : s.e0, s.l);
                if constexpr (b >= 1 && b - 1 < NH) ((b & 1) ? s.e0 : s.e1) = __builtin_amdgcn_exp2f((b & 1) ? s.d0 : s.d1);
                if constexpr (b < NH) ((b & 1) ? s.d1 : s.d0) = s.sv[b][0] - s.m;
            } else if constexpr (u == U_FS) {
                if constexpr (MODE != 2) s.l = xhalf_sum(s.l);   // the two halves' partial sums of the row
                const bool rvalid = j < rows_ok;
                if constexpr (MODE == 1) {   // this chunk's pair of the row; nothing else happens to tile i in a statistics pass
                    if (rvalid && hf == 0 && w == 0) P.stats[((int64_t)chunk * P.h + cno.a) * P.n + cno.t * TR + j] = f32x2{s.m, s.l};
                    s.fscale = 0.f;
                    return;
                }
                if constexpr (AUX)
                    if (P.lse && chunk == 0 && rvalid && hf == 0 && w == 0)
                        P.lse[(int64_t)cno.a * P.n + cno.t * TR + j] = (s.m + __log2f(s.l)) * 0.69314718055994530942f;
                float fs = __builtin_amdgcn_exp2f(mw - s.m) * __builtin_amdgcn_rcpf(s.l);
                asm volatile("" : "+v"(fs));              // keep the select below a select (no branch around the exp / rcp)
                s.fscale = rvalid ? fs : 0.f;
            } else if constexpr (u < N_FIRST) {
                if constexpr (MODE == 1) return;
#ifdef X3P_NORM_SKEW
                // skewed order (dev knob): the three parts of a 4-key chunk sit two units apart -- no unit waits for its predecessor
                constexpr int ch = nmap.ch[u - U_NORM], part = nmap.part[u - U_NORM], r = ch / 4, c4 = ch % 4;
                NS& s = sv_[ch & 3];                     // (shadows the iteration's state: this chunk's own registers)
#else
                constexpr int r = (u - U_NORM) / 12, c4 = ((u - U_NORM) % 12) / 3, part = (u - U_NORM) % 3;
#endif
                unsigned char* pb = smem + ((waddr0 ^ (16 * c4)) + r * PBUF);
                if constexpr (part == 0) {
                    const f32x2 fs2 = {s0.fscale, s0.fscale};
#ifndef X3P_PK
                    const f32x2 e01 = {E2[r][2 * c4][0] * s0.fscale, E2[r][2 * c4][1] * s0.fscale};
                    const f32x2 e23 = {E2[r][2 * c4 + 1][0] * s0.fscale, E2[r][2 * c4 + 1][1] * s0.fscale};
                    (void)fs2;
#else
                    const f32x2 e01 = E2[r][2 * c4] * fs2, e23 = E2[r][2 * c4 + 1] * fs2;
#endif
                    s.p4 = f32x4{e01[0], e01[1], e23[0], e23[1]};
                    // P is ROUNDED to fp32 here in every variant: without this the compiler contracts the product into the subtraction
                    // of the split below (fma) in the variants that do not store A, and their O differs in the last bits
                    asm volatile("" : "+v"(s.p4));
                    if constexpr (AUX) {
                        if (attn_base && j < rows_ok) {
                            const int key0 = 32 * (gkb0 + r) + 8 * c4 + 4 * hf;
                            float* arow = attn_base + ((int64_t)cno.a * P.n + cno.t * TR + j) * P.attn_ld + key0;
                            if (attn_vec && key0 + 4 <= kk) {
                                *reinterpret_cast<f32x4*>(arow) = s.p4;
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (key0 + e < kk) arow[e] = s.p4[e];
                            }
                        }
                    }
                    s.h01 = cvt_pk(s.p4[0], s.p4[1]), s.h23 = cvt_pk(s.p4[2], s.p4[3]);
                    *reinterpret_cast<u32x2*>(pb) = u32x2{s.h01, s.h23};
                }
#ifdef X3P_ABL_NOLO
                else if constexpr (part >= 1) {
                }
#endif
                else if constexpr (part == 1) {
#ifndef X3P_PK
                    s.r0 = s.p4[0] - __uint_as_float(s.h01 << 16), s.r1 = s.p4[1] - __uint_as_float(s.h01 & 0xffff0000u);
                    s.r2 = s.p4[2] - __uint_as_float(s.h23 << 16), s.r3 = s.p4[3] - __uint_as_float(s.h23 & 0xffff0000u);
#else
                    const f32x2 ra = f32x2{s.p4[0], s.p4[1]} - f32x2{__uint_as_float(s.h01 << 16), __uint_as_float(s.h01 & 0xffff0000u)};
                    const f32x2 rb = f32x2{s.p4[2], s.p4[3]} - f32x2{__uint_as_float(s.h23 << 16), __uint_as_float(s.h23 & 0xffff0000u)};
                    s.r0 = ra[0], s.r1 = ra[1], s.r2 = rb[0], s.r3 = rb[1];
#endif
                } else {
                    *reinterpret_cast<u32x2*>(pb + PBUF / 2) = u32x2{cvt_pk(s.r0, s.r1), cvt_pk(s.r2, s.r3)};
                }
            } else {
                constexpr int kind = smap.kind[u - N_FIRST], arg = smap.arg[u - N_FIRST];
                if constexpr (kind == 0) {
                    if constexpr (NDU > 0) s.doff = dma_tab[0];
                } else if constexpr (kind == 1) {
                    constexpr int r = arg >> 1, r0 = 8 * (arg & 1);
#pragma unroll
                    for (int i = r0; i < r0 + 8; ++i) Ta[r][i] = NACC == 2 ? Tacc[r][0][i] + Tacc[r][NACC - 1][i] : Tacc[r][0][i];
                    if constexpr (LASTW) {            // padded keys (the launch's last key block; every key block past a short last chunk)
#pragma unroll
                        for (int i = r0; i < r0 + 8; ++i)
                            Ta[r][i] = (32 * (gkb0 + r) + (i & 3) + 8 * (i >> 2) + 4 * hf) < kk ? Ta[r][i] : -INFINITY;
                    }
#ifdef X3P_ABL_NOMAX
                    float mx = Ta[r][r0];
#else
                    float mx = fmaxf(fmaxf(Ta[r][r0], Ta[r][r0 + 1]), Ta[r][r0 + 2]);
                    mx = fmaxf(fmaxf(mx, Ta[r][r0 + 3]), Ta[r][r0 + 4]);
                    mx = fmaxf(fmaxf(mx, Ta[r][r0 + 5]), Ta[r][r0 + 6]);
                    mx = fmaxf(mx, Ta[r][r0 + 7]);
#endif
                    if constexpr (arg == 0) s.mx = mx; else s.mx = fmaxf(s.mx, mx);
                } else if constexpr (kind == 2) {
                    mw = xhalf_max(s.mx);
                    if constexpr (LASTW) mw = fmaxf(mw, -1e30f);   // a wave whose keys are all padding: exp2(-inf - mw) = 0, not NaN
                    s.l2 = f32x2{0.f, 0.f};
                } else if constexpr (kind == 3) {
                    constexpr int q = arg;        // stage q: sub of score pair q, exp of pair q - 1, add of pair q - 2
#ifndef X3P_ABL_NOSUM
#ifndef X3P_PK
                    if constexpr (q >= 2) s.l2[0] += E2[(q - 2) >> 3][(q - 2) & 7][0], s.l2[1] += E2[(q - 2) >> 3][(q - 2) & 7][1];
#else
                    if constexpr (q >= 2) s.l2 += E2[(q - 2) >> 3][(q - 2) & 7];
#endif
#endif
                    if constexpr (q >= 1 && q - 1 < NPAIR) {
                        const f32x2 x = (q & 1) ? s.x0 : s.x1;
#ifdef X3P_ABL_NOEXP
                        E2[(q - 1) >> 3][(q - 1) & 7] = x * f32x2{0.5f, 0.5f};
#else
                        E2[(q - 1) >> 3][(q - 1) & 7] = f32x2{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
#endif
                    }
#ifndef X3P_PK
                    if constexpr (q < NPAIR) ((q & 1) ? s.x1 : s.x0) = f32x2{Ta[q >> 3][2 * (q & 7)] - mw, Ta[q >> 3][2 * (q & 7) + 1] - mw};
#else
                    if constexpr (q < NPAIR) ((q & 1) ? s.x1 : s.x0) = f32x2{Ta[q >> 3][2 * (q & 7)], Ta[q >> 3][2 * (q & 7) + 1]} - f32x2{mw, mw};
#endif
                } else if constexpr (kind == 4) {
                    const int off = s.doff;
                    if constexpr (arg + 1 < NDU) s.doff = dma_tab[(arg + 1) * 64];
                    dma_unit(std::integral_constant<int, arg>{}, dc, off);
                } else {
                    if constexpr (MODE != 2) {
                        const float lsum = xhalf_sum(s.l2[0] + s.l2[1]);
                        *reinterpret_cast<f32x2*>(smem + st_lane + ((par_n ^ 1) * NW + w) * (TR * 8)) = f32x2{mw, lsum};   // both halves: same pair
                    }
                }
            }
        };
        // MFMA slots of an iteration: 3 NKS R of GEMM1, then 3 NKS R of GEMM2; one MFMA, then the units of its slot.  The first-half
        // units ride behind slots 0 .. H - 3 R - 1, so that the P images are complete before the first transpose-read is requested;
        // the second-half units behind slots H + 2 .. 2 H - 1 (the last GEMM1 results are in flight).
        constexpr int H = 3 * NKS * R;
#ifndef X3P_PFQ
#define X3P_PFQ 1
#endif
#ifndef X3P_PFV
#define X3P_PFV 1
#endif
        constexpr int PFQ = X3P_PFQ < NKS ? X3P_PFQ : NKS - 1, PFV = X3P_PFV < NCB ? X3P_PFV : NCB - 1;
#ifndef X3P_FLO
#define X3P_FLO 0
#endif
#ifndef X3P_FHI_EXTRA
#define X3P_FHI_EXTRA 0
#endif
#ifndef X3P_SLO
#define X3P_SLO 2
#endif
#ifndef X3P_SHI_EXTRA
#define X3P_SHI_EXTRA 0
#endif
        // (dev knobs: where the unit windows start / end among the MFMA slots; the defaults are the round-4 placement)
        constexpr int F_LO = X3P_FLO, F_HI = H - 3 * R * PFV - X3P_FHI_EXTRA, S_LO = H + X3P_SLO * R, S_HI = 2 * H - X3P_SHI_EXTRA;
#define X3P_UB1(k) ((k) <= F_LO ? 0 : (k) >= F_HI ? N_FIRST : (((k) - F_LO) * N_FIRST + (F_HI - F_LO) / 2) / (F_HI - F_LO))
#define X3P_UB2(k) ((k) <= S_LO ? N_FIRST : (k) >= S_HI ? NUNITS : N_FIRST + (((k) - S_LO) * (NUNITS - N_FIRST) + (S_HI - S_LO) / 2) / (S_HI - S_LO))
#define X3P_UB(k) ((k) < H ? X3P_UB1(k) : X3P_UB2(k))

        // ---- one pipeline iteration: GEMM1(i + 1) | normalise(i), then GEMM2(i) | statistics(i + 1) | DMA of Q(i + 3), V(i + 2).
        // Tile x lives in ring slot x % 3; statistics of tile x in buffer x & 1.
        // The SAME body runs the fill (i = -1: tile -1 does not exist -> rows_ok = 0, P = 0; its accumulators are zeroed again
        // before tile 0) and the drain (i = T - 1: GEMM1 / statistics of a tile T that does not exist run on stale operands and
        // are never consumed): one code path means one register allocation, and no spill code anywhere near the loop.
        auto iteration = [&](int i, int slot_q, int slot_v, const Cur cno, const Cur cnx, const DmaCtx& dc, int rows_ok) __attribute__((always_inline)) {
            const int par_n = i & 1;
            const int qa = q_lane + slot_q * QSLOT;         // Q(i + 1): slot (i + 1) % 3
            int va[NCB];                                    // V(i): slot i % 3
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) va[cb] = (voff0 + slot_v * VSLOT) ^ (64 * cb);
            X3P_STAMP(i + 1, 0);
            VS vs;
            bf16x8 ql[NKS], qh[NKS], vl[2 * NCB], vh[2 * NCB], ph[R][2], pl[R][2];
            // Fragment reads run PFQ k-steps (GEMM1) / PFV steps (GEMM2) ahead of their MFMAs.  One step is enough: distances of 2
            // and 3 steps (X3P_PFQ / X3P_PFV, 24 more registers) measured the same within noise at one and at two key blocks per wave
            // (profiles/r05_attn_x3p_kbw.txt) -- the LDS round trip is not what the MFMAs wait for.
            static_for<0, PFQ>([&](auto s_t) __attribute__((always_inline)) {
                constexpr int s0 = decltype(s_t)::value;
                if constexpr (s0 < NKS) ql[s0] = q_frag(qa, s0, 1), qh[s0] = q_frag(qa, s0, 0);
            });
            // the fragments of GEMM2 step E (part 0 .. 3: Vl / Vl / Vh / Vh of its two column blocks; the P fragments of key block r
            // when the step opens a new 16-row k-step)
            auto g2_load = [&](auto e_t2, auto part_t, auto r_t2) __attribute__((always_inline)) {
                constexpr int E = decltype(e_t2)::value, part = decltype(part_t)::value, r = decltype(r_t2)::value;
                constexpr int NPq = NCB / 2, skE = E / NPq, cE = 2 * (E % NPq);
                if constexpr (E < NCB && MODE != 1) {
                    if constexpr (r == 0) {
                        if constexpr (part == 0) vl[2 * E] = v_frag(va[cE], skE, 1);
                        if constexpr (part == 1) vl[2 * E + 1] = v_frag(va[cE + 1], skE, 1);
                        if constexpr (part == 2) vh[2 * E] = v_frag(va[cE], skE, 0);
                        if constexpr (part == 3) vh[2 * E + 1] = v_frag(va[cE + 1], skE, 0);
                    }
                    if constexpr (cE == 0 && part == 0) ph[r][skE] = p_frag(r, skE, 0);
                    if constexpr (cE == 0 && part == 1) pl[r][skE] = p_frag(r, skE, 1);
                }
            };
            // ---------------- first half: GEMM1.  k-step e: products Kh Ql, Kl Qh, Kh Qh of every key block, the key blocks alternating
            // (KBW = 2: no MFMA follows its predecessor on the same accumulator)
            static_for<0, NKS>([&](auto e_t) __attribute__((always_inline)) {
                constexpr int e = decltype(e_t)::value;
                X3P_STAMP2(i + 1, e);
                if constexpr (e + PFQ < NKS) {              // a later k-step's fragments, PFQ steps ahead of their first use
                    X3P_FENCE();
                    ql[e + PFQ] = q_frag(qa, e + PFQ, 1);
                    qh[e + PFQ] = q_frag(qa, e + PFQ, 0);
                }
                static_for<0, 3 * R>([&](auto m_t) __attribute__((always_inline)) {
                    constexpr int mm = decltype(m_t)::value, mi = mm / R, r = mm % R, k = 3 * R * e + mm;
                    X3P_FENCE();
                    constexpr int kr = 3 * e + mi, ai = kr % NACC;       // the key block's MFMA kr goes to accumulator kr % NACC
                    f32x16& acc = Tacc[r][ai];
                    const f32x16 c0 = kr < NACC ? f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f} : acc;
                    if constexpr (mi == 0) acc = X3P_MFMA1(kph[r][e], ql[e], c0);
                    if constexpr (mi == 1) acc = X3P_MFMA1(kpl[r][e], qh[e], c0);
                    if constexpr (mi == 2) acc = X3P_MFMA1(kph[r][e], qh[e], c0);
                    if constexpr (e + PFV >= NKS) {         // all P chunks are written: the first PFV steps of GEMM2
                        constexpr int E = e + PFV - NKS;
                        if constexpr (mi < 2) {
                            g2_load(std::integral_constant<int, E>{}, std::integral_constant<int, mi>{}, std::integral_constant<int, r>{});
                        } else {
                            if constexpr (r == 0) g2_load(std::integral_constant<int, E>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
                            if constexpr (r == R - 1) g2_load(std::integral_constant<int, E>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
                        }
                    }
                    static_for<X3P_UB(k), X3P_UB(k + 1)>([&](auto u_t) __attribute__((always_inline)) {
                        unit(u_t, vs, par_n, cno, cnx, rows_ok, dc);
                    });
                });
            });
            X3P_STAMP(i + 1, 2);
            // ---------------- second half: GEMM2.  step e = (16-row k-step e / NP, column-block pair e % NP): the products Ph Vl, Pl Vh,
            // Ph Vh of the pair's two column blocks and of the wave's key blocks alternate, so that an MFMA never follows its predecessor
            // on the same accumulator behind a gap (it would wait out the whole matrix pipeline)
            constexpr int NP = NCB / 2;
            static_for<0, NCB>([&](auto e_t) __attribute__((always_inline)) {
                constexpr int e = decltype(e_t)::value, sk = e / NP, c0 = 2 * (e % NP);
                X3P_STAMP2(i + 1, 8 + e);
                static_for<0, 6 * R>([&](auto m_t) __attribute__((always_inline)) {
                    constexpr int mm = decltype(m_t)::value, mi = mm / R, r = mm % R, k = H + 6 * R * e + mm, prod = mi / 2, cb = c0 + (mi & 1);
                    X3P_FENCE();
                    if constexpr (MODE != 1) {
                        if constexpr (prod == 0) acc_o[r][cb] = X3P_MFMA2(ph[r][sk], vl[2 * e + (mi & 1)], acc_o[r][cb]);
                        if constexpr (prod == 1) acc_o[r][cb] = X3P_MFMA2(pl[r][sk], vh[2 * e + (mi & 1)], acc_o[r][cb]);
                        if constexpr (prod == 2) acc_o[r][cb] = X3P_MFMA2(ph[r][sk], vh[2 * e + (mi & 1)], acc_o[r][cb]);
                    }
                    // the fragments of step e + PFV, that far ahead of their first use
                    if constexpr (mi < 4) g2_load(std::integral_constant<int, e + PFV>{}, std::integral_constant<int, mi>{}, std::integral_constant<int, r>{});
                    static_for<X3P_UB(k), X3P_UB(k + 1)>([&](auto u_t) __attribute__((always_inline)) {
                        unit(u_t, vs, par_n, cno, cnx, rows_ok, dc);
                    });
                });
            });
            X3P_FENCE();
            X3P_STAMP(i + 1, 3);
        };
        // wait until at most n of this wave's LDS-DMA instructions are in flight: they complete in order, so everything issued before
        // the last n has landed.  (AUX builds also have stores in flight, which are not ordered with the loads: drain everything.)
        auto wait_dma = [&](int n) __attribute__((always_inline)) {
            if constexpr (DRAIN) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
#define X3P_VMC(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
                switch (n) {
                    X3P_VMC(0) X3P_VMC(1) X3P_VMC(2) X3P_VMC(3) X3P_VMC(4) X3P_VMC(5) X3P_VMC(6) X3P_VMC(7) X3P_VMC(8) X3P_VMC(9)
                    X3P_VMC(10) X3P_VMC(11) X3P_VMC(12) X3P_VMC(13) X3P_VMC(14) X3P_VMC(15) X3P_VMC(16) X3P_VMC(17) X3P_VMC(18) X3P_VMC(19)
                    X3P_VMC(20) X3P_VMC(21) X3P_VMC(22) X3P_VMC(23) X3P_VMC(24) X3P_VMC(25) X3P_VMC(26) X3P_VMC(27) X3P_VMC(28) X3P_VMC(29)
                    X3P_VMC(30) X3P_VMC(31) X3P_VMC(32) X3P_VMC(33) X3P_VMC(34) X3P_VMC(35) X3P_VMC(36) X3P_VMC(37) X3P_VMC(38) X3P_VMC(39)
                    default: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
                }
#undef X3P_VMC
            }
        };

        // ---- the pipeline.  The second half of iteration i issues the DMA of Q(i + 3) -> Q slot i % 3 (GEMM1 read Q(i) out of it
        // in iteration i - 1) and of V(i + 2) -> V slot (i + 2) % 3 (GEMM2 read V(i - 1) out of it in iteration i - 1): both slots
        // are free since the last barrier.  These instructions stay in flight across the barrier that closes iteration i; the wait
        // at the end of iteration i + 1 retires them (vmcnt counts down in issue order: "at most this iteration's own" means every
        // older one has landed), and the barrier behind it makes them visible to the other waves, two iterations before the
        // first read (GEMM1(i + 3) / GEMM2(i + 2) in iteration i + 2).  Before the loop: Q(0), V(0), Q(1).
        Cur c0, c1, c2, c3;                   // items i, i + 1, i + 2, i + 3
        c1.a = first_head, c1.t = f_begin - first_head * P.tiles_per_head;   // item 0
        c0 = c1;                              // item -1 does not exist (any valid cursor)
        c2 = cur_next(c1);
        c3 = cur_next(c2);
        if constexpr (ISS) {
            const DmaCtx d0 = dma_ctx(true, c1, 0, MODE != 1, c1, 0), d1 = dma_ctx(1 < T, c2, 1, false, c2, 1);
            static_for<0, UQ + UV>([&](auto u_t) __attribute__((always_inline)) { dma_unit(u_t, d0, dma_tab[decltype(u_t)::value * 64]); });
            static_for<0, UQ>([&](auto u_t) __attribute__((always_inline)) { dma_unit(u_t, d1, dma_tab[decltype(u_t)::value * 64]); });
        }
        int head_g1 = c1.a, head_g2 = -1;     // heads whose Kp fragments / accumulators are in registers
        load_kp(head_g1);
        X3P_WAIT_VM0();
        __builtin_amdgcn_s_barrier();
        int slot_q = 0, slot_v = 2;           // (i + 1) % 3, i % 3 for i = -1
#pragma clang loop unroll(disable)
        for (int i = -1; i < T; ++i) {
            // head changes (wave-uniform, rare): accumulators of a finished head out, Kp fragments of the next head in
            if (MODE != 1 && i >= 0 && c0.a != head_g2) {
                if (head_g2 >= 0) flush(head_g2);
                zero_acc();
                head_g2 = c0.a;
                X3P_WAIT_VM0();
            }
            if (i + 1 < T && c1.a != head_g1) {
                load_kp(c1.a);
                head_g1 = c1.a;
                X3P_WAIT_VM0();
            }
            int rows_ok = 0;
            if (i >= 0) {
                rows_ok = n32 - c0.t * TR;
                if (rows_ok > TR) rows_ok = TR;
            }
            const bool dov = MODE != 1 && i + 2 < T;   // a statistics pass never touches V
            DmaCtx dc = {};
            if constexpr (ISS) dc = dma_ctx(i + 3 < T, c3, slot_v, dov, c2, slot_q == 2 ? 0 : slot_q + 1);
            iteration(i, slot_q, slot_v, c0, i + 1 < T ? c1 : c0, dc, rows_ok);   // (item T does not exist: any valid cursor)
            // everything issued BEFORE this iteration has landed for this wave; together with the barrier: for every wave
            if constexpr (ISS) {
                const uint64_t want = full_mask & ((i + 3 < T ? QBITS : 0ull) | (dov ? VBITS : 0ull));
                wait_dma(__builtin_popcountll(want));
            }   // (no DMA of this wave's own: nothing to wait for; its global loads / stores are the compiler's to track)
            X3P_STAMP(i + 1, 4);
            __builtin_amdgcn_s_barrier();
            X3P_STAMP(i + 1, 5);
            slot_v = slot_q;
            slot_q = slot_q == 2 ? 0 : slot_q + 1;
            c0 = c1, c1 = c2, c2 = c3, c3 = cur_next(c3);
        }
        if constexpr (MODE != 1) flush(head_g2);
#undef X3P_UB
#undef X3P_UB1
#undef X3P_UB2
    };
    // KBW = 1: the second-dispatched waves of a SIMD lose every arbitration to the first at equal priority and set the iteration time
    // (in-kernel trace: first half 1450 ticks for waves 0-3, 2480 for waves 4-6): static priority for them (guide T5, static form)
#ifndef X3P_NOPRIO
    if (KBW == 1 && w >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    using KB = std::integral_constant<int, KBW>;
    using KL = std::integral_constant<int, RLAST>;
    // LASTW (the masking variant) for every wave that holds padded keys: the last wave, and -- in merged key-chunk launches whose last
    // chunk is shorter -- every wave past that chunk's keys (wave-uniform, decided once)
    // (MODE 0 covers all keys in one launch: only its last wave has padded keys, and that wave's copy of the pipeline is specialised on
    // its constant wave index; a second, wave-generic masking copy cost the config-B instantiation 21 spilled registers)
    const bool pads = MODE != 0 && 32 * KBW * (w + 1) > kk;
    if constexpr (LOADER) {
        if (w == NW) {
            // ---- the loader wave: the DMA schedule of the pipeline (see run()) and its barriers, nothing else.  The per-lane source
            // offsets of a full tile live in registers (this wave has them to spare).
            __builtin_amdgcn_s_setprio(2);
            int offs[UQ + UV];
#pragma unroll
            for (int u = 0; u < UQ; ++u) {
                int row, chunk;
                q_rc(u, row, chunk);
                offs[u] = row * ldq_b + 16 * chunk;
            }
#pragma unroll
            for (int u = 0; u < UV; ++u) {
                int row, chunk;
                v_rc(u, row, chunk);
                offs[UQ + u] = row * ldv_b + 16 * chunk;
            }
            Cur c1, c2, c3;
            c1.a = first_head, c1.t = f_begin - first_head * P.tiles_per_head;
            c2 = cur_next(c1);
            c3 = cur_next(c2);
            {
                const DmaCtx d0 = dma_ctx(true, c1, 0, MODE != 1, c1, 0), d1 = dma_ctx(1 < T, c2, 1, false, c2, 1);
                static_for<0, UQ + UV>([&](auto u_t) __attribute__((always_inline)) { dma_unit(u_t, d0, offs[decltype(u_t)::value]); });
                static_for<0, UQ>([&](auto u_t) __attribute__((always_inline)) { dma_unit(u_t, d1, offs[decltype(u_t)::value]); });
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            int slot_q = 0, slot_v = 2;
            for (int i = -1; i < T; ++i) {
                const bool dov = MODE != 1 && i + 2 < T;
                const DmaCtx dc = dma_ctx(i + 3 < T, c3, slot_v, dov, c2, slot_q == 2 ? 0 : slot_q + 1);
                static_for<0, UQ + UV>([&](auto u_t) __attribute__((always_inline)) { dma_unit(u_t, dc, offs[decltype(u_t)::value]); });
                // what the PREVIOUS iteration issued has to have landed before this iteration's barrier (run(): wait_dma)
                const uint64_t want = full_mask & ((i + 3 < T ? QBITS : 0ull) | (dov ? VBITS : 0ull));
                const int n_own = __builtin_popcountll(want);
#define X3P_VMC(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
                switch (n_own) {
                    X3P_VMC(0) X3P_VMC(1) X3P_VMC(2) X3P_VMC(3) X3P_VMC(4) X3P_VMC(5) X3P_VMC(6) X3P_VMC(7) X3P_VMC(8) X3P_VMC(9)
                    X3P_VMC(10) X3P_VMC(11) X3P_VMC(12) X3P_VMC(13) X3P_VMC(14) X3P_VMC(15) X3P_VMC(16) X3P_VMC(17) X3P_VMC(18) X3P_VMC(19)
                    X3P_VMC(20) X3P_VMC(21) X3P_VMC(22) X3P_VMC(23) X3P_VMC(24) X3P_VMC(25) X3P_VMC(26) X3P_VMC(27) X3P_VMC(28) X3P_VMC(29)
                    X3P_VMC(30) X3P_VMC(31) X3P_VMC(32) X3P_VMC(33) X3P_VMC(34) X3P_VMC(35) X3P_VMC(36) X3P_VMC(37) X3P_VMC(38) X3P_VMC(39)
                    default: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
                }
#undef X3P_VMC
                __builtin_amdgcn_s_barrier();
                slot_v = slot_q;
                slot_q = slot_q == 2 ? 0 : slot_q + 1;
                c2 = c3, c3 = cur_next(c3);
            }
        } else if (w == NW - 1) {
            run(KL{}, std::true_type{}, std::false_type{});
        } else if (MODE != 0 && pads) {
            if constexpr (MODE != 0) run(KB{}, std::true_type{}, std::false_type{});
        } else {
            run(KB{}, std::false_type{}, std::false_type{});
        }
    } else if constexpr (LIGHT >= 0) {
        // the last wave carries one key block where the others carry two: it issues all the DMA
        if (w == NW - 1)
            run(KL{}, std::true_type{}, std::true_type{});
        else if (MODE != 0 && pads) {
            if constexpr (MODE != 0) run(KB{}, std::true_type{}, std::false_type{});
        } else
            run(KB{}, std::false_type{}, std::false_type{});
    } else {
        if (w < L0) {                                               // statistics pass, wave 0: stores the pairs, issues no DMA
            if (MODE != 0 && pads) {
                if constexpr (MODE != 0) run(KB{}, std::true_type{}, std::false_type{});
            } else {
                run(KB{}, std::false_type{}, std::false_type{});
            }
        } else if (w == NW - 1) {
            run(KL{}, std::true_type{}, std::true_type{});
        } else if (MODE != 0 && pads) {
            if constexpr (MODE != 0) run(KB{}, std::true_type{}, std::true_type{});
        } else {
            run(KB{}, std::false_type{}, std::true_type{});
        }
    }
}

template <int DK>
__global__ __launch_bounds__(64) void x3p_prep_kp_kernel(const float* __restrict__ kp, int64_t ldkp, int k, int nkb, float c_exp,
                                                          u32x4* __restrict__ out, int chunk_size, int64_t out_stride) {
    constexpr int NKS = DK / 16;
    const int a = blockIdx.y, b = blockIdx.x, lane = threadIdx.x;
    {   // key chunk blockIdx.z (merged launches; one chunk otherwise: chunk_size = k)
        const int k0 = blockIdx.z * chunk_size;
        kp += (int64_t)k0 * ldkp, out += blockIdx.z * out_stride;
        k = k - k0 < chunk_size ? k - k0 : chunk_size;
    }
    int key = 32 * b + (lane & 31);
    const bool pad = key >= k;
    if (pad) key = k - 1;
    const int hf = lane >> 5;
    u32x4* dst = out + ((int64_t)(a * nkb + b) * NKS * 2) * 64 + lane;
    f32x8 raw[NKS];
#pragma unroll
    for (int kb = 0; kb < NKS; ++kb) raw[kb] = load8(kp + (int64_t)key * ldkp + a * DK + 16 * kb + 8 * hf);
#pragma unroll
    for (int kb = 0; kb < NKS; ++kb) {
        u32x4 hi, lo;
        split8(raw[kb] * c_exp, hi, lo);
        if (pad) hi = lo = u32x4{0u, 0u, 0u, 0u};
        dst[(2 * kb) * 64] = hi;
        dst[(2 * kb + 1) * 64] = lo;
    }
}

// out[key, a*DK + col] = sum over the (workgroup, segment) partials of head a, ascending workgroup order (fixed: bit-reproducible)
// Four waves per unit: wave q sums every fourth batch of 16 partials (all 16 loads of a batch in flight), the four sums meet in LDS
// and are added in wave order -- a fixed order whatever the timing.  (One wave per unit walked the ~43 partials of a config-B head
// in three dependent round trips: 8 us for 27 MB.)
template <int DK>
__global__ __launch_bounds__(256) void x3p_reduce_kernel(const float* __restrict__ partial, int nkb, int num_wg, int seg_count,
                                                          int tiles_per_head, int tiles_per_wg, int k, int h, float* __restrict__ out,
                                                          int chunk_size, int64_t partial_stride) {
    constexpr int NCB = DK / 32;
    const int tiles = nkb * NCB;
    const int a = blockIdx.y;
    {   // key chunk blockIdx.z (merged launches; one chunk otherwise: chunk_size = k)
        const int k0 = blockIdx.z * chunk_size;
        partial += blockIdx.z * partial_stride, out += (int64_t)k0 * (h * DK);
        k = k - k0 < chunk_size ? k - k0 : chunk_size;
    }
    const int unit = blockIdx.x;   // (tile, q4): one small workgroup per unit, so that the units spread over all CUs
    const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
    const int t_idx = unit >> 2, q4 = unit & 3;
    if (32 * (t_idx / NCB) + 8 * q4 >= k) return;
    __shared__ f32x4 part[3][64];
    const int f_lo = a * tiles_per_head, f_hi = (a + 1) * tiles_per_head - 1;
    const int b_lo = f_lo / tiles_per_wg;
    int b_hi = f_hi / tiles_per_wg;
    if (b_hi > num_wg - 1) b_hi = num_wg - 1;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const int64_t off = ((int64_t)(t_idx * 4 + q4) * 64 + lane) * 4;
    for (int b = b_lo + 16 * wq; b <= b_hi; b += 64) {
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (b + u <= b_hi) {
                const int seg = a - ((b + u) * tiles_per_wg) / tiles_per_head;
                v[u] = *reinterpret_cast<const f32x4*>(partial + ((int64_t)(b + u) * seg_count + seg) * (int64_t)tiles * 1024 + off);
            }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    if (wq > 0) part[wq - 1][lane] = s;
    __syncthreads();
    if (wq > 0) return;
    s = ((s + part[0][lane]) + part[1][lane]) + part[2][lane];
    const int kb = t_idx / NCB, cbk = t_idx - kb * NCB;
    const int col = a * DK + 32 * cbk + (lane & 31);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = 32 * kb + i + 8 * q4 + 4 * (lane >> 5);
        if (key < k) out[(int64_t)key * (h * DK) + col] = s[i];
    }
}

template <int DK, int NKB, int KBW, bool AUX, int MODE>
int x3p_launch(const X3PParams& P, const X3PPlan& pl, float* out, hipStream_t s) {
    constexpr int lds = x3p_lds_bytes(DK, NKB, KBW);
    static_assert(lds <= 160 * 1024, "sparse_attn_x3p: LDS budget");
    auto kern = sparse_attn_x3p_kernel<DK, NKB, KBW, AUX, MODE>;
    static thread_local unsigned long long attr_set_mask = 0;   // devices (bit = device id) that have the opt-in
    const unsigned long long attr_set_bit = snf::device_bit();
    if (!(attr_set_mask & attr_set_bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            snf::set_error("sparse_attn_x3p: cannot reserve %d bytes of LDS", lds);
            (void)hipGetLastError();
            return SNF_ELAUNCH;
        }
        attr_set_mask |= attr_set_bit;
    }
    const int nch = P.merged > 1 ? P.merged : 1;
    const int csize = P.merged > 1 ? P.chunk_size : P.k;
    if (P.kp && (MODE != 2 || P.merged <= 1)) {   // (the merged statistics launch has already made the fragments of every chunk;
                                                   //  no kp: the caller's key projection wrote the fragment image itself)
        hipLaunchKernelGGL((x3p_prep_kp_kernel<DK>), dim3(NKB, P.h, nch), dim3(64), 0, s, P.kp, P.ldkp, P.k, NKB,
                           P.scale * 1.44269504088896340736f, const_cast<u32x4*>(P.kp_frag), csize, P.kpfrag_stride);
        int rc0 = snf::check_launch("x3p_prep_kp_kernel");
        if (rc0) return rc0;
    }
    // merged: 8 XCDs x (row ranges per XCD) x chunks, the chunk index innermost within an XCD's slots
    const int grid = P.merged > 1 ? 8 * ((pl.num_wg + 7) / 8) * nch : pl.num_wg;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * x3p_waves(NKB, KBW)), lds, s, P);
    int rc = snf::check_launch("sparse_attn_x3p_kernel");
    if (rc || MODE == 1) return rc;
    hipLaunchKernelGGL((x3p_reduce_kernel<DK>), dim3(NKB * (DK / 32) * 4, P.h, nch), dim3(256), 0, s, P.partial, NKB, pl.num_wg, pl.seg_count,
                       pl.tiles_per_head, pl.tiles_per_wg, P.k, P.h, out, csize, P.partial_stride);
    return snf::check_launch("x3p_reduce_kernel");
}
template <int DK, int NB, int KBW>
int x3p_modes(const X3PParams& P, const X3PPlan& pl, float* out, hipStream_t s, int mode) {
    const bool aux = P.attn != nullptr || P.lse != nullptr;
    if (mode == 1) return x3p_launch<DK, NB, KBW, false, 1>(P, pl, out, s);
    if (mode == 2) return aux ? x3p_launch<DK, NB, KBW, true, 2>(P, pl, out, s) : x3p_launch<DK, NB, KBW, false, 2>(P, pl, out, s);
    return aux ? x3p_launch<DK, NB, KBW, true, 0>(P, pl, out, s) : x3p_launch<DK, NB, KBW, false, 0>(P, pl, out, s);
}

}  // namespace

