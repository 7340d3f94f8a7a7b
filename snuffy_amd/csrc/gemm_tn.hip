// K16: the weight-gradient contraction of the training step, C [p, q] f32 = A^T B over the BAG axis, on the matrix cores straight
// from the row-major operands the backward chain already holds (reference: the autograd of nn.Linear under train.py:259,468-473 --
// dW = dY^T X).  A [n, .] and B [n, .] are bf16 images whose rows are bag rows; the contraction index is the ROW, so both MFMA
// operands are "k-strided" in memory.  gfx950's transposing LDS read (ds_read_b64_tr_b16) delivers exactly that fragment out of a
// row-major [32 rows][256 columns] step image, so the images are staged as they lie in HBM: full 512-byte row segments by LDS-DMA,
// no transposed copy of any operand (the library's TN kernels reach 0.86 PF/s on these shapes; a transposing pass over the five
// operands of a config-B step would cost more than it saves).
//
// fp32-class (X3): the operands are split images with a hi and a lo plane (column offsets inside the same rows: the [hi | hi | lo]
// images of snf_split3_colsum_f32 / the FFN-in epilogue / layernorm_rows_split3); every 32-row step issues hi hi + hi lo + lo hi
// out of four staged images (A hi, A lo, B hi, B lo: 64 KiB), 96 MFMAs per wave behind 48 transposing reads -- the arithmetic
// intensity of gemm_hl_kernel.  Plain bf16 (X3 = false): one product, two images.
//
// Decomposition: the output is small (768 x 3072 at config B: 36 tiles of 256 x 256) and the contraction long (32 768 rows), so
// every tile is cut into `parts` row ranges, one workgroup each (tiles x parts ~ the CU count: one round); a workgroup parks its
// fp32 accumulators in its own slab in FRAGMENT order (one coalesced 16-byte store per lane and fragment), and tn_reduce_kernel
// sums the parts of a tile in part order (bit-reproducible) and writes C.  256 x 256 tiles, 8 waves as 2 (p) x 4 (q), 128 x 64 per
// wave, two step buffers (128 KiB), all waves in lockstep like gemm_hl_kernel.
//
// LDS image: row r (0 .. 31) of 512 bytes; the 16-byte chunk at position c holds true chunk c ^ 2 (r & 7) (swizzle applied on the
// DMA source address: the LDS side of LDS-DMA is lane-linear).  A transposing read of a 16-lane group covers 4 rows x 32 bytes; the
// two groups of a 32-lane half cover rows with 8 different (r & 7): 8 x 32 bytes on 16 different chunk positions mod 16 -- every
// bank once.
//
// HL (layout of gemm_hl_kernel's operands: [hi(32) | lo(32)] per 32 columns, hi and lo of a column 64 bytes apart in one 128-byte line):
// an operand's step image is 32 rows x 1024 bytes holding both planes as they lie in HBM -- full-line DMA, one row per instruction --
// and the same swizzle on 64 chunk positions; only the fragment addresses differ (block bi, plane pl -> chunk 8 (bi >> 1) + 4 pl +
// 2 (bi & 1) + half).  The training chain's images in this layout are 4 bytes per element instead of 6.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int TM = 256, TN_ = 256, KS = 32;          // tile (p x q), rows per step
constexpr int IMG = KS * TM * 2;                      // one plane's step image: 32 rows x 512 B = 16 KiB
constexpr int TN_MIN_STEPS = 16;                      // a row part is at least this many steps

struct TnParams {
    const unsigned short* a;   // [n, lda] bf16
    const unsigned short* b;   // [n, ldb] bf16
    int64_t lda, ldb;
    int a_hi, a_lo, b_hi, b_lo;   // column offsets of the planes (elements)
    int p, q;                      // output rows (columns of A's planes), output columns (columns of B's planes)
    int steps;                     // n / 32
    int tiles_p, tiles_q, parts;
    float* slab;                   // [parts][tiles][256 x 256] f32 in fragment order
    float* c;                      // reduce: [p, ldc]
    int64_t ldc;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

template <int PITCH>
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p) {   // rows r .. r + 3 and r + 16 .. r + 19 of one 16-column block
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * PITCH));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <bool X3, bool HL>
__global__ __launch_bounds__(512, 2) void gemm_tn_kernel(const TnParams P) {
    static_assert(X3 || !HL, "gemm_tn: the interleaved layout carries both planes");
    constexpr int NPL = X3 ? 2 : 1;                // planes per operand
    constexpr int STEP_BYTES = 2 * NPL * IMG;      // [A hi | A lo | B hi | B lo]   (HL: [A rows of 1024 B | B rows of 1024 B])
    constexpr int B_OFF = NPL * IMG;
    constexpr int PITCH = HL ? 1024 : 512;         // LDS row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    // work item of this workgroup: consecutive items (the q tiles of one p tile of one row part first) on one XCD, which then streams
    // each A row range once per p tile from HBM and shares it through its L2
    const int ntiles = P.tiles_p * P.tiles_q, nitems = ntiles * P.parts;
    const int xcd = blockIdx.x & 7, in_xcd = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int item = xcd * per_xcd + in_xcd;
    if (item >= nitems) return;
    const int part = item / ntiles, tile = item - part * ntiles;
    const int tp = tile / P.tiles_q, tq = tile - tp * P.tiles_q;
    const int s_begin = (int)((int64_t)part * P.steps / P.parts), ns = (int)((int64_t)(part + 1) * P.steps / P.parts) - s_begin;

    // ---- LDS-DMA sources.  A plane image is 16 pieces of 2 rows x 512 B; wave wid stages pieces 2 wid, 2 wid + 1 (rows 4 wid ..
    // 4 wid + 3) of every image; lane l lands at row 2 piece + (l >> 5), chunk position l & 31 and fetches true chunk
    // (l & 31) ^ 2 (row & 7).  Columns past the plane read a valid address (column 0); their products are never stored.
    // HL: a piece is ONE row of 1024 B (both planes of 256 columns); wave wid stages rows 4 wid .. 4 wid + 3 of A and of B; lane l lands at
    // chunk position l and fetches chunk l ^ 2 (row & 7) of the tile's 1024-byte span of its row.
    int src_a[HL ? 4 : 2], src_b[HL ? 4 : 2];
    if constexpr (HL) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 4 * wid + j;
            const int ch = lane ^ (2 * (row & 7));
            const int tc = 32 * (ch >> 3) + 8 * (ch & 3);          // true column of the chunk inside the tile (either plane)
            src_a[j] = row * (int)P.lda + 2 * tp * TM + (tp * TM + tc + 8 <= P.p ? 8 * ch : 0);
            src_b[j] = row * (int)P.ldb + 2 * tq * TN_ + (tq * TN_ + tc + 8 <= P.q ? 8 * ch : 0);
        }
    } else {
        const int pos = lane & 31;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 2 * (2 * wid + j) + (lane >> 5);
            const int ch = pos ^ (2 * (row & 7));
            int ca = tp * TM + 8 * ch, cb = tq * TN_ + 8 * ch;
            if (ca + 8 > P.p) ca = 0;
            if (cb + 8 > P.q) cb = 0;
            src_a[j] = row * (int)P.lda + ca;
            src_b[j] = row * (int)P.ldb + cb;
        }
    }
    const unsigned short* const a_base = P.a + (int64_t)s_begin * KS * P.lda;
    const unsigned short* const b_base = P.b + (int64_t)s_begin * KS * P.ldb;
    auto stage = [&](int kstep, int buf) __attribute__((always_inline)) {
        unsigned char* base = smem + buf * STEP_BYTES;
        const unsigned short* ap = a_base + (int64_t)kstep * KS * P.lda;
        const unsigned short* bp = b_base + (int64_t)kstep * KS * P.ldb;
        if constexpr (HL) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int piece = (4 * wid + j) * 1024;
                __builtin_amdgcn_global_load_lds((glb_void*)(ap + src_a[j] + P.a_hi), (lds_void*)(base + piece), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_void*)(bp + src_b[j] + P.b_hi), (lds_void*)(base + B_OFF + piece), 16, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < (HL ? 0 : 2); ++j) {
            const int piece = (2 * wid + j) * 1024;
            __builtin_amdgcn_global_load_lds((glb_void*)(ap + src_a[j] + P.a_hi), (lds_void*)(base + piece), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(bp + src_b[j] + P.b_hi), (lds_void*)(base + B_OFF + piece), 16, 0, 0);
            if constexpr (X3) {
                __builtin_amdgcn_global_load_lds((glb_void*)(ap + src_a[j] + P.a_lo), (lds_void*)(base + IMG + piece), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_void*)(bp + src_b[j] + P.b_lo), (lds_void*)(base + B_OFF + IMG + piece), 16, 0, 0);
            }
        }
    };

    // ---- fragment addresses: lane (g = l >> 4, i = l & 15) supplies the 8-byte chunk of row 4 g + (i >> 2), columns 4 (i & 3) .. + 3 of a
    // 16-column block; block index bi (16 wr' + ... in units of 16 columns) sits at chunk position 2 (bi ^ s7) + ((i & 3) >> 1)
    const int fi = lane & 15, fg = lane >> 4;
    const int s7 = 4 * (fg & 1) + (fi >> 2);
    // from a hi fragment's address to its lo twin: the next plane image, or (HL) the other 64-byte half of the 128-byte column group --
    // unit u + 2 sits at (u + 2) ^ s7 = (u ^ s7) ^ 2: 64 bytes up or down with bit 1 of s7
    const int LO = HL ? ((s7 & 2) ? -64 : 64) : IMG;
    const int frow = (4 * fg + (fi >> 2)) * PITCH + 16 * ((fi & 3) >> 1) + 8 * (fi & 1);
    // (HL: half-chunk index of block bi's hi values = 8 (bi >> 1) + (bi & 1) in units of 32 bytes, the lo twin 2 units = 64 bytes further)
    auto a_addr = [&](const unsigned char* base, int mi) __attribute__((always_inline)) {
        const int bi = 8 * wr + mi;
        return base + frow + 32 * ((HL ? 4 * (bi >> 1) + (bi & 1) : bi) ^ s7);
    };
    auto b_addr = [&](const unsigned char* base, int ni) __attribute__((always_inline)) {
        const int bi = 4 * wc + ni;
        return base + B_OFF + frow + 32 * ((HL ? 4 * (bi >> 1) + (bi & 1) : bi) ^ s7);
    };

    f32x4 acc[8][4];
    bf16x8 af[8], bh[4], bl[X3 ? 4 : 1];

    stage(0, 0);
    int buf = 0;
    for (int s = 0; s < ns; ++s) {
        // this step's images have landed (every wave waits for its own pieces, then all meet); the other buffer is free: its last
        // fragment reads belong to the previous step's MFMAs, which every wave has issued before this barrier
        wait_vmcnt<0>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* base = smem + buf * STEP_BYTES;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bh[ni] = tr_frag<PITCH>(b_addr(base, ni));
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) af[mi] = tr_frag<PITCH>(a_addr(base, mi));
        if constexpr (X3) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bl[ni] = tr_frag<PITCH>(b_addr(base, ni) + LO);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < ns) stage(s + 1, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        // acc[mi][ni] (lane (i, g)) = C[p = 16 mi' + i][q = 16 ni' + 4 g .. + 3]: B's fragment is the MFMA's first operand
        if (s == 0) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[ni], af[mi], (f32x4{0.f, 0.f, 0.f, 0.f}), 0, 0, 0);
        } else {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[ni], af[mi], acc[mi][ni], 0, 0, 0);
        }
        if constexpr (X3) {
            // hi (A) x lo (B); the A-lo fragment of a row block replaces its A-hi fragment as soon as that block's products are issued
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[ni], af[mi], acc[mi][ni], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                af[mi] = tr_frag<PITCH>(a_addr(base, mi) + LO);
                __builtin_amdgcn_sched_barrier(0);
            }
            // lo (A) x hi (B)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[ni], af[mi], acc[mi][ni], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        buf ^= 1;
    }
    // park the accumulators: fragment (mi, ni) of thread t at slab[(mi * 4 + ni) * 512 + t] (16 bytes each, coalesced)
    unsigned tix = threadIdx.x;
    asm volatile("" : "+v"(tix));
    f32x4* mine = reinterpret_cast<f32x4*>(P.slab + ((size_t)part * ntiles + tile) * (size_t)(TM * TN_)) + tix;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) mine[(mi * 4 + ni) * 512] = acc[mi][ni];
}

// C = sum over the row parts, in part order; one thread per 16-byte fragment element: thread (tile, f = mi * 4 + ni, t) owns
// C[256 tp + 128 wr + 16 mi + i][256 tq + 64 wc + 16 ni + 4 g .. + 3]  (t = 64 (4 wr + wc) + 16 g + i)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const TnParams P) {
    const int ntiles = P.tiles_p * P.tiles_q;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;          // over ntiles * 32 * 512
    if (e >= (int64_t)ntiles * 32 * 512) return;
    const int tile = (int)(e / (32 * 512)), r = (int)(e - (int64_t)tile * (32 * 512));
    const int f = r >> 9, t = r & 511;
    const int mi = f >> 2, ni = f & 3, wid = t >> 6, g = (t >> 4) & 3, i = t & 15;
    const int tp = tile / P.tiles_q, tq = tile - tp * P.tiles_q;
    const int row = tp * TM + 128 * (wid >> 2) + 16 * mi + i, col = tq * TN_ + 64 * (wid & 3) + 16 * ni + 4 * g;
    const f32x4* src = reinterpret_cast<const f32x4*>(P.slab) + (size_t)tile * (TM * TN_ / 4) + r;
    f32x4 sum = src[0];
    for (int pp = 1; pp < P.parts; ++pp) sum += src[(size_t)pp * ntiles * (TM * TN_ / 4)];
    if (row < P.p && col + 4 <= P.q) *reinterpret_cast<f32x4*>(P.c + (int64_t)row * P.ldc + col) = sum;
}

int tn_parts(int64_t steps, int ntiles) {
    int parts = snf::cu_count() / (ntiles > 0 ? ntiles : 1);
    if (parts < 1) parts = 1;
    if (parts > steps / TN_MIN_STEPS) parts = (int)(steps / TN_MIN_STEPS);
    if (parts < 1) parts = 1;
    return parts;
}

template <bool X3, bool HL = false>
int launch_tn(const TnParams& P, hipStream_t s) {
    constexpr int lds = 2 * 2 * (X3 ? 2 : 1) * IMG;
    static thread_local unsigned long long attr_set_mask = 0;   // devices (bit = device id) that have the opt-in
    const unsigned long long bit = snf::device_bit();
    auto kern = gemm_tn_kernel<X3, HL>;
    if (!(attr_set_mask & bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            snf::set_error("gemm_tn: cannot reserve %d bytes of LDS", lds);
            (void)hipGetLastError();
            return SNF_ELAUNCH;
        }
        attr_set_mask |= bit;
    }
    const int nitems = P.tiles_p * P.tiles_q * P.parts;
    const int grid = ((nitems + 7) / 8) * 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, P);
    int rc = snf::check_launch("gemm_tn_kernel");
    if (rc) return rc;
    const int64_t elems = (int64_t)P.tiles_p * P.tiles_q * 32 * 512;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, s, P);
    return snf::check_launch("tn_reduce_kernel");
}

}  // namespace

extern "C" size_t snf_gemm_tn_ws_bytes(int64_t n, int p, int q) {
    if (n < KS || p < 1 || q < 1) return 0;
    const int ntiles = ((p + TM - 1) / TM) * ((q + TN_ - 1) / TN_);
    return (size_t)tn_parts(n / KS, ntiles) * ntiles * (size_t)(TM * TN_) * sizeof(float);
}

extern "C" int snf_gemm_tn_f32(const void* a, int64_t lda, int a_hi, int a_lo, const void* b, int64_t ldb, int b_hi, int b_lo, int hl,
                               int64_t n, int p, int q, float* c, int64_t ldc, void* workspace, size_t workspace_bytes, snf_stream_t stream) {
    SNF_REQUIRE(a && b && c && workspace, "snf_gemm_tn_f32: null pointer");
    SNF_REQUIRE(n >= 1 && p >= 1 && q >= 1, "snf_gemm_tn_f32: bad shape n=%lld p=%d q=%d", (long long)n, p, q);
    if (hl) a_lo = a_hi + 32, b_lo = b_hi + 32;
    const bool x3 = a_lo >= 0 || b_lo >= 0;
    SNF_REQUIRE(!x3 || (a_lo >= 0 && b_lo >= 0), "snf_gemm_tn_f32: both operands carry a lo plane, or neither");
    const int64_t a_w = hl ? (int64_t)a_hi + 2 * (int64_t)p : (int64_t)(a_lo > a_hi ? a_lo : a_hi) + p;
    const int64_t b_w = hl ? (int64_t)b_hi + 2 * (int64_t)q : (int64_t)(b_lo > b_hi ? b_lo : b_hi) + q;
    if ((hl && (p % 32 || q % 32 || a_hi % 64 || b_hi % 64)) || n % KS || p % 8 || q % 8 || a_hi < 0 || b_hi < 0 || a_hi % 8 || b_hi % 8 || (x3 && (a_lo % 8 || b_lo % 8)) || lda % 8 || ldb % 8 ||
        lda < a_w || ldb < b_w || ldc < q || ldc % 4 ||
        (reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(workspace)) % 16 ||
        (KS + 1) * lda >= 0x7fffffffll || (KS + 1) * ldb >= 0x7fffffffll) {
        snf::set_error("snf_gemm_tn_f32: shape n=%lld p=%d q=%d (lda %lld ldb %lld ldc %lld, planes %d %d / %d %d) outside the kernel's domain "
                       "(n %% 32, p %% 8, q %% 8, plane offsets %% 8, 16-byte aligned rows)",
                       (long long)n, p, q, (long long)lda, (long long)ldb, (long long)ldc, a_hi, a_lo, b_hi, b_lo);
        return SNF_EUNSUPPORTED;
    }
    const size_t need = snf_gemm_tn_ws_bytes(n, p, q);
    if (workspace_bytes < need) {
        snf::set_error("snf_gemm_tn_f32: workspace %zu < %zu", workspace_bytes, need);
        return SNF_EWORKSPACE;
    }
    TnParams P;
    P.a = reinterpret_cast<const unsigned short*>(a), P.b = reinterpret_cast<const unsigned short*>(b);
    P.lda = lda, P.ldb = ldb;
    P.a_hi = a_hi, P.a_lo = a_lo, P.b_hi = b_hi, P.b_lo = b_lo;
    P.p = p, P.q = q;
    P.steps = (int)(n / KS);
    P.tiles_p = (p + TM - 1) / TM, P.tiles_q = (q + TN_ - 1) / TN_;
    P.parts = tn_parts(P.steps, P.tiles_p * P.tiles_q);
    P.slab = reinterpret_cast<float*>(workspace);
    P.c = c, P.ldc = ldc;
    hipStream_t s = snf::as_stream(stream);
    if (hl) return launch_tn<true, true>(P, s);
    return x3 ? launch_tn<true>(P, s) : launch_tn<false>(P, s);
}
