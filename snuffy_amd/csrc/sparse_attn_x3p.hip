// K7, fp32-class: host side of the pipelined split-bf16 x3 sparse attention on pre-split operands (kernels: sparse_attn_x3p_impl.h) --
// launch plans, key chunks, workspace layout, the C entry points; and the kernel family dk = 128 / one key block per wave (round 4).
// The other families live in sparse_attn_x3p_k2.hip (dk = 128, two key blocks per wave) and sparse_attn_x3p_dk64.hip (dk = 64).
#include <atomic>

#include "sparse_attn_x3p_impl.h"

namespace snf {
namespace x3p {
// Key blocks per wave of the dk = 128 family for launches of 5 .. 8 key blocks: 1 = one wave per key block, two waves per SIMD (round
// 4), 2 = one wave per SIMD with two key blocks each (round 5: half the LDS fragment reads and no issue sharing, but nothing hides a
// wave's own latencies -- measured equal at 160 / 200 keys and 10 % behind at 256, profiles/r05_attn_x3p_kbw.txt, so 1 stays the
// default).  snf_debug_x3p_kbw() switches it for A / B measurements and the parity tests.
std::atomic<int> g_kbw_dk128{1};

int run_dk128_k1(const X3PParams& P, const X3PPlan& pl, float* out, hipStream_t s, int mode) {
#define SNF_X3P_CASE(NB) \
    case NB: return x3p_modes<128, NB, 1>(P, pl, out, s, mode);
    switch (pl.nkb) {
#ifndef SNF_ATTN_DEV
        SNF_X3P_CASE(4)
        SNF_X3P_CASE(5)
        SNF_X3P_CASE(6)
#endif
        SNF_X3P_CASE(7)
#ifndef SNF_ATTN_DEV
        SNF_X3P_CASE(8)
#endif
        default: break;
    }
#undef SNF_X3P_CASE
    snf::set_error("sparse_attn_x3p: key-block count %d not built (dk = 128, one key block per wave)", pl.nkb);
    return SNF_EUNSUPPORTED;
}
}  // namespace x3p
}  // namespace snf

namespace {

bool x3p_plan(int64_t n, int k, int h, int dk, X3PPlan* pl, int max_wg = 0) {
    if ((dk != 128 && dk != 64) || k < 97 || k > 256 || n < 1 || h < 1) return false;   // 4 .. 8 key blocks (fewer: too few waves per CU, and too many DMA instructions per wave)
    const int64_t tph = (n + TR - 1) / TR, total = tph * h;
    if (total > 0x7fffffff) return false;
    const int cus = max_wg > 0 ? max_wg : snf::cu_count();
    int64_t num_wg = total < cus ? total : cus;
    const int64_t tpw = (total + num_wg - 1) / num_wg;
    num_wg = (total + tpw - 1) / tpw;
    pl->num_wg = (int)num_wg;
    pl->tiles_per_head = (int)tph;
    pl->tiles_per_wg = (int)tpw;
    pl->total_tiles = (int)total;
    pl->seg_count = (int)((tpw + tph - 1) / tph + 1);
    pl->nkb = (k + 31) / 32;
    return true;
}
size_t x3p_partial_bytes(const X3PPlan& pl, int dk) { return (size_t)pl.num_wg * pl.seg_count * (size_t)(pl.nkb * (dk / 32)) * 1024 * sizeof(float); }
size_t x3p_kpfrag_bytes(const X3PPlan& pl, int h, int dk) { return (size_t)h * pl.nkb * (dk / 16) * 2 * 64 * 16; }
size_t x3p_workspace(const X3PPlan& pl, int h, int dk) { return x3p_partial_bytes(pl, dk) + x3p_kpfrag_bytes(pl, h, dk); }

// the kernel family of a launch: dk = 64 has one; dk = 128 runs two key blocks per wave from 5 key blocks up (g_kbw_dk128)
int x3p_dispatch(int dk, const X3PParams& P, const X3PPlan& pl, float* out, hipStream_t s, int mode) {
    if (dk == 64) return snf::x3p::run_dk64_k1(P, pl, out, s, mode);
    if (snf::x3p::g_kbw_dk128 == 2 && pl.nkb >= 5) return snf::x3p::run_dk128_k2(P, pl, out, s, mode);
    return snf::x3p::run_dk128_k1(P, pl, out, s, mode);
}

// keys per launch: 256 at most (8 key blocks); more keys run as up to 8 chunks of (almost) equal size, a multiple of 4:
// statistics pass per chunk, then the chunks' main passes with the softmax exact over all keys
struct X3PChunks {
    int count, size;
};
bool x3p_chunks(int k, int dk, X3PChunks* c) {
    if ((dk != 128 && dk != 64) || k < 97 || k > 8 * 256) return false;
    c->count = (k + 255) / 256;
    // (round 5: whole key blocks per chunk, so that every chunk starts on a key-block boundary -- the key projection can write the
    //  fragment image of any chunked launch -- and all chunks share ONE kernel instantiation: a shorter last chunk runs with its
    //  trailing key blocks masked)
    c->size = c->count == 1 ? k : (((k + c->count - 1) / c->count + 31) / 32) * 32;
    return true;
}
struct X3PLayout {   // workspace: partial accumulators | Kp fragment image | statistics
    size_t partial, kpfrag, stats;
    bool merged;       // all chunks in one statistics launch and one main launch (partial / kpfrag then hold every chunk's region)
    X3PPlan plan;      // of the first chunk (merged: of every chunk)
};
// Merged launches need every chunk's workgroups co-resident (one per CU: the kernel's LDS) and the same block size for all chunks.
int x3p_merged_ranges(const X3PChunks& ch, int k) {
    if (ch.count < 2) return 0;
    const int last = k - (ch.count - 1) * ch.size;
    if (last < 1) return 0;
    const int per_xcd = snf::cu_count() / 8 / ch.count;
    return per_xcd >= 1 ? 8 * per_xcd : 0;
}
bool x3p_layout(int64_t n, int k, int h, int dk, X3PChunks* ch, X3PLayout* lay) {
    if (h < 1 || !x3p_chunks(k, dk, ch)) return false;
    const int ranges = x3p_merged_ranges(*ch, k);
    lay->merged = ranges > 0;
    // chunks are whole key blocks, so the last one can be short (k = 1345: 5 x 256 + 65).  Only the merged launch masks a short last
    // chunk; a launch of its own needs the kernel's 97 keys (devices with fewer than 8 x chunks CUs have no merged form)
    if (!lay->merged && ch->count > 1 && k - (ch->count - 1) * ch->size < 97) return false;
    if (!x3p_plan(n, ch->size, h, dk, &lay->plan, ranges)) return false;
    const size_t copies = lay->merged ? ch->count : 1;
    lay->partial = copies * x3p_partial_bytes(lay->plan, dk);        // the first chunks are the largest
    lay->kpfrag = copies * x3p_kpfrag_bytes(lay->plan, h, dk);
    lay->stats = ch->count > 1 ? (size_t)ch->count * h * n * sizeof(f32x2) : 0;
    return true;
}

}  // namespace

extern "C" {

size_t snf_sparse_attn_fwd_x3_hl_workspace_bytes(int64_t n, int k, int h, int dk) {
    X3PChunks ch;
    X3PLayout lay;
    if (!x3p_layout(n, k, h, dk, &ch, &lay)) return 0;
    return lay.partial + lay.kpfrag + lay.stats;
}

// chunk geometry of an external Kp fragment image (snf_linear_rows_x3_kpfrag_f32): chunks must start on key-block boundaries
static bool x3p_kpfrag_geometry(int k, int h, int dk, X3PChunks* ch, size_t* chunk_bytes) {
    X3PPlan pl;
    if (h < 1 || !x3p_chunks(k, dk, ch) || (ch->count > 1 && ch->size % 32) || !x3p_plan(1, ch->size, h, dk, &pl)) return false;
    *chunk_bytes = x3p_kpfrag_bytes(pl, h, dk);
    return true;
}

size_t snf_sparse_attn_x3_hl_kpfrag_bytes(int k, int h, int dk) {
    X3PChunks ch;
    size_t cb;
    return x3p_kpfrag_geometry(k, h, dk, &ch, &cb) ? cb * ch.count : 0;
}

int snf_linear_rows_x3_kpfrag_f32(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int k_keys, int h, int dk,
                                  int kdim, float scale, void* kp_frag, size_t kp_frag_bytes, snf_stream_t stream) {
    SNF_REQUIRE(x && w && kp_frag, "snf_linear_rows_x3_kpfrag_f32: null pointer");
    X3PChunks ch;
    size_t cb;
    if (!x3p_kpfrag_geometry(k_keys, h, dk, &ch, &cb)) {
        snf::set_error("snf_linear_rows_x3_kpfrag_f32: k=%d h=%d dk=%d outside the fused form (dk = 64 / 128, 97 <= k <= 2048, key chunks of a "
                       "multiple of 32 keys)", k_keys, h, dk);
        return SNF_EUNSUPPORTED;
    }
    SNF_REQUIRE(kp_frag_bytes >= cb * ch.count && (reinterpret_cast<uintptr_t>(kp_frag) & 15) == 0,
                "snf_linear_rows_x3_kpfrag_f32: fragment buffer %zu < %zu (or not 16-byte aligned)", kp_frag_bytes, cb * ch.count);
    SNF_REQUIRE(kdim >= 16 && kdim % 16 == 0 && ldx >= kdim && ldw >= kdim && (ldx % 4) == 0 && (ldw % 4) == 0 &&
                    ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) == 0,
                "snf_linear_rows_x3_kpfrag_f32: x / w rows must be 16-byte aligned, k %% 16 == 0");
    return snf::skinny_linear_x3_kpfrag(x, ldx, w, ldw, bias, k_keys, h * dk, kdim, dk, ch.count > 1 ? ch.size : k_keys,
                                        (int64_t)(cb / sizeof(u32x4)), scale * 1.44269504088896340736f, kp_frag, snf::as_stream(stream));
}

int snf_gather_linear_rows_x3_kpfrag_f32(const float* x, int64_t ldx, int64_t n, const int64_t* idx, const float* w, int64_t ldw,
                                         const float* bias, int k_keys, int h, int dk, int kdim, float scale, void* kp_frag,
                                         size_t kp_frag_bytes, float* xs, int32_t* slot_map, snf_stream_t stream) {
    SNF_REQUIRE(x && idx && w && kp_frag, "snf_gather_linear_rows_x3_kpfrag_f32: null pointer");
    SNF_REQUIRE(n >= 1 && n <= 0x7fffffffll, "snf_gather_linear_rows_x3_kpfrag_f32: bad row count %lld", (long long)n);
    X3PChunks ch;
    size_t cb;
    if (!x3p_kpfrag_geometry(k_keys, h, dk, &ch, &cb)) {
        snf::set_error("snf_gather_linear_rows_x3_kpfrag_f32: k=%d h=%d dk=%d outside the fused form (dk = 64 / 128, 97 <= k <= 2048, key "
                       "chunks of a multiple of 32 keys)", k_keys, h, dk);
        return SNF_EUNSUPPORTED;
    }
    SNF_REQUIRE(kp_frag_bytes >= cb * ch.count && (reinterpret_cast<uintptr_t>(kp_frag) & 15) == 0,
                "snf_gather_linear_rows_x3_kpfrag_f32: fragment buffer %zu < %zu (or not 16-byte aligned)", kp_frag_bytes, cb * ch.count);
    SNF_REQUIRE(kdim >= 16 && kdim % 16 == 0 && ldx >= kdim && ldw >= kdim && (ldx % 4) == 0 && (ldw % 4) == 0 &&
                    ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(xs)) & 15) == 0,
                "snf_gather_linear_rows_x3_kpfrag_f32: x / w / xs rows must be 16-byte aligned, k %% 16 == 0");
    return snf::skinny_linear_x3_kpfrag(x, ldx, w, ldw, bias, k_keys, h * dk, kdim, dk, ch.count > 1 ? ch.size : k_keys,
                                        (int64_t)(cb / sizeof(u32x4)), scale * 1.44269504088896340736f, kp_frag, snf::as_stream(stream), idx, n,
                                        xs, kdim, slot_map);
}

static int x3p_forward(const void* q_hl, int64_t ldq, const void* v_hl, int64_t ldv, const float* kp, const void* kp_frag_ext, int64_t n,
                       int k, int h, int dk, float scale, float* out, float* attn, float* lse, void* workspace, size_t workspace_bytes,
                       snf_stream_t stream);

int snf_sparse_attn_fwd_x3_hl(const void* q_hl, int64_t ldq, const void* v_hl, int64_t ldv, const float* kp, int64_t n, int k, int h,
                              int dk, float scale, float* out, float* attn, float* lse, void* workspace, size_t workspace_bytes,
                              snf_stream_t stream) {
    SNF_REQUIRE(kp, "snf_sparse_attn_fwd_x3_hl: null pointer");
    return x3p_forward(q_hl, ldq, v_hl, ldv, kp, nullptr, n, k, h, dk, scale, out, attn, lse, workspace, workspace_bytes, stream);
}

int snf_sparse_attn_fwd_x3_hl_kpfrag(const void* q_hl, int64_t ldq, const void* v_hl, int64_t ldv, const void* kp_frag, int64_t n, int k,
                                     int h, int dk, float* out, float* attn, float* lse, void* workspace, size_t workspace_bytes,
                                     snf_stream_t stream) {
    SNF_REQUIRE(kp_frag && (reinterpret_cast<uintptr_t>(kp_frag) & 15) == 0, "snf_sparse_attn_fwd_x3_hl_kpfrag: null / unaligned fragment image");
    if (!snf_sparse_attn_x3_hl_kpfrag_bytes(k, h, dk)) {
        snf::set_error("snf_sparse_attn_fwd_x3_hl_kpfrag: k=%d h=%d dk=%d has no fragment-image form", k, h, dk);
        return SNF_EUNSUPPORTED;
    }
    return x3p_forward(q_hl, ldq, v_hl, ldv, nullptr, kp_frag, n, k, h, dk, 0.f, out, attn, lse, workspace, workspace_bytes, stream);
}

static int x3p_forward(const void* q_hl, int64_t ldq, const void* v_hl, int64_t ldv, const float* kp, const void* kp_frag_ext, int64_t n,
                       int k, int h, int dk, float scale, float* out, float* attn, float* lse, void* workspace, size_t workspace_bytes,
                       snf_stream_t stream) {
    SNF_REQUIRE(q_hl && v_hl && out, "snf_sparse_attn_fwd_x3_hl: null pointer");
    X3PChunks ch;
    X3PLayout lay;
    if (n < 1 || !x3p_layout(n, k, h, dk, &ch, &lay)) {
        snf::set_error("snf_sparse_attn_fwd_x3_hl: unsupported shape n=%lld k=%d h=%d dk=%d (dk = 64 / 128, 97 <= k <= 2048)", (long long)n, k, h, dk);
        return SNF_EUNSUPPORTED;
    }
    const int64_t d = (int64_t)h * dk;
    SNF_REQUIRE(ldq >= 2 * d && ldv >= 2 * d && (ldq % 8) == 0 && (ldv % 8) == 0, "snf_sparse_attn_fwd_x3_hl: ldq=%lld / ldv=%lld must be "
                ">= 2*h*dk bf16 and keep rows 16-byte aligned", (long long)ldq, (long long)ldv);
    SNF_REQUIRE(((reinterpret_cast<uintptr_t>(q_hl) | reinterpret_cast<uintptr_t>(v_hl) | (kp ? reinterpret_cast<uintptr_t>(kp) : 0)) & 15) == 0,
                "snf_sparse_attn_fwd_x3_hl: q / v / kp must be 16-byte aligned");
    const size_t need = lay.partial + lay.kpfrag + lay.stats;
    if (!workspace || workspace_bytes < need) {
        snf::set_error("snf_sparse_attn_fwd_x3_hl: workspace %zu < %zu", workspace_bytes, need);
        return SNF_EWORKSPACE;
    }
    unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
    X3PParams P;
    P.q = reinterpret_cast<const unsigned short*>(q_hl), P.v = reinterpret_cast<const unsigned short*>(v_hl), P.kp = kp;
    P.n = n, P.ldq = ldq, P.ldv = ldv, P.ldkp = d;
    P.k = k, P.h = h, P.scale = scale;
    P.attn = attn, P.attn_ld = k, P.lse = lse;
    P.trace = snf::g_attn_trace, P.trace_wg = snf::g_attn_trace_wg;
    P.partial = reinterpret_cast<float*>(ws);
    P.kp_frag = reinterpret_cast<const u32x4*>(ws + lay.partial);
    P.stats = reinterpret_cast<f32x2*>(ws + lay.partial + lay.kpfrag);
    size_t ext_stride = 0;   // u32x4 units between the chunks of an external fragment image
    if (kp_frag_ext) {
        X3PChunks ch2;
        size_t cb;
        if (!x3p_kpfrag_geometry(k, h, dk, &ch2, &cb)) return SNF_EUNSUPPORTED;
        ext_stride = cb / sizeof(u32x4);
        P.kp_frag = reinterpret_cast<const u32x4*>(kp_frag_ext);
    }
    P.nchunks = ch.count, P.chunk = 0;
    P.merged = 0, P.chunk_size = k, P.kpfrag_stride = 0, P.partial_stride = 0;
    hipStream_t s = snf::as_stream(stream);
    if (lay.merged) {   // statistics of all chunks in one launch, then all main passes in one launch (+ one prep, one reduction)
        const X3PPlan& pl = lay.plan;
        P.merged = ch.count, P.chunk_size = ch.size;
        P.kpfrag_stride = kp_frag_ext ? (int64_t)ext_stride : (int64_t)(lay.kpfrag / ch.count / sizeof(u32x4));
        P.partial_stride = (int64_t)(lay.partial / ch.count / sizeof(float));
        P.tiles_per_head = pl.tiles_per_head, P.tiles_per_wg = pl.tiles_per_wg, P.total_tiles = pl.total_tiles;
        P.seg_count = pl.seg_count;
        X3PParams S1 = P;
        S1.attn = nullptr, S1.lse = nullptr;
        int rc = x3p_dispatch(dk, S1, pl, out, s, 1);
        if (rc) return rc;
        return x3p_dispatch(dk, P, pl, out, s, 2);
    }
    // pass 0: everything in one launch (one chunk).  Otherwise pass 1: statistics of every chunk, pass 2: the chunks' main passes
    for (int pass = ch.count == 1 ? 0 : 1; pass <= (ch.count == 1 ? 0 : 2); ++pass)
        for (int c = 0; c < ch.count; ++c) {
            const int k0 = c * ch.size, kc = k - k0 < ch.size ? k - k0 : ch.size;
            X3PPlan pl;
            if (!x3p_plan(n, kc, h, dk, &pl)) {
                snf::set_error("snf_sparse_attn_fwd_x3_hl: chunk of %d keys outside the kernel", kc);
                return SNF_EUNSUPPORTED;
            }
            X3PParams C = P;
            C.kp = kp ? kp + (int64_t)k0 * d : nullptr;
            if (kp_frag_ext) C.kp_frag = P.kp_frag + (size_t)c * ext_stride;
            C.k = kc, C.chunk = c;
            C.attn = (pass != 1 && attn) ? attn + k0 : nullptr;
            C.lse = (pass != 1 && c == 0) ? lse : nullptr;
            C.tiles_per_head = pl.tiles_per_head, C.tiles_per_wg = pl.tiles_per_wg, C.total_tiles = pl.total_tiles;
            C.seg_count = pl.seg_count;
            int rc = x3p_dispatch(dk, C, pl, out + (int64_t)k0 * d, s, pass);
            if (rc) return rc;
        }
    return SNF_OK;
}

// development hook (tools/x3p_dev.py, tests): key blocks per wave of the dk = 128 family (2; anything else: the default, 1)
void snf_debug_x3p_kbw(int kbw) { snf::x3p::g_kbw_dk128 = kbw == 2 ? 2 : 1; }

}  // extern "C"
