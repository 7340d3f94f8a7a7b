// K7 (fast form), translation unit 1: the dk = 128 kernel variants and the C entry points (see sparse_attn_mfma_impl.h).
#include "sparse_attn_mfma_impl.h"

namespace snf {
unsigned long long* g_attn_trace = nullptr;  // debug hook, see snf_debug_attn_trace (also read by sparse_attn_x3p.hip)
int g_attn_trace_wg = 0;
}  // namespace snf
using snf::g_attn_trace;
using snf::g_attn_trace_wg;

namespace snf {
int attn_launch_dk128(int qv_dtype, bool stats_pass, const AttnParams& P, const Plan& pl, float* out, hipStream_t s) {
    if (stats_pass)
        return qv_dtype == SNF_DT_F32 ? launch_stats<128, float>(P, pl, s) : launch_stats<128, unsigned short>(P, pl, s);
    return qv_dtype == SNF_DT_F32 ? launch_nkb<128, float>(P, pl, out, s) : launch_nkb<128, unsigned short>(P, pl, out, s);
}
}  // namespace snf

extern "C" {

// debug only (not part of the public header): device buffer of >= 64*8*4 u64 receiving s_memtime stamps of workgroup 0
void snf_debug_attn_trace(void* buf) { g_attn_trace = reinterpret_cast<unsigned long long*>(buf); }
void snf_debug_attn_trace_wg(int wg) { g_attn_trace_wg = wg; }

size_t snf_sparse_attn_fwd_workspace_bytes(int64_t n, int k, int h, int dk, int mfma) {
    if (n < 1 || k < 1 || h < 1 || dk < 1) return 0;
    if (mfma) {
        size_t pb, sb;
        if (!chunked_workspace(n, k, h, dk, &pb, &sb)) return 0;
        return pb + sb + kp_staging_bytes(k, h, dk);   // partial tiles | chunk statistics | bf16 copy of an f32 Kp
    }
    return snf::generic_attn_workspace_bytes(n, k, h, dk);
}

int snf_sparse_attn_fwd_mfma(const void* q, int64_t ldq, const void* v, int64_t ldv, int qv_dtype, const void* kp,
                             int kp_dtype, int64_t n, int k, int h, int dk, float scale, float* out, float* attn,
                             float* lse, void* workspace, size_t workspace_bytes, snf_stream_t stream) {
    return snf_sparse_attn_fwd_mfma_dropout(q, ldq, v, ldv, qv_dtype, kp, kp_dtype, n, k, h, dk, scale, out, attn, lse, 0.f, 0, 0,
                                            workspace, workspace_bytes, stream);
}

int snf_sparse_attn_fwd_mfma_dropout(const void* q, int64_t ldq, const void* v, int64_t ldv, int qv_dtype, const void* kp,
                                     int kp_dtype, int64_t n, int k, int h, int dk, float scale, float* out, float* attn,
                                     float* lse, float dropout_p, uint64_t seed, uint64_t offset, void* workspace,
                                     size_t workspace_bytes, snf_stream_t stream) {
    SNF_REQUIRE(q && v && kp && out, "snf_sparse_attn_fwd_mfma: null pointer");
    SNF_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "snf_sparse_attn_fwd_mfma: dropout_p=%f outside [0, 1)", dropout_p);
    SNF_REQUIRE(!(dropout_p > 0.f) || attn || lse, "snf_sparse_attn_fwd_mfma: dropout needs the lse (or attn) output -- "
                "the backward regenerates the mask and recomputes P from lse");
    SNF_REQUIRE(n >= 1 && k >= 1 && h >= 1, "snf_sparse_attn_fwd_mfma: bad shape");
    SNF_REQUIRE(qv_dtype == SNF_DT_F32 || qv_dtype == SNF_DT_BF16, "snf_sparse_attn_fwd_mfma: bad dtype %d", qv_dtype);
    SNF_REQUIRE(kp_dtype == SNF_DT_F32 || kp_dtype == SNF_DT_BF16, "snf_sparse_attn_fwd_mfma: bad kp dtype %d", kp_dtype);
    ChunkPlan cp;
    size_t partial_bytes = 0, stats_bytes = 0;
    if (!make_chunks(k, dk, &cp) || !chunked_workspace(n, k, h, dk, &partial_bytes, &stats_bytes)) {
        snf::set_error("snf_sparse_attn_fwd_mfma: unsupported shape k=%d dk=%d (need dk in {64, 128} and k <= %d)", k, dk,
                       MAX_CHUNKS * (dk == 128 ? 224 : 256));
        return SNF_EUNSUPPORTED;
    }
    if (dropout_p > 0.f && cp.n_chunks > 1) {
        // the in-register mask is keyed on the launch's own key index and the chunks' main passes differ in their outputs:
        // the training shapes (k <= 224 / 256) are one chunk; more keys with dropout take the exact kernels + snf_dropout_mask_f32
        snf::set_error("snf_sparse_attn_fwd_mfma: dropout is supported for one key chunk only (k <= %d at dk = %d), got k=%d",
                       dk == 128 ? 224 : 256, dk, k);
        return SNF_EUNSUPPORTED;
    }
    const int64_t d = (int64_t)h * dk;
    if (ldq >= (1 << 24) || ldv >= (1 << 24) || n * (ldq > ldv ? ldq : ldv) >= 0x7fffffffll) {
        snf::set_error("snf_sparse_attn_fwd_mfma: n * row pitch = %lld elements exceeds the 32-bit offsets of the kernel",
                       (long long)(n * (ldq > ldv ? ldq : ldv)));
        return SNF_EUNSUPPORTED;
    }
    const int align = qv_dtype == SNF_DT_BF16 ? 8 : 4;   // 16-byte rows
    SNF_REQUIRE(ldq >= d && ldv >= d && (ldq % align) == 0 && (ldv % align) == 0,
                "snf_sparse_attn_fwd_mfma: ldq=%lld / ldv=%lld must be >= h*dk and keep rows 16-byte aligned",
                (long long)ldq, (long long)ldv);
    SNF_REQUIRE((reinterpret_cast<uintptr_t>(q) & 15) == 0 && (reinterpret_cast<uintptr_t>(v) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(kp) & 15) == 0,
                "snf_sparse_attn_fwd_mfma: q / v / kp must be 16-byte aligned");
    const size_t need = partial_bytes + stats_bytes + (kp_dtype == SNF_DT_F32 ? kp_staging_bytes(k, h, dk) : 0);
    if (!workspace || workspace_bytes < need) {
        snf::set_error("snf_sparse_attn_fwd_mfma: workspace %zu < %zu", workspace_bytes, need);
        return SNF_EWORKSPACE;
    }
    AttnParams P;
    P.q = q;
    P.v = v;
    P.n = n;
    P.ldq = ldq;
    P.ldv = ldv;
    P.ldkp = d;
    P.h = h;
    P.scale = scale;
    P.attn_ld = k;
    P.n_stride = n;
    P.vl = nullptr;
    P.vl_bags = 0;
    P.out_direct = nullptr;
    P.partial = reinterpret_cast<float*>(workspace);
    P.trace = g_attn_trace;
    P.trace_wg = g_attn_trace_wg;
    P.drop = snf::make_dropout(dropout_p, seed, offset);
    float* stats = cp.n_chunks > 1 ? reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + partial_bytes)
                                   : nullptr;
    P.n_chunks = cp.n_chunks;
    hipStream_t s = snf::as_stream(stream);
    // the kernels read Kp as bf16 (half the bytes of the cold prologue burst every workgroup starts with); an f32 Kp is
    // rounded once here, to the same values the kernels used to produce in flight
    const unsigned short* kp16 = reinterpret_cast<const unsigned short*>(kp);
    if (kp_dtype == SNF_DT_F32) {
        unsigned short* stage = reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(workspace) + partial_bytes +
                                                                  stats_bytes);
        const int64_t groups = (int64_t)k * d / 8;
        hipLaunchKernelGGL(kp_to_bf16_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<const float*>(kp), stage, groups);
        int rc = snf::check_launch("kp_to_bf16_kernel");
        if (rc) return rc;
        kp16 = stage;
    }
    auto plan_chunk = [&](int c, Plan* pl) -> int {   // keys of chunk c; fills the launch geometry
        const int k0 = c * cp.chunk_k;
        const int kc = (k - k0 < cp.chunk_k) ? k - k0 : cp.chunk_k;
        make_plan(n, kc, h, dk, pl);
        P.kp = kp16 + (int64_t)k0 * d;
        P.k = kc;
        P.tiles_per_head = pl->tiles_per_head;
        P.tiles_per_wg = pl->tiles_per_wg;
        P.total_tiles = pl->total_tiles;
        P.seg_count = pl->seg_count;
        return k0;
    };
    Plan pl;
    if (stats) {   // pass 1: row statistics of every chunk
        P.attn = nullptr;
        P.lse = nullptr;
        P.stats = nullptr;
        for (int c = 0; c < cp.n_chunks; ++c) {
            plan_chunk(c, &pl);
            P.stats_out = stats + (size_t)c * h * n * 2;
            int rc = dk == 128 ? snf::attn_launch_dk128(qv_dtype, true, P, pl, nullptr, s)
                               : snf::attn_launch_dk64(qv_dtype, true, P, pl, nullptr, s);
            if (rc) return rc;
        }
    }
    P.stats = stats;
    P.stats_out = nullptr;
    for (int c = 0; c < cp.n_chunks; ++c) {
        const int k0 = plan_chunk(c, &pl);
        P.attn = attn ? attn + k0 : nullptr;
        P.lse = c == 0 ? lse : nullptr;
        float* out_c = out + (int64_t)k0 * d;
        int rc = dk == 128 ? snf::attn_launch_dk128(qv_dtype, false, P, pl, out_c, s)
                           : snf::attn_launch_dk64(qv_dtype, false, P, pl, out_c, s);
        if (rc) return rc;
    }
    return SNF_OK;
}

// ---- varlen: many bags in one launch (small bags are launch-latency bound: SURVEY 7 step 8) ---------------------------------
// offsets[bags + 1] (HOST memory) = first packed row of every bag.  Sizes of the plan table (int32 words, built on the host by
// this call when `table` is given and uploaded by the caller once per batch composition) and of the launch workspace.
int snf_sparse_attn_varlen_plan(const int64_t* offsets, int bags, int k, int h, int dk, int32_t* table, size_t table_ints,
                                size_t* table_ints_needed, size_t* workspace_bytes) {
    SNF_REQUIRE(offsets && bags >= 1 && k >= 1 && h >= 1, "snf_sparse_attn_varlen_plan: bad arguments");
    VarlenPlan vp;
    if (!make_varlen_plan(offsets, bags, k, h, dk, &vp, nullptr, 0)) {
        snf::set_error("snf_sparse_attn_varlen_plan: unsupported shape (bags=%d k=%d dk=%d: need dk in {64, 128}, k <= %d, "
                       "non-empty bags)", bags, k, dk, dk == 128 ? 224 : 256);
        return SNF_EUNSUPPORTED;
    }
    const size_t need = (size_t)snf_attn::VL_DESC * bags + (size_t)vp.total_wg;
    if (table_ints_needed) *table_ints_needed = need;
    if (workspace_bytes)
        *workspace_bytes = ((size_t)vp.partial_slots * (size_t)(vp.nkb * (dk / 32)) * 1024 * sizeof(float) + 255) / 256 * 256 +
                           kp_staging_bytes(k * bags, h, dk);
    if (table) {
        SNF_REQUIRE(table_ints >= need, "snf_sparse_attn_varlen_plan: table %zu < %zu ints", table_ints, need);
        make_varlen_plan(offsets, bags, k, h, dk, &vp, table, table_ints);
    }
    return SNF_OK;
}

// q, v [T, ld] bf16 (T = offsets[bags] packed rows), kp [bags * k, h * dk] (bag b's keys in rows b k .. b k + k - 1),
// out [bags * k, h * dk] f32, attn [h, T, k] / lse [h, T] or null.  table_dev = the uploaded plan table.
int snf_sparse_attn_fwd_mfma_varlen(const void* q, int64_t ldq, const void* v, int64_t ldv, const void* kp, int kp_dtype,
                                    const int64_t* offsets, int bags, int k, int h, int dk, float scale, float* out, float* attn,
                                    float* lse, const int32_t* table_dev, void* workspace, size_t workspace_bytes,
                                    snf_stream_t stream) {
    SNF_REQUIRE(q && v && kp && out && offsets && table_dev, "snf_sparse_attn_fwd_mfma_varlen: null pointer");
    SNF_REQUIRE(kp_dtype == SNF_DT_F32 || kp_dtype == SNF_DT_BF16, "snf_sparse_attn_fwd_mfma_varlen: bad kp dtype %d", kp_dtype);
    VarlenPlan vp;
    if (bags < 1 || !make_varlen_plan(offsets, bags, k, h, dk, &vp, nullptr, 0)) {
        snf::set_error("snf_sparse_attn_fwd_mfma_varlen: unsupported shape (bags=%d k=%d dk=%d)", bags, k, dk);
        return SNF_EUNSUPPORTED;
    }
    const int64_t d = (int64_t)h * dk, total = offsets[bags];
    int64_t nmax = 0;
    for (int b = 0; b < bags; ++b) nmax = offsets[b + 1] - offsets[b] > nmax ? offsets[b + 1] - offsets[b] : nmax;
    if (ldq >= (1 << 24) || ldv >= (1 << 24) || nmax * (ldq > ldv ? ldq : ldv) >= 0x7fffffffll) {
        snf::set_error("snf_sparse_attn_fwd_mfma_varlen: bag rows * row pitch exceeds the 32-bit offsets of the kernel");
        return SNF_EUNSUPPORTED;
    }
    SNF_REQUIRE(ldq >= d && ldv >= d && (ldq % 8) == 0 && (ldv % 8) == 0,
                "snf_sparse_attn_fwd_mfma_varlen: ldq=%lld / ldv=%lld must be >= h*dk and keep rows 16-byte aligned",
                (long long)ldq, (long long)ldv);
    SNF_REQUIRE((reinterpret_cast<uintptr_t>(q) & 15) == 0 && (reinterpret_cast<uintptr_t>(v) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(kp) & 15) == 0,
                "snf_sparse_attn_fwd_mfma_varlen: q / v / kp must be 16-byte aligned");
    const size_t partial_bytes = ((size_t)vp.partial_slots * (size_t)(vp.nkb * (dk / 32)) * 1024 * sizeof(float) + 255) / 256 * 256;
    const size_t need = partial_bytes + (kp_dtype == SNF_DT_F32 ? kp_staging_bytes(k * bags, h, dk) : 0);
    if (!workspace || workspace_bytes < need) {
        snf::set_error("snf_sparse_attn_fwd_mfma_varlen: workspace %zu < %zu", workspace_bytes, need);
        return SNF_EWORKSPACE;
    }
    hipStream_t s = snf::as_stream(stream);
    const unsigned short* kp16 = reinterpret_cast<const unsigned short*>(kp);
    if (kp_dtype == SNF_DT_F32) {
        unsigned short* stage = reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(workspace) + partial_bytes);
        const int64_t groups = (int64_t)k * bags * d / 8;
        hipLaunchKernelGGL(kp_to_bf16_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<const float*>(kp), stage, groups);
        int rc = snf::check_launch("kp_to_bf16_kernel");
        if (rc) return rc;
        kp16 = stage;
    }
    AttnParams P;
    P.q = q, P.v = v, P.kp = kp16;
    P.n = total, P.ldq = ldq, P.ldv = ldv, P.ldkp = d;
    P.k = k, P.h = h, P.scale = scale;
    P.attn = attn, P.attn_ld = k, P.lse = lse;
    P.stats = nullptr, P.stats_out = nullptr, P.n_chunks = 1;
    P.partial = reinterpret_cast<float*>(workspace);
    P.tiles_per_head = P.tiles_per_wg = P.total_tiles = P.seg_count = 0;   // per bag, from the table
    P.trace = nullptr, P.trace_wg = 0;
    P.drop = snf::make_dropout(0.f, 0, 0);
    P.n_stride = total;
    P.vl = table_dev, P.vl_bags = bags;
    P.out_direct = out;
    Plan pl;
    pl.num_wg = (int)vp.total_wg, pl.nkb = vp.nkb;
    pl.tiles_per_head = pl.tiles_per_wg = pl.total_tiles = pl.seg_count = 0;
    if (vp.all_direct) pl.tiles_per_head = -1;   // launch_variant: no reduction pass
    return dk == 128 ? snf::attn_launch_varlen_dk128(P, pl, out, s) : snf::attn_launch_varlen_dk64(P, pl, out, s);
}

}  // extern "C"
