// K2: top-k patch selector (replaces torch.sort(c, 1, descending=True)[:k], snuffy.py:128-130).
//
// Exact, deterministic, tie rule = descending score then ascending index (torch.sort(stable=True) order).
// Each score becomes a 64-bit composite key  (orderable(score) << 32) | ~index ; larger composite == earlier in the
// output.  Round r: every workgroup bitonic-sorts a chunk of 4096 composites in LDS and keeps its top min(k, chunk);
// rounds repeat on the survivors until one chunk is left, whose top k indices are the answer.  Integer compare-exchange
// only -- bit-exact on every run.
#include "common.h"

namespace {

constexpr int CHUNK = 4096;
constexpr int TPB = 1024;

__device__ __forceinline__ unsigned int orderable_desc(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;  // NaN sorts first (torch semantics)
    if (u == 0x80000000u) u = 0u;                               // -0.0 == +0.0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// in-LDS bitonic sort, descending, of CHUNK 64-bit keys by TPB threads
__device__ __forceinline__ void bitonic_sort_desc(unsigned long long* s) {
    for (int size = 2; size <= CHUNK; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
#pragma unroll
            for (int t = 0; t < CHUNK / 2 / TPB; ++t) {
                int p = threadIdx.x + t * TPB;                       // pair id in [0, CHUNK/2)
                int lo = 2 * p - (p & (stride - 1));                 // index with the `stride` bit clear
                int hi = lo + stride;
                bool desc = ((lo & size) == 0);                      // direction of this bitonic block
                unsigned long long a = s[lo], b = s[hi];
                bool swap = desc ? (a < b) : (a > b);
                if (swap) {
                    s[lo] = b;
                    s[hi] = a;
                }
            }
        }
    }
    __syncthreads();
}

// FROM_SCORES: src is float scores (stride elements apart), else src is a composite list of length m.
// TO_INDEX   : write int64 indices (final round), else write composites.
template <bool FROM_SCORES, bool TO_INDEX>
__global__ __launch_bounds__(TPB) void topk_round_kernel(const void* __restrict__ src, int64_t m, int64_t stride, int k,
                                                        void* __restrict__ dst) {
    __shared__ unsigned long long s[CHUNK];
    const int64_t base = (int64_t)blockIdx.x * CHUNK;
#pragma unroll
    for (int t = 0; t < CHUNK / TPB; ++t) {
        int li = threadIdx.x + t * TPB;
        int64_t gi = base + li;
        unsigned long long key = 0ull;  // below every real composite (index part of a real key is never ~0 here)
        if (gi < m) {
            if (FROM_SCORES) {
                float v = reinterpret_cast<const float*>(src)[gi * stride];
                key = ((unsigned long long)orderable_desc(v) << 32) | (unsigned long long)(0xffffffffu - (unsigned int)gi);
            } else {
                key = reinterpret_cast<const unsigned long long*>(src)[gi];
            }
        }
        s[li] = key;
    }
    bitonic_sort_desc(s);
    int64_t remain = m - base;
    int cnt = (int)(remain < CHUNK ? remain : CHUNK);
    int keep = cnt < k ? cnt : k;
    for (int li = threadIdx.x; li < keep; li += TPB) {
        unsigned long long key = s[li];
        if (TO_INDEX) {
            reinterpret_cast<int64_t*>(dst)[li] = (int64_t)(0xffffffffu - (unsigned int)(key & 0xffffffffull));
        } else {
            reinterpret_cast<unsigned long long*>(dst)[(int64_t)blockIdx.x * k + li] = key;
        }
    }
}

// survivors after a round over m items with chunk keep k (every chunk but the last is full)
inline int64_t survivors(int64_t m, int k) {
    int64_t full = m / CHUNK, rem = m % CHUNK;
    return full * (int64_t)(k < CHUNK ? k : CHUNK) + (rem < k ? rem : k);
}

}  // namespace

extern "C" {

size_t snf_topk_workspace_bytes(int64_t n, int k) {
    if (n <= CHUNK || k < 1) return 0;
    int64_t s0 = ((n + CHUNK - 1) / CHUNK) * (int64_t)k;
    int64_t s1 = ((s0 + CHUNK - 1) / CHUNK) * (int64_t)k;
    return (size_t)(s0 + s1) * sizeof(unsigned long long);
}

int snf_topk_f32(const float* scores, int64_t n, int64_t stride, int k, int64_t* idx_out, void* workspace,
                 size_t workspace_bytes, snf_stream_t stream) {
    SNF_REQUIRE(scores && idx_out, "snf_topk_f32: null pointer");
    SNF_REQUIRE(n >= 1 && stride >= 1, "snf_topk_f32: bad n=%lld stride=%lld", (long long)n, (long long)stride);
    SNF_REQUIRE(k >= 1 && k <= n, "snf_topk_f32: need 1 <= k <= n (k=%d n=%lld)", k, (long long)n);
    SNF_REQUIRE(k <= SNF_TOPK_MAX_K, "snf_topk_f32: k=%d exceeds SNF_TOPK_MAX_K=%d", k, SNF_TOPK_MAX_K);
    SNF_REQUIRE(n < 0xffffffffll, "snf_topk_f32: n too large");
    hipStream_t s = snf::as_stream(stream);
    if (n <= CHUNK) {
        hipLaunchKernelGGL((topk_round_kernel<true, true>), dim3(1), dim3(TPB), 0, s, scores, n, stride, k, idx_out);
        return snf::check_launch("topk_round_kernel<scores,index>");
    }
    if (!workspace || workspace_bytes < snf_topk_workspace_bytes(n, k)) {
        snf::set_error("snf_topk_f32: workspace %zu < %zu", workspace_bytes, snf_topk_workspace_bytes(n, k));
        return SNF_EWORKSPACE;
    }
    unsigned long long* buf0 = reinterpret_cast<unsigned long long*>(workspace);
    int64_t cap0 = ((n + CHUNK - 1) / CHUNK) * (int64_t)k;
    unsigned long long* buf1 = buf0 + cap0;
    // round 0: scores -> composites
    int64_t m = n;
    int blocks = (int)((m + CHUNK - 1) / CHUNK);
    hipLaunchKernelGGL((topk_round_kernel<true, false>), dim3(blocks), dim3(TPB), 0, s, scores, m, stride, k, buf0);
    int rc = snf::check_launch("topk_round_kernel<scores,pairs>");
    if (rc) return rc;
    // NB: chunks keep exactly k (all full) except possibly the last; lists are stored at block*k so a short last
    // chunk leaves a gap only at the very end -> the survivor list is contiguous of length survivors(m, k).
    m = survivors(m, k);
    unsigned long long* cur = buf0;
    unsigned long long* nxt = buf1;
    while (m > CHUNK) {
        blocks = (int)((m + CHUNK - 1) / CHUNK);
        hipLaunchKernelGGL((topk_round_kernel<false, false>), dim3(blocks), dim3(TPB), 0, s, cur, m, (int64_t)1, k, nxt);
        rc = snf::check_launch("topk_round_kernel<pairs,pairs>");
        if (rc) return rc;
        int64_t m2 = survivors(m, k);
        SNF_REQUIRE(m2 < m, "snf_topk_f32: reduction does not converge (k=%d)", k);
        m = m2;
        unsigned long long* t = cur;
        cur = nxt;
        nxt = t;
    }
    hipLaunchKernelGGL((topk_round_kernel<false, true>), dim3(1), dim3(TPB), 0, s, cur, m, (int64_t)1, k, idx_out);
    return snf::check_launch("topk_round_kernel<pairs,index>");
}

int snf_topk_gather_f32(const float* scores, int64_t n, int64_t stride, int k, int64_t* idx_out, const float* x, int d,
                        float* xs, void* workspace, size_t workspace_bytes, snf_stream_t stream) {
    int rc = snf_topk_f32(scores, n, stride, k, idx_out, workspace, workspace_bytes, stream);
    if (rc) return rc;
    if (x && xs) rc = snf_gather_rows_f32(x, n, d, idx_out, k, xs, stream);
    return rc;
}

}  // extern "C"
