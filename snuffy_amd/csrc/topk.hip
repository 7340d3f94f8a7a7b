// K2: top-k patch selector (replaces torch.sort(c, 1, descending=True)[:k], snuffy.py:128-130).
//
// Exact, deterministic, tie rule = descending score then ascending index (torch.sort(stable=True) order).
// Each score becomes a 64-bit composite key  (orderable(score) << 32) | ~index ; larger composite == earlier in the
// output.  Round r: every workgroup bitonic-sorts a chunk of 4096 composites (registers + wave shuffles, LDS only for
// the cross-wave steps) and keeps its top min(k, chunk);
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

// Bitonic sort (descending) of CHUNK = 4 * TPB 64-bit keys, 4 consecutive keys per thread held in REGISTERS:
//   partner distance 1, 2        -> inside the thread
//   partner distance 4 .. 128    -> another lane of the same wave (64-bit shuffle, no barrier)
//   partner distance >= 256      -> another wave: through LDS (10 of the 78 network steps)
// Element i of the chunk lives in thread i / 4, slot i % 4.
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
    unsigned int lo = (unsigned int)v, hi = (unsigned int)(v >> 32);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ void cmpx(unsigned long long& mine, unsigned long long other, bool take_max) {
    const bool other_gt = other > mine;
    mine = (other_gt == take_max) ? other : mine;
}

__device__ __forceinline__ void bitonic_sort_desc_regs(unsigned long long (&v)[4], unsigned long long* s) {
    const int tid = threadIdx.x;
    for (int size = 2; size <= CHUNK; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 256) {
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 4; ++e) s[4 * tid + e] = v[e];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * tid + e;
                    const bool lower = (i & stride) == 0, desc = (i & size) == 0;
                    cmpx(v[e], s[i ^ stride], lower == desc);
                }
            } else if (stride >= 4) {
                const int lm = stride >> 2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * tid + e;
                    const bool lower = (i & stride) == 0, desc = (i & size) == 0;
                    cmpx(v[e], shfl_xor_u64(v[e], lm), lower == desc);
                }
            } else {
                const bool desc = ((4 * tid) & size) == 0;  // size >= 2: all 4 slots of a thread share it unless size < 8
                if (stride == 2) {
                    const bool d0 = ((4 * tid + 0) & size) == 0, d1 = ((4 * tid + 1) & size) == 0;
                    unsigned long long a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
                    cmpx(v[0], a2, d0);        // slot 0 is the lower element of the pair (0,2)
                    cmpx(v[2], a0, !d0);
                    cmpx(v[1], a3, d1);
                    cmpx(v[3], a1, !d1);
                } else {
                    const bool d0 = ((4 * tid + 0) & size) == 0, d2 = ((4 * tid + 2) & size) == 0;
                    unsigned long long a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
                    cmpx(v[0], a1, d0);
                    cmpx(v[1], a0, !d0);
                    cmpx(v[2], a3, d2);
                    cmpx(v[3], a2, !d2);
                }
                (void)desc;
            }
        }
    }
}

// FROM_SCORES: src is float scores (stride elements apart), else src is a composite list of length m.
// TO_INDEX   : write int64 indices (final round), else write composites.
template <bool FROM_SCORES, bool TO_INDEX>
__global__ __launch_bounds__(TPB) void topk_round_kernel(const void* __restrict__ src, int64_t m, int64_t stride, int k,
                                                        void* __restrict__ dst) {
    __shared__ unsigned long long s[CHUNK];
    const int64_t base = (int64_t)blockIdx.x * CHUNK;
    unsigned long long v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int li = 4 * threadIdx.x + e;
        const int64_t gi = base + li;
        unsigned long long key = 0ull;  // below every real composite (index part of a real key is never ~0 here)
        if (gi < m) {
            if (FROM_SCORES) {
                float f = reinterpret_cast<const float*>(src)[gi * stride];
                key = ((unsigned long long)orderable_desc(f) << 32) | (unsigned long long)(0xffffffffu - (unsigned int)gi);
            } else {
                key = reinterpret_cast<const unsigned long long*>(src)[gi];
            }
        }
        v[e] = key;
    }
    bitonic_sort_desc_regs(v, s);
    int64_t remain = m - base;
    int cnt = (int)(remain < CHUNK ? remain : CHUNK);
    int keep = cnt < k ? cnt : k;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int li = 4 * threadIdx.x + e;
        if (li < keep) {
            if (TO_INDEX)
                reinterpret_cast<int64_t*>(dst)[li] = (int64_t)(0xffffffffu - (unsigned int)(v[e] & 0xffffffffull));
            else
                reinterpret_cast<unsigned long long*>(dst)[(int64_t)blockIdx.x * k + li] = v[e];
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
