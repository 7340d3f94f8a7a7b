// K2: top-k patch selector (replaces torch.sort(c, 1, descending=True)[:k], snuffy.py:128-130).
//
// Exact, deterministic, tie rule = descending score then ascending index (torch.sort(stable=True) order).
// Each score becomes a 32-bit orderable key; larger key == earlier in the output, equal keys by ascending index.  One
// workgroup radix-selects the k-th largest key (below) and rank-sorts the k survivors as 64-bit composites
// (orderable(score) << 32) | ~index.  Integer compares and integer LDS atomics only -- bit-exact on every run.
// (History: a multi-round chunked bitonic sort, 60 us at n = 32768, then a register/shuffle bitonic network for n <= 4096
// -- 29 us at any size, against 11 us for the radix select -- both gone.)
//
// Two forms.  (1) ONE workgroup, everything in its LDS / registers (topk_radix_kernel): any n, the form behind snf_topk_f32 up
// to 64 k scores.  (2) The fused selector (topk_select_kernel): the critic pass that writes the scores has already counted the
// first radix digit of every key into a small global histogram (selector.h); ceil(n / 4096) workgroups read that histogram,
// classify their slice (above the threshold bin / inside it) and append the few hundred survivors to two short lists; the
// workgroup that arrives last finishes the selection on those lists alone.  Same result, bit for bit.
#include "selector.h"

namespace {

using snf::orderable_desc;

// ---------------------------------------------------------------------------------------------------------------
// Radix select: ONE workgroup, O(n) work.  Three counting passes (11 + 11 + 10 bits of the orderable key, LDS
// histogram with integer atomics -> exact and deterministic) find the k-th largest key T; one more pass collects every
// key > T plus the lowest-index (k - #greater) keys == T; the k survivors are sorted as 64-bit composites (bitonic, LDS).
// ---------------------------------------------------------------------------------------------------------------
constexpr int RS_MAXK = 2048;
// debug: phase stamps of thread 0 (tools/topk_trace.py); compiled in only with -DSNF_TOPK_TRACE
#ifdef SNF_TOPK_TRACE
__device__ unsigned long long g_topk_trace[16];
#define TSTAMP(ix) do { if (threadIdx.x == 0) g_topk_trace[ix] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TSTAMP(ix) do { } while (0)
#endif

__device__ __forceinline__ unsigned int block_excl_scan_1024(unsigned int v, unsigned int* wave_tot, unsigned int* total) {
    // exclusive prefix sum over the 1024 threads (ascending thread id); *total = sum of all
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        unsigned int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    __syncthreads();
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    unsigned int base = 0, tot = 0;
    for (int i = 0; i < 16; ++i) {
        unsigned int t = wave_tot[i];
        if (i < wv) base += t;
        tot += t;
    }
    *total = tot;
    return base + incl - v;
}

// IPT > 0: n <= 1024 * IPT and every thread keeps its IPT keys in registers -- the scores are read from memory ONCE (all
// loads in flight together) instead of once per counting pass plus once for the collection; one workgroup pays a full
// memory latency per dependent read round, which is what this kernel's time is made of.  IPT == 0: streaming form, any n.
// order the k survivors in sel[] (descending composite == descending score, ascending index) and write their indices
__device__ __forceinline__ void sort_emit(unsigned long long* sel, int k, int64_t* __restrict__ idx_out) {
    const int tid = threadIdx.x;
    if (k <= 512) {
        // rank sort: the composites are distinct, so #(greater) IS the output position.  k broadcast LDS reads per thread and
        // no further barrier -- the bitonic network below costs log2(k)^2 / 2 workgroup barriers (36 at k = 200).
        // `parts` adjacent lanes share one survivor (k * parts <= 1024): each counts over a strided quarter of the list, four
        // independent LDS reads per trip (one dependent read per trip is a full LDS latency each), then a lane-group sum
        const int parts = k <= 128 ? 8 : (k <= 256 ? 4 : 2);
        const int cand = tid / parts, part = tid % parts;
        const unsigned long long mine = cand < k ? sel[cand] : ~0ull;
        int rank = 0, j2 = part;
        for (; j2 + 3 * parts < k; j2 += 4 * parts) {
            const unsigned long long v0 = sel[j2], v1 = sel[j2 + parts], v2 = sel[j2 + 2 * parts], v3 = sel[j2 + 3 * parts];
            rank += (v0 > mine) + (v1 > mine) + (v2 > mine) + (v3 > mine);
        }
        for (; j2 < k; j2 += parts) rank += (sel[j2] > mine) ? 1 : 0;
        for (int o = 1; o < parts; o <<= 1) rank += __shfl_xor(rank, o, 64);
        if (cand < k && part == 0) idx_out[rank] = (int64_t)(0xffffffffu - (unsigned int)(mine & 0xffffffffull));
        return;
    }
    int p2 = 1;
    while (p2 < k) p2 <<= 1;
    for (int i = k + tid; i < p2; i += 1024) sel[i] = 0ull;
    for (int size = 2; size <= p2; size <<= 1) {
        for (int st = size >> 1; st > 0; st >>= 1) {
            __syncthreads();
            for (int p = tid; p < p2 / 2; p += 1024) {
                const int lo = 2 * p - (p & (st - 1));
                const int hi = lo + st;
                const bool desc = (lo & size) == 0;
                const unsigned long long a = sel[lo], b = sel[hi];
                if (desc ? (a < b) : (a > b)) {
                    sel[lo] = b;
                    sel[hi] = a;
                }
            }
        }
    }
    __syncthreads();
    for (int j = tid; j < k; j += 1024) idx_out[j] = (int64_t)(0xffffffffu - (unsigned int)(sel[j] & 0xffffffffull));
}

// LDS of one selection: histogram words (4 x 2048 in the register form, 2048 otherwise), the survivor list, scan scratch
template <int HIST_WORDS>
struct TopkLds {
    __attribute__((aligned(16))) unsigned int hist[HIST_WORDS];
    unsigned long long sel[RS_MAXK];
    unsigned int wave_tot[16];
    unsigned int s_digit, s_above, s_cnt_sel, s_eq_base, s_a, s_b, s_c, s_d;
};

template <int IPT, int HIST_WORDS>
__device__ __forceinline__ void topk_radix_body(const float* __restrict__ scores, int64_t n, int64_t stride, int k,
                                                int64_t* __restrict__ idx_out, TopkLds<HIST_WORDS>& L) {
    constexpr bool REG = IPT > 0;
    static_assert(HIST_WORDS >= (REG ? 4 * 2048 : 2048), "histogram too small");
    unsigned int* const hist = L.hist;
    unsigned long long* const sel = L.sel;
    unsigned int* const wave_tot = L.wave_tot;
    unsigned int& s_digit = L.s_digit;
    unsigned int& s_above = L.s_above;
    unsigned int& s_cnt_sel = L.s_cnt_sel;
    unsigned int& s_eq_base = L.s_eq_base;
    // Register form: slots past n hold the key 0, which no score maps to (the smallest orderable key, -inf, is 0x007fffff)
    // and which sits alone in digit 0 of the first pass -- it can never be selected (k <= n), never matches a later prefix
    // and never passes the final threshold, so the per-key loops below carry no bounds test.
    TSTAMP(0);
    // slot u of thread t holds score index 4 (t + 1024 (u / 4)) + u % 4 with vector loads, t + 1024 u otherwise
    const bool vec = REG && stride == 1 && (reinterpret_cast<uintptr_t>(scores) & 15) == 0;
    auto slot_index = [&](int u) __attribute__((always_inline)) -> unsigned int {
        return vec ? 4u * ((unsigned int)threadIdx.x + 1024u * (unsigned int)(u >> 2)) + (unsigned int)(u & 3)
                   : (unsigned int)threadIdx.x + 1024u * (unsigned int)u;
    };
    unsigned int rkey[REG ? IPT : 1];
    if constexpr (REG) {
        if (vec) {   // uniform: 16-byte loads, 4 consecutive scores per slot group (one CU pulls dword loads at ~11 B/clk)
#pragma unroll
            for (int g4 = 0; g4 < IPT / 4; ++g4) {
                const int base = 4 * ((int)threadIdx.x + 1024 * g4);
                if (base + 3 < n) {
                    const float4 f = *reinterpret_cast<const float4*>(scores + base);
                    rkey[4 * g4] = orderable_desc(f.x);
                    rkey[4 * g4 + 1] = orderable_desc(f.y);
                    rkey[4 * g4 + 2] = orderable_desc(f.z);
                    rkey[4 * g4 + 3] = orderable_desc(f.w);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rkey[4 * g4 + e] = (base + e < n) ? orderable_desc(scores[base + e]) : 0u;
                }
            }
        } else if (stride == 1) {   // uniform: unit-stride scores need no 64-bit index multiply per load
#pragma unroll
            for (int u = 0; u < IPT; ++u) {
                const int i = (int)threadIdx.x + u * 1024;
                rkey[u] = (i < n) ? orderable_desc(scores[i]) : 0u;
            }
        } else {
#pragma unroll
            for (int u = 0; u < IPT; ++u) {
                const int64_t i = threadIdx.x + (int64_t)u * 1024;
                rkey[u] = (i < n) ? orderable_desc(scores[i * stride]) : 0u;
            }
        }
    }
    // first pass: 4 replicas interleaved per digit (word 4 d + lane % 4; 8 replicas measured slower) -- real scores crowd into a few dozen of the 2048
    // (sign, exponent, 2 mantissa bits) bins, and lanes adding to the SAME word serialise; later passes use words 0..2047
    const int tid = threadIdx.x;
    unsigned int prefix = 0, mask = 0, krem = (unsigned int)k, cnt_eq = 0;
    const int shifts[3] = {21, 10, 0};
    const int nbits[3] = {11, 11, 10};
#ifdef SNF_TOPK_TRACE
    if constexpr (REG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
    TSTAMP(1);
    for (int pass = 0; pass < 3; ++pass) {
        TSTAMP(2 + 2 * pass);
        const int shift = shifts[pass];
        const unsigned int nb = 1u << nbits[pass];
        for (int i = tid; i < ((REG && pass == 0) ? 4 * 2048 : 2048); i += 1024) hist[i] = 0;
        __syncthreads();
        if constexpr (REG) {
            if (pass == 0) {   // no prefix yet: every key counts (the padding keys land in digit 0, below every score)
                const unsigned int rep = (unsigned int)tid & 3u;
#pragma unroll
                for (int u = 0; u < IPT; ++u) atomicAdd(&hist[((rkey[u] >> 21) << 2) | rep], 1u);
                __syncthreads();
                // fold the replicas: thread t owns digits 2t, 2t+1 (two 16-byte reads), result back in words 0..2047
                const uint4 a = *reinterpret_cast<const uint4*>(&hist[8 * tid]), b = *reinterpret_cast<const uint4*>(&hist[8 * tid + 4]);
                __syncthreads();
                hist[2 * tid] = a.x + a.y + a.z + a.w;
                hist[2 * tid + 1] = b.x + b.y + b.z + b.w;
            } else {
#pragma unroll
                for (int u = 0; u < IPT; ++u)
                    if ((rkey[u] & mask) == prefix) atomicAdd(&hist[(rkey[u] >> shift) & (nb - 1)], 1u);
            }
        } else
        for (int64_t i0 = tid; i0 < n; i0 += 8 * 1024) {   // 8 independent loads in flight per thread
            float f[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t i = i0 + (int64_t)u * 1024;
                f[u] = (i < n) ? scores[i * stride] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t i = i0 + (int64_t)u * 1024;
                const unsigned int key = orderable_desc(f[u]);
                if (i < n && (key & mask) == prefix) atomicAdd(&hist[(key >> shift) & (nb - 1)], 1u);
            }
        }
        __syncthreads();
        TSTAMP(3 + 2 * pass);
        // suffix counts: thread t owns digits dpt*t .. dpt*t + dpt-1 ; S(d) = #keys (in the prefix class) with digit >= d
        const int dpt = (int)(nb / 1024);
        unsigned int own[2] = {0, 0}, local = 0;
        for (int e = 0; e < dpt; ++e) {
            own[e] = hist[dpt * tid + e];
            local += own[e];
        }
        unsigned int total;
        const unsigned int below = block_excl_scan_1024(local, wave_tot, &total);   // keys with a smaller digit
        unsigned int s_hi = total - below - local;                                   // keys in digits above this thread's
        for (int e = dpt - 1; e >= 0; --e) {
            const unsigned int s_d = s_hi + own[e];                                  // S(d) for d = dpt*tid + e
            if (s_d >= krem && s_hi < krem) {
                s_digit = (unsigned int)(dpt * tid + e);
                s_above = s_hi;
            }
            s_hi = s_d;
        }
        __syncthreads();
        cnt_eq = hist[s_digit];
        prefix |= s_digit << shift;
        mask |= (nb - 1) << shift;
        krem -= s_above;
        __syncthreads();
    }
    TSTAMP(8);
    // threshold key T = prefix: #(key > T) = k - krem, #(key == T) = cnt_eq >= krem
    const unsigned int T = prefix;
    if (tid == 0) {
        s_cnt_sel = 0;
        s_eq_base = 0;
    }
    __syncthreads();
    const bool take_all_eq = (cnt_eq == krem);
    if (take_all_eq && REG) {   // common case (no tie straddles the k-th place): unordered append straight from registers
        if constexpr (REG) {
#pragma unroll
            for (int u = 0; u < IPT; ++u) {
                if (rkey[u] >= T) {
                    const unsigned int pos = atomicAdd(&s_cnt_sel, 1u);
                    sel[pos] = ((unsigned long long)rkey[u] << 32) | (unsigned long long)(0xffffffffu - slot_index(u));
                }
            }
        }
    } else     if (take_all_eq) {   // common case (no tie straddles the k-th place): unordered append, 8 loads in flight
        for (int64_t i0 = tid; i0 < n; i0 += 8 * 1024) {
            float f[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t i = i0 + (int64_t)u * 1024;
                f[u] = (i < n) ? scores[i * stride] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t i = i0 + (int64_t)u * 1024;
                const unsigned int key = orderable_desc(f[u]);
                if (i < n && key >= T) {
                    const unsigned int pos = atomicAdd(&s_cnt_sel, 1u);
                    sel[pos] = ((unsigned long long)key << 32) | (unsigned long long)(0xffffffffu - (unsigned int)i);
                }
            }
        }
    } else
    for (int64_t i0 = 0; i0 < n; i0 += 1024) {
        const int64_t i = i0 + tid;
        unsigned int key = 0;
        bool gt = false, eq = false;
        if (i < n) {
            key = orderable_desc(scores[i * stride]);
            gt = key > T;
            eq = key == T;
        }
        if (gt || (eq && take_all_eq)) {
            const unsigned int pos = atomicAdd(&s_cnt_sel, 1u);
            sel[pos] = ((unsigned long long)key << 32) | (unsigned long long)(0xffffffffu - (unsigned int)i);
        }
        if (!take_all_eq) {   // ties straddle the k-th place: keep the lowest indices (uniform branch)
            unsigned int tot;
            const unsigned int rank = s_eq_base + block_excl_scan_1024(eq ? 1u : 0u, wave_tot, &tot);
            if (eq && rank < krem) {
                const unsigned int pos = atomicAdd(&s_cnt_sel, 1u);
                sel[pos] = ((unsigned long long)key << 32) | (unsigned long long)(0xffffffffu - (unsigned int)i);
            }
            __syncthreads();
            if (tid == 0) s_eq_base += tot;
            __syncthreads();
        }
    }
    __syncthreads();
    TSTAMP(9);
    sort_emit(sel, k, idx_out);
    TSTAMP(10);
}

template <int IPT>
__global__ __launch_bounds__(1024) void topk_radix_kernel(const float* __restrict__ scores, int64_t n, int64_t stride, int k,
                                                          int64_t* __restrict__ idx_out) {
    __shared__ TopkLds<(IPT > 0) ? 4 * 2048 : 2048> L;
    topk_radix_body<IPT>(scores, n, stride, k, idx_out, L);
}

// many bags, one launch: workgroup b runs the one-workgroup form on scores[offsets[b] .. offsets[b + 1]) and writes the bag's k
// indices (relative to the bag's first row) to idx_out[b * k ..] -- the same kernel body as a per-bag launch, same results
template <int IPT>
__global__ __launch_bounds__(1024) void topk_radix_segmented_kernel(const float* __restrict__ scores,
                                                                    const int64_t* __restrict__ offsets, int k,
                                                                    int64_t* __restrict__ idx_out) {
    __shared__ TopkLds<(IPT > 0) ? 4 * 2048 : 2048> L;
    const int64_t lo = offsets[blockIdx.x], hi = offsets[blockIdx.x + 1];
    // a bag shorter than k selects (and orders) all of its rows: entries k_b .. k - 1 of its output row are not written
    const int kb = hi - lo < k ? (int)(hi - lo) : k;
    topk_radix_body<IPT>(scores + lo, hi - lo, 1, kb, idx_out + (int64_t)blockIdx.x * k, L);
}

// ---------------------------------------------------------------------------------------------------------------
// Fused selector (form 2)
// ---------------------------------------------------------------------------------------------------------------
constexpr int SEL_SLICE = 4096;   // scores per workgroup: 4 consecutive per thread

// first-digit histogram of a score vector that did not come out of the critic pass (snf_topk_f32 above 64 k scores)
__global__ __launch_bounds__(1024) void sel_hist_kernel(const float* __restrict__ scores, int64_t n, int64_t stride,
                                                        snf::SelectorState* __restrict__ st) {
    __shared__ unsigned int hist[snf::SEL_BINS];
    const int tid = threadIdx.x;
    hist[2 * tid] = 0;
    hist[2 * tid + 1] = 0;
    __syncthreads();
    const int64_t base = ((int64_t)blockIdx.x * 1024 + tid) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (base + e < n) atomicAdd(&hist[orderable_desc(scores[(base + e) * stride]) >> 21], 1u);
    __syncthreads();
    unsigned int* g = st->hist + (blockIdx.x & (snf::SEL_REPL - 1)) * snf::SEL_BINS;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const unsigned int c = hist[2 * tid + e];
        if (c) __hip_atomic_fetch_add(&g[2 * tid + e], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(1024) void topk_select_kernel(const float* __restrict__ scores, int64_t n, int64_t stride, int k,
                                                           int64_t* __restrict__ idx_out,
                                                           snf::SelectorState* __restrict__ st) {
    __shared__ TopkLds<2048> L;
    const int tid = threadIdx.x;
    // this workgroup's slice of the scores, requested before anything else
    const int64_t base = ((int64_t)blockIdx.x * 1024 + tid) * 4;
    unsigned int key[4];
    if (stride == 1 && base + 3 < n && (reinterpret_cast<uintptr_t>(scores) & 15) == 0) {
        const float4 f = *reinterpret_cast<const float4*>(scores + base);
        key[0] = orderable_desc(f.x);
        key[1] = orderable_desc(f.y);
        key[2] = orderable_desc(f.z);
        key[3] = orderable_desc(f.w);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) key[e] = (base + e < n) ? orderable_desc(scores[(base + e) * stride]) : 0u;   // key 0: no score
    }
    // ---- the first digit's histogram: thread t owns bins 2t, 2t + 1 (summed over the replicas)
    unsigned int own[2] = {0u, 0u};
#pragma unroll
    for (int r = 0; r < snf::SEL_REPL; ++r) {
        const uint2 h = *reinterpret_cast<const uint2*>(&st->hist[r * snf::SEL_BINS + 2 * tid]);
        own[0] += h.x;
        own[1] += h.y;
    }
    if (tid == 0) {
        L.s_a = 0;   // winners of this workgroup
        L.s_b = 0;   // candidates of this workgroup
        L.s_digit = 0;
        L.s_above = 0;
        L.s_cnt_sel = 0;
    }
    const unsigned int local = own[0] + own[1];
    unsigned int total;
    const unsigned int below = block_excl_scan_1024(local, L.wave_tot, &total);
    {
        unsigned int s_hi = total - below - local;   // keys in the bins above this thread's
#pragma unroll
        for (int e = 1; e >= 0; --e) {
            const unsigned int s_d = s_hi + own[e];   // S(d) = #keys with first digit >= d
            if (s_d >= (unsigned int)k && s_hi < (unsigned int)k) {
                L.s_digit = (unsigned int)(2 * tid + e);
                L.s_above = s_hi;
                L.s_cnt_sel = own[e];
            }
            s_hi = s_d;
        }
    }
    __syncthreads();
    const unsigned int bstar = L.s_digit, n_above = L.s_above, cnt_eq0 = L.s_cnt_sel;
    // every workgroup reads the same histogram, so all of them take the same decision.  total != n: the state was left dirty
    // by an aborted call; a crowded threshold bin (massive ties): both go to the exact one-workgroup selection on the scores
    const bool fast = total == (unsigned int)n && cnt_eq0 <= (unsigned int)snf::SEL_CAND_CAP;
    if (fast) {
        unsigned int pw[4], pc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned int d = key[e] >> 21;
            pw[e] = pc[e] = 0xffffffffu;
            if (d > bstar) pw[e] = atomicAdd(&L.s_a, 1u);
            else if (d == bstar) pc[e] = atomicAdd(&L.s_b, 1u);
        }
        __syncthreads();
        if (tid == 0) {
            L.s_c = L.s_a ? __hip_atomic_fetch_add(&st->n_win, L.s_a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            L.s_d = L.s_b ? __hip_atomic_fetch_add(&st->n_cand, L.s_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        }
        __syncthreads();
        const unsigned int bw = L.s_c, bc = L.s_d;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned long long comp = ((unsigned long long)key[e] << 32) | (unsigned long long)(0xffffffffu - (unsigned int)(base + e));
            // 8-byte agent-scope stores: written through to where every other CU reads them (no release fence needed)
            if (pw[e] != 0xffffffffu && bw + pw[e] < (unsigned int)snf::SEL_MAXK)
                __hip_atomic_store(&st->win[bw + pw[e]], comp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (pc[e] != 0xffffffffu && bc + pc[e] < (unsigned int)snf::SEL_CAND_CAP)
                __hip_atomic_store(&st->cand[bc + pc[e]], comp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- arrive; the last workgroup finishes alone
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned int ticket = __hip_atomic_fetch_add(&st->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        L.s_eq_base = (ticket == gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!L.s_eq_base) return;
    if (fast) {
        // candidates of the threshold bin in registers (<= 4 per thread), read where the other workgroups wrote them
        unsigned long long c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned int j = (unsigned int)tid + 1024u * (unsigned int)u;
            c[u] = j < cnt_eq0 ? __hip_atomic_load(&st->cand[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
        for (unsigned int j = tid; j < n_above; j += 1024)
            L.sel[j] = __hip_atomic_load(&st->win[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // radix select on the remaining digits of the 64-bit composite (key << 32 | ~index): the low word orders equal keys by
        // ascending index, so a tie that straddles the k-th place is resolved by the same loop; it ends as soon as the whole
        // remaining class is taken (normally after the two key digits)
        unsigned long long prefix = (unsigned long long)bstar << 53, mask = 0x7ffull << 53;
        unsigned int krem = (unsigned int)k - n_above, cnt_eq = cnt_eq0;
        const int shifts[5] = {42, 32, 21, 10, 0};
        const int nbits[5] = {11, 10, 11, 11, 10};
        for (int pass = 0; pass < 5 && cnt_eq != krem; ++pass) {
            const int shift = shifts[pass];
            const unsigned int nb = 1u << nbits[pass];
            L.hist[2 * tid] = 0;
            L.hist[2 * tid + 1] = 0;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if ((c[u] & mask) == prefix) atomicAdd(&L.hist[(unsigned int)(c[u] >> shift) & (nb - 1)], 1u);
            __syncthreads();
            const int dpt = (int)(nb / 1024);
            unsigned int o2[2] = {0, 0}, loc = 0;
            for (int e = 0; e < dpt; ++e) {
                o2[e] = L.hist[dpt * tid + e];
                loc += o2[e];
            }
            unsigned int tot;
            const unsigned int bel = block_excl_scan_1024(loc, L.wave_tot, &tot);
            unsigned int s_hi = tot - bel - loc;
            for (int e = dpt - 1; e >= 0; --e) {
                const unsigned int s_d = s_hi + o2[e];
                if (s_d >= krem && s_hi < krem) {
                    L.s_digit = (unsigned int)(dpt * tid + e);
                    L.s_above = s_hi;
                }
                s_hi = s_d;
            }
            __syncthreads();
            cnt_eq = L.hist[L.s_digit];
            prefix |= (unsigned long long)L.s_digit << shift;
            mask |= (unsigned long long)(nb - 1) << shift;
            krem -= L.s_above;
            __syncthreads();
        }
        if (tid == 0) L.s_cnt_sel = n_above;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c[u] != 0ull && (c[u] & mask) >= prefix) {
                const unsigned int pos = atomicAdd(&L.s_cnt_sel, 1u);
                if (pos < (unsigned int)RS_MAXK) L.sel[pos] = c[u];
            }
        __syncthreads();
        sort_emit(L.sel, k, idx_out);
    } else {
        topk_radix_body<0>(scores, n, stride, k, idx_out, L);
        if (tid == 0) st->fallbacks += 1;
    }
    // leave the counted state zeroed for the next call (nobody else reads or writes it any more)
    uint4* hz = reinterpret_cast<uint4*>(st->hist);
    for (int i = tid; i < snf::SEL_REPL * snf::SEL_BINS / 4; i += 1024) hz[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid == 0) {
        st->n_win = 0;
        st->n_cand = 0;
        st->arrive = 0;
    }
}

}  // namespace

extern "C" {

#ifdef SNF_TOPK_TRACE
void snf_debug_topk_trace(unsigned long long* host16) { (void)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_topk_trace), 16 * sizeof(unsigned long long)); }
#endif

// above this many scores snf_topk_f32 takes the multi-workgroup form (histogram launch + select launch); below it the one
// workgroup with its keys in registers is as fast or faster (15 us at 32 k scores vs 61 us for the streaming form at 100 k)
constexpr int64_t TOPK_SINGLE_WG_MAX_N = 65536;

size_t snf_selector_state_bytes(void) { return sizeof(snf::SelectorState); }

size_t snf_topk_workspace_bytes(int64_t n, int k) {
    (void)k;
    return n > TOPK_SINGLE_WG_MAX_N ? sizeof(snf::SelectorState) : 0;   // small inputs keep their state in LDS
}

static int launch_select(const float* scores, int64_t n, int64_t stride, int k, int64_t* idx_out, snf::SelectorState* st,
                         hipStream_t s) {
    const unsigned grid = (unsigned)((n + SEL_SLICE - 1) / SEL_SLICE);
    hipLaunchKernelGGL(topk_select_kernel, dim3(grid), dim3(1024), 0, s, scores, n, stride, k, idx_out, st);
    return snf::check_launch("topk_select_kernel");
}

int snf_topk_select_f32(const float* scores, int64_t n, int k, int64_t* idx_out, void* selector_state, snf_stream_t stream) {
    SNF_REQUIRE(scores && idx_out && selector_state, "snf_topk_select_f32: null pointer");
    SNF_REQUIRE(n >= 1 && n < 0x3fffffffll, "snf_topk_select_f32: bad n=%lld", (long long)n);
    SNF_REQUIRE(k >= 1 && k <= n && k <= RS_MAXK, "snf_topk_select_f32: need 1 <= k <= min(n, %d) (k=%d n=%lld)", RS_MAXK, k,
                (long long)n);
    SNF_REQUIRE((reinterpret_cast<uintptr_t>(selector_state) & 15) == 0, "snf_topk_select_f32: state must be 16-byte aligned");
    return launch_select(scores, n, 1, k, idx_out, reinterpret_cast<snf::SelectorState*>(selector_state), snf::as_stream(stream));
}

int snf_topk_hist_select_f32(const float* scores, int64_t n, int64_t stride, int k, int64_t* idx_out, void* state,
                             snf_stream_t stream) {
    SNF_REQUIRE(scores && idx_out && state, "snf_topk_hist_select_f32: null pointer");
    SNF_REQUIRE(n >= 1 && n < 0x3fffffffll && stride >= 1, "snf_topk_hist_select_f32: bad n=%lld stride=%lld", (long long)n,
                (long long)stride);
    SNF_REQUIRE(k >= 1 && k <= n && k <= RS_MAXK, "snf_topk_hist_select_f32: need 1 <= k <= min(n, %d) (k=%d n=%lld)", RS_MAXK, k,
                (long long)n);
    SNF_REQUIRE((reinterpret_cast<uintptr_t>(state) & 15) == 0, "snf_topk_hist_select_f32: state must be 16-byte aligned");
    hipStream_t s = snf::as_stream(stream);
    // scratch state: clear the counted part, count the first digit of every score, select
    snf::SelectorState* st = reinterpret_cast<snf::SelectorState*>(state);
    if (hipMemsetAsync(st, 0, offsetof(snf::SelectorState, win), s) != hipSuccess) return snf::check_launch("hipMemsetAsync");
    const unsigned grid = (unsigned)((n + SEL_SLICE - 1) / SEL_SLICE);
    hipLaunchKernelGGL(sel_hist_kernel, dim3(grid), dim3(1024), 0, s, scores, n, stride, st);
    int rc = snf::check_launch("sel_hist_kernel");
    if (rc) return rc;
    return launch_select(scores, n, stride, k, idx_out, st, s);
}

int snf_topk_f32(const float* scores, int64_t n, int64_t stride, int k, int64_t* idx_out, void* workspace,
                 size_t workspace_bytes, snf_stream_t stream) {
    SNF_REQUIRE(scores && idx_out, "snf_topk_f32: null pointer");
    SNF_REQUIRE(n >= 1 && stride >= 1, "snf_topk_f32: bad n=%lld stride=%lld", (long long)n, (long long)stride);
    SNF_REQUIRE(k >= 1 && k <= n, "snf_topk_f32: need 1 <= k <= n (k=%d n=%lld)", k, (long long)n);
    SNF_REQUIRE(k <= SNF_TOPK_MAX_K, "snf_topk_f32: k=%d exceeds SNF_TOPK_MAX_K=%d", k, SNF_TOPK_MAX_K);
    SNF_REQUIRE(n < 0x3fffffffll, "snf_topk_f32: n too large");
    hipStream_t s = snf::as_stream(stream);
    SNF_REQUIRE(k <= RS_MAXK, "snf_topk_f32: k=%d exceeds %d", k, RS_MAXK);
    if (n > TOPK_SINGLE_WG_MAX_N && workspace && workspace_bytes >= sizeof(snf::SelectorState) &&
        (reinterpret_cast<uintptr_t>(workspace) & 15) == 0)
        return snf_topk_hist_select_f32(scores, n, stride, k, idx_out, workspace, stream);
    if (n <= 1024 * 8)
        hipLaunchKernelGGL(topk_radix_kernel<8>, dim3(1), dim3(1024), 0, s, scores, n, stride, k, idx_out);
    else if (n <= 1024 * 16)
        hipLaunchKernelGGL(topk_radix_kernel<16>, dim3(1), dim3(1024), 0, s, scores, n, stride, k, idx_out);
    else if (n <= 1024 * 32)
        hipLaunchKernelGGL(topk_radix_kernel<32>, dim3(1), dim3(1024), 0, s, scores, n, stride, k, idx_out);
    else if (n <= 1024 * 64)
        hipLaunchKernelGGL(topk_radix_kernel<64>, dim3(1), dim3(1024), 0, s, scores, n, stride, k, idx_out);
    else
        hipLaunchKernelGGL(topk_radix_kernel<0>, dim3(1), dim3(1024), 0, s, scores, n, stride, k, idx_out);
    return snf::check_launch("topk_radix_kernel");
}

// top-k of every bag of a packed score vector in ONE launch (varlen path).  offsets [bags + 1] int64 in DEVICE memory, max_n =
// the longest bag (picks the register budget).  idx_out [bags, k]: indices inside the bag; a bag with n_b < k scores fills only
// its first n_b entries (all of its rows, ordered).
int snf_topk_segmented_f32(const float* scores, const int64_t* offsets_dev, int bags, int64_t max_n, int k, int64_t* idx_out,
                           snf_stream_t stream) {
    SNF_REQUIRE(scores && offsets_dev && idx_out, "snf_topk_segmented_f32: null pointer");
    SNF_REQUIRE(bags >= 1 && max_n >= 1 && max_n < 0x3fffffffll, "snf_topk_segmented_f32: bad bags=%d max_n=%lld", bags,
                (long long)max_n);
    SNF_REQUIRE(k >= 1 && k <= RS_MAXK, "snf_topk_segmented_f32: need 1 <= k <= %d (k=%d)", RS_MAXK, k);
    hipStream_t s = snf::as_stream(stream);
    const dim3 grid((unsigned)bags), wg(1024);
    if (max_n <= 1024 * 8)
        hipLaunchKernelGGL(topk_radix_segmented_kernel<8>, grid, wg, 0, s, scores, offsets_dev, k, idx_out);
    else if (max_n <= 1024 * 16)
        hipLaunchKernelGGL(topk_radix_segmented_kernel<16>, grid, wg, 0, s, scores, offsets_dev, k, idx_out);
    else if (max_n <= 1024 * 32)
        hipLaunchKernelGGL(topk_radix_segmented_kernel<32>, grid, wg, 0, s, scores, offsets_dev, k, idx_out);
    else if (max_n <= 1024 * 64)
        hipLaunchKernelGGL(topk_radix_segmented_kernel<64>, grid, wg, 0, s, scores, offsets_dev, k, idx_out);
    else
        hipLaunchKernelGGL(topk_radix_segmented_kernel<0>, grid, wg, 0, s, scores, offsets_dev, k, idx_out);
    return snf::check_launch("topk_radix_segmented_kernel");
}

int snf_topk_gather_f32(const float* scores, int64_t n, int64_t stride, int k, int64_t* idx_out, const float* x, int d,
                        float* xs, void* workspace, size_t workspace_bytes, snf_stream_t stream) {
    int rc = snf_topk_f32(scores, n, stride, k, idx_out, workspace, workspace_bytes, stream);
    if (rc) return rc;
    if (x && xs) rc = snf_gather_rows_f32(x, n, d, idx_out, k, xs, stream);
    return rc;
}

}  // extern "C"
