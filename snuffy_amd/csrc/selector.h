// Fused selector, shared pieces: the critic pass (rowops.hip) accumulates the FIRST radix digit of every score into a small
// global histogram while it writes the scores; the select kernel (topk.hip) starts from that histogram.
#pragma once
#include "common.h"

namespace snf {

constexpr int SEL_BINS = 2048;       // first radix digit = the top 11 bits of the orderable key (sign, exponent, 2 mantissa bits)
constexpr int SEL_REPL = 4;          // global histogram replicas (workgroup id & 3): real scores crowd into a few dozen bins, and
                                     // device-scope atomics to ONE address serialise at ~11 ns each
constexpr int SEL_CAND_CAP = 4096;   // keys of the threshold bin the multi-workgroup path holds; more -> single-workgroup path
constexpr int SEL_MAXK = 2048;

// Lives in caller-owned device memory (snf_selector_state_bytes), zeroed ONCE by the caller; every select call leaves the
// counted part zeroed again (the last workgroup clears it after all others have arrived), so no memset rides in the bag
// pipeline.  A histogram whose total is not n (an aborted call left it dirty) is detected and the exact single-workgroup
// selection on the scores themselves takes over.
struct SelectorState {
    unsigned int hist[SEL_REPL * SEL_BINS];
    unsigned int n_win, n_cand, arrive, fallbacks;   // fallbacks: diagnostics (counts selections that left the fast path)
    unsigned long long win[SEL_MAXK];                // (key << 32 | ~index) of every score above the threshold bin
    unsigned long long cand[SEL_CAND_CAP];           // ... of every score inside it
};

// float -> 32-bit key whose unsigned order is the DESCENDING selection order's inverse: larger key == earlier in the output
__device__ __forceinline__ unsigned int orderable_desc(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;  // NaN sorts first (torch semantics)
    if (u == 0x80000000u) u = 0u;                               // -0.0 == +0.0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

}  // namespace snf
