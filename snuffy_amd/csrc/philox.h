// Counter-based dropout mask for the attention probabilities (reference snuffy.py:166-167: nn.Dropout(p) on p_attn).
//
// The mask is a pure function of (seed, offset, head a, row, key): Philox4x32-10 (Salmon et al., "Parallel random numbers: as
// easy as 1, 2, 3", SC'11) with key = seed and counter = (group index, offset), one call per group of 4 consecutive keys of a
// row.  The forward and backward kernels regenerate it in registers (it is never stored), and a host implementation
// (oracle/philox_ref.py) reproduces it bit for bit.
//   group index g = (a * n + row) * ceil(k / 4) + key / 4        counter = {g.lo, g.hi, offset.lo, offset.hi}
//   element key & 3 of the 4 outputs; keep iff output >= floor(p * 2^32); kept probabilities are scaled by 1 / (1 - p).
#pragma once
#include <stdint.h>

namespace snf {

struct DropoutState {
    unsigned seed_lo, seed_hi, off_lo, off_hi;
    unsigned thresh;   // floor(p * 2^32); 0 = dropout off
    float scale;       // 1 / (1 - p)
};

inline DropoutState make_dropout(float p, uint64_t seed, uint64_t offset) {
    DropoutState d;
    d.seed_lo = (unsigned)seed, d.seed_hi = (unsigned)(seed >> 32);
    d.off_lo = (unsigned)offset, d.off_hi = (unsigned)(offset >> 32);
    if (!(p > 0.f)) {
        d.thresh = 0u;
        d.scale = 1.f;
    } else {
        const double t = (double)p * 4294967296.0;
        d.thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
        d.scale = (float)(1.0 / (1.0 - (double)p));
    }
    return d;
}

#ifdef __HIPCC__
typedef __attribute__((ext_vector_type(4))) unsigned int philox_u4;
typedef __attribute__((ext_vector_type(4))) float philox_f4;

__device__ __forceinline__ philox_u4 philox4x32_10(philox_u4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        c = philox_u4{hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// multipliers (0 or 1 / (1 - p)) of the 4 keys key0 .. key0 + 3 (key0 a multiple of 4) of row `row` of head a
__device__ __forceinline__ philox_f4 dropout_mask4(const DropoutState& d, int a, int64_t n, int64_t row, int k, int key0) {
    const unsigned long long g = ((unsigned long long)a * (unsigned long long)n + (unsigned long long)row) *
                                     (unsigned long long)((k + 3) >> 2) + (unsigned long long)(key0 >> 2);
    const philox_u4 r = philox4x32_10(philox_u4{(unsigned)g, (unsigned)(g >> 32), d.off_lo, d.off_hi}, d.seed_lo, d.seed_hi);
    return philox_f4{r[0] >= d.thresh ? d.scale : 0.f, r[1] >= d.thresh ? d.scale : 0.f, r[2] >= d.thresh ? d.scale : 0.f,
                     r[3] >= d.thresh ? d.scale : 0.f};
}
#endif

}  // namespace snf
