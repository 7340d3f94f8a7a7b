// K3: device-side random patch share -- the opt-in "fast mode" of EncoderLayer's random selection (reference
// snuffy.py:136-147: np.random.choice(remaining, k2, replace=False) on the host, behind a device -> host copy of the top rows).
//
// Every row gets a 30-bit random key from Philox4x32-10 (csrc/philox.h: key = seed, counter = (row / 4, offset), element row & 3 of
// the four outputs), written as the POSITIVE FINITE float with that bit pattern, so that float order == key order; the rows already
// selected by the critic (top) get -1.  The k2 largest keys (snf_topk_f32, ties by ascending row: 2^-30 per pair) are then a uniform
// sample without replacement of the remaining rows, in uniformly random order -- the distribution of the reference's draw, from a
// different stream (the reference's MT19937 draws stay the default: bit-exact parity mode).
// (seed, offset) come from a 16-byte device record so that a captured HIP graph draws fresh rows on every replay:
// snf_sampler_advance increments the offset on the device.  oracle/philox_ref.py: random_share_keys() is the host twin.
#include "common.h"
#include "philox.h"

namespace {

__global__ __launch_bounds__(256) void sampler_advance_kernel(unsigned long long* state) {
    if (threadIdx.x == 0 && blockIdx.x == 0) state[1] += 1ull;
}

__global__ __launch_bounds__(256) void sampler_keys_kernel(const unsigned long long* __restrict__ state, unsigned long long layer,
                                                            int64_t n, float* __restrict__ keys) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;      // rows 4 g .. 4 g + 3
    if (4 * g >= n) return;
    const unsigned long long seed = state[0], off = state[1] + (layer << 48);   // layers of one forward draw from disjoint streams
    const snf::philox_u4 r = snf::philox4x32_10(snf::philox_u4{(unsigned)g, (unsigned)((unsigned long long)g >> 32), (unsigned)off,
                                                               (unsigned)(off >> 32)}, (unsigned)seed, (unsigned)(seed >> 32));
    float* dst = keys + 4 * g;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (4 * g + e < n) dst[e] = __uint_as_float(r[e] >> 2);      // 0 .. 2^30 - 1: bit patterns of non-negative finite floats
}

__global__ __launch_bounds__(256) void sampler_exclude_kernel(const int64_t* __restrict__ rows, int k, int64_t n, float* __restrict__ keys) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < k) {
        const int64_t r = rows[i];
        if (r >= 0 && r < n) keys[r] = -1.f;
    }
}

}  // namespace

extern "C" {

int snf_sampler_advance(void* state, snf_stream_t stream) {
    SNF_REQUIRE(state && (reinterpret_cast<uintptr_t>(state) & 7) == 0, "snf_sampler_advance: null / unaligned state");
    hipLaunchKernelGGL(sampler_advance_kernel, dim3(1), dim3(64), 0, snf::as_stream(stream), reinterpret_cast<unsigned long long*>(state));
    return snf::check_launch("sampler_advance_kernel");
}

int snf_random_share_keys_f32(const void* state, int layer, int64_t n, const int64_t* exclude_rows, int n_exclude, float* keys,
                              snf_stream_t stream) {
    SNF_REQUIRE(state && keys && (reinterpret_cast<uintptr_t>(state) & 7) == 0, "snf_random_share_keys_f32: null / unaligned pointer");
    SNF_REQUIRE(n >= 1 && n < (1ll << 40) && layer >= 0 && layer < 4096 && n_exclude >= 0 && (n_exclude == 0 || exclude_rows),
                "snf_random_share_keys_f32: bad arguments n=%lld layer=%d n_exclude=%d", (long long)n, layer, n_exclude);
    hipStream_t s = snf::as_stream(stream);
    const int64_t groups = (n + 3) / 4;
    hipLaunchKernelGGL(sampler_keys_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const unsigned long long*>(state), (unsigned long long)layer, n, keys);
    int rc = snf::check_launch("sampler_keys_kernel");
    if (rc || n_exclude == 0) return rc;
    hipLaunchKernelGGL(sampler_exclude_kernel, dim3((unsigned)((n_exclude + 255) / 256)), dim3(256), 0, s, exclude_rows, n_exclude, n, keys);
    return snf::check_launch("sampler_exclude_kernel");
}

}  // extern "C"
