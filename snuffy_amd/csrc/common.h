// Shared host/device helpers for libsnuffy_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/snuffy_hip.h"

namespace snf {

void set_error(const char* fmt, ...);
// Returns SNF_OK or SNF_ELAUNCH (and records the HIP error string) after a kernel launch.
int check_launch(const char* what);
int cu_count();
unsigned long long device_bit();
// gemm.hip: x [r, k] w [c, k]^T + bias written as the Kp fragment image of sparse_attn_x3p.hip (see SkinnyFrag there)
// idx (nullable): input row i is row idx[i] of x [n_rows, ldx]; xs (nullable) receives the gathered rows, map (nullable) the row -> slot map
int skinny_linear_x3_kpfrag(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int r, int c, int k, int dk,
                            int chunk_size, int64_t chunk_stride, float c_exp, void* frag, hipStream_t s, const int64_t* idx = nullptr,
                            int64_t n_rows = 0, float* xs = nullptr, int64_t ldxs = 0, int32_t* map = nullptr);

static inline hipStream_t as_stream(snf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace snf

#define SNF_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            snf::set_error(__VA_ARGS__); \
            return SNF_EINVAL;          \
        }                               \
    } while (0)

#define SNF_WAVE 64

// ---- device helpers -------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// round-to-nearest-even f32 -> bf16 bits (NaN preserved as quiet NaN)
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned int)b) << 16); }
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    return (unsigned int)f32_to_bf16_bits(lo) | ((unsigned int)f32_to_bf16_bits(hi) << 16);
}
