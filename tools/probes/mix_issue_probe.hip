// One wave per SIMD: cycles of a repeating group "1 MFMA (32x32x16 bf16) + K fillers" for K VALU / LDS-read fillers.
// Answers: what does a filler cost beside an MFMA, and where is the knee?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define MF(i) "v_mfma_f32_32x32x16_bf16 %" #i ", %8, %9, %" #i "\n"
#define F1 "v_fma_f32 %4, %4, %10, %10\n"
#define F2 F1 "v_fma_f32 %5, %5, %10, %10\n"
#define F4 F2 "v_fma_f32 %6, %6, %10, %10\nv_fma_f32 %7, %7, %10, %10\n"
#define E1 "v_exp_f32 %4, %4\n"
#define L1 "ds_read_b64 %11, %12\n"
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

#define PROBE(NAME, FILL, NFILL)                                                                                  \
    __global__ __launch_bounds__(256, 1) void NAME(unsigned long long* out, float seed) {                         \
        __shared__ float lds[1024];                                                                               \
        lds[threadIdx.x] = seed;                                                                                  \
        f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};                                                                \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, cc = 0.5f;                                  \
        u32x4 a = {threadIdx.x, 1, 2, 3}, b = {4, 5, 6, threadIdx.x};                                             \
        unsigned long long ld = 0;                                                                                \
        unsigned addr = (threadIdx.x & 63) * 8;                                                                   \
        __syncthreads();                                                                                          \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                                     \
        asm volatile(REP16(MF(0) FILL MF(1) FILL MF(2) FILL MF(3) FILL) "s_waitcnt lgkmcnt(0)\n"                  \
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)             \
                     : "v"(a), "v"(b), "v"(cc), "v"(ld), "v"(addr));                                              \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                                     \
        if ((threadIdx.x & 63) == 0) { out[0] = t1 - t0; out[1] = NFILL; }                                        \
        if (c0[0] + c1[1] + c2[2] + c3[3] + a0 + a1 + a2 + a3 + (float)ld == 12345.f) out[2] = 1;                 \
    }
PROBE(m_0, "", 0)
PROBE(m_f2, F2, 2)
PROBE(m_f4, F4, 4)
PROBE(m_f6, F4 F2, 6)
PROBE(m_f8, F4 F4, 8)
PROBE(m_f12, F4 F4 F4, 12)
PROBE(m_f4e2, F4 E1 E1, 6)
PROBE(m_f4l2, F4 L1 L1, 6)
PROBE(m_f6l2, F4 F2 L1 L1, 8)
#define RUN(NAME)                                                                                  \
    {                                                                                              \
        for (int it = 0; it < 3; ++it) NAME<<<256, 256>>>(d, 1.0f);                                \
        hipDeviceSynchronize();                                                                    \
        unsigned long long h[3];                                                                   \
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);                                         \
        printf("%-8s %2llu fillers per MFMA: %6.1f cycles per group\n", #NAME, h[1], h[0] / 64.0); \
    }
int main() {
    unsigned long long* d;
    hipMalloc(&d, 64);
    RUN(m_0) RUN(m_f2) RUN(m_f4) RUN(m_f6) RUN(m_f8) RUN(m_f12) RUN(m_f4e2) RUN(m_f4l2) RUN(m_f6l2)
    return 0;
}
