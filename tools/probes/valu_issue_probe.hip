// Single-wave issue-rate probe for gfx950: cycles per instruction (s_memtime) of short VALU patterns, one wave on one SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/valu_issue_probe.hip -o tools/probes/valu_issue_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define PROBE(NAME, BODY, NINSTR)                                                                   \
    __global__ void NAME(unsigned long long* out, float seed) {                                     \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, \
              a6 = seed + 6, a7 = seed + 7, c = 0.5f;                                               \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                       \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                       \
        asm volatile(REP64(BODY) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                     : "v"(c));                                                                     \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                       \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                       \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = NINSTR * 64; }                           \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) out[2] = 1;                           \
    }

// dependent chain of fma
PROBE(fma_dep, "v_fma_f32 %0, %0, %8, %8\n", 1)
// 2 / 4 / 8 independent chains
PROBE(fma_ind2, "v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %1, %1, %8, %8\n", 2)
PROBE(fma_ind4, "v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %1, %1, %8, %8\nv_fma_f32 %2, %2, %8, %8\nv_fma_f32 %3, %3, %8, %8\n", 4)
PROBE(fma_ind8, "v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %1, %1, %8, %8\nv_fma_f32 %2, %2, %8, %8\nv_fma_f32 %3, %3, %8, %8\n"
                "v_fma_f32 %4, %4, %8, %8\nv_fma_f32 %5, %5, %8, %8\nv_fma_f32 %6, %6, %8, %8\nv_fma_f32 %7, %7, %8, %8\n", 8)
PROBE(exp_dep, "v_exp_f32 %0, %0\n", 1)
PROBE(exp_ind4, "v_exp_f32 %0, %0\nv_exp_f32 %1, %1\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\n", 4)
PROBE(exp_ind8, "v_exp_f32 %0, %0\nv_exp_f32 %1, %1\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\n"
                "v_exp_f32 %4, %4\nv_exp_f32 %5, %5\nv_exp_f32 %6, %6\nv_exp_f32 %7, %7\n", 8)
// softmax element pattern: fma -> exp -> add, 1 / 2 / 4 elements interleaved by stage
PROBE(sm_1, "v_fma_f32 %0, %0, %8, %8\nv_exp_f32 %0, %0\nv_add_f32 %4, %4, %0\n", 3)
PROBE(sm_2, "v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %1, %1, %8, %8\nv_exp_f32 %0, %0\nv_exp_f32 %1, %1\n"
            "v_add_f32 %4, %4, %0\nv_add_f32 %5, %5, %1\n", 6)
PROBE(sm_4, "v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %1, %1, %8, %8\nv_fma_f32 %2, %2, %8, %8\nv_fma_f32 %3, %3, %8, %8\n"
            "v_exp_f32 %0, %0\nv_exp_f32 %1, %1\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\n"
            "v_add_f32 %4, %4, %0\nv_add_f32 %5, %5, %1\nv_add_f32 %6, %6, %2\nv_add_f32 %7, %7, %3\n", 12)
PROBE(max3_ind4, "v_max3_f32 %0, %0, %8, %4\nv_max3_f32 %1, %1, %8, %5\nv_max3_f32 %2, %2, %8, %6\nv_max3_f32 %3, %3, %8, %7\n", 4)
PROBE(cvt_ind4, "v_cvt_pk_bf16_f32 %0, %4, %5\nv_cvt_pk_bf16_f32 %1, %5, %6\nv_cvt_pk_bf16_f32 %2, %6, %7\nv_cvt_pk_bf16_f32 %3, %7, %4\n", 4)
PROBE(snop0, "s_nop 0\n", 1)
PROBE(snop1, "s_nop 1\n", 1)

__global__ void pkmul_ind4(unsigned long long* out, float seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {seed, seed + 1}, a1 = {seed + 2, seed + 3}, a2 = {seed + 4, seed + 5}, a3 = {seed + 6, seed + 7}, c = {0.5f, 0.5f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    asm volatile(REP64("v_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = 4 * 64; }
    if (a0[0] + a1[0] + a2[1] + a3[1] == 12345.f) out[2] = 1;
}

#define RUN(NAME)                                                                          \
    {                                                                                      \
        double r[3];                                                                       \
        const int cfg[3][2] = {{1, 64}, {1, 256}, {256, 256}};   /* one wave | one wave per SIMD | every CU busy */ \
        for (int c = 0; c < 3; ++c) {                                                      \
            for (int it = 0; it < 3; ++it) NAME<<<cfg[c][0], cfg[c][1]>>>(d, 1.0f);        \
            hipDeviceSynchronize();                                                        \
            unsigned long long h[3];                                                       \
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);                             \
            r[c] = (double)h[0] / h[1];                                                    \
        }                                                                                  \
        printf("%-12s cyc/instr: 1 wave %5.2f | 4 waves (1 per SIMD) %5.2f | 256 WG x 4 waves %5.2f\n", #NAME, r[0], r[1], r[2]); \
    }

int main() {
    unsigned long long* d;
    hipMalloc(&d, 64);
    RUN(fma_dep) RUN(fma_ind2) RUN(fma_ind4) RUN(fma_ind8) RUN(exp_dep) RUN(exp_ind4) RUN(exp_ind8)
    RUN(sm_1) RUN(sm_2) RUN(sm_4) RUN(max3_ind4) RUN(cvt_ind4) RUN(pkmul_ind4) RUN(snop0) RUN(snop1)
    return 0;
}
