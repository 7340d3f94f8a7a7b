// MFMA issue-rate probe for gfx950: cycles per v_mfma_f32_32x32x16_bf16 (s_memtime), accumulators in VGPRs vs AGPRs,
// 7 independent accumulators round-robin (the GEMM1 pattern), one wave per SIMD (256 threads) or a single wave.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define M7V "v_mfma_f32_32x32x16_bf16 %0, %7, %8, %0\nv_mfma_f32_32x32x16_bf16 %1, %7, %8, %1\nv_mfma_f32_32x32x16_bf16 %2, %7, %8, %2\n" \
            "v_mfma_f32_32x32x16_bf16 %3, %7, %8, %3\nv_mfma_f32_32x32x16_bf16 %4, %7, %8, %4\nv_mfma_f32_32x32x16_bf16 %5, %7, %8, %5\n" \
            "v_mfma_f32_32x32x16_bf16 %6, %7, %8, %6\n"
#define REP8(x) x x x x x x x x

template <bool AGPR>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out) {
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {};
    u32x4 a = {threadIdx.x, 1, 2, 3}, b = {4, 5, 6, threadIdx.x};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (AGPR)
        asm volatile(REP8(M7V) : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3), "+a"(c4), "+a"(c5), "+a"(c6) : "v"(a), "v"(b));
    else
        asm volatile(REP8(M7V) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6) : "v"(a), "v"(b));
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
    float s = c0[0] + c1[1] + c2[2] + c3[3] + c4[4] + c5[5] + c6[6];
    if (s == 12345.f) out[8] = 1;
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 128);
    for (int threads : {64, 256}) {
        for (int ag = 0; ag < 2; ++ag) {
            unsigned long long h[4];
            for (int it = 0; it < 3; ++it) {
                if (ag) probe<true><<<1, threads>>>(d); else probe<false><<<1, threads>>>(d);
            }
            hipDeviceSynchronize();
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("%3d threads, acc in %s: %5llu cycles / 56 MFMA = %.1f cyc each\n", threads, ag ? "AGPR" : "VGPR", h[0], h[0] / 56.0);
        }
    }
    // many workgroups: all CUs busy (power / clock effects)
    for (int ag = 0; ag < 2; ++ag) {
        unsigned long long h[4];
        for (int it = 0; it < 3; ++it) {
            if (ag) probe<true><<<256, 256>>>(d); else probe<false><<<256, 256>>>(d);
        }
        hipDeviceSynchronize();
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("256 WG x 256 threads, acc in %s: %5llu cycles / 56 MFMA = %.1f cyc each (some WG)\n", ag ? "AGPR" : "VGPR", h[0], h[0] / 56.0);
    }
    return 0;
}
