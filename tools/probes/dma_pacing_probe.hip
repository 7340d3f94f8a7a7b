// LDS-DMA pacing probe: the config-B Q|V operand stream (256 workgroups, XCD-contiguous tile ranges, 256 A rows + 256 W rows per
// tile, 128 bytes per row and step = 64 KiB per step, 128-byte segments) under the things a GEMM loop does between its DMA
// instructions: barriers, LDS fragment reads, MFMA bursts, shallower in-flight depth.  Which of them collapses the 23 TB/s of
// the pure stream to the 5-8 TB/s seen inside the GEMM kernels?
// build: hipcc --offload-arch=gfx950 -O3 -o dma_pacing_probe.bin dma_pacing_probe.hip ; run: ./dma_pacing_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// GROUPS: the 8 DMA instructions of a wave and step are issued in GROUPS batches with a workgroup barrier after each batch
// INFL:   DMA instructions left in flight by the per-step wait (8 = one whole step ahead, 0 = drain)
// READS:  ds_read_b128 per wave and step (fragment-read traffic), MFMAS: v_mfma_f32_16x16x32_bf16 per wave and step
template <int GROUPS, int INFL, int READS, int MFMAS, int NBUF>
__global__ __launch_bounds__(512) void pace_kernel(const char* a, const char* w, int kbytes, int tiles_n, int per_xcd,
                                                   int wgs_per_xcd, int ntiles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STEPB = 128, STEP_BYTES = 512 * STEPB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int t_lo = xcd * per_xcd, t_hi = min(t_lo + per_xcd, ntiles);
    const int nsteps = kbytes / STEPB;
    int buf = 0;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fr[4];
    for (int i = 0; i < 4; ++i) fr[i] = __builtin_bit_cast(bf16x8, u32x4{1u, 2u, 3u, (unsigned)lane});
    for (int tile = t_lo + j; tile < t_hi; tile += wgs_per_xcd) {
        const int tm = tile / tiles_n, tn = tile % tiles_n;
        for (int s = 0; s < nsteps; ++s) {
            char* dst = smem + buf * STEP_BYTES;
#pragma unroll
            for (int gq = 0; gq < GROUPS; ++gq) {
#pragma unroll
                for (int i = gq * (8 / GROUPS); i < (gq + 1) * (8 / GROUPS); ++i) {
                    const int instr = wave * 8 + i;                          // 64 pieces of 8 rows x 128 B
                    const int row = instr * 8 + (lane >> 3);                 // 0..511: A rows then W rows
                    const int off = s * STEPB + (lane & 7) * 16;
                    const char* src = row < 256 ? a + (size_t)(tm * 256 + row) * kbytes + off
                                                : w + (size_t)(tn * 256 + row - 256) * kbytes + off;
                    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(dst + instr * 1024), 16, 0, 0);
                }
                if (GROUPS > 1 && gq + 1 < GROUPS) __builtin_amdgcn_s_barrier();
            }
            if constexpr (READS > 0) {
                const char* rb = smem + (buf ^ (NBUF > 1 ? 1 : 0)) * STEP_BYTES;
#pragma unroll
                for (int i = 0; i < READS; ++i)
                    fr[i & 3] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(rb + ((wave * 32 + i) & 63) * 1024 + lane * 16));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if constexpr (MFMAS > 0) {
#pragma unroll
                for (int i = 0; i < MFMAS; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[i & 3], fr[(i + 1) & 3], acc[i & 7], 0, 0, 0);
            }
            buf = buf + 1 == NBUF ? 0 : buf + 1;
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(INFL) : "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < 8; ++i) t += acc[i][0];
    if (t == 123.456f) sink[blockIdx.x] = t;
}

template <int GROUPS, int INFL, int READS, int MFMAS, int NBUF>
void run(const char* name, const char* a, const char* w, float* sink) {
    const int m = 32768, n = 1536, kbytes = 1536;
    const int tiles_m = m / 256, tiles_n = n / 256, ntiles = tiles_m * tiles_n;
    const int grid = 256, per_xcd = (ntiles + 7) / 8, wgs_per_xcd = grid / 8;
    const int lds = NBUF * 512 * 128;
    auto kern = pace_kernel<GROUPS, INFL, READS, MFMAS, NBUF>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, a, w, kbytes, tiles_n, per_xcd, wgs_per_xcd, ntiles, sink);
    hipEventRecord(e0);
    const int iters = 20;
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, a, w, kbytes, tiles_n, per_xcd, wgs_per_xcd, ntiles, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters;
    const double bytes = (double)ntiles * 512.0 * kbytes;
    printf("%-64s %7.1f us  L2->LDS %6.2f TB/s  (%s)\n", name, us, bytes / us / 1e6, hipGetErrorString(hipGetLastError()));
}

int main() {
    char *a, *w; float* sink;
    hipMalloc(&a, (size_t)32768 * 1536); hipMalloc(&w, (size_t)1536 * 1536); hipMalloc(&sink, 4096);
    hipMemset(a, 1, (size_t)32768 * 1536); hipMemset(w, 1, (size_t)1536 * 1536);
    run<1, 8, 0, 0, 2>("pure stream: 8 DMA back to back, one step in flight", a, w, sink);
    run<1, 0, 0, 0, 2>("drained every step (vmcnt 0)", a, w, sink);
    run<1, 4, 0, 0, 2>("half a step in flight (vmcnt 4)", a, w, sink);
    run<2, 8, 0, 0, 2>("2 batches of 4 DMA, barrier between", a, w, sink);
    run<4, 8, 0, 0, 2>("4 batches of 2 DMA, barriers between", a, w, sink);
    run<1, 8, 24, 0, 2>("+ 24 ds_read_b128 per wave and step", a, w, sink);
    run<1, 8, 0, 64, 2>("+ 64 MFMA per wave and step", a, w, sink);
    run<1, 8, 24, 64, 2>("+ 24 reads + 64 MFMA", a, w, sink);
    run<4, 8, 24, 64, 2>("4 batches + 24 reads + 64 MFMA", a, w, sink);
    run<1, 8, 24, 192, 2>("+ 24 reads + 192 MFMA (x3 one-pass ratio)", a, w, sink);
    // the CORRECT two-buffer dependency: step s + 1's DMA is issued first, flies under step s's reads + MFMAs, drained before the barrier
    run<1, 0, 24, 64, 2>("drained each step + 24 reads + 64 MFMA (correct 2-buffer loop)", a, w, sink);
    run<1, 0, 24, 192, 2>("drained each step + 24 reads + 192 MFMA", a, w, sink);
    run<1, 0, 12, 32, 2>("drained each step + 12 reads + 32 MFMA (32-deep steps' ratio, 64 KiB DMA)", a, w, sink);
    // random operand bytes instead of a constant fill (clock / power behaviour of the MFMA pipe)
    {
        unsigned* h = (unsigned*)malloc((size_t)32768 * 1536);
        unsigned x = 12345u;
        for (size_t i = 0; i < (size_t)32768 * 1536 / 4; ++i) { x = x * 1664525u + 1013904223u; h[i] = (x & 0x3f7f3f7fu) | 0x3c003c00u; }   // bf16 pairs in [0.008, 1)
        hipMemcpy(a, h, (size_t)32768 * 1536, hipMemcpyHostToDevice);
        hipMemcpy(w, h, (size_t)1536 * 1536, hipMemcpyHostToDevice);
        free(h);
    }
    run<1, 8, 0, 0, 2>("random data: pure stream", a, w, sink);
    run<1, 0, 24, 64, 2>("random data: drained + 24 reads + 64 MFMA", a, w, sink);
    run<1, 0, 24, 192, 2>("random data: drained + 24 reads + 192 MFMA", a, w, sink);
    return 0;
}
