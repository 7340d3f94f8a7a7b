// L2 -> LDS fill-rate probe: how fast can 256 workgroups stream GEMM operand tiles into LDS with global_load_lds_dwordx4, as a
// function of the contiguous bytes ONE wave-instruction takes from a matrix row (SEG = 64 / 128 / 256 B: 4 / 8 / 16 lanes per
// row)?  Emulates the Q|V GEMM of config B: A [32768, 768] bf16 (row pitch 1536 B), W [1536, 768]; workgroup w walks tiles
// like gemm.hip (XCD-contiguous ranges), per tile 256 A rows + 256 W rows, K in steps of STEPB bytes per row.
// build: hipcc --offload-arch=gfx950 -O3 -o fill_probe.bin fill_probe.hip ; run: ./fill_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int SEG, int STEPB, int NBUF>
__global__ __launch_bounds__(512) void fill_kernel(const char* a, const char* w, int m, int n, int kbytes, int tiles_n,
                                                   int per_xcd, int wgs_per_xcd, int ntiles, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS_PER_INSTR = 1024 / SEG;              // a wave-instruction moves 1 KiB
    constexpr int INSTR_PER_STEP = 512 * STEPB / 1024;      // A (256 rows) + W (256 rows) of one step, all waves together
    constexpr int PER_WAVE = INSTR_PER_STEP / 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int t_lo = xcd * per_xcd, t_hi = min(t_lo + per_xcd, ntiles);
    const int nsteps = kbytes / STEPB;
    int buf = 0;
    for (int tile = t_lo + j; tile < t_hi; tile += wgs_per_xcd) {
        const int tm = tile / tiles_n, tn = tile % tiles_n;
        for (int s = 0; s < nsteps; ++s) {
            char* dst = smem + buf * (512 * STEPB);
#pragma unroll
            for (int i = 0; i < PER_WAVE; ++i) {
                const int instr = wave * PER_WAVE + i;                 // 0 .. INSTR_PER_STEP-1
                const int chunk_per_row = STEPB / SEG;                 // instrs needed to cover one row's step bytes
                const int rowgrp = instr / chunk_per_row, part = instr % chunk_per_row;
                const int row = rowgrp * ROWS_PER_INSTR + lane / (SEG / 16);   // 0..511 : A rows then W rows
                const int off = s * STEPB + part * SEG + (lane % (SEG / 16)) * 16;
                const char* src = row < 256 ? a + (size_t)(tm * 256 + row) * kbytes + off
                                            : w + (size_t)(tn * 256 + row - 256) * kbytes + off;
                __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(dst + instr * 1024), 16, 0, 0);
            }
            buf = buf + 1 == NBUF ? 0 : buf + 1;
            // keep NBUF-1 steps in flight
            if constexpr (NBUF == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(3 * PER_WAVE) : "memory");
            if constexpr (NBUF == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(1 * PER_WAVE) : "memory");
            if constexpr (NBUF == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * PER_WAVE) : "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && smem[5] == 77) sink[blockIdx.x] = 1;
}

template <int SEG, int STEPB, int NBUF>
void run(const char* name, const char* a, const char* w, unsigned* sink) {
    const int m = 32768, n = 1536, kbytes = 1536;
    const int tiles_m = m / 256, tiles_n = n / 256, ntiles = tiles_m * tiles_n;
    const int grid = 256, per_xcd = (ntiles + 7) / 8, wgs_per_xcd = grid / 8;
    const int lds = NBUF * 512 * STEPB;
    auto kern = fill_kernel<SEG, STEPB, NBUF>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, a, w, m, n, kbytes, tiles_n, per_xcd, wgs_per_xcd, ntiles, sink);
    hipEventRecord(e0);
    const int iters = 20;
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, a, w, m, n, kbytes, tiles_n, per_xcd, wgs_per_xcd, ntiles, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters;
    const double bytes = (double)ntiles * 512.0 * kbytes;
    printf("%-34s %7.1f us  L2->LDS %6.2f TB/s  (%s)\n", name, us, bytes / us / 1e6, hipGetErrorString(hipGetLastError()));
}

int main() {
    char *a, *w; unsigned* sink;
    hipMalloc(&a, (size_t)32768 * 1536); hipMalloc(&w, (size_t)1536 * 1536); hipMalloc(&sink, 4096);
    hipMemset(a, 1, (size_t)32768 * 1536); hipMemset(w, 1, (size_t)1536 * 1536);
    run<64, 64, 4>("seg 64 B, step 64 B/row, ring 4", a, w, sink);
    run<64, 128, 2>("seg 64 B, step 128 B/row, ring 2", a, w, sink);
    run<128, 128, 2>("seg 128 B, step 128 B/row, ring 2", a, w, sink);
    run<128, 128, 3>("seg 128 B, step 128 B/row, ring 3 (wait)", a, w, sink);
    run<256, 256, 1>("seg 256 B, step 256 B/row, ring 1", a, w, sink);
    run<128, 256, 1>("seg 128 B, step 256 B/row, ring 1", a, w, sink);
    return 0;
}
