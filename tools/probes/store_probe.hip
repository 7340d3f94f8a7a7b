// Store-pattern probe (round 6): what does a GEMM epilogue's store burst cost per CU, by the shape of one wave-instruction?
// 256 workgroups x 512 threads; workgroup b writes the 256 x 256 bf16 tiles b, b + 256, .. of a [M, N] matrix.
//   pattern 0: the GEMM's own -- a wave owns 128 rows x 64 columns; an instruction = 16 rows x 64 bytes (lane: row l & 15, 16-byte chunk l >> 4)
//   pattern 1: same region, an instruction = 8 rows x 128 bytes (lane: row l >> 3, chunk l & 7): full 128-byte lines
//   pattern 2: a wave owns 32 rows x 256 columns; an instruction = 2 rows x 512 bytes (lane: row l >> 5, chunk l & 31)
//   pattern 3: pattern 0 with nontemporal stores
// hipcc --offload-arch=gfx950 -O3 -o store_probe tools/probes/store_probe.hip && ./store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int PAT>
__global__ __launch_bounds__(512) void store_kernel(unsigned short* out, int m, int n, int tiles_n, int ntiles, int spin) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 2, wc = w & 3;
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tm = t / tiles_n, tn = t - tm * tiles_n;
        // something between the bursts (the main loop's place): spin cycles of sleep
        for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(16);
        if (PAT == 0 || PAT == 3) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = tm * 256 + 128 * wr + 16 * mi + (lane & 15);
                    const int col = tn * 256 + 64 * wc + 32 * h + 8 * (lane >> 4);
                    if (row < m && col + 8 <= n) {
                        u32x4* dst = reinterpret_cast<u32x4*>(out + (size_t)row * n + col);
                        if (PAT == 3) __builtin_nontemporal_store(v, dst); else *dst = v;
                    }
                }
        } else if (PAT == 1) {
#pragma unroll
            for (int mi = 0; mi < 16; ++mi) {
                const int row = tm * 256 + 128 * wr + 8 * mi + (lane >> 3);
                const int col = tn * 256 + 64 * wc + 8 * (lane & 7);
                if (row < m && col + 8 <= n) *reinterpret_cast<u32x4*>(out + (size_t)row * n + col) = v;
            }
        } else {
#pragma unroll
            for (int mi = 0; mi < 16; ++mi) {
                const int row = tm * 256 + 32 * w + 2 * mi + (lane >> 5);
                const int col = tn * 256 + 8 * (lane & 31);
                if (row < m && col + 8 <= n) *reinterpret_cast<u32x4*>(out + (size_t)row * n + col) = v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the epilogue's stores have to be done before the stream goes on
        __syncthreads();
    }
}

template <int PAT>
float run(unsigned short* out, int m, int n, int spin, int reps) {
    const int tiles_m = (m + 255) / 256, tiles_n = (n + 255) / 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(store_kernel<PAT>, dim3(256), dim3(512), 0, 0, out, m, n, tiles_n, tiles_m * tiles_n, spin);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(store_kernel<PAT>, dim3(256), dim3(512), 0, 0, out, m, n, tiles_n, tiles_m * tiles_n, spin);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    const int m = 100864;
    for (int n : {1152, 1536, 384}) {
        unsigned short* out;
        hipMalloc(&out, (size_t)m * n * 2);
        for (int spin : {0, 8}) {
            const float t0 = run<0>(out, m, n, spin, 10), t1 = run<1>(out, m, n, spin, 10), t2 = run<2>(out, m, n, spin, 10), t3 = run<3>(out, m, n, spin, 10);
            const double mb = (double)m * n * 2 / 1e6;
            printf("n=%4d spin=%d  %.0f MB:  16x64B %.1f us (%.2f TB/s) | 8x128B %.1f us (%.2f) | 2x512B %.1f us (%.2f) | 16x64B nt %.1f us (%.2f)\n", n, spin, mb,
                   t0, mb / t0 / 1e6 * 1e6 / 1e6, t1, mb / t1, t2, mb / t2, t3, mb / t3);
        }
        hipFree(out);
    }
    return 0;
}
