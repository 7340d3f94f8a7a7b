#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s16x4;
// every lane supplies its own byte address into a 16 KB LDS image filled with image[i] = i (u16)
__global__ void k(const int* addr, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short img[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) img[i] = (unsigned short)i;
    __syncthreads();
    const int a = addr[threadIdx.x];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + a / 2));
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = (unsigned short)v[e];
}
int main() {
    int* d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 64 * 4 * 2);
    for (int test = 0; test < 3; ++test) {
        std::vector<int> addr(64);
        for (int l = 0; l < 64; ++l) {
            if (test == 0) addr[l] = l * 8;                                        // lane-linear
            if (test == 1) addr[l] = ((l >> 4) * 4 + ((l & 15) >> 2)) * 448 + (l & 3) * 8;  // 4 rows x 32 B per group, row stride 448
            if (test == 2) addr[l] = (l * 37 % 64) * 8 + 1024;                     // arbitrary permutation
        }
        hipMemcpy(d_addr, addr.data(), 256, hipMemcpyHostToDevice);
        k<<<1, 64>>>(d_addr, d_out);
        std::vector<unsigned short> out(256);
        hipMemcpy(out.data(), d_out, 512, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
            // model: R[i][e] = D[4e + (i>>2)][i&3] inside each 16-lane group
            int g = l >> 4, i = l & 15;
            int src_lane = g * 16 + 4 * e + (i >> 2);
            int expect = addr[src_lane] / 2 + (i & 3);
            if (out[l * 4 + e] != expect) { if (bad < 8) printf("test %d lane %d e %d got %d expect %d\n", test, l, e, out[l*4+e], expect); ++bad; }
        }
        printf("test %d: %s (%d mismatches)\n", test, bad ? "MODEL WRONG" : "model ok", bad);
        if (test == 0) { for (int l = 0; l < 20; ++l) printf("lane %d: %d %d %d %d\n", l, out[l*4], out[l*4+1], out[l*4+2], out[l*4+3]); }
    }
    return 0;
}
