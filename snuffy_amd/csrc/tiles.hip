// Tile preprocessing on the device: what reference compute_feats.py:104-152,173-177 does per tile on DataLoader
// workers -- Resize(224) of a PIL image (bilinear with PIL's antialiasing box), ToTensor (uint8 -> float / 255), optional
// ImageNet normalisation -- for a whole batch of decoded uint8 tiles in one launch.
//
// The resize is Pillow's ImagingResample for 8-bit bands restated (Pillow is a dependency of the reference through
// torchvision; src/libImaging/Resample.c): two separable passes, horizontal then vertical, each a convolution with
// per-output-pixel integer coefficients (the double-precision triangle-filter weights scaled by 2^22 and rounded -- computed on
// the host, snuffy_amd/tiles.py), accumulated from 2^21 and shifted back with clamping to 0..255; the intermediate image is
// uint8, exactly as in Pillow.  Bit-exact against PIL-produced fixtures (tests/golden/f10_tiles.npz).
//
// One workgroup per (image, channel): the horizontally resized channel [H_in, W_out] lives in LDS.  Outputs (either may be
// null): the fp32 tensor [B, 3, H_out, W_out] the extractor's fp32 path takes, and / or the patch-embedding GEMM's A operand
// [B * P, 3 * ps * ps] bf16 (column order (c, i, j) = the flattening of the conv weight), so the bf16 path needs no
// separate patchify pass.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;   // arithmetic shift, as Pillow's table lookup index
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

struct TileParams {
    const unsigned char* img;   // [B, H, W, C] uint8 (C = 3: RGB as decoded)
    int b, h, w, c;
    int oh, ow;
    const int* hbounds;         // [ow][2] = (xmin, count)
    const int* hcoef;           // [ow][hks]
    int hks;
    const int* vbounds;         // [oh][2]
    const int* vcoef;           // [oh][vks]
    int vks;
    float mean[4], inv_std[4];  // (v / 255 - mean) / std when normalize, with the division done as in torch
    float std_[4];
    int normalize;
    float* out_f32;             // [B, C, oh, ow] or null
    unsigned short* cols;       // [B * (oh/ps) * (ow/ps), C * ps * ps] bf16 or null
    int ps;
};

__global__ __launch_bounds__(256) void tile_preprocess_kernel(TileParams P) {
    extern __shared__ unsigned char tmp[];   // [h][ow] uint8: the horizontally resized channel
    const int c = blockIdx.x, b = blockIdx.y;
    const unsigned char* src = P.img + (int64_t)b * P.h * P.w * P.c + c;
    for (int idx = threadIdx.x; idx < P.h * P.ow; idx += 256) {
        const int y = idx / P.ow, ox = idx - y * P.ow;
        const int xmin = P.hbounds[2 * ox], cnt = P.hbounds[2 * ox + 1];
        const int* k = P.hcoef + ox * P.hks;
        const unsigned char* line = src + ((int64_t)y * P.w + xmin) * P.c;
        int ss = 1 << (PRECISION_BITS - 1);
        for (int x = 0; x < cnt; ++x) ss += (int)line[x * P.c] * k[x];
        tmp[idx] = (unsigned char)clip8(ss);
    }
    __syncthreads();
    const int gw = P.ps ? P.ow / P.ps : 0, gh = P.ps ? P.oh / P.ps : 0;
    for (int idx = threadIdx.x; idx < P.oh * P.ow; idx += 256) {
        const int oy = idx / P.ow, ox = idx - oy * P.ow;
        const int ymin = P.vbounds[2 * oy], cnt = P.vbounds[2 * oy + 1];
        const int* k = P.vcoef + oy * P.vks;
        int ss = 1 << (PRECISION_BITS - 1);
        for (int y = 0; y < cnt; ++y) ss += (int)tmp[(ymin + y) * P.ow + ox] * k[y];
        float v = (float)clip8(ss) / 255.0f;                    // ToTensor
        if (P.normalize) v = (v - P.mean[c]) / P.std_[c];       // NormalizeImage
        if (P.out_f32) P.out_f32[(((int64_t)b * P.c + c) * P.oh + oy) * P.ow + ox] = v;
        if (P.cols) {
            const int gy = oy / P.ps, i = oy - gy * P.ps, gx = ox / P.ps, j = ox - gx * P.ps;
            const int64_t row = ((int64_t)b * gh + gy) * gw + gx;
            P.cols[row * ((int64_t)P.c * P.ps * P.ps) + (c * P.ps + i) * P.ps + j] = f32_to_bf16_bits(v);
        }
    }
}

}  // namespace

extern "C" int snf_tile_preprocess_u8(const void* img_u8, int b, int h, int w, int c, int oh, int ow, const int* hbounds,
                                      const int* hcoef, int hks, const int* vbounds, const int* vcoef, int vks, int normalize,
                                      const float* mean, const float* std_, float* out_f32, void* cols_bf16, int patch,
                                      snf_stream_t stream) {
    SNF_REQUIRE(img_u8 && hbounds && hcoef && vbounds && vcoef && (out_f32 || cols_bf16), "snf_tile_preprocess_u8: null pointer");
    SNF_REQUIRE(b >= 1 && h >= 1 && w >= 1 && c >= 1 && c <= 4 && oh >= 1 && ow >= 1 && hks >= 1 && vks >= 1,
                "snf_tile_preprocess_u8: bad shape");
    SNF_REQUIRE(!normalize || (mean && std_), "snf_tile_preprocess_u8: normalisation needs mean and std (host arrays of c floats)");
    SNF_REQUIRE(!cols_bf16 || (patch >= 1 && oh % patch == 0 && ow % patch == 0),
                "snf_tile_preprocess_u8: output %dx%d is not a multiple of the patch size %d", oh, ow, patch);
    const size_t lds = (size_t)h * ow;
    if (lds > 160 * 1024) {
        snf::set_error("snf_tile_preprocess_u8: a %d x %d intermediate channel does not fit the 160 KiB LDS", h, ow);
        return SNF_EUNSUPPORTED;
    }
    TileParams P;
    P.img = reinterpret_cast<const unsigned char*>(img_u8);
    P.b = b, P.h = h, P.w = w, P.c = c, P.oh = oh, P.ow = ow;
    P.hbounds = hbounds, P.hcoef = hcoef, P.hks = hks, P.vbounds = vbounds, P.vcoef = vcoef, P.vks = vks;
    for (int i = 0; i < 4; ++i) {
        P.mean[i] = (normalize && i < c) ? mean[i] : 0.f;
        P.std_[i] = (normalize && i < c) ? std_[i] : 1.f;
        P.inv_std[i] = 1.f / P.std_[i];
    }
    P.normalize = normalize;
    P.out_f32 = out_f32;
    P.cols = reinterpret_cast<unsigned short*>(cols_bf16);
    P.ps = cols_bf16 ? patch : 0;
    auto kern = tile_preprocess_kernel;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            snf::set_error("snf_tile_preprocess_u8: cannot reserve %zu bytes of LDS", lds);
            (void)hipGetLastError();
            return SNF_ELAUNCH;
        }
    }
    hipLaunchKernelGGL(kern, dim3(c, b), dim3(256), lds, snf::as_stream(stream), P);
    return snf::check_launch("tile_preprocess_kernel");
}
