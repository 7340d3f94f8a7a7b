"""Batched tile preprocessing for the extractor (reference compute_feats.py:104-152,173-177): decoded uint8 tiles
go to the GPU as they are and ONE HIP kernel does Resize(224) + ToTensor + optional ImageNet normalisation for the whole batch
(snf_tile_preprocess_u8), bit-identical to the per-tile PIL / torch code of the reference.

Host side here: the per-output-pixel integer coefficient tables of Pillow's 8-bit resampler (Resample.c: precompute_coeffs +
normalize_coeffs_8bpc, triangle filter) in double precision, cached per (in, out) size."""
import math

import numpy as np
import torch

from . import _ffi, ops

PRECISION_BITS = 32 - 8 - 2
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
_tables = {}


def _triangle(x):
    x = -x if x < 0.0 else x
    return 1.0 - x if x < 1.0 else 0.0


def resample_coeffs(in_size, out_size):
    """(bounds [out, 2] int32 = (first input pixel, count), coefficients [out, ksize] int32, ksize) of Pillow's bilinear
    resampler with its antialiasing support for in_size -> out_size."""
    scale = in_size / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_triangle((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            coef[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, coef, ksize


def resize_target(h, w, size):
    """Output (oh, ow) of torchvision's Resize(int) on an h x w image: the shorter side becomes `size`."""
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


def _device_tables(n_in, n_out, device):
    key = (n_in, n_out, str(device))
    t = _tables.get(key)
    if t is None:
        b, c, ks = resample_coeffs(n_in, n_out)
        t = _tables[key] = (torch.from_numpy(b).to(device), torch.from_numpy(c).to(device), ks)
    return t


def preprocess_tiles(tiles_u8, size=224, normalize=False, want="f32", patch=16):
    """tiles_u8 [B, H, W, 3] uint8 on the GPU (decoded tiles, RGB) -> Resize(size) (shorter side, PIL bilinear with
    antialiasing), / 255, optional ImageNet normalisation.  want = "f32": [B, 3, oh, ow] fp32 (the reference's tensor);
    "cols": the patch-embedding GEMM operand [B * P, 3 * patch * patch] bf16; "both": the pair."""
    if not (isinstance(tiles_u8, torch.Tensor) and tiles_u8.is_cuda and tiles_u8.dtype == torch.uint8 and tiles_u8.dim() == 4):
        raise _ffi.SnuffyHipError("preprocess_tiles: need a [B, H, W, C] uint8 GPU tensor (no CPU fallback)")
    tiles_u8 = tiles_u8.contiguous()
    b, h, w, c = tiles_u8.shape
    oh, ow = (h, w) if size is None else resize_target(h, w, size)
    hb, hc, hks = _device_tables(w, ow, tiles_u8.device)
    vb, vc, vks = _device_tables(h, oh, tiles_u8.device)
    out = torch.empty(b, c, oh, ow, dtype=torch.float32, device=tiles_u8.device) if want in ("f32", "both") else None
    cols = None
    if want in ("cols", "both"):
        if oh % patch or ow % patch:
            raise ValueError("resized tile %dx%d is not a multiple of the patch size %d" % (oh, ow, patch))
        cols = torch.empty(b * (oh // patch) * (ow // patch), c * patch * patch, dtype=torch.bfloat16, device=tiles_u8.device)
    import ctypes
    mean = (ctypes.c_float * 4)(*(list(IMAGENET_MEAN) + [0.0]))
    std = (ctypes.c_float * 4)(*(list(IMAGENET_STD) + [1.0]))
    p = ops._p
    _ffi.check(_ffi.load().snf_tile_preprocess_u8(p(tiles_u8), b, h, w, c, oh, ow, p(hb), p(hc), hks, p(vb), p(vc), vks,
                                                  1 if normalize else 0, mean, std, p(out), p(cols), patch, ops._stream()),
               "snf_tile_preprocess_u8")
    return {"f32": out, "cols": cols, "both": (out, cols)}[want]
