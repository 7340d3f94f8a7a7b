"""CPU restatement of the dropout mask of the attention kernels (csrc/philox.h) -- TEST INFRASTRUCTURE, never imported by the
product.  Philox4x32-10 (Salmon et al., SC'11; the constants are the published ones), numpy uint64 arithmetic.

mask[a, row, key] = 1 / (1 - p) if philox(counter = (g, offset), key = seed)[key & 3] >= floor(p * 2^32) else 0,
g = (a * n + row) * ceil(k / 4) + key // 4.  This is the contract the kernels regenerate in registers; the reference draws its
mask from torch's generator (snuffy.py:166-167) -- a different stream with the same distribution."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over the counter words (uint64 arrays holding 32-bit values); returns the 4 output words."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & MASK32 for c in (c0, c1, c2, c3)]
    k0, k1 = np.uint64(k0) & MASK32, np.uint64(k1) & MASK32
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK32, p1 >> np.uint64(32), p1 & MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & MASK32, lo1, (hi0 ^ c3 ^ k1) & MASK32, lo0
        k0, k1 = (k0 + W0) & MASK32, (k1 + W1) & MASK32
    return c0, c1, c2, c3


def dropout_mask(h, n, k, p, seed, offset):
    """[h, n, k] float32 multipliers (0 or 1 / (1 - p)) -- what the kernels apply to the probabilities."""
    if not p > 0:
        return np.ones((h, n, k), dtype=np.float32)
    kg = (k + 3) // 4
    g = (np.arange(h * n, dtype=np.uint64)[:, None] * np.uint64(kg) + np.arange(kg, dtype=np.uint64)[None, :]).reshape(-1)
    off = np.uint64(offset & 0xFFFFFFFFFFFFFFFF)
    sd = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
    out = philox4x32_10(g & MASK32, g >> np.uint64(32), np.full_like(g, off & MASK32), np.full_like(g, off >> np.uint64(32)),
                        sd & MASK32, sd >> np.uint64(32))
    words = np.stack(out, axis=1).reshape(h * n, kg * 4)[:, :k]
    p32 = float(np.float32(p))                       # the C ABI takes dropout_p as a float
    t = p32 * 4294967296.0
    thresh = np.uint64(4294967295 if t >= 4294967295.0 else int(t))
    scale = np.float32(1.0 / (1.0 - p32))
    return np.where(words >= thresh, scale, np.float32(0)).astype(np.float32).reshape(h, n, k)


def random_share_keys(n, seed, offset, layer=0, exclude=()):
    """Host twin of snf_random_share_keys_f32 (csrc/sampler.hip): float32 keys [n]; the k2 largest (ties by ascending row) are the
    device sampler's random patch share."""
    g = np.arange((n + 3) // 4, dtype=np.uint64)
    off = (np.uint64(offset & 0xFFFFFFFFFFFFFFFF) + (np.uint64(layer) << np.uint64(48))) & np.uint64(0xFFFFFFFFFFFFFFFF)
    sd = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
    out = philox4x32_10(g & MASK32, g >> np.uint64(32), np.full_like(g, off & MASK32), np.full_like(g, off >> np.uint64(32)),
                        sd & MASK32, sd >> np.uint64(32))
    words = np.stack(out, axis=1).reshape(-1)[:n]
    keys = (words >> np.uint64(2)).astype(np.uint32).view(np.float32).copy()
    keys[np.asarray(exclude, dtype=np.int64)] = -1.0
    return keys


def random_share_draw(n, k2, seed, offset, layer=0, exclude=()):
    """The k2 rows the device sampler selects: descending key, ties by ascending row (snf_topk_f32's rule)."""
    keys = random_share_keys(n, seed, offset, layer, exclude)
    order = np.argsort(-keys.astype(np.float64), kind="stable")
    return order[:k2]
