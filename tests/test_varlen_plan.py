"""Host-side geometry of the varlen path (no GPU: the plan entry points are pure host code; cu_count() falls back to the MI355X's
256 without a device): every tile of every (bag, head) is covered exactly once, a bag's plan does not depend on what it is packed
with, the per-workgroup bag table matches the descriptors, the ragged index helpers address the padded top-k output correctly."""
import ctypes

import numpy as np
import pytest

from snuffy_amd import _ffi

VL_DESC = 12


def _attn_plan(fn, sizes, k, h, dk):
    off = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=off[1:])
    need, ws = ctypes.c_size_t(0), ctypes.c_size_t(0)
    rc = fn(ctypes.c_void_p(off.ctypes.data), len(sizes), k, h, dk, None, 0, ctypes.byref(need), ctypes.byref(ws))
    if rc:
        return rc, None, None, None
    table = np.full(need.value, -7, dtype=np.int32)
    rc = fn(ctypes.c_void_p(off.ctypes.data), len(sizes), k, h, dk, ctypes.c_void_p(table.ctypes.data), table.size,
            ctypes.byref(need), ctypes.byref(ws))
    assert rc == 0
    return 0, table, ws.value, off


@pytest.mark.parametrize("which,tile_rows,small_tiles", [("snf_sparse_attn_varlen_plan", 128, 8), ("snf_sparse_attn_x3_varlen_plan", 64, 16)])
@pytest.mark.parametrize("sizes", [[1000] * 5, [200, 1000, 129, 5000, 2048, 777], [40000, 300, 65536, 8192], [224]])
def test_attention_plan_covers_every_tile_once(which, tile_rows, small_tiles, sizes):
    lib = _ffi.load()
    k, h, dk = 200, 6, 128
    rc, table, ws, off = _attn_plan(getattr(lib, which), sizes, k, h, dk)
    assert rc == 0
    b = len(sizes)
    desc = table[:VL_DESC * b].reshape(b, VL_DESC)
    wg_bag = table[VL_DESC * b:]
    wg0 = part0 = 0
    for i, n in enumerate(sizes):
        d = desc[i]
        tph = -(-n // tile_rows)
        assert d[0] == wg0 and d[1] == off[i] and d[2] == n and d[3] == i * k
        assert d[4] == tph and d[6] == tph * h and d[8] == part0
        tpw, num_wg, seg = int(d[5]), int(d[9]), int(d[7])
        assert num_wg * tpw >= d[6] > (num_wg - 1) * tpw                      # every tile, no empty workgroup
        if d[6] <= 256:
            assert tpw == min(tph, small_tiles)                              # a bag that cannot fill the chip: >= 1024 rows / workgroup
        assert seg >= -(-tpw // tph) + (1 if tpw % tph else 0) and seg >= 1   # partial slots for every head a workgroup can touch
        assert np.all(wg_bag[wg0:wg0 + num_wg] == i)
        assert d[10] == (1 if tpw == tph else 0)        # one workgroup per head: the bag's output is stored without a reduction
        # a bag's plan does not depend on the batch: the same bag alone gives the same descriptor (but for the batch offsets)
        rc1, t1, _, _ = _attn_plan(getattr(lib, which), [n], k, h, dk)
        assert rc1 == 0 and np.array_equal(t1[4:10][[0, 1, 2, 3, 5]], d[4:10][[0, 1, 2, 3, 5]])
        wg0 += num_wg
        part0 += num_wg * seg
    assert wg_bag.size == wg0
    nkb = 7
    assert ws >= part0 * nkb * (dk // 32) * 1024 * 4


def test_attention_plan_refuses_what_the_kernels_do_not_take():
    lib = _ffi.load()
    for which in ("snf_sparse_attn_varlen_plan", "snf_sparse_attn_x3_varlen_plan"):
        fn = getattr(lib, which)
        assert _attn_plan(fn, [1000, 2000], 225, 6, 128)[0] != 0      # more keys than one LDS image (dk = 128: 224)
        assert _attn_plan(fn, [1000, 2000], 200, 6, 83)[0] != 0       # head width outside the MFMA kernels
        off = np.array([0, 100, 100], dtype=np.int64)                 # an empty bag
        need = ctypes.c_size_t(0)
        assert fn(ctypes.c_void_p(off.ctypes.data), 2, 50, 6, 128, None, 0, ctypes.byref(need), None) != 0
        assert _attn_plan(fn, [300, 300], 256, 6, 64)[0] == 0          # dk = 64 holds 256 keys


@pytest.mark.parametrize("sizes", [[1, 3, 300, 5000], [100000, 2, 4097]])
def test_head_plan_keeps_every_bags_own_partition(sizes):
    lib = _ffi.load()
    off = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=off[1:])
    need, ws = ctypes.c_size_t(0), ctypes.c_size_t(0)
    d = 384
    assert lib.snf_ln_mean_head_varlen_plan(ctypes.c_void_p(off.ctypes.data), len(sizes), d, None, 0, ctypes.byref(need),
                                            ctypes.byref(ws)) == 0
    table = np.zeros(need.value, dtype=np.int32)
    assert lib.snf_ln_mean_head_varlen_plan(ctypes.c_void_p(off.ctypes.data), len(sizes), d, ctypes.c_void_p(table.ctypes.data),
                                            table.size, ctypes.byref(need), ctypes.byref(ws)) == 0
    b = len(sizes)
    desc = table[:4 * b].reshape(b, 4)
    wg = 0
    for i, n in enumerate(sizes):
        parts = min(-(-n // 4), 256 * 4)            # the single-bag launch: one workgroup per 4 rows, capped at 4 per CU
        assert list(desc[i]) == [wg, off[i], n, parts]
        assert np.all(table[4 * b + wg:4 * b + wg + parts] == i)
        wg += parts
    assert table.size == 4 * b + wg
    assert ws.value == (wg + 16 * b) * d * 4


def test_packed_bags_and_ragged_key_helpers_on_the_host():
    import torch

    from snuffy_amd import ops
    sizes = [5, 300, 1, 40, 200]
    pk = ops.PackedBags(sizes, "cpu")
    assert pk.total == sum(sizes) and pk.max_n == 300 and list(pk.host) == [0, 5, 305, 306, 346, 546]
    with pytest.raises(ValueError):
        ops.PackedBags([3, 0, 2], "cpu")
    kbs = [min(200, n) for n in sizes]
    rag = pk.ragged(kbs)
    assert rag is pk.ragged(kbs)                                   # cached per key-count tuple
    assert rag.kmax == 200 and rag.total == sum(kbs) and list(rag.koff) == [0, 5, 205, 206, 246, 446]
    assert rag.desc.tolist() == [[0, 5, 0, 5], [5, 300, 5, 200], [305, 1, 205, 1], [306, 40, 206, 40], [346, 200, 246, 200]]
    pos, base = rag.flat_index(200)
    # the padded [B, 200] top-k output -> flat selected rows in packed coordinates
    top = torch.full((len(sizes), 200), -1, dtype=torch.int64)
    for b, k in enumerate(kbs):
        top[b, :k] = torch.arange(k - 1, -1, -1)                    # any permutation of the bag's rows
    sel = top.reshape(-1)[pos] + base
    assert sel.shape[0] == sum(kbs) and int(sel.min()) >= 0
    for b, k in enumerate(kbs):
        seg = sel[int(rag.koff[b]):int(rag.koff[b + 1])]
        assert seg.tolist() == [int(pk.host[b]) + j for j in range(k - 1, -1, -1)]
    with pytest.raises(ValueError):
        pk.ragged([6, 200, 1, 40, 200])                            # more keys than rows
    assert ops.ragged_attn_supported(200, 83) and ops.ragged_attn_supported(256, 128) and not ops.ragged_attn_supported(257, 64)
    assert not ops.ragged_attn_supported(256, 160)
