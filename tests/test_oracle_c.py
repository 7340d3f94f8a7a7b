"""The C restatement (oracle/snuffy_oracle.c) against the golden vectors and the torch oracle.  CPU only."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import snuffy_oracle as orc
from tests.helpers import golden_files, load_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "liboracle.so")


@pytest.fixture(scope="module")
def clib():
    if not os.path.exists(LIB):
        import subprocess
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    lib = ctypes.CDLL(LIB)
    lib.orc_topk_desc_stable.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    lib.orc_k_split.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    lib.orc_sparse_attention.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                                 ctypes.c_void_p, ctypes.c_void_p]
    return lib


def c_topk(lib, c, k):
    c = np.ascontiguousarray(c, dtype=np.float32)
    idx = np.empty(k, dtype=np.int64)
    assert lib.orc_topk_desc_stable(c.ctypes.data, c.size, 1, k, idx.ctypes.data) == 0
    return idx


def test_c_topk_golden(clib):
    z = np.load(golden_files("f2_")[0])
    for n in (5000, 32768):
        assert np.array_equal(c_topk(clib, z[f"tiefree_c_{n}"], 1024), z[f"tiefree_order_{n}"])
    assert np.array_equal(c_topk(clib, z["ties_c"], 1024), z["ties_stable_order"])
    assert np.array_equal(c_topk(clib, z["special_c"], z["special_c"].size), z["special_stable_order"])
    for lam, r, n, k1, k2 in z["k_table"]:
        a, b = ctypes.c_int64(), ctypes.c_int64()
        clib.orc_k_split(int(lam), float(r), int(n), ctypes.byref(a), ctypes.byref(b))
        assert (a.value, b.value) == (int(k1), int(k2))


def test_c_attention_matches_reference_A(clib):
    """A of the reference (golden F1) re-derived from the golden weights by the C restatement."""
    z, sd = load_case(golden_files("f1_n1000")[0])
    N, D, h, lam, depth, seed = [int(v) for v in z["cfg"]]
    x = torch.from_numpy(z["x"])[0]
    pre = "b_classifier.encoder.layers.0."
    sel = torch.from_numpy(z["top"])
    xn = orc.layer_norm(x, sd[pre + "sublayer.0.norm.weight"], sd[pre + "sublayer.0.norm.bias"])
    lin = pre + "self_attn.linears."
    q = torch.nn.functional.linear(xn, sd[lin + "0.weight"], sd[lin + "0.bias"]).contiguous().numpy()
    kp = torch.nn.functional.linear(x[sel], sd[lin + "1.weight"], sd[lin + "1.bias"]).contiguous().numpy()
    v = torch.nn.functional.linear(xn, sd[lin + "2.weight"], sd[lin + "2.bias"]).contiguous().numpy()
    k = kp.shape[0]
    out = np.empty((k, D), dtype=np.float32)
    attn = np.empty((h, N, k), dtype=np.float32)
    assert clib.orc_sparse_attention(q.ctypes.data, kp.ctypes.data, v.ctypes.data, N, k, h, D // h, out.ctypes.data,
                                     attn.ctypes.data) == 0
    np.testing.assert_allclose(attn[:, z["A_rows"], :], z["A_sub"][0], rtol=0, atol=1e-6)
    np.testing.assert_allclose(attn.astype(np.float64).sum(axis=1), z["A_colsum"][0], rtol=1e-5)
    o_t, p_t = orc.sparse_attention(torch.from_numpy(q), torch.from_numpy(kp), torch.from_numpy(v), h)
    np.testing.assert_allclose(out, o_t.numpy(), rtol=0, atol=2e-5)
