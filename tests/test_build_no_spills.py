"""Build guard (no GPU): the hot-path kernels of the built library keep every value in registers.

Round 5's end-of-round counter pass found the headline attention instantiation spilling 21 registers (a second, never-taken copy of the
pipeline in the single-launch kernels: PMC traffic x1.37 instead of x1.29, +2.4 us per launch).  The AMDGPU metadata of the built objects
says so without a GPU and without recompiling (tools/scan_spills.py); this test keeps the kernels of the config-B / config-A bag at zero."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

# gemm_hl_kernel is held to a bound instead.  Until the end of round 5 it kept the per-lane row / swizzle terms of its DMA sources in scratch
# (72 - 124 bytes) and re-read them, one L2 round trip behind the other, in front of the last step's MFMAs of every tile: 3 % of the FFN
# launches (same-box A / B).  They are recomputed per tile now; what is left (<= 20 bytes) is read once per tile in the epilogue.
BOUNDED = {"gemm_hl_kernel<": 24}
HOT = (
    "gemm_bf16_kernel<",                                   # bf16 projections, concatenated-K fp32-class form of small bags
    "critic_kernel<", "topk_select_kernel", "topk_hist", "gather_slot_map_kernel", "skinny_linear_x3_kernel<",
    "ln_colsum_kernel<", "ln_colreduce_kernel", "head_gemv_kernel", "layernorm_rows_kernel<", "x3p_reduce_kernel<",
    "reduce_partials_kernel<", "split3_colsum_kernel<", "pt_v_lds_kernel<", "scores_softmax_x3u_kernel<",
)


def _single_launch_x3p(name):
    # sparse_attn_x3p_kernel<DK, NKB, KBW = 1, AUX, MODE = 0>: the launch that covers all keys (attention() of snuffy.py:160-168 at K <= 256)
    return name.startswith("void (anonymous namespace)::sparse_attn_x3p_kernel<") and name.split(">")[0].endswith(", 0") and \
        name.split("<")[1].split(",")[2].strip() == "1"


def _bf16_headline(name):
    return name.startswith("void (anonymous namespace)::sparse_attn_mfma_kernel<128, 7, unsigned short, false, false, 8, false>")


def test_hot_path_kernels_do_not_spill():
    import scan_spills
    objdir = os.path.join(scan_spills.ROOT, "snuffy_amd", "build")
    if not os.path.isdir(objdir) or not any(f.endswith(".o") for f in os.listdir(objdir)):
        pytest.skip("no build objects here (the library was built elsewhere)")
    try:
        ks = scan_spills.kernels(objdir)
    except RuntimeError as exc:
        pytest.skip(str(exc))
    names = scan_spills.demangle([k[1] for k in ks])
    assert len(ks) > 300, "the metadata of the built objects could not be read"
    checked, bad = 0, []
    for (obj, _, scratch, spilled, _vg), name in zip(ks, names):
        bound = next((b for h, b in BOUNDED.items() if h in name), None)
        hot = _single_launch_x3p(name) or _bf16_headline(name) or any(h in name for h in HOT)
        if not hot and bound is None:
            continue
        checked += 1
        if scratch > (bound or 0):
            bad.append((obj, name[:120], scratch, spilled))
    assert checked >= 60, checked
    assert not bad, bad
