"""snf_gemm_bf16 (hand-written MFMA GEMM, csrc/gemm.hip) against an fp64 contraction of the SAME bf16-rounded operands:
every activation, both tile widths, ragged M, partial N tiles, strided operands, fp32 / bf16 outputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ref_act(y, act):
    import torch.nn.functional as F
    return {"none": lambda t: t, "relu": F.relu, "gelu": F.gelu, "leakyrelu": lambda t: F.leaky_relu(t, 0.01),
            "selu": F.selu}[act](y)


def run_case(m, n, k, act, tile_n, out_dtype, bias=True, seed=0, lda=None):
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(seed)
    a_full = (torch.randn(m, lda or k, generator=g)).to(torch.bfloat16)
    a = a_full[:, :k]
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, generator=g) if bias else None
    ref = a.double() @ w.double().t()
    if bias:
        ref = ref + b.double()
    ref = ref_act(ref, act)
    ad = a_full.to(DEV)[:, :k]
    out = ops.gemm_bf16(ad, w.to(DEV), b.to(DEV) if bias else None, act, out_dtype, tile_n=tile_n)
    assert out.dtype == out_dtype and tuple(out.shape) == (m, n)
    err = (out.cpu().double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    tol = 2e-5 * max(1.0, scale) if out_dtype == torch.float32 else 4.5e-3 * max(1.0, scale)   # bf16 out: one rounding (2^-8)
    assert err <= tol, (m, n, k, act, tile_n, out_dtype, err, scale)
    return out


@pytest.mark.parametrize("tile_n", [256, 128])
@pytest.mark.parametrize("m,n,k", [(256, 256, 96), (512, 512, 128), (300, 256, 192), (1000, 384, 768), (257, 1152, 384),
                                   (4096, 1536, 768), (777, 768, 3072), (1, 128, 96), (255, 16, 160), (640, 48, 128), (513, 264, 96),
                                   (70000, 256, 96)])
def test_gemm_shapes(m, n, k, tile_n):
    run_case(m, n, k, "none", tile_n, torch.float32, seed=m + n + k)
    run_case(m, n, k, "none", tile_n, torch.bfloat16, seed=m + n + k + 1)


@pytest.mark.parametrize("act", ["relu", "gelu", "leakyrelu", "selu", "none"])
def test_gemm_epilogues(act):
    run_case(900, 512, 256, act, 256, torch.float32, seed=3)
    run_case(900, 384, 256, act, 128, torch.bfloat16, seed=4)
    run_case(520, 256, 128, act, 0, torch.float32, bias=False, seed=5)


def test_gemm_strided_views_and_determinism():
    """a as a column slice of a wider buffer (row pitch > k), out into a column slice; repeated calls are bit-identical."""
    from snuffy_amd import ops
    out1 = run_case(700, 256, 128, "relu", 256, torch.bfloat16, seed=9, lda=384)
    out2 = run_case(700, 256, 128, "relu", 256, torch.bfloat16, seed=9, lda=384)
    assert torch.equal(out1, out2)
    g = torch.Generator().manual_seed(1)
    a = torch.randn(512, 128, generator=g).to(torch.bfloat16).to(DEV)
    w = torch.randn(256, 128, generator=g).to(torch.bfloat16).to(DEV)
    wide = torch.zeros(512, 512, dtype=torch.bfloat16, device=DEV)
    ops.gemm_bf16(a, w, None, "none", out=wide[:, 256:])
    assert torch.equal(wide[:, 256:], ops.gemm_bf16(a, w)) and float(wide[:, :256].abs().max()) == 0.0


def test_gemm_config_b_shapes_match_library():
    """The three projections of a config-B bag (N = 32768, D = 768) against the library GEMM on the same operands."""
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(2)
    for n, k, act in [(1536, 768, "none"), (3072, 768, "relu"), (768, 3072, "none")]:
        a = torch.randn(32768, k, generator=g).to(torch.bfloat16).to(DEV)
        w = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16).to(DEV)
        b = torch.randn(n, generator=g).to(DEV)
        out = ops.gemm_bf16(a, w, b, act)
        lib = torch.addmm(b, a.float(), w.float().t())
        lib = torch.relu(lib) if act == "relu" else lib
        assert (out.float() - lib).abs().max().item() <= 4e-3 * max(1.0, lib.abs().max().item())


def test_gemm_rejects_unsupported_shapes():
    from snuffy_amd import ops
    from snuffy_amd._ffi import SnuffyHipError
    a = torch.zeros(64, 48, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(32, 48, dtype=torch.bfloat16, device=DEV)
    assert not ops.gemm_supported(64, 32, 48) and ops.gemm_supported(64, 32, 96)
    with pytest.raises(SnuffyHipError):
        ops.gemm_bf16(a, w)
    with pytest.raises(SnuffyHipError):
        ops.gemm_bf16(a.cpu(), w.cpu())
