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


# ---- fp32-class projections: split-bf16 x3 over a tripled K axis ---------------------------------------------------------
def _split_host(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi, lo


@pytest.mark.parametrize("n,d", [(700, 384), (513, 768), (64, 96), (1000, 100)])
def test_layernorm_rows_split3_is_the_split_of_the_fp32_rows(n, d):
    """snf_layernorm_rows_split3_f32 == [hi | hi | lo] of snf_layernorm_rows_f32's fp32 output, bit for bit (incl. patch rows)."""
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(n + d)
    x = (torch.randn(n, d, generator=g) * 3 + 1).to(DEV)
    gam, bet = torch.randn(d, generator=g).to(DEV), torch.randn(d, generator=g).to(DEV)
    slot = torch.full((n,), -1, dtype=torch.int32)
    slot[[3, n // 2, n - 1]] = torch.tensor([0, 1, 2], dtype=torch.int32)
    patch = torch.randn(3, d, generator=g).to(DEV)
    for kw in ({}, {"slot": slot.to(DEV), "patch_rows": patch}):
        ref = ops.layernorm_rows(x, gam, bet, 1e-5, **kw)
        got = ops.layernorm_rows_split3(x, gam, bet, 1e-5, **kw)
        hi, lo = _split_host(ref)
        assert got.shape == (n, 3 * d) and got.dtype == torch.bfloat16
        assert torch.equal(got[:, :d], hi) and torch.equal(got[:, d:2 * d], hi) and torch.equal(got[:, 2 * d:], lo)


@pytest.mark.parametrize("m,n,k,act", [(1000, 768, 384, "none"), (4096, 1536, 768, "none"), (777, 3072, 768, "relu"),
                                       (900, 768, 3072, "none"), (300, 512, 96, "gelu")])
def test_gemm_x3_is_fp32_class(m, n, k, act):
    """[hi | hi | lo] x [Wh | Wl | Wh] on the bf16 MFMA kernel against an fp64 contraction of the fp32 operands: error of the
    order of 2^-17 per product (measured <= 4e-6 relative to the result scale), 500x below a plain bf16 GEMM's."""
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(m + k)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g)
    ref = ref_act(a.double() @ w.double().t() + b.double(), act)
    a3 = ops.split3_rows(a.to(DEV))
    w3 = ops.split3_weight(w.to(DEV))
    assert a3.shape == (m, 3 * k) and w3.shape == (n, 3 * k)
    out = ops.gemm_bf16(a3, w3, b.to(DEV), act, torch.float32)
    scale = max(1.0, ref.abs().max().item())
    err = (out.cpu().double() - ref).abs().max().item() / scale
    assert err <= 8e-6, (m, n, k, act, err)
    # the split image of the result straight from the epilogue == the split of the fp32 output, bit for bit
    img = ops.gemm_bf16(a3, w3, b.to(DEV), act, split3=True)
    hi, lo = _split_host(out)
    assert img.shape == (m, 3 * n)
    assert torch.equal(img[:, :n], hi) and torch.equal(img[:, n:2 * n], hi)
    if act == "gelu":   # the erf polynomial may round differently in the two kernel variants: hi + lo reproduces the value
        assert ((img[:, :n].float() + img[:, 2 * n:].float()) - out).abs().max().item() <= 2.0 ** -15 * scale
    else:
        assert torch.equal(img[:, 2 * n:], lo)
    img128 = ops.gemm_bf16(a3, w3, b.to(DEV), act, split3=True, tile_n=128)
    out128 = ops.gemm_bf16(a3, w3, b.to(DEV), act, torch.float32, tile_n=128)
    hi, lo = _split_host(out128)
    assert torch.equal(img128[:, :n], hi) and (act == "gelu" or torch.equal(img128[:, 2 * n:], lo))
    # ... and the interleaved hl image (round 5: what the pipelined attention streams when the bag is too small for the one-pass GEMM)
    for tile_n, ref_out in ((0, out), (128, out128)):
        hl = ops.gemm_bf16(a3, w3, b.to(DEV), act, hl_out=True, tile_n=tile_n).view(m, n // 32, 2, 32)
        hi, lo = _split_host(ref_out)
        assert torch.equal(hl[:, :, 0].reshape(m, n), hi)
        if act == "gelu":
            assert ((hl[:, :, 0].reshape(m, n).float() + hl[:, :, 1].reshape(m, n).float()) - ref_out).abs().max().item() <= 2.0 ** -15 * scale
        else:
            assert torch.equal(hl[:, :, 1].reshape(m, n), lo)


@pytest.mark.parametrize("m,n,k,act", [(1000, 768, 384, "none"), (4096, 1536, 768, "none"), (777, 3072, 768, "relu"),
                                       (900, 768, 3072, "none"), (300, 512, 96, "gelu"), (257, 264, 32, "selu"),
                                       (32768, 1536, 768, "none"), (20000, 256, 64, "leakyrelu")])
def test_gemm_hl_one_pass_kernel(m, n, k, act):
    """snf_gemm_hl_bf16 (interleaved [hi(32) | lo(32)] images, one full-line staging per K step, hi hi + hi lo + lo hi out of it)
    against fp64 and against the concatenated form: fp32-class error, fp32 / bf16 / hl-image outputs, ragged M, partial column
    tiles, 1 .. 96 K steps; the producers of the format (split kernel, LayerNorm, host weight split) against their definition."""
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(m + k)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g)
    ref = ref_act(a.double() @ w.double().t() + b.double(), act)
    ad, wd, bd = a.to(DEV), w.to(DEV), b.to(DEV)
    a_hl, w_hl = ops.split_hl_rows(ad), ops.split_hl_weight(wd)
    hi, lo = _split_host(ad)
    assert a_hl.shape == (m, 2 * k) and w_hl.shape == (n, 2 * k)
    v = a_hl.view(m, k // 32, 2, 32)
    assert torch.equal(v[:, :, 0].reshape(m, k), hi) and torch.equal(v[:, :, 1].reshape(m, k), lo)
    out = ops.gemm_hl(a_hl, w_hl, bd, act)
    scale = max(1.0, ref.abs().max().item())
    err = (out.cpu().double() - ref).abs().max().item() / scale
    assert err <= 8e-6, (m, n, k, act, err)
    cat = ops.gemm_x3(ops.split3_rows(ad), ops.split3_weight(wd), bd, act)   # same products, different summation order
    assert (out - cat).abs().max().item() <= 4e-6 * scale
    assert torch.equal(out, ops.gemm_hl(a_hl, w_hl, bd, act))                # deterministic
    res = torch.randn(m, n, generator=g).to(DEV)
    assert (ops.gemm_hl(a_hl, w_hl, bd, act, resid=res) - (out + res)).abs().max().item() <= 1e-6 * scale   # residual in the epilogue
    ob = ops.gemm_hl(a_hl, w_hl, bd, act, out_dtype=torch.bfloat16)
    assert (ob.float() - out).abs().max().item() <= 2.0 ** -8 * scale
    if n % 32 == 0:
        img = ops.gemm_hl(a_hl, w_hl, bd, act, hl_out=True).view(m, n // 32, 2, 32)
        hi, lo = _split_host(out)
        assert torch.equal(img[:, :, 0].reshape(m, n), hi)
        assert ((img[:, :, 0].reshape(m, n).float() + img[:, :, 1].reshape(m, n).float()) - out).abs().max().item() <= 2.0 ** -15 * scale


@pytest.mark.parametrize("m,n,k,act", [(32768, 768, 3072, "none"), (100000, 1536, 768, "relu"), (5000, 768, 3072, "gelu"),
                                       (33000, 520, 1024, "none"), (900, 768, 3072, "none"), (70000, 768, 256, "none")])
def test_gemm_hl_split_k_of_the_last_round(m, n, k, act, monkeypatch):
    """snf_gemm_hl_ws_bf16: the tiles of the last, partly filled round (config B's FFN output projection: 384 tiles on 256 CUs) run as
    2 .. 4 K parts on otherwise idle workgroups, combined by the last arriver in part order -- against fp64, against the plain tile
    walk, bit-reproducible, every output type, tickets left clean."""
    from snuffy_amd import _ffi, ops
    g = torch.Generator().manual_seed(m + n + k)
    a = (torch.randn(m, k, generator=g) * 0.5).to(DEV)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(DEV)
    b = torch.randn(n, generator=g).to(DEV)
    res = torch.randn(m, n, generator=g).to(DEV)
    a_hl, w_hl = ops.split_hl_rows(a), ops.split_hl_weight(w)
    splits = int(_ffi.load().snf_gemm_hl_ws_bytes(m, n, k)) > 0
    if (m, n, k) in ((32768, 768, 3072), (900, 768, 3072), (33000, 520, 1024)):
        assert splits          # 1.5 rounds / a 12-tile launch on 16 workgroups / 390 tiles: the last round is split
    if m in (100000, 70000):
        assert not splits      # many full rounds (not worth it) / parts would be shorter than 8 K steps
    monkeypatch.setattr(ops, "GEMM_HL_SPLITK", True)
    out = ops.gemm_hl(a_hl, w_hl, b, act, resid=res)
    assert torch.equal(out, ops.gemm_hl(a_hl, w_hl, b, act, resid=res))
    if splits:
        # the scratch buffer comes from the allocator per call and may be dirty: the call zeroes its own tickets
        nb = int(_ffi.load().snf_gemm_hl_ws_bytes(m, n, k))
        for _ in range(2):
            junk = torch.full((nb,), 0xAB, dtype=torch.uint8, device=DEV)
            del junk                                   # the next allocation of this size gets the same block back, dirty
            assert torch.equal(out, ops.gemm_hl(a_hl, w_hl, b, act, resid=res))
    img = ops.gemm_hl(a_hl, w_hl, b, act, hl_out=True) if n % 32 == 0 else None
    ob = ops.gemm_hl(a_hl, w_hl, b, act, out_dtype=torch.bfloat16)
    monkeypatch.setattr(ops, "GEMM_HL_SPLITK", False)
    plain = ops.gemm_hl(a_hl, w_hl, b, act, resid=res)
    scale = max(1.0, plain.abs().max().item())
    assert (out - plain).abs().max().item() <= 2e-6 * scale
    rows = torch.cat([torch.arange(0, 300), torch.arange(m - 300, m)])           # fp64 on a sample of rows (first / last tiles)
    ref = ref_act(a[rows].cpu().double() @ w.cpu().double().t() + b.cpu().double(), act) + res[rows].cpu().double()
    assert (out[rows].cpu().double() - ref).abs().max().item() <= 8e-6 * scale
    assert (ob.float() - (plain - res)).abs().max().item() <= 2.0 ** -7 * scale
    if img is not None:
        v = img.view(m, n // 32, 2, 32)
        assert ((v[:, :, 0].reshape(m, n).float() + v[:, :, 1].reshape(m, n).float()) - (plain - res)).abs().max().item() <= 2.0 ** -15 * scale


def test_gemm_hl_split_k_graph_replay_survives_a_larger_shape_and_other_streams():
    """ADVICE r4 (high / medium): the split-K scratch used to be ONE growing buffer per device -- a graph captured at shape A replayed
    against freed memory once a larger shape B had replaced the buffer, and two streams shared one set of tickets.  Now every call
    takes its scratch from the allocator (a capture: from the graph's pool): capture A, run a larger B eagerly, churn the allocator,
    replay A == eager A; and two streams running split-K GEMMs concurrently both get the single-stream result."""
    from snuffy_amd import _ffi, ops
    lib = _ffi.load()
    g = torch.Generator().manual_seed(5)
    shapes = [(900, 768, 3072), (32768, 768, 3072)]             # 12 tiles on 16 workgroups / 384 tiles on 256: both split
    ops_in = []
    for m, n, k in shapes:
        assert int(lib.snf_gemm_hl_ws_bytes(m, n, k)) > 0
        a = (torch.randn(m, k, generator=g) * 0.5).to(DEV)
        w = (torch.randn(n, k, generator=g) / k ** 0.5).to(DEV)
        ops_in.append((ops.split_hl_rows(a), ops.split_hl_weight(w), torch.randn(n, generator=g).to(DEV)))
    assert int(lib.snf_gemm_hl_ws_bytes(*shapes[1])) > int(lib.snf_gemm_hl_ws_bytes(*shapes[0]))
    eager = [ops.gemm_hl(a, w, b) for a, w, b in ops_in]
    torch.cuda.synchronize()
    a0, w0, b0 = ops_in[0]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_g = ops.gemm_hl(a0, w0, b0)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_g, eager[0])
    big = ops.gemm_hl(*ops_in[1])                                # a larger scratch request after the capture
    assert torch.equal(big, eager[1])
    del big
    junk = [torch.full((64 << 20,), 0xCD, dtype=torch.uint8, device=DEV) for _ in range(4)]   # reuse whatever was freed, dirty
    del junk
    torch.cuda.empty_cache()
    out_g.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_g, eager[0])
    # two streams, split-K GEMMs in flight on both at once
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    res = {}
    for rep in range(3):
        for s, i in ((s1, 1), (s2, 0)):
            with torch.cuda.stream(s):
                res[i] = ops.gemm_hl(*ops_in[i])
        torch.cuda.synchronize()
        assert torch.equal(res[0], eager[0]) and torch.equal(res[1], eager[1])


@pytest.mark.parametrize("n,d", [(700, 384), (513, 768), (64, 96), (1000, 2048)])
def test_layernorm_rows_hl_is_the_interleaved_split_of_the_fp32_rows(n, d):
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(n + d)
    x = (torch.randn(n, d, generator=g) * 3 + 1).to(DEV)
    gam, bet = torch.randn(d, generator=g).to(DEV), torch.randn(d, generator=g).to(DEV)
    slot = torch.full((n,), -1, dtype=torch.int32)
    slot[[3, n // 2, n - 1]] = torch.tensor([0, 1, 2], dtype=torch.int32)
    patch = torch.randn(3, d, generator=g).to(DEV)
    for kw in ({}, {"slot": slot.to(DEV), "patch_rows": patch}):
        hi, lo = _split_host(ops.layernorm_rows(x, gam, bet, 1e-5, **kw))
        got = ops.layernorm_rows_hl(x, gam, bet, 1e-5, **kw).view(n, d // 32, 2, 32)
        assert torch.equal(got[:, :, 0].reshape(n, d), hi) and torch.equal(got[:, :, 1].reshape(n, d), lo)


def test_layernorm_rows_hl_patch_writes_the_rows_in_place():
    """snf_layernorm_rows_hl_patch_f32: LayerNorm(x + addend) of K rows written over chosen rows of an existing hl image == the
    rows layernorm_rows_hl produces for the sum, bit for bit; the other rows untouched."""
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(3)
    n, k, d = 5000, 200, 768
    img = ops.layernorm_rows_hl(torch.randn(n, d, generator=g).to(DEV), None, None, 1e-5)
    before = img.clone()
    xs, dl = torch.randn(k, d, generator=g).to(DEV), torch.randn(k, d, generator=g).to(DEV)
    rows = torch.randperm(n, generator=g)[:k].to(DEV)
    gam, bet = torch.rand(d, generator=g).to(DEV) + 0.5, torch.randn(d, generator=g).to(DEV)
    for kw in ({}, {"gamma": gam, "beta": bet}):
        img.copy_(before)
        ops.layernorm_rows_hl_patch_(img, rows, xs, dl, eps=1e-5, **kw)
        want = ops.layernorm_rows_hl(xs + dl, kw.get("gamma"), kw.get("beta"), 1e-5)
        assert torch.equal(img[rows], want)
        keep = torch.ones(n, dtype=torch.bool, device=DEV)
        keep[rows] = False
        assert torch.equal(img[keep], before[keep])
    img.copy_(before)
    ops.layernorm_rows_hl_patch_(img, rows, xs, None)
    assert torch.equal(img[rows], ops.layernorm_rows_hl(xs, None, None, 1e-5))


def test_fp32_path_x3_projections_against_library_projections(monkeypatch):
    """One encoder layer of the fp32 path at config-A size: split-bf16 x3 projections vs the fp32 library GEMMs."""
    from snuffy_amd import functional as SF
    torch.manual_seed(0)
    n, d = 8192, 384
    from tests.helpers import build_amd_milnet
    net = build_amd_milnet(d, 6, "gelu", 200, 0.0, 2).to(DEV).eval()
    net.configure(precision="fp32")
    x = torch.randn(1, n, d, device=DEV) * 0.7
    with torch.no_grad():
        monkeypatch.setattr(SF, "FP32_GEMM", "x3")
        c3, l3, a3 = net(x)
        monkeypatch.setattr(SF, "FP32_GEMM", "library")
        c0, l0, a0 = net(x)
    assert torch.equal(c3, c0)
    assert (l3 - l0).abs().max().item() <= 2e-5 * max(1.0, l0.abs().max().item())
    assert (a3 - a0).abs().max().item() <= 2e-5


def test_fp32_path_shared_normalisation_against_two_layernorm_passes(monkeypatch):
    """fp32 hl branch at config-B width: ONE normalised image + LayerNorm affines folded into Wq | Wv and W1 (FP32_SHARED_NORM) against
    the two full LayerNorm passes with affine; non-trivial gamma / beta; depth 2 so that a layer without the critic hand-over runs too."""
    from snuffy_amd import functional as SF
    torch.manual_seed(1)
    n, d = 20000, 768
    from tests.helpers import build_amd_milnet
    net = build_amd_milnet(d, 6, "relu", 200, 0.0, 2).to(DEV).eval()
    with torch.no_grad():
        for layer in net.b_classifier.encoder.layers:
            for sub in layer.sublayer:
                sub.norm.weight.uniform_(0.5, 1.5)
                sub.norm.bias.uniform_(-0.3, 0.3)
    net.invalidate()
    net.configure(precision="fp32")
    x = torch.randn(1, n, d, device=DEV) * 0.7
    with torch.no_grad():
        monkeypatch.setattr(SF, "FP32_SHARED_NORM", True)
        c1, l1, a1 = net(x)
        assert SF.shared_norm_layer(net.b_classifier.encoder.layers[0]) and SF.hl_layer_eligible(net.b_classifier.encoder.layers[0], n, d)
        monkeypatch.setattr(SF, "FP32_SHARED_NORM", False)
        c0, l0, a0 = net(x)
    assert torch.equal(c1, c0)
    assert (l1 - l0).abs().max().item() <= 1e-5 * max(1.0, l0.abs().max().item())
    assert (a1 - a0).abs().max().item() <= 1e-5


@pytest.mark.parametrize("r,c,k", [(200, 768, 768), (1, 384, 384), (37, 100, 128), (512, 1536, 64), (224, 768, 3072)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_linear_rows_x3_skinny_kernel_vs_fp64(r, c, k, out_dtype):
    """snf_linear_rows_x3_f32: the key / output projections of the K selected rows -- fp32-class (split-bf16 x3, split in registers),
    any r <= 8192 and c, k % 64 == 0; against fp64, and deterministic."""
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(r * 7 + c)
    x = torch.randn(r, k, generator=g).to(DEV)
    w = (torch.randn(c, k, generator=g) / k ** 0.5).to(DEV)
    b = torch.randn(c, generator=g).to(DEV)
    y = ops.linear_rows_x3(x, w, b, out_dtype=out_dtype)
    assert y.shape == (r, c) and y.dtype == out_dtype
    ref = x.double() @ w.double().t() + b.double()
    tol = 2e-5 if out_dtype == torch.float32 else 5e-3
    assert (y.double() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    assert torch.equal(y, ops.linear_rows_x3(x, w, b, out_dtype=out_dtype))
    y0 = ops.linear_rows_x3(x[:, :k], w, None, out_dtype=torch.float32)       # no bias
    assert (y0.double() - (ref - b.double())).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


# ---- round 6: the epilogue variants of the bf16 GEMM behind the fused ViT block (LayerNorm folded into the consumer, residual
# stream updated by the producer) -- vd:97-127
@pytest.mark.parametrize("act", ["none", "gelu"])
@pytest.mark.parametrize("m,n,k", [(394, 384, 128), (1000, 1152, 384), (256, 1536, 384), (1577, 192, 96), (100, 64, 1600), (3000, 768, 768)])
def test_gemm_bf16_lnfold_is_layernorm_then_linear(m, n, k, act):
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(m + n + k)
    x = torch.randn(m, k, generator=g) * (0.5 + torch.rand(m, 1, generator=g)) + 0.7 * torch.randn(m, 1, generator=g)   # row means ~ row spreads
    gam, bet = 1.0 + 0.2 * torch.randn(k, generator=g), 0.3 * torch.randn(k, generator=g)
    w0, b0 = torch.randn(n, k, generator=g) / k ** 0.5, torch.randn(n, generator=g)
    eps = 1e-6
    wf = (w0.double() * gam.double()).float().to(torch.bfloat16)
    cs = wf.double().sum(1).float()
    bfold = (w0.double() @ bet.double() + b0.double()).float()
    stats, xb = ops.vit_row_stats(x.to(DEV), eps=eps, want_bf16=True)
    mean, var = x.double().mean(1), x.double().var(1, unbiased=False)
    assert torch.allclose(stats[:, 0].cpu().double(), mean, atol=1e-5) and torch.allclose(stats[:, 1].cpu().double(), (var + eps).rsqrt(), rtol=1e-5)
    assert torch.equal(xb.cpu(), x.to(torch.bfloat16))
    out = ops.gemm_bf16_lnfold(xb, wf.to(DEV), cs.to(DEV), bfold.to(DEV), stats, act)
    # the kernel's own arithmetic in fp64: rounded operands, exact statistics
    emu = (var + eps).rsqrt()[:, None] * (x.to(torch.bfloat16).double() @ wf.double().t() - mean[:, None] * cs.double()) + bfold.double()
    emu = ref_act(emu, act)
    assert (out.cpu().double() - emu).abs().max() <= 4.5e-3 * max(1.0, emu.abs().max().item())        # one bf16 rounding of the result
    # ... and it IS LayerNorm -> Linear within the bf16 class
    ref = ref_act(torch.nn.functional.layer_norm(x.double(), (k,), gam.double(), bet.double(), eps) @ w0.double().t() + b0.double(), act)
    assert (out.cpu().double() - ref).abs().max() <= 2e-2 * max(1.0, ref.abs().max().item())
    out2 = ops.gemm_bf16_lnfold(xb, wf.to(DEV), cs.to(DEV), bfold.to(DEV), stats, act)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("m,n,k", [(394, 384, 128), (1000, 384, 1600), (256, 128, 96), (1577, 192, 768), (100, 64, 96), (3000, 768, 3072), (700, 320, 256), (515, 640, 192)])
def test_gemm_bf16_resid_updates_the_stream_in_place(m, n, k):
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(m * 3 + n + k)
    x = torch.randn(m, n, generator=g) * 2.0
    a_full = torch.randn(m, k + 64, generator=g).to(torch.bfloat16)         # a row-strided operand view
    w, b = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16), torch.randn(n, generator=g)
    ref = x.double() + a_full[:, :k].double() @ w.double().t() + b.double()
    xd = x.to(DEV)
    xb = torch.full((m, n), 7.0, dtype=torch.bfloat16, device=DEV)
    part = torch.full((m, n // 32, 2), -1.0, device=DEV)
    ops.gemm_bf16_resid_(xd, a_full.to(DEV)[:, :k], w.to(DEV), b.to(DEV), xb, part)
    assert (xd.cpu().double() - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(xb.cpu(), xd.cpu().to(torch.bfloat16))
    xg = xd.cpu().double().view(m, n // 32, 32)
    assert torch.allclose(part[..., 0].cpu().double(), xg.sum(-1), atol=1e-3) and torch.allclose(part[..., 1].cpu().double(), (xg * xg).sum(-1), rtol=1e-5, atol=1e-3)
    stats, _ = ops.vit_row_stats(part=part, d=n, eps=1e-6)
    assert torch.allclose(stats[:, 0].cpu().double(), xd.cpu().double().mean(1), atol=1e-5)
    assert torch.allclose(stats[:, 1].cpu().double(), (xd.cpu().double().var(1, unbiased=False) + 1e-6).rsqrt(), rtol=2e-4)
    # bit-reproducible
    xd2, xb2, part2 = x.to(DEV), torch.empty_like(xb), torch.empty_like(part)
    ops.gemm_bf16_resid_(xd2, a_full.to(DEV)[:, :k], w.to(DEV), b.to(DEV), xb2, part2)
    assert torch.equal(xd, xd2) and torch.equal(xb, xb2) and torch.equal(part, part2)


@pytest.mark.parametrize("n,p,q", [(32768, 768, 3072), (4096, 256, 256), (2048, 520, 264), (8192, 1536, 768), (1024, 384, 1536)])
@pytest.mark.parametrize("x3", [True, False])
def test_gemm_tn_weight_gradient_contraction(n, p, q, x3):
    """snf_gemm_tn_f32: a^T b over the bag axis straight from row-major bf16 images (transposing LDS reads), fp32-class from split
    images [hi | hi | lo] or one bf16 product -- against fp64 of the same operands, bit-reproducible, ragged tile edges."""
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(n + p + q)
    a = torch.randn(n, p, generator=g).to(DEV)
    b = torch.randn(n, q, generator=g).to(DEV)
    if x3:
        a3, b3 = ops.split3_rows(a), ops.split3_rows(b)
        out = ops.gemm_tn(a3, b3, p, q, (p, 2 * p), (q, 2 * q))
        ref = a.double().t() @ b.double()
        tol = 2e-5
        assert torch.equal(out, ops.gemm_tn(a3, b3, p, q, (0, 2 * p), (q, 2 * q)))       # either copy of the hi plane, run to run
    else:
        a16, b16 = a.to(torch.bfloat16), b.to(torch.bfloat16)
        pad = torch.zeros(n, 8, dtype=torch.bfloat16, device=DEV)
        a_v = torch.cat([pad, a16, pad], 1)                                                # a plane inside wider rows
        out = ops.gemm_tn(a_v, b16, p, q, (8, -1), (0, -1))
        ref = a16.double().t() @ b16.double()
        tol = 2e-6
        assert torch.equal(out, ops.gemm_tn(a16, b16, p, q))
    scale = ref.abs().max().item()
    assert out.shape == (p, q) and (out.double() - ref).abs().max().item() <= tol * scale


@pytest.mark.parametrize("m,k,scaled", [(1536, 768, True), (3072, 768, False), (768, 3072, False), (100, 40, True)])
def test_split3_weight_image_in_one_launch(m, k, scaled):
    """snf_split3_weight_f32 == the elementwise formulation, bit for bit: [Wh | Wl | Wh] of W (x diag(gamma) in fp32)."""
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(m + k)
    w = torch.randn(m, k, generator=g).to(DEV)
    gam = (torch.rand(k, generator=g) + 0.5).to(DEV) if scaled else None
    out = ops.split3_weight(w, gam)
    wf = w * gam if scaled else w
    hi = wf.to(torch.bfloat16)
    lo = (wf - hi.float()).to(torch.bfloat16)
    assert torch.equal(out, torch.cat([hi, lo, hi], 1))


@pytest.mark.parametrize("n,p,q", [(32768, 768, 3072), (2048, 512, 288), (4096, 1536, 768)])
def test_gemm_tn_interleaved_images(n, p, q):
    """snf_gemm_tn_f32 on hl images ([hi(32) | lo(32)] per 32 columns): the same products as on plane images, an operand that starts
    inside a wider image, fp64 reference."""
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(n + p + q + 1)
    a = torch.randn(n, p, generator=g).to(DEV)
    b = torch.randn(n, q, generator=g).to(DEV)
    a_hl, b_hl = ops.split_hl_rows(a), ops.split_hl_rows(b)
    out = ops.gemm_tn(a_hl, b_hl, p, q, hl=True)
    ref = a.double().t() @ b.double()
    assert (out.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    planes = ops.gemm_tn(ops.split3_rows(a), ops.split3_rows(b), p, q, (p, 2 * p), (q, 2 * q))
    assert (out - planes).abs().max().item() <= 2e-6 * ref.abs().max().item()            # same products, another summation order
    assert torch.equal(out, ops.gemm_tn(a_hl, b_hl, p, q, hl=True))
    if p % 64 == 0:
        half = ops.gemm_tn(a_hl, b_hl, p // 2, q, (p, -1), (0, -1), hl=True)                # the right half of a's columns
        assert (half.double() - ref[p // 2:]).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("m,k,gated", [(32768, 3072, True), (5000, 768, False), (1000, 1536, True)])
def test_split_hl_colsum(m, k, gated):
    """snf_split_hl_colsum_f32 == split_hl_rows of the gated matrix, plus its column sums; two matrices sharing one image."""
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(m + k)
    x = torch.randn(m, k, generator=g).to(DEV)
    act = torch.randn(m, k, generator=g).to(DEV)
    gate_hl = ops.split_hl_rows(torch.relu(act)) if gated else None
    img, cs = ops.split_hl_colsum(x, gate_hl)
    xg = x * (torch.relu(act).to(torch.bfloat16) > 0) if gated else x
    assert torch.equal(img, ops.split_hl_rows(xg.contiguous()))
    assert torch.allclose(cs.double().cpu(), xg.double().sum(0).cpu(), rtol=1e-5, atol=1e-3)
    wide = torch.zeros(m, 4 * k, dtype=torch.bfloat16, device=DEV)
    ops.split_hl_colsum(x, out=wide, col=0, want_colsum=False)
    ops.split_hl_colsum(xg.contiguous(), out=wide, col=k, want_colsum=False)
    assert torch.equal(wide[:, :2 * k], ops.split_hl_rows(x)) and torch.equal(wide[:, 2 * k:], ops.split_hl_rows(xg.contiguous()))


@pytest.mark.parametrize("m,n,k", [(32768, 3072, 768), (5000, 544, 256), (700, 1024, 96)])
def test_gemm_hl_gated_epilogue(m, n, k):
    """snf_gemm_hl_gated_bf16 == the hl image of gemm_hl's result with the elements behind a closed gate zeroed (the gate: hi values of the
    activation's own hl image), bit for bit; hl_colsum of it == the column sums."""
    from snuffy_amd import ops
    g = torch.Generator().manual_seed(m + n + k)
    a = (torch.randn(m, k, generator=g) * 0.5).to(DEV)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(DEV)
    act = torch.relu(torch.randn(m, n, generator=g)).to(DEV)
    a_hl, w_hl, gate_hl = ops.split_hl_rows(a), ops.split_hl_weight(w), ops.split_hl_rows(act)
    out = ops.gemm_hl_gated(a_hl, w_hl, gate_hl)
    plain = ops.gemm_hl(a_hl, w_hl)
    ref = plain * (act.to(torch.bfloat16) > 0)
    assert torch.equal(out, ops.split_hl_rows(ref.contiguous()))
    assert torch.equal(out, ops.gemm_hl_gated(a_hl, w_hl, gate_hl))
    if 2 * n <= 8192:
        cs = ops.hl_colsum(out)
        assert torch.allclose(cs.double().cpu(), ref.double().sum(0).cpu(), rtol=1e-4, atol=1e-2 * max(1.0, ref.abs().max().item()))
