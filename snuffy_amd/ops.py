"""Tensor-level wrappers over the C ABI (include/snuffy_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every kernel is ours.  All functions require CUDA
(ROCm) tensors and raise otherwise -- there is no CPU path in the product.
"""
import ctypes
import math

import torch

from . import _ffi
from ._ffi import ACT_CODES, DT_BF16, DT_BF16_HL, DT_BF16_SPLIT3, DT_F32, TOPK_MAX_K, check


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a Python Stream
    object through several device-index helpers (~9 us per call, ~50 us per bag); the raw getter is a single C call."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on_current_device(t, name):
    """Kernels launch on torch's current stream of the CURRENT device: a tensor living on another GPU would be addressed from
    the wrong device -- refuse loudly (wrap the call in ``torch.cuda.device(t.device)`` or call ``torch.cuda.set_device``)."""
    if t.device.index != torch.cuda.current_device():
        raise _ffi.SnuffyHipError("%s lives on %s but the current device is cuda:%d: select the tensor's device first "
                                  "(torch.cuda.set_device / torch.cuda.device)" % (name, t.device, torch.cuda.current_device()))


def _req(t, dtype, name, dim=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _ffi.SnuffyHipError(
            "%s must be a GPU tensor: snuffy_amd runs on MI355X only (no CPU fallback)" % name)
    _on_current_device(t, name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if dim is not None and t.dim() != dim:
        raise ValueError("%s must be %d-D, got shape %s" % (name, dim, tuple(t.shape)))
    return t if t.is_contiguous() else t.contiguous()


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ----------------------------------------------------------------------------------------------------------------------
def critic(x, w, b=None, want_max=False):
    """scores = x w^T + b  (FCLayer, snuffy.py:39-41); optionally the column max / argmax (train.py:831-834)."""
    x = _req(x, torch.float32, "x", 2)
    w = _req(w, torch.float32, "w", 2)
    if b is not None:
        b = _req(b, torch.float32, "b", 1)
    n, d = x.shape
    c = w.shape[0]
    if w.shape[1] != d:
        raise ValueError("critic: w is %s but x has %d features" % (tuple(w.shape), d))
    scores = torch.empty(n, c, dtype=torch.float32, device=x.device)
    mv = mi = None
    if want_max:
        mv = torch.empty(c, dtype=torch.float32, device=x.device)
        mi = torch.empty(c, dtype=torch.int64, device=x.device)
    check(_ffi.load().snf_critic_f32(_p(x), n, d, _p(w), _p(b), c, _p(scores), _p(mv), _p(mi), _stream()),
          "snf_critic_f32")
    return (scores, mv, mi) if want_max else scores


def critic_ln(x, w, b, eps):
    """One pass over the bag: (scores [n, c] f32, xhat [n, d] bf16 = affine-free LayerNorm of x).  See snf_critic_ln_f32."""
    x = _req(x, torch.float32, "x", 2)
    w = _req(w, torch.float32, "w", 2)
    if b is not None:
        b = _req(b, torch.float32, "b", 1)
    n, d = x.shape
    c = w.shape[0]
    if w.shape[1] != d:
        raise ValueError("critic_ln: w is %s but x has %d features" % (tuple(w.shape), d))
    scores = torch.empty(n, c, dtype=torch.float32, device=x.device)
    xhat = torch.empty(n, d, dtype=torch.bfloat16, device=x.device)
    check(_ffi.load().snf_critic_ln_f32(_p(x), n, d, _p(w), _p(b), c, _p(scores), float(eps), _p(xhat), _stream()),
          "snf_critic_ln_f32")
    return scores, xhat


# ----------------------------------------------------------------------------------------------------------------------
# fused selector (snf_critic_select_f32 + snf_topk_select_f32): the critic pass counts the first radix digit of its scores
# into a small device-resident state; the selection that follows starts from it.  One state per (device, stream) -- a selection is
# issued on the stream of its critic pass, right behind it; `pending` remembers which score tensor the state describes.
# ----------------------------------------------------------------------------------------------------------------------
SELECT_FUSED_MIN_N = 16385       # up to 16 k scores the one-workgroup selection (keys in registers, 9-12 us) is as fast or faster
_SELECTORS = {}


class _Selector:
    def __init__(self, device):
        lib = _ffi.load()
        self.nbytes = int(lib.snf_selector_state_bytes())
        self.state = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)   # zeroed ONCE; every select re-zeroes it
        self.pending = None


def _selector_key(device):
    """One selector state per (device, stream): the critic pass and the selection that consumes its histogram run back to back on ONE
    stream, and two streams (threads) of a device must not count into the same histogram (re-entrant)."""
    raw = _raw_stream(device.index) if _raw_stream is not None else torch.cuda.current_stream(device).cuda_stream
    return (device.index, int(raw))


def selector(device, create=True):
    """The selector state of the current stream of `device`; None while a HIP graph is being captured and none exists yet for the
    capturing stream (the graph warm-up forwards, which run on that stream, allocate it outside the capture)."""
    key = _selector_key(device)
    sel = _SELECTORS.get(key)
    if sel is None and create and not torch.cuda.is_current_stream_capturing():
        sel = _SELECTORS[key] = _Selector(device)
    return sel


def fresh_selector(device):
    """A NEW selector state for the current stream of `device` (replacing the one registered for it): what a HIP-graph capture calls
    on its capture stream before the warm-up forwards -- the graph bakes the state's address in, so every graph gets a state of its
    own (two graphs replayed on different streams never share a histogram) and keeps the returned object alive."""
    sel = _SELECTORS[_selector_key(device)] = _Selector(device)
    return sel


def critic_select(x, w, b, eps=None):
    """The critic pass with the selector's histogram: scores [n, 1] f32 (and xhat [n, d] bf16, the affine-free LayerNorm of x,
    when eps is given -- as critic_ln).  The following topk(scores.view(-1), k) finishes the selection from the histogram.
    Returns (scores, xhat or None), or None when the fused form does not apply (then use critic / critic_ln)."""
    if w.shape[0] != 1 or x.shape[0] < SELECT_FUSED_MIN_N or x.shape[0] >= (1 << 30):
        return None
    sel = selector(x.device)
    if sel is None:
        return None
    x = _req(x, torch.float32, "x", 2)
    w = _req(w, torch.float32, "w", 2)
    if b is not None:
        b = _req(b, torch.float32, "b", 1)
    n, d = x.shape
    if w.shape[1] != d:
        raise ValueError("critic_select: w is %s but x has %d features" % (tuple(w.shape), d))
    if sel.pending is not None:          # an earlier histogram was never consumed: start from a clean state
        sel.state.zero_()
        sel.pending = None
    scores = torch.empty(n, 1, dtype=torch.float32, device=x.device)
    xhat = torch.empty(n, d, dtype=torch.bfloat16, device=x.device) if eps is not None else None
    check(_ffi.load().snf_critic_select_f32(_p(x), n, d, _p(w), _p(b), _p(scores), float(eps or 0.0), _p(xhat), _p(sel.state),
                                            _stream()), "snf_critic_select_f32")
    sel.pending = (scores.data_ptr(), n, scores._version)
    return scores, xhat


def critic_ln_hl(x, w, b, gamma, beta, eps):
    """One pass over the bag for the fp32-class path: (scores [n, c] f32, LayerNorm(x) with its affine as the interleaved hi / lo
    image [n, 2 d] bf16 -- the operand layernorm_rows_hl() would produce, to the last fp32 place of the normalised value).  With one class and a bag long enough for
    the fused selector the pass also counts the selector's histogram (the following topk() starts from it)."""
    x = _req(x, torch.float32, "x", 2)
    w = _req(w, torch.float32, "w", 2)
    if b is not None:
        b = _req(b, torch.float32, "b", 1)
    if (gamma is None) != (beta is None):
        raise ValueError("critic_ln_hl: gamma and beta: both or neither (neither = the affine-free image)")
    if gamma is not None:
        gamma = _req(gamma, torch.float32, "gamma", 1)
        beta = _req(beta, torch.float32, "beta", 1)
    n, d = x.shape
    c = w.shape[0]
    if w.shape[1] != d or d % 32:
        raise ValueError("critic_ln_hl: w is %s, x has %d features (need d %% 32 == 0)" % (tuple(w.shape), d))
    sel = selector(x.device) if (c == 1 and SELECT_FUSED_MIN_N <= n < (1 << 30)) else None
    if sel is not None and sel.pending is not None:
        sel.state.zero_()
        sel.pending = None
    scores = torch.empty(n, c, dtype=torch.float32, device=x.device)
    img = torch.empty(n, 2 * d, dtype=torch.bfloat16, device=x.device)
    check(_ffi.load().snf_critic_ln_hl_f32(_p(x), n, d, _p(w), _p(b), c, _p(scores), _p(gamma), _p(beta), float(eps), _p(img),
                                           _p(sel.state) if sel is not None else None, _stream()), "snf_critic_ln_hl_f32")
    if sel is not None:
        sel.pending = (scores.data_ptr(), n, scores._version)
    return scores, img


def topk(scores, k, x=None):
    """Indices of the k largest scores, descending, ties by ascending index (snuffy.py:128-130).

    scores: 1-D (any stride) f32.  With x [n, d] also returns the gathered rows x[idx] (snuffy.py:131).
    """
    if not scores.is_cuda:
        raise _ffi.SnuffyHipError("topk: scores must be a GPU tensor (no CPU fallback)")
    if scores.dtype != torch.float32 or scores.dim() != 1:
        raise TypeError("topk: scores must be 1-D float32")
    n = scores.shape[0]
    stride = scores.stride(0) if n > 1 else 1
    if stride < 1:
        scores = scores.contiguous()
        stride = 1
    k = int(k)
    if not (1 <= k <= n):
        raise ValueError("topk: need 1 <= k <= n (k=%d, n=%d)" % (k, n))
    lib = _ffi.load()
    idx = torch.empty(k, dtype=torch.int64, device=scores.device)
    sel = _SELECTORS.get(_selector_key(scores.device))
    if sel is not None and sel.pending is not None:
        if x is None and stride == 1 and k <= TOPK_MAX_K and sel.pending == (scores.data_ptr(), n, scores._version):
            # these scores came out of critic_select: their first-digit histogram is waiting in the selector state
            _on_current_device(scores, "scores")
            sel.pending = None
            check(lib.snf_topk_select_f32(_p(scores), n, k, _p(idx), _p(sel.state), _stream()), "snf_topk_select_f32")
            return idx
    wsb = lib.snf_topk_workspace_bytes(n, k)
    ws = _ws(wsb, scores.device)
    if x is None:
        check(lib.snf_topk_f32(_p(scores), n, stride, k, _p(idx), _p(ws), wsb, _stream()), "snf_topk_f32")
        return idx
    x = _req(x, torch.float32, "x", 2)
    xs = torch.empty(k, x.shape[1], dtype=torch.float32, device=x.device)
    check(lib.snf_topk_gather_f32(_p(scores), n, stride, k, _p(idx), _p(x), x.shape[1], _p(xs), _p(ws), wsb,
                                  _stream()), "snf_topk_gather_f32")
    return idx, xs


def topk_hist_select(scores, k):
    """topk() through the multi-workgroup form whatever n is (histogram launch + select launch on a scratch state)."""
    if not scores.is_cuda or scores.dtype != torch.float32 or scores.dim() != 1:
        raise TypeError("topk_hist_select: scores must be a 1-D float32 GPU tensor")
    _on_current_device(scores, "scores")
    n = scores.shape[0]
    stride = scores.stride(0) if n > 1 else 1
    lib = _ffi.load()
    idx = torch.empty(int(k), dtype=torch.int64, device=scores.device)
    st = _ws(lib.snf_selector_state_bytes(), scores.device)
    check(lib.snf_topk_hist_select_f32(_p(scores), n, stride, int(k), _p(idx), _p(st), _stream()), "snf_topk_hist_select_f32")
    return idx


class DeviceSampler:
    """State of the device-side random patch share (csrc/sampler.hip): a 16-byte device record {seed, offset}.  The seed is torch's
    CPU seed at construction and the first offset is drawn from torch's CPU generator (reproducible under torch.manual_seed, no
    device sync); every forward advances the offset ON THE DEVICE, inside a captured graph as well."""

    def __init__(self, device, seed=None, offset=None):
        seed = int(torch.initial_seed()) if seed is None else int(seed)
        offset = int(torch.randint(0, 2 ** 40, (1,), dtype=torch.int64).item()) if offset is None else int(offset)
        self.seed, self.offset0 = seed & (2 ** 63 - 1), offset & (2 ** 47 - 1)
        self.state = torch.tensor([self.seed, self.offset0], dtype=torch.int64, device=device)

    def advance(self):
        check(_ffi.load().snf_sampler_advance(_p(self.state), _stream()), "snf_sampler_advance")

    def draw(self, n, k2, exclude, layer=0):
        """k2 rows of 0 .. n - 1 outside `exclude` [k1] int64: uniform without replacement, random order -> [k2] int64."""
        keys = torch.empty(n, dtype=torch.float32, device=self.state.device)
        ne = 0 if exclude is None else int(exclude.shape[0])
        if ne:
            exclude = _req(exclude, torch.int64, "exclude", 1)
        if not (0 <= k2 <= n - ne):
            raise ValueError("DeviceSampler.draw: %d rows wanted, %d available" % (k2, n - ne))
        check(_ffi.load().snf_random_share_keys_f32(_p(self.state), int(layer), n, _p(exclude) if ne else None, ne, _p(keys), _stream()),
              "snf_random_share_keys_f32")
        if k2 == 0:
            return torch.empty(0, dtype=torch.int64, device=keys.device)
        return topk(keys, k2)


def gather_rows(x, idx):
    x = _req(x, torch.float32, "x", 2)
    idx = _req(idx, torch.int64, "idx", 1)
    out = torch.empty(idx.shape[0], x.shape[1], dtype=torch.float32, device=x.device)
    check(_ffi.load().snf_gather_rows_f32(_p(x), x.shape[0], x.shape[1], _p(idx), idx.shape[0], _p(out), _stream()),
          "snf_gather_rows_f32")
    return out


def gather_slot_map(x, idx, bf16_copy=False):
    """(x[idx] [k, d], map [n] int32 with map[idx[j]] = j, -1 elsewhere) in one launch; idx must be duplicate-free.
    bf16_copy: also return x[idx] rounded to bf16 (third element) -- the operand of the bf16 key projection."""
    x = _req(x, torch.float32, "x", 2)
    idx = _req(idx, torch.int64, "idx", 1)
    n, d = x.shape
    k = idx.shape[0]
    if k > 2048:
        xs = gather_rows(x, idx)
        return (xs, slot_map(idx, n), xs.to(torch.bfloat16)) if bf16_copy else (xs, slot_map(idx, n))
    xs = torch.empty(k, d, dtype=torch.float32, device=x.device)
    m = torch.empty(n, dtype=torch.int32, device=x.device)
    xs16 = torch.empty(k, d, dtype=torch.bfloat16, device=x.device) if bf16_copy else None
    check(_ffi.load().snf_gather_slot_map_f32(_p(x), n, d, _p(idx), k, _p(xs), _p(m), _p(xs16), _stream()),
          "snf_gather_slot_map_f32")
    return (xs, m, xs16) if bf16_copy else (xs, m)


def scatter_rows(x, idx, rows, inplace=False):
    """y = x.clone(); y[idx] = rows  (snuffy.py:154-155); in place when asked."""
    x = _req(x, torch.float32, "x", 2)
    idx = _req(idx, torch.int64, "idx", 1)
    rows = _req(rows, torch.float32, "rows", 2)
    y = x if inplace else torch.empty_like(x)
    check(_ffi.load().snf_scatter_rows_f32(_p(x), x.shape[0], x.shape[1], _p(idx), idx.shape[0], _p(rows), _p(y),
                                           _stream()), "snf_scatter_rows_f32")
    return y


def scatter_add_rows_(z, idx, delta):
    z = _req(z, torch.float32, "z", 2)
    idx = _req(idx, torch.int64, "idx", 1)
    delta = _req(delta, torch.float32, "delta", 2)
    check(_ffi.load().snf_scatter_add_rows_f32(_p(z), z.shape[0], z.shape[1], _p(idx), idx.shape[0], _p(delta),
                                               _stream()), "snf_scatter_add_rows_f32")
    return z


def slot_map(idx, n):
    idx = _req(idx, torch.int64, "idx", 1)
    m = torch.empty(n, dtype=torch.int32, device=idx.device)
    check(_ffi.load().snf_slot_map_i32(_p(idx), idx.shape[0], n, _p(m), _stream()), "snf_slot_map_i32")
    return m


def layernorm_rows(x, gamma=None, beta=None, eps=1e-5, slot=None, patch_rows=None, out_dtype=torch.float32,
                   want_stats=False, out=None, out_row_idx=None):
    """LayerNorm over rows (snuffy.py:97,107); rows listed in `slot` are read from patch_rows instead of x.

    out / out_row_idx: write row i of the result to out[out_row_idx[i]] (re-normalise patched rows in place)."""
    x = _req(x, torch.float32, "x", 2)
    n, d = x.shape
    if gamma is not None:
        gamma = _req(gamma, torch.float32, "gamma", 1)
    if beta is not None:
        beta = _req(beta, torch.float32, "beta", 1)
    if slot is not None:
        slot = _req(slot, torch.int32, "slot", 1)
        patch_rows = _req(patch_rows, torch.float32, "patch_rows", 2)
    if out is None:
        if out_row_idx is not None:
            raise ValueError("layernorm_rows: out_row_idx needs an explicit out buffer")
        out = torch.empty(n, d, dtype=out_dtype, device=x.device)
    else:
        out_dtype = out.dtype
        if not out.is_contiguous() or out.shape[1] != d:
            raise ValueError("layernorm_rows: bad out buffer")
    if out_row_idx is not None:
        out_row_idx = _req(out_row_idx, torch.int64, "out_row_idx", 1)
        if want_stats:
            raise ValueError("layernorm_rows: stats with out_row_idx not supported")
    of32 = out if out_dtype == torch.float32 else None
    obf = out if out_dtype == torch.bfloat16 else None
    if of32 is None and obf is None:
        raise TypeError("layernorm_rows: out_dtype must be float32 or bfloat16")
    mean = rstd = None
    if want_stats:
        mean = torch.empty(n, dtype=torch.float32, device=x.device)
        rstd = torch.empty(n, dtype=torch.float32, device=x.device)
    check(_ffi.load().snf_layernorm_rows_f32(_p(x), n, d, _p(slot), _p(patch_rows), _p(gamma), _p(beta), float(eps),
                                             _p(of32), _p(obf), _p(mean), _p(rstd), _p(out_row_idx), _stream()),
          "snf_layernorm_rows_f32")
    return (out, mean, rstd) if want_stats else out


def layernorm_rows_bwd(x, dy, gamma=None, eps=1e-5, residual=None, want_dx=True, want_dx_bf16=False, want_param_grads=True):
    """Backward of layernorm_rows (snf_layernorm_rows_bwd_f32): x [n, d] f32 (the forward's input), dy [n, d] f32 / bf16 or
    ONE row [d] broadcast to all rows.  Returns (dx f32 or None, dx bf16 or None, dgamma_unscaled [d], dbeta [d]) where
    dgamma_unscaled = sum_rows dy * xhat and dbeta = sum_rows dy (None when not asked for)."""
    x = _req(x, torch.float32, "x", 2)
    n, d = x.shape
    if dy.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("layernorm_rows_bwd: dy must be float32 or bfloat16")
    if dy.dim() == 1:
        dy = _req(dy, dy.dtype, "dy", 1)
        stride = 0
        if dy.shape[0] != d:
            raise ValueError("layernorm_rows_bwd: broadcast dy has %d entries, need %d" % (dy.shape[0], d))
    else:
        dy = _rows16(dy, "dy")
        if dy.shape != x.shape:
            raise ValueError("layernorm_rows_bwd: dy shape %s does not match x %s" % (tuple(dy.shape), tuple(x.shape)))
        stride = dy.stride(0)
    if gamma is not None:
        gamma = _req(gamma, torch.float32, "gamma", 1)
    if residual is not None:
        residual = _req(residual, torch.float32, "residual", 2)
    lib = _ffi.load()
    dx = torch.empty_like(x) if want_dx else None
    dxb = torch.empty(n, d, dtype=torch.bfloat16, device=x.device) if want_dx_bf16 else None
    part = torch.empty(lib.snf_layernorm_bwd_blocks(n), 2, d, dtype=torch.float32, device=x.device) if want_param_grads else None
    check(lib.snf_layernorm_rows_bwd_f32(_p(x), n, d, _p(dy), DT_F32 if dy.dtype == torch.float32 else DT_BF16, stride,
                                         _p(gamma), float(eps), _p(residual), _p(dx), _p(dxb), _p(part), _stream()),
          "snf_layernorm_rows_bwd_f32")
    if part is None:
        return dx, dxb, None, None
    sums = part.sum(0)
    return dx, dxb, sums[0], sums[1]


def colsum_fused(src, row_weight=None, gate=None, want_bf16=False, inplace=False):
    """Column sums of src [n, d] (f32 or bf16) fused with an optional row weight [n] (f32, any stride), an optional ReLU gate
    (bf16 [n, d]: elements whose gate is <= 0 are zeroed) and an optional bf16 copy of the result (inplace: written over a
    bf16 src).  Returns (colsum [d] f32, bf16 copy or None).  See snf_colsum_fused."""
    if src.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("colsum_fused: src must be float32 or bfloat16")
    src = _req(src, src.dtype, "src", 2)
    n, d = src.shape
    if d % 8 or d > 8192:
        raise ValueError("colsum_fused: d=%d must be a multiple of 8 and <= 8192" % d)
    ws = 0
    if row_weight is not None:
        if row_weight.dtype != torch.float32 or not row_weight.is_cuda or row_weight.dim() != 1 or row_weight.shape[0] != n:
            raise ValueError("colsum_fused: row_weight must be a 1-D float32 GPU tensor of %d values (any stride)" % n)
        ws = row_weight.stride(0)
    if gate is not None:
        gate = _req(gate, torch.bfloat16, "gate", 2)
        if gate.shape != src.shape:
            raise ValueError("colsum_fused: gate shape %s does not match src %s" % (tuple(gate.shape), tuple(src.shape)))
    dst = None
    if inplace:
        if src.dtype != torch.bfloat16:
            raise TypeError("colsum_fused: inplace needs a bfloat16 src")
        dst = src
    elif want_bf16:
        dst = torch.empty(n, d, dtype=torch.bfloat16, device=src.device)
    lib = _ffi.load()
    part = torch.empty(lib.snf_colsum_blocks(n), d, dtype=torch.float32, device=src.device)
    check(lib.snf_colsum_fused(_p(src), DT_F32 if src.dtype == torch.float32 else DT_BF16, n, d, _p(row_weight), ws, _p(gate),
                               _p(dst), _p(part), _stream()), "snf_colsum_fused")
    return part.sum(0), dst


def fold_linear(w, gamma, beta, bias, out_w, out_bf=None, out_bf16=None):
    """out_w[r, c] = bf16(w[r, c] * gamma[c]) written into out_w (a [r, c] bf16 view, rows may be strided);
    out_bf / out_bf16 [r] = w @ beta + bias as f32 / bf16 (either may be None).  See snf_fold_linear_f32."""
    w = _req(w, torch.float32, "w", 2)
    r, c = w.shape
    gamma = _req(gamma, torch.float32, "gamma", 1)
    beta = _req(beta, torch.float32, "beta", 1)
    if bias is not None:
        bias = _req(bias, torch.float32, "bias", 1)
    if out_w.dtype != torch.bfloat16 or out_w.shape != (r, c) or out_w.stride(1) != 1:
        raise ValueError("fold_linear: out_w must be a [%d, %d] bfloat16 view with unit column stride" % (r, c))
    check(_ffi.load().snf_fold_linear_f32(_p(w), r, c, _p(gamma), _p(beta), _p(bias), _p(out_w), out_w.stride(0), _p(out_bf),
                                          _p(out_bf16), _stream()), "snf_fold_linear_f32")


def unfold_linear(dwf, w, gamma, beta, dbf):
    """Gradients of a folded projection back to its parameters: returns (dW [r, c], dgamma [c], dbeta [c]) with
    dW = dWf * gamma + outer(dbf, beta), dgamma = sum_r dWf * W, dbeta = W^T dbf.  See snf_unfold_linear_f32."""
    dwf = _req(dwf, torch.float32, "dwf", 2)
    w = _req(w, torch.float32, "w", 2)
    r, c = w.shape
    gamma = _req(gamma, torch.float32, "gamma", 1)
    beta = _req(beta, torch.float32, "beta", 1)
    dbf = _req(dbf, torch.float32, "dbf", 1)
    lib = _ffi.load()
    dw = torch.empty(r, c, dtype=torch.float32, device=w.device)
    part = torch.empty(lib.snf_fold_blocks(r), 2, c, dtype=torch.float32, device=w.device)
    check(lib.snf_unfold_linear_f32(_p(dwf), _p(w), r, c, _p(gamma), _p(beta), _p(dbf), _p(dw), _p(part), _stream()),
          "snf_unfold_linear_f32")
    sums = part.sum(0)
    return dw, sums[0], sums[1]


def bias_act_(h, bias, act):
    """h = act(h + bias) in place (snuffy.py:224-225)."""
    if not h.is_cuda or not h.is_contiguous() or h.dim() != 2:
        raise _ffi.SnuffyHipError("bias_act_: h must be a contiguous 2-D GPU tensor")
    dt = {torch.float32: DT_F32, torch.bfloat16: DT_BF16}.get(h.dtype)
    if dt is None:
        raise TypeError("bias_act_: h must be float32 or bfloat16")
    if bias is not None:
        bias = _req(bias, torch.float32, "bias", 1)
    check(_ffi.load().snf_bias_act(_p(h), dt, h.shape[0], h.shape[1], _p(bias), ACT_CODES[act], _stream()),
          "snf_bias_act")
    return h


def ln_mean_head(z, gamma, beta, eps, w_head, b_head, add_bf16=None, add_bias=None, slot=None, delta_rows=None,
                 want_z=False):
    """logits = W_head mean_n(LN(z')) + b_head  (snuffy.py:86,71) where z' = z (+ add_bf16) (+ add_bias)
    (+ delta_rows[slot]).  Returns (logits [C], pooled [D], z' or None)."""
    z = _req(z, torch.float32, "z", 2)
    n, d = z.shape
    if add_bf16 is not None:
        add_bf16 = _req(add_bf16, torch.bfloat16, "add_bf16", 2)
    if add_bias is not None:
        add_bias = _req(add_bias, torch.float32, "add_bias", 1)
    if slot is not None:
        slot = _req(slot, torch.int32, "slot", 1)
        delta_rows = _req(delta_rows, torch.float32, "delta_rows", 2)
    z_out = torch.empty_like(z) if want_z else None
    gamma = _req(gamma, torch.float32, "gamma", 1)
    beta = _req(beta, torch.float32, "beta", 1)
    w_head = _req(w_head, torch.float32, "w_head", 2)
    if b_head is not None:
        b_head = _req(b_head, torch.float32, "b_head", 1)
    c = w_head.shape[0]
    lib = _ffi.load()
    logits = torch.empty(c, dtype=torch.float32, device=z.device)
    pooled = torch.empty(d, dtype=torch.float32, device=z.device)
    wsb = lib.snf_ln_mean_head_workspace_bytes(d)
    ws = _ws(wsb, z.device)
    check(lib.snf_ln_mean_head_f32(_p(z), n, d, _p(add_bf16), _p(add_bias), _p(slot), _p(delta_rows), _p(z_out),
                                   _p(gamma), _p(beta), float(eps), _p(w_head), _p(b_head), c, _p(logits), _p(pooled),
                                   _p(ws), wsb, _stream()), "snf_ln_mean_head_f32")
    return logits, pooled, z_out


def attn_bwd_dropout_supported(k, dk):
    """sparse_attn_bwd regenerates the forward's dropout mask in its kernels (dropout=...) instead of reading a mask tensor."""
    return 1 <= k <= 1024 and k % 4 == 0 and dk % 8 == 0


def sparse_attn_bwd(q, kp, v, p, dout, h, mask=None, scale=None, dropout=None):
    """Exact-fp32 backward of sparse_attn_fwd: (dq [n,d], dkp [k,d], dv [n,d]).  p [h,n,k] = forward probabilities,
    mask (optional) = dropout keep-mask already divided by (1 - p_drop), dout [k, d].  q / v may be the row-strided halves of a fused
    [Q | V] projection output where the matrix-core kernels apply (dk % 8 == 0, k <= 1024); otherwise they are made contiguous.
    dropout = (p_drop, seed, offset) instead of mask: the kernels regenerate the forward's Philox mask (attn_bwd_dropout_supported)."""
    if q.dtype != torch.float32 or v.dtype != torch.float32:
        raise TypeError("sparse_attn_bwd: q and v must be float32")
    kp = _req(kp, torch.float32, "kp", 2)
    if q.dim() == 2 and kp.dim() == 2 and (q.shape[1] // h) % 8 == 0 and kp.shape[0] <= 1024:
        q, v = _rows16(q, "q"), _rows16(v, "v")
    else:
        q, v = _req(q, torch.float32, "q", 2), _req(v, torch.float32, "v", 2)
    p = _req(p, torch.float32, "p", 3)
    dout = _req(dout, torch.float32, "dout", 2)
    if mask is not None:
        mask = _req(mask, torch.float32, "mask", 3)
    n, d = q.shape
    k = kp.shape[0]
    dk = d // h
    if p.shape != (h, n, k) or dout.shape != (k, d) or (mask is not None and mask.shape != p.shape):
        raise ValueError("sparse_attn_bwd: inconsistent shapes")
    scale = 1.0 / math.sqrt(dk) if scale is None else scale
    lib = _ffi.load()
    dq = torch.empty(n, d, dtype=torch.float32, device=q.device)
    dv = torch.empty(n, d, dtype=torch.float32, device=q.device)
    dkp = torch.empty_like(kp)
    wsb = lib.snf_sparse_attn_bwd_workspace_bytes(n, k, h, dk)
    ws = _ws(wsb, q.device)
    if dropout is not None and mask is None and float(dropout[0]) > 0.0:
        check(lib.snf_sparse_attn_bwd_dropout_f32(_p(q), q.stride(0), _p(kp), _p(v), v.stride(0), _p(p), float(dropout[0]),
                                                  int(dropout[1]) & (2 ** 64 - 1), int(dropout[2]) & (2 ** 64 - 1), _p(dout), n, k, h, dk,
                                                  float(scale), _p(dq), _p(dkp), _p(dv), _p(ws), wsb, _stream()),
              "snf_sparse_attn_bwd_dropout_f32")
        return dq, dkp, dv
    check(lib.snf_sparse_attn_bwd_ld_f32(_p(q), q.stride(0), _p(kp), _p(v), v.stride(0), _p(p), _p(mask), _p(dout), n, k, h, dk, float(scale),
                                         _p(dq), _p(dkp), _p(dv), _p(ws), wsb, _stream()), "snf_sparse_attn_bwd_ld_f32")
    return dq, dkp, dv


def mfma_attn_bwd_supported(k, dk):
    return (dk == 128 and 1 <= k <= 224) or (dk == 64 and 1 <= k <= 256)


def sparse_attn_bwd_mfma(q, v, kp, dout, lse, h, mask=None, scale=None, dropout=None, fused_bf16_grads=False):
    """MFMA backward (bf16 operands): (dq [n,d] f32, dkp [k,d] f32, dv [n,d] f32) from q, v (f32 or bf16, row-strided views
    allowed), kp [k,d] f32, dout [k,d] f32 and the forward's lse [h,n].  mask: dropout keep-mask / (1 - p) or None.
    fused_bf16_grads: dq and dv are the two column halves of ONE bf16 buffer [n, 2 d] (returned as views of it)."""
    if q.dtype not in (torch.float32, torch.bfloat16) or v.dtype != q.dtype:
        raise TypeError("sparse_attn_bwd_mfma: q and v must both be float32 or both bfloat16")
    q = _rows16(q, "q")
    v = _rows16(v, "v")
    kp = _req(kp, torch.float32, "kp", 2)
    dout = _req(dout, torch.float32, "dout", 2)
    lse = _req(lse, torch.float32, "lse", 2)
    if mask is not None:
        mask = _req(mask, torch.float32, "mask", 3)
    n, d = q.shape
    k = kp.shape[0]
    dk = d // h
    scale = 1.0 / math.sqrt(dk) if scale is None else scale
    lib = _ffi.load()
    if fused_bf16_grads:
        dqv = torch.empty(n, 2 * d, dtype=torch.bfloat16, device=q.device)
        dq, dv, ldd, gdt = dqv[:, :d], dqv[:, d:], 2 * d, DT_BF16
    else:
        dq = torch.empty(n, d, dtype=torch.float32, device=q.device)
        dv = torch.empty(n, d, dtype=torch.float32, device=q.device)
        ldd, gdt = d, DT_F32
    # bf16 operands: dS is written as bf16 and dKp = dS^T Q is one batched bf16 library GEMM (fp32 accumulate) -- half the
    # dS traffic and ~40 us instead of ~310 us for the fp32 slice kernel; f32 operands keep the fp32 route
    bf16 = q.dtype == torch.bfloat16
    ds = torch.empty(h, n, k, dtype=torch.bfloat16 if bf16 else torch.float32, device=q.device)
    dt = DT_F32 if q.dtype == torch.float32 else DT_BF16
    pdrop, seed, offset = dropout if dropout is not None else (0.0, 0, 0)
    check(lib.snf_sparse_attn_bwd_mfma_ex(_p(q), q.stride(0), _p(v), v.stride(0), dt, _p(kp), _p(dout), _p(lse), _p(mask),
                                          float(pdrop), int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), n, k, h, dk,
                                          float(scale), _p(dq), _p(dv), ldd, gdt, _p(ds), DT_BF16 if bf16 else DT_F32, _stream()),
          "snf_sparse_attn_bwd_mfma")
    if bf16:
        # dKp = dS^T Q per head: a [k, dk] output over a contraction of n rows -- split over 16 row chunks (one batched GEMM,
        # partials summed in fp32) so that the library has 16 h tiles to spread instead of h (121 -> 85 us at config B)
        c = 16
        if n >= 8192 and n % c == 0:
            dsv = ds.view(h, c, n // c, k).transpose(2, 3)                          # [h, c, k, n/c]
            qh = q.reshape(n, h, dk).view(c, n // c, h, dk).permute(2, 0, 1, 3)     # [h, c, n/c, dk]
            dkp = torch.matmul(dsv, qh).sum(1, dtype=torch.float32).transpose(0, 1).reshape(k, d)
        else:
            qh = q.view(n, h, dk).transpose(0, 1)                   # [h, n, dk] (strided view)
            dkp = torch.bmm(ds.transpose(1, 2), qh).float().transpose(0, 1).reshape(k, d)
        return dq, dkp, dv
    dkp = torch.empty(k, d, dtype=torch.float32, device=q.device)
    wsb = lib.snf_sparse_attn_bwd_workspace_bytes(n, k, h, dk)
    ws = _ws(wsb, q.device)
    qf = q if q.is_contiguous() else q.contiguous()
    check(lib.snf_sparse_attn_dkp_f32(_p(ds), _p(qf), n, k, h, dk, _p(dkp), _p(ws), wsb, _stream()), "snf_sparse_attn_dkp_f32")
    return dq, dkp, dv


def mfma_attn_supported(k, dk, n=None, ld=None):
    """Shapes the MFMA attention kernel takes.  One launch holds 256 (dk = 64) / 224 (dk = 128) keys next to the P and V
    images in the 160 KiB LDS of a CU; up to 8 key chunks are run back to back with exact cross-chunk softmax statistics.
    n / ld (rows and row pitch of q, v) are optional: the kernel addresses rows with 32-bit element offsets."""
    if not ((dk == 64 and 1 <= k <= 8 * 256) or (dk == 128 and 1 <= k <= 8 * 224)):
        return False
    if n is not None and (n > 0xffff00 or (ld is not None and (ld >= (1 << 24) or n * ld >= 0x7fffffff))):
        return False
    return True


def mfma_attn_dropout_supported(k, dk):
    """In-kernel Philox dropout (forward and the MFMA backward) covers ONE key chunk; more keys use the exact kernels with
    the mask tensor of dropout_mask()."""
    return (dk == 64 and 1 <= k <= 256) or (dk == 128 and 1 <= k <= 224)


def _rows16(t, name):
    """[n, d] view whose rows are unit-stride and 16-byte aligned (e.g. one half of a fused projection output): the
    kernels take a row pitch, so such views are used in place.  Anything else is made contiguous."""
    t = _req_nc(t, name)
    elt = t.element_size()
    if t.stride(1) != 1 or (t.stride(0) * elt) % 16 or (t.data_ptr() % 16) or t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def _req_nc(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _ffi.SnuffyHipError(
            "%s must be a GPU tensor: snuffy_amd runs on MI355X only (no CPU fallback)" % name)
    _on_current_device(t, name)
    if t.dim() != 2:
        raise ValueError("%s must be 2-D, got shape %s" % (name, tuple(t.shape)))
    return t


def sparse_attn_fwd(q, kp, v, h, scale=None, need_attn=False, need_lse=False):
    """Exact-fp32 sparse attention (snuffy.py:160-168). q, v [n, d]; kp [k, d] -> (out [k, d], attn [h,n,k], lse)."""
    q = _req(q, torch.float32, "q", 2)
    kp = _req(kp, torch.float32, "kp", 2)
    v = _req(v, torch.float32, "v", 2)
    n, d = q.shape
    k = kp.shape[0]
    if d % h:
        raise ValueError("d_model %d not divisible by h %d" % (d, h))
    dk = d // h
    scale = 1.0 / math.sqrt(dk) if scale is None else scale
    lib = _ffi.load()
    out = torch.empty(k, d, dtype=torch.float32, device=q.device)
    attn = torch.empty(h, n, k, dtype=torch.float32, device=q.device) if need_attn else None
    lse = torch.empty(h, n, dtype=torch.float32, device=q.device) if need_lse else None
    wsb = lib.snf_sparse_attn_fwd_workspace_bytes(n, k, h, dk, 0)
    if need_attn:  # P lives in `attn`; only the partial-sum part of the workspace is needed
        wsb -= ((h * n * k * 4 + 255) // 256) * 256
    ws = _ws(wsb, q.device)
    check(lib.snf_sparse_attn_fwd_f32(_p(q), _p(kp), _p(v), n, k, h, dk, float(scale), _p(out), _p(attn), _p(lse),
                                      _p(ws), wsb, _stream()), "snf_sparse_attn_fwd_f32")
    return out, attn, lse


def x3u_attn_supported(k, dk):
    """Shapes of the unfused fp32-class attention (snf_sparse_attn_fwd_x3u_f32): the head widths outside the pipelined kernels."""
    return dk % 16 == 0 and 16 <= dk <= 256 and 1 <= k <= 1024


def sparse_attn_fwd_x3u(q, kp, v, h, scale=None, need_attn=False, need_lse=False):
    """fp32-class sparse attention for head widths like 192 (snf_sparse_attn_fwd_x3u_f32: scores + softmax in split-bf16 x 3 on the
    bf16 matrix cores with the operands split on the fly, P^T V exact on the f32 matrix cores).  q, v [n, d] f32 (row-strided views
    allowed: the halves of a fused [Q | V] projection), kp [k, d] f32 -> (out, attn or None, lse or None)."""
    if q.dtype != torch.float32 or v.dtype != torch.float32:
        raise TypeError("sparse_attn_fwd_x3u: q and v must be float32")
    q = _rows16(q, "q")
    v = _rows16(v, "v")
    kp = _req(kp, torch.float32, "kp", 2)
    n, d = q.shape
    k = kp.shape[0]
    if d % h:
        raise ValueError("d_model %d not divisible by h %d" % (d, h))
    dk = d // h
    scale = 1.0 / math.sqrt(dk) if scale is None else scale
    lib = _ffi.load()
    out = torch.empty(k, d, dtype=torch.float32, device=q.device)
    attn = torch.empty(h, n, k, dtype=torch.float32, device=q.device) if need_attn else None
    lse = torch.empty(h, n, dtype=torch.float32, device=q.device) if need_lse else None
    wsb = lib.snf_sparse_attn_fwd_workspace_bytes(n, k, h, dk, 0)
    if need_attn:
        wsb -= ((h * n * k * 4 + 255) // 256) * 256
    ws = _ws(wsb, q.device)
    check(lib.snf_sparse_attn_fwd_x3u_f32(_p(q), q.stride(0), _p(kp), _p(v), v.stride(0), n, k, h, dk, float(scale), _p(out), _p(attn),
                                          _p(lse), _p(ws), wsb, _stream()), "snf_sparse_attn_fwd_x3u_f32")
    return out, attn, lse


def x3_attn_supported(k, dk):
    """Shapes of the fp32-class (split-bf16 x 3) MFMA attention kernel: 224 (dk = 128) / 256 (dk = 64) keys per launch, up to 8
    key chunks with exact cross-chunk softmax statistics."""
    return (dk == 128 and 1 <= k <= 8 * 224) or (dk == 64 and 1 <= k <= 8 * 256)


def x3_attn_dropout_supported(k, dk):
    """The fp32-class kernel applies the training dropout mask itself (one key chunk)."""
    return (dk == 128 and 1 <= k <= 224) or (dk == 64 and 1 <= k <= 256)


def sparse_attn_fwd_x3(q, v, kp, h, scale=None, need_attn=False, need_lse=False, dropout=None):
    """fp32-class sparse attention on the matrix cores (snf_sparse_attn_fwd_x3): q, v [n, d] f32 (row-strided views allowed),
    kp [k, d] f32 -> (out [k, d], attn [h, n, k] or None, lse [h, n] or None).  dropout = (p, seed, offset): the training forward
    (snf_sparse_attn_fwd_x3_dropout, x3_attn_dropout_supported shapes): out = (P o M)^T V with M the mask ops.dropout_mask writes for the
    same (p, seed, offset); attn stays the UNDROPPED P."""
    if q.dtype != torch.float32 or v.dtype != torch.float32:
        raise TypeError("sparse_attn_fwd_x3: q and v must be float32")
    q = _rows16(q, "q")
    v = _rows16(v, "v")
    kp = _req(kp, torch.float32, "kp", 2)
    n, d = q.shape
    k = kp.shape[0]
    if d % h or kp.shape[1] != d or v.shape != q.shape:
        raise ValueError("sparse_attn_fwd_x3: inconsistent shapes q %s v %s kp %s h %d" % (tuple(q.shape), tuple(v.shape),
                                                                                            tuple(kp.shape), h))
    dk = d // h
    scale = 1.0 / math.sqrt(dk) if scale is None else scale
    lib = _ffi.load()
    out = torch.empty(k, d, dtype=torch.float32, device=q.device)
    attn = torch.empty(h, n, k, dtype=torch.float32, device=q.device) if need_attn else None
    lse = torch.empty(h, n, dtype=torch.float32, device=q.device) if need_lse else None
    wsb = lib.snf_sparse_attn_fwd_x3_workspace_bytes(n, k, h, dk)
    ws = _ws(wsb, q.device)
    if dropout is not None and dropout[0] > 0.0:
        check(lib.snf_sparse_attn_fwd_x3_dropout(_p(q), q.stride(0), _p(v), v.stride(0), _p(kp), n, k, h, dk, float(scale), float(dropout[0]),
                                                 int(dropout[1]) & (2 ** 64 - 1), int(dropout[2]) & (2 ** 64 - 1), _p(out), _p(attn), _p(lse), _p(ws), wsb, _stream()),
              "snf_sparse_attn_fwd_x3_dropout")
        return out, attn, lse
    check(lib.snf_sparse_attn_fwd_x3(_p(q), q.stride(0), _p(v), v.stride(0), _p(kp), n, k, h, dk, float(scale), _p(out), _p(attn),
                                     _p(lse), _p(ws), wsb, _stream()), "snf_sparse_attn_fwd_x3")
    return out, attn, lse


_X3_HL_OK = {}


def x3_hl_attn_supported(k, dk):
    """Shapes of the pipelined fp32-class attention kernel on pre-split operands (snf_sparse_attn_fwd_x3_hl): dk = 64 / 128,
    97 .. 2048 keys -- asked of the library itself, whose answer also depends on the device (a key count whose last chunk is
    shorter than 97 keys needs the merged launch: >= 8 x chunks CUs)."""
    if dk not in (64, 128) or not 97 <= k <= 2048:
        return False
    key = (int(k), int(dk), torch.cuda.current_device() if torch.cuda.is_available() else -1)
    if key not in _X3_HL_OK:
        _X3_HL_OK[key] = _ffi.load().snf_sparse_attn_fwd_x3_hl_workspace_bytes(4096, int(k), 1, int(dk)) > 0
    return _X3_HL_OK[key]


class KpFrag:
    """The key projection of one bag as the fragment image of the pipelined attention kernel (linear_rows_x3_kpfrag): the keys'
    count and width travel with the opaque buffer; the softmax scale is already folded in."""

    def __init__(self, buf, k, d, h):
        self.buf, self.k, self.d, self.h = buf, k, d, h


def x3_hl_kpfrag_supported(k, h, dk):
    """The key projection can write the attention kernel's fragment image directly (no fp32 Kp, no prep launch)."""
    return 1 <= k <= 8192 and _ffi.load().snf_sparse_attn_x3_hl_kpfrag_bytes(int(k), int(h), int(dk)) > 0


def linear_rows_x3_kpfrag(x, w, bias, h, scale=None):
    """Kp = x [k, kdim] f32 @ w [d, kdim]^T f32 + bias, fp32-class (as linear_rows_x3), written as the scaled, split fragment image
    sparse_attn_fwd_x3_hl takes as `kp` (snf_linear_rows_x3_kpfrag_f32) -> KpFrag."""
    if x.dtype != torch.float32 or w.dtype != torch.float32:
        raise TypeError("linear_rows_x3_kpfrag: x and w must be float32")
    x = _rows16(x, "x")
    w = _rows16(w, "w")
    k, kdim = x.shape
    d = w.shape[0]
    if w.shape[1] != kdim or d % h:
        raise ValueError("linear_rows_x3_kpfrag: x %s w %s h %d" % (tuple(x.shape), tuple(w.shape), h))
    dk = d // h
    lib = _ffi.load()
    nbytes = lib.snf_sparse_attn_x3_hl_kpfrag_bytes(k, h, dk)
    if not nbytes or kdim % 16:
        raise ValueError("linear_rows_x3_kpfrag: k=%d h=%d dk=%d kdim=%d outside the fused form" % (k, h, dk, kdim))
    if bias is not None:
        bias = _req(bias, torch.float32, "bias", 1)
    scale = 1.0 / math.sqrt(dk) if scale is None else scale
    buf = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    check(lib.snf_linear_rows_x3_kpfrag_f32(_p(x), x.stride(0), _p(w), w.stride(0), _p(bias), k, h, dk, kdim, float(scale), _p(buf),
                                            nbytes, _stream()), "snf_linear_rows_x3_kpfrag_f32")
    return KpFrag(buf, k, d, h)


def gather_linear_rows_x3_kpfrag(x, idx, w, bias, h, scale=None, want_map=True):
    """gather_slot_map + linear_rows_x3_kpfrag in ONE launch (snf_gather_linear_rows_x3_kpfrag_f32): x [n, kdim] f32 (the bag), idx [k]
    int64 (the selected rows) -> (KpFrag of x[idx] @ w^T + bias, xs = x[idx] [k, kdim] f32, row -> slot map [n] int32 or None).
    Same bytes as the two launches it replaces."""
    if x.dtype != torch.float32 or w.dtype != torch.float32:
        raise TypeError("gather_linear_rows_x3_kpfrag: x and w must be float32")
    x = _rows16(x, "x")
    w = _rows16(w, "w")
    idx = _req(idx, torch.int64, "idx", 1)
    n, kdim = x.shape
    k = idx.shape[0]
    d = w.shape[0]
    if w.shape[1] != kdim or d % h:
        raise ValueError("gather_linear_rows_x3_kpfrag: x %s w %s h %d" % (tuple(x.shape), tuple(w.shape), h))
    dk = d // h
    lib = _ffi.load()
    nbytes = lib.snf_sparse_attn_x3_hl_kpfrag_bytes(k, h, dk)
    if not nbytes or kdim % 16:
        raise ValueError("gather_linear_rows_x3_kpfrag: k=%d h=%d dk=%d kdim=%d outside the fused form" % (k, h, dk, kdim))
    if bias is not None:
        bias = _req(bias, torch.float32, "bias", 1)
    scale = 1.0 / math.sqrt(dk) if scale is None else scale
    buf = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    xs = torch.empty(k, kdim, dtype=torch.float32, device=x.device)
    slot = torch.empty(n, dtype=torch.int32, device=x.device) if want_map else None
    check(lib.snf_gather_linear_rows_x3_kpfrag_f32(_p(x), x.stride(0), n, _p(idx), _p(w), w.stride(0), _p(bias), k, h, dk, kdim, float(scale),
                                                   _p(buf), nbytes, _p(xs), _p(slot), _stream()), "snf_gather_linear_rows_x3_kpfrag_f32")
    return KpFrag(buf, k, d, h), xs, slot


def sparse_attn_fwd_x3_hl(q_hl, v_hl, kp, h, scale=None, need_attn=False, need_lse=False):
    """fp32-class sparse attention on PRE-SPLIT operands (snf_sparse_attn_fwd_x3_hl): q_hl, v_hl [n, 2 d] bf16 interleaved split
    images (split_hl_rows / gemm_hl(hl_out=True); row-strided views allowed, e.g. the two halves of the [Q | V] projection's
    image), kp [k, d] f32 -- or the KpFrag of linear_rows_x3_kpfrag (scale then already applied) -> (out [k, d] f32,
    attn [h, n, k] or None, lse [h, n] or None)."""
    if q_hl.dtype != torch.bfloat16 or v_hl.dtype != torch.bfloat16:
        raise TypeError("sparse_attn_fwd_x3_hl: q_hl and v_hl must be bfloat16 hl images")
    q_hl = _rows16(q_hl, "q_hl")
    v_hl = _rows16(v_hl, "v_hl")
    frag = kp if isinstance(kp, KpFrag) else None
    if frag is None:
        kp = _req(kp, torch.float32, "kp", 2)
    elif frag.h != h or scale is not None:
        raise ValueError("sparse_attn_fwd_x3_hl: the fragment image was made for h = %d with its scale folded in" % frag.h)
    n, d2 = q_hl.shape
    k, d = (frag.k, frag.d) if frag is not None else kp.shape
    if d2 != 2 * d or d % h or v_hl.shape != q_hl.shape:
        raise ValueError("sparse_attn_fwd_x3_hl: inconsistent shapes q_hl %s v_hl %s kp %s h %d" % (tuple(q_hl.shape), tuple(v_hl.shape),
                                                                                                   (k, d), h))
    dk = d // h
    scale = 1.0 / math.sqrt(dk) if scale is None else scale
    lib = _ffi.load()
    out = torch.empty(k, d, dtype=torch.float32, device=q_hl.device)
    attn = torch.empty(h, n, k, dtype=torch.float32, device=q_hl.device) if need_attn else None
    lse = torch.empty(h, n, dtype=torch.float32, device=q_hl.device) if need_lse else None
    wsb = lib.snf_sparse_attn_fwd_x3_hl_workspace_bytes(n, k, h, dk)
    ws = _ws(wsb, q_hl.device)
    if frag is not None:
        check(lib.snf_sparse_attn_fwd_x3_hl_kpfrag(_p(q_hl), q_hl.stride(0), _p(v_hl), v_hl.stride(0), _p(frag.buf), n, k, h, dk, _p(out),
                                                   _p(attn), _p(lse), _p(ws), wsb, _stream()), "snf_sparse_attn_fwd_x3_hl_kpfrag")
    else:
        check(lib.snf_sparse_attn_fwd_x3_hl(_p(q_hl), q_hl.stride(0), _p(v_hl), v_hl.stride(0), _p(kp), n, k, h, dk, float(scale),
                                            _p(out), _p(attn), _p(lse), _p(ws), wsb, _stream()), "snf_sparse_attn_fwd_x3_hl")
    return out, attn, lse


# ----------------------------------------------------------------------------------------------------------------------
# varlen path: many bags per launch (include/snuffy_hip.h "Varlen path")
# ----------------------------------------------------------------------------------------------------------------------
class PackedBags:
    """Row offsets of B bags packed into one [T, .] tensor, on the host (launch geometry) and on the device (segmented top-k),
    plus the launch plans derived from them (built on the host once per kernel shape, uploaded once)."""

    def __init__(self, sizes, device):
        import numpy as np
        sizes = [int(n) for n in sizes]
        if not sizes or min(sizes) < 1:
            raise ValueError("PackedBags: need at least one bag and no empty bag, got sizes %s" % (sizes,))
        self.sizes = sizes
        self.bags = len(sizes)
        self.max_n = max(sizes)
        self.host = np.zeros(self.bags + 1, dtype=np.int64)
        np.cumsum(sizes, out=self.host[1:])
        self.total = int(self.host[-1])
        self.device = torch.device(device)
        self.dev = torch.from_numpy(self.host).to(self.device)
        self._plans = {}

    def _host_ptr(self):
        return ctypes.c_void_p(self.host.ctypes.data)

    def ragged(self, kbs):
        """RaggedKeys for per-bag key counts `kbs` (cached)."""
        key = ("ragged",) + tuple(int(k) for k in kbs)
        hit = self._plans.get(key)
        if hit is None:
            hit = self._plans[key] = RaggedKeys(self, kbs)
        return hit

    def plan(self, kind, *shape):
        """(device table int32, workspace bytes) of a segmented launch: kind in {"mfma", "x3", "head"}."""
        import numpy as np
        key = (kind,) + tuple(shape)
        hit = self._plans.get(key)
        if hit is not None:
            return hit
        lib = _ffi.load()
        need, wsb = ctypes.c_size_t(0), ctypes.c_size_t(0)
        fn = {"mfma": lib.snf_sparse_attn_varlen_plan, "x3": lib.snf_sparse_attn_x3_varlen_plan,
              "head": lib.snf_ln_mean_head_varlen_plan}[kind]
        check(fn(self._host_ptr(), self.bags, *shape, None, 0, ctypes.byref(need), ctypes.byref(wsb)), "varlen plan (%s)" % kind)
        table = np.zeros(need.value, dtype=np.int32)
        check(fn(self._host_ptr(), self.bags, *shape, ctypes.c_void_p(table.ctypes.data), table.size, ctypes.byref(need),
                 ctypes.byref(wsb)), "varlen plan (%s)" % kind)
        hit = self._plans[key] = (torch.from_numpy(table).to(self.device), int(wsb.value))
        return hit


class RaggedKeys:
    """Per-bag key counts of a packed batch whose bags do not all select the same number of rows (bags shorter than Lambda select
    all of theirs): key-row offsets, the kernel's per-bag descriptors and the index helpers that turn the padded [B, k] output of
    topk_segmented() into the flat list of selected rows in packed coordinates."""

    def __init__(self, packed, kbs):
        import numpy as np
        kbs = [int(k) for k in kbs]
        if len(kbs) != packed.bags or any(k < 1 or k > n for k, n in zip(kbs, packed.sizes)):
            raise ValueError("RaggedKeys: need 1 <= k_b <= n_b for every bag")
        self.kbs = kbs
        self.kmax = max(kbs)
        self.koff = np.zeros(packed.bags + 1, dtype=np.int64)
        np.cumsum(kbs, out=self.koff[1:])
        self.total = int(self.koff[-1])
        desc = np.stack([packed.host[:-1], np.asarray(packed.sizes), self.koff[:-1], np.asarray(kbs)], axis=1).astype(np.int32)
        self.desc = torch.from_numpy(np.ascontiguousarray(desc)).to(packed.device)
        self._row0 = packed.host[:-1].copy()
        self._np = np

    def flat_index(self, pitch):
        """(positions of the valid entries in a padded [B, pitch] array, first packed row of each entry's bag), device int64."""
        np = self._np
        hit = getattr(self, "_flat", None)
        if hit is None or hit[0] != pitch:
            pos = np.concatenate([b * pitch + np.arange(k, dtype=np.int64) for b, k in enumerate(self.kbs)])
            base = np.repeat(self._row0, self.kbs)
            hit = self._flat = (pitch, torch.from_numpy(pos).to(self.desc.device), torch.from_numpy(base).to(self.desc.device))
        return hit[1], hit[2]


def ragged_attn_supported(kmax, dk):
    return 1 <= kmax <= 256 and (16 + dk) * kmax * 4 <= 160 * 1024


def sparse_attn_fwd_ragged(q, v, kp, packed, rag, h, scale=None, need_attn=False, need_lse=False):
    """Exact-fp32 sparse attention of B packed SMALL bags with per-bag key counts (snf_sparse_attn_fwd_ragged_f32).  q, v [T, d]
    f32 (row-strided views allowed), kp [sum K_b, d] f32 -> (out [sum K_b, d], attn [h, T, kmax] or None, lse [h, T] or None);
    bag b's probabilities are attn[:, row0 : row0 + n_b, :K_b]."""
    if q.dtype != torch.float32 or v.dtype != torch.float32:
        raise TypeError("sparse_attn_fwd_ragged: q and v must be float32")
    q = _req_nc(q, "q")
    v = _req_nc(v, "v")
    if q.stride(1) != 1:
        q = q.contiguous()
    if v.stride(1) != 1:
        v = v.contiguous()
    kp = _req(kp, torch.float32, "kp", 2)
    t, d = q.shape
    if t != packed.total or v.shape != q.shape or kp.shape != (rag.total, d) or d % h:
        raise ValueError("sparse_attn_fwd_ragged: q %s v %s kp %s do not match %d packed rows / %d keys"
                         % (tuple(q.shape), tuple(v.shape), tuple(kp.shape), packed.total, rag.total))
    dk = d // h
    scale = 1.0 / math.sqrt(dk) if scale is None else scale
    out = torch.empty(rag.total, d, dtype=torch.float32, device=q.device)
    attn = torch.empty(h, t, rag.kmax, dtype=torch.float32, device=q.device) if need_attn else None
    lse = torch.empty(h, t, dtype=torch.float32, device=q.device) if need_lse else None
    check(_ffi.load().snf_sparse_attn_fwd_ragged_f32(_p(q), q.stride(0), _p(v), v.stride(0), _p(kp), _p(rag.desc), packed.bags, t,
                                                     rag.kmax, h, dk, float(scale), _p(out), _p(attn), _p(lse), _stream()),
          "snf_sparse_attn_fwd_ragged_f32")
    return out, attn, lse


def varlen_attn_supported(precision_kind, k, dk):
    """Single key chunk only: k <= 224 (dk = 128) / 256 (dk = 64); the fp32-class kernel is built for 2, 4, 7 (8) key blocks."""
    return (dk == 128 and 1 <= k <= 224) or (dk == 64 and 1 <= k <= 256)


def topk_segmented(scores, packed, k):
    """Top-k of every bag of a packed score vector in one launch: [B, k] int64 indices INSIDE each bag (descending score, ties
    by ascending index -- the same one-workgroup kernel body as topk(), one workgroup per bag).  A bag with fewer than k rows fills
    only its first n_b entries (all of its rows, ordered); the rest of its output row is uninitialised."""
    scores = _req(scores, torch.float32, "scores", 1)
    k = int(k)
    if scores.shape[0] != packed.total or not (1 <= k <= TOPK_MAX_K):
        raise ValueError("topk_segmented: %d scores for %d packed rows, k=%d" % (scores.shape[0], packed.total, k))
    idx = torch.empty(packed.bags, k, dtype=torch.int64, device=scores.device)
    check(_ffi.load().snf_topk_segmented_f32(_p(scores), _p(packed.dev), packed.bags, packed.max_n, k, _p(idx), _stream()),
          "snf_topk_segmented_f32")
    return idx


def sparse_attn_fwd_mfma_varlen(q, v, kp, packed, k, h, scale=None, need_attn=False, need_lse=False):
    """bf16-MFMA sparse attention of B packed bags in one launch.  q, v [T, d] bf16 (row-strided views allowed), kp [B * k, d]
    (bag b's keys in rows b k ..) -> (out [B * k, d] f32, attn [h, T, k] or None, lse [h, T] or None).  A bag's result
    does not depend on what it is packed with (bit for bit); against sparse_attn_fwd_mfma() bag by bag P / lse are identical and
    O differs by the fp32 order of the partial sums only (small bags get more rows per workgroup here)."""
    if q.dtype != torch.bfloat16 or v.dtype != torch.bfloat16:
        raise TypeError("sparse_attn_fwd_mfma_varlen: q and v must be bfloat16")
    q = _rows16(q, "q")
    v = _rows16(v, "v")
    if kp.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("sparse_attn_fwd_mfma_varlen: kp must be float32 or bfloat16")
    kp = _req(kp, kp.dtype, "kp", 2)
    t, d = q.shape
    if t != packed.total or v.shape != q.shape or kp.shape != (packed.bags * k, d) or d % h:
        raise ValueError("sparse_attn_fwd_mfma_varlen: q %s v %s kp %s do not match %d packed rows, %d bags x %d keys"
                         % (tuple(q.shape), tuple(v.shape), tuple(kp.shape), packed.total, packed.bags, k))
    dk = d // h
    scale = 1.0 / math.sqrt(dk) if scale is None else scale
    table, wsb = packed.plan("mfma", k, h, dk)
    ws = _ws(wsb, q.device)
    out = torch.empty(packed.bags * k, d, dtype=torch.float32, device=q.device)
    attn = torch.empty(h, t, k, dtype=torch.float32, device=q.device) if need_attn else None
    lse = torch.empty(h, t, dtype=torch.float32, device=q.device) if need_lse else None
    kdt = DT_F32 if kp.dtype == torch.float32 else DT_BF16
    check(_ffi.load().snf_sparse_attn_fwd_mfma_varlen(_p(q), q.stride(0), _p(v), v.stride(0), _p(kp), kdt, packed._host_ptr(),
                                                      packed.bags, k, h, dk, float(scale), _p(out), _p(attn), _p(lse), _p(table),
                                                      _p(ws), wsb, _stream()), "snf_sparse_attn_fwd_mfma_varlen")
    return out, attn, lse


def sparse_attn_fwd_x3_varlen(q, v, kp, packed, k, h, scale=None, need_attn=False, need_lse=False):
    """fp32-class sparse attention of B packed bags in one launch (f32 q, v [T, d], kp [B * k, d]); composition-independent bit
    for bit, O within the fp32 summation order of sparse_attn_fwd_x3() bag by bag."""
    if q.dtype != torch.float32 or v.dtype != torch.float32:
        raise TypeError("sparse_attn_fwd_x3_varlen: q and v must be float32")
    q = _rows16(q, "q")
    v = _rows16(v, "v")
    kp = _req(kp, torch.float32, "kp", 2)
    t, d = q.shape
    if t != packed.total or v.shape != q.shape or kp.shape != (packed.bags * k, d) or d % h:
        raise ValueError("sparse_attn_fwd_x3_varlen: q %s v %s kp %s do not match %d packed rows, %d bags x %d keys"
                         % (tuple(q.shape), tuple(v.shape), tuple(kp.shape), packed.total, packed.bags, k))
    dk = d // h
    scale = 1.0 / math.sqrt(dk) if scale is None else scale
    table, wsb = packed.plan("x3", k, h, dk)
    ws = _ws(wsb, q.device)
    out = torch.empty(packed.bags * k, d, dtype=torch.float32, device=q.device)
    attn = torch.empty(h, t, k, dtype=torch.float32, device=q.device) if need_attn else None
    lse = torch.empty(h, t, dtype=torch.float32, device=q.device) if need_lse else None
    check(_ffi.load().snf_sparse_attn_fwd_x3_varlen(_p(q), q.stride(0), _p(v), v.stride(0), _p(kp), packed._host_ptr(), packed.bags,
                                                    k, h, dk, float(scale), _p(out), _p(attn), _p(lse), _p(table), _p(ws), wsb,
                                                    _stream()), "snf_sparse_attn_fwd_x3_varlen")
    return out, attn, lse


def ln_mean_head_varlen(z, packed, gamma, beta, eps, w_head, b_head, add_bf16=None, add_bias=None, slot=None, delta_rows=None):
    """ln_mean_head() of every packed bag in three launches: (logits [B, C], pooled [B, D]); bit-identical bag by bag."""
    z = _req(z, torch.float32, "z", 2)
    t, d = z.shape
    if t != packed.total:
        raise ValueError("ln_mean_head_varlen: %d rows for %d packed rows" % (t, packed.total))
    if add_bf16 is not None:
        add_bf16 = _req(add_bf16, torch.bfloat16, "add_bf16", 2)
    if add_bias is not None:
        add_bias = _req(add_bias, torch.float32, "add_bias", 1)
    if slot is not None:
        slot = _req(slot, torch.int32, "slot", 1)
        delta_rows = _req(delta_rows, torch.float32, "delta_rows", 2)
    gamma = _req(gamma, torch.float32, "gamma", 1)
    beta = _req(beta, torch.float32, "beta", 1)
    w_head = _req(w_head, torch.float32, "w_head", 2)
    if b_head is not None:
        b_head = _req(b_head, torch.float32, "b_head", 1)
    c = w_head.shape[0]
    table, wsb = packed.plan("head", d)
    ws = _ws(wsb, z.device)
    logits = torch.empty(packed.bags, c, dtype=torch.float32, device=z.device)
    pooled = torch.empty(packed.bags, d, dtype=torch.float32, device=z.device)
    check(_ffi.load().snf_ln_mean_head_varlen_f32(_p(z), packed._host_ptr(), packed.bags, d, _p(add_bf16), _p(add_bias), _p(slot),
                                                  _p(delta_rows), None, _p(gamma), _p(beta), float(eps), _p(w_head), _p(b_head), c,
                                                  _p(logits), _p(pooled), _p(table), _p(ws), wsb, _stream()),
          "snf_ln_mean_head_varlen_f32")
    return logits, pooled


def dropout_mask(h, n, k, p, seed, offset, device):
    """The attention kernels' dropout mask as a tensor [h, n, k] f32 (0 or 1 / (1 - p)) -- see snf_dropout_mask_f32."""
    m = torch.empty(h, n, k, dtype=torch.float32, device=device)
    with torch.cuda.device(m.device):
        check(_ffi.load().snf_dropout_mask_f32(float(p), int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), h, n, k, _p(m),
                                               _stream()), "snf_dropout_mask_f32")
    return m


def sparse_attn_fwd_mfma(q, v, kp, n, h, scale=None, need_attn=False, need_lse=False, dropout=None):
    """bf16-MFMA sparse attention.  q, v [n, d] row-major (both f32 or both bf16; row-strided views such as the two halves
    of a fused [n, 2d] projection are taken in place); kp [k, d] f32 or bf16 (f32 is rounded to bf16 by the library, one
    extra small launch)."""
    if q.dtype not in (torch.float32, torch.bfloat16) or v.dtype != q.dtype:
        raise TypeError("sparse_attn_fwd_mfma: q and v must both be float32 or both bfloat16")
    q = _rows16(q, "q")
    v = _rows16(v, "v")
    if kp.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("sparse_attn_fwd_mfma: kp must be float32 or bfloat16")
    kp = _req(kp, kp.dtype, "kp", 2)
    d = q.shape[1]
    if kp.shape[1] != d:
        raise ValueError("sparse_attn_fwd_mfma: kp %s does not match the width %d of q" % (tuple(kp.shape), d))
    if v.shape[1] != d or q.shape[0] < n or v.shape[0] < n:
        raise ValueError("sparse_attn_fwd_mfma: q %s / v %s do not hold %d rows of width %d"
                         % (tuple(q.shape), tuple(v.shape), n, d))
    k = kp.shape[0]
    dk = d // h
    scale = 1.0 / math.sqrt(dk) if scale is None else scale
    lib = _ffi.load()
    out = torch.empty(k, d, dtype=torch.float32, device=q.device)
    attn = torch.empty(h, n, k, dtype=torch.float32, device=q.device) if need_attn else None
    lse = torch.empty(h, n, dtype=torch.float32, device=q.device) if need_lse else None
    wsb = lib.snf_sparse_attn_fwd_workspace_bytes(n, k, h, dk, 1)
    ws = _ws(wsb, q.device)
    dt = DT_F32 if q.dtype == torch.float32 else DT_BF16
    kdt = DT_F32 if kp.dtype == torch.float32 else DT_BF16
    pdrop, seed, offset = dropout if dropout is not None else (0.0, 0, 0)
    if pdrop > 0 and not mfma_attn_dropout_supported(k, dk):
        raise ValueError("sparse_attn_fwd_mfma: in-kernel dropout needs a single key chunk (k <= %d at dk = %d), got k = %d"
                         % (224 if dk == 128 else 256, dk, k))
    check(lib.snf_sparse_attn_fwd_mfma_dropout(_p(q), q.stride(0), _p(v), v.stride(0), dt, _p(kp), kdt, n, k, h, dk, float(scale),
                                               _p(out), _p(attn), _p(lse), float(pdrop), int(seed) & (2 ** 64 - 1),
                                               int(offset) & (2 ** 64 - 1), _p(ws), wsb, _stream()), "snf_sparse_attn_fwd_mfma")
    return out, attn, lse


def gemm_supported(m, n, k, lda=None, ldw=None):
    """Shapes snf_gemm_bf16 takes (see include/snuffy_hip.h)."""
    lda = k if lda is None else lda
    ldw = k if ldw is None else ldw
    return (m >= 1 and k >= 64 and k % 32 == 0 and n >= 8 and n % 8 == 0 and lda % 8 == 0 and ldw % 8 == 0
            and m * lda < 0x7fffffff and n * ldw < 0x7fffffff)


def gemm_prefers_native(m, n, k):
    """Shape policy measured on MI355X (tools/gemm_bench.py, profiles/r02_gemm_bench.txt): the hand-written kernel wins where
    the K loop is short (k <= 512: ViT qkv / proj / fc1, bags of <= 8k patches: 1.2 - 1.5x the library); at k >= 768 with
    large m * n both are bound by the L2 -> LDS fill rate and the library's main loop is ~20 % ahead."""
    return gemm_supported(m, n, k) and (k <= 512 or (k <= 768 and m * n <= (16 << 20)))


def linear_bf16(a, w, bias_f32=None, bias_bf16=None, act="none", prefer_native=None):
    """act(a @ w.T + bias) -> bf16, on the hand-written MFMA kernel or the library GEMM by the shape policy above.  The
    activations the library cannot fuse exactly (erf GELU, leaky ReLU, SELU) always take the native kernel when the shape is
    in its domain."""
    m, k = a.shape
    n = w.shape[0]
    native = gemm_prefers_native(m, n, k) if prefer_native is None else (prefer_native and gemm_supported(m, n, k))
    if act in ("gelu", "leakyrelu", "selu") and gemm_supported(m, n, k):
        native = True
    if native:
        if bias_f32 is None and bias_bf16 is not None:
            bias_f32 = bias_bf16.float()
        return gemm_bf16(a, w, bias_f32, act)
    if bias_bf16 is None and bias_f32 is not None:
        bias_bf16 = bias_f32.to(torch.bfloat16)
    if act == "relu" and bias_bf16 is not None:
        return torch._addmm_activation(bias_bf16, a, w.t())
    if act == "none":
        return torch.addmm(bias_bf16, a, w.t()) if bias_bf16 is not None else torch.mm(a, w.t())
    # activations the library cannot fuse: ONE rounding of the biased pre-activation, like the native epilogue (fp32 bias applied
    # together with the activation in a single pass over the product)
    out = torch.mm(a, w.t())
    bias_act_(out, bias_f32 if bias_f32 is not None else (bias_bf16.float() if bias_bf16 is not None else None), act)
    return out


def gemm_bf16(a, w, bias=None, act="none", out_dtype=torch.bfloat16, out=None, tile_n=0, split3=False, hl_out=False):
    """act(a @ w.T + bias) on the hand-written MFMA kernel.  a [m, k] bf16 (row-strided views allowed), w [n, k] bf16 (the
    nn.Linear layout), bias [n] f32 or None, act in relu | gelu (erf) | leakyrelu | selu | none -> [m, n] bf16 or f32.
    split3: the result leaves as its bf16 image [hi | hi | lo], [m, 3 n] (operand of a following x3 GEMM).
    hl_out: the result leaves as its interleaved hl image [m, 2 n] (n % 32 == 0; operand of sparse_attn_fwd_x3_hl / gemm_hl)."""
    if a.dtype != torch.bfloat16 or w.dtype != torch.bfloat16:
        raise TypeError("gemm_bf16: a and w must be bfloat16")
    a = _rows16(a, "a")
    w = _rows16(w, "w")
    m, k = a.shape
    n = w.shape[0]
    if w.shape[1] != k:
        raise ValueError("gemm_bf16: a is %s but w is %s" % (tuple(a.shape), tuple(w.shape)))
    if bias is not None:
        bias = _req(bias, torch.float32, "bias", 1)
        if bias.shape[0] != n:
            raise ValueError("gemm_bf16: bias has %d entries for %d columns" % (bias.shape[0], n))
    if hl_out:
        if split3 or n % 32:
            raise ValueError("gemm_bf16: an hl-image output needs n %% 32 == 0 and excludes split3 (n = %d)" % n)
        if out is None:
            out = torch.empty(m, 2 * n, dtype=torch.bfloat16, device=a.device)
        elif out.shape != (m, 2 * n) or out.stride(1) != 1 or out.dtype != torch.bfloat16:
            raise ValueError("gemm_bf16: bad out buffer for the hl image")
        odt = DT_BF16_HL
    elif split3:
        if out is None:
            out = torch.empty(m, 3 * n, dtype=torch.bfloat16, device=a.device)
        elif out.shape != (m, 3 * n) or out.stride(1) != 1 or out.dtype != torch.bfloat16:
            raise ValueError("gemm_bf16: bad out buffer for the split image")
        odt = DT_BF16_SPLIT3
    else:
        if out is None:
            out = torch.empty(m, n, dtype=out_dtype, device=a.device)
        elif out.shape != (m, n) or out.stride(1) != 1 or out.dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("gemm_bf16: bad out buffer")
        odt = DT_F32 if out.dtype == torch.float32 else DT_BF16
    check(_ffi.load().snf_gemm_bf16(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), m, n, k, ACT_CODES[act], _p(out),
                                    out.stride(0), odt, int(tile_n), _stream()), "snf_gemm_bf16")
    return out


def gemm_lnfold_supported(n, k):
    """Domain of gemm_bf16_lnfold / gemm_bf16_resid_ (256-wide tiles of the bf16 GEMM with the epilogue variants of round 6)."""
    return n % 64 == 0 and k % 32 == 0 and k >= 96


def gemm_bf16_lnfold(a, w, colsum, bias, rowstats, act="none", out=None):
    """act(LN(x) W0^T + b0) with the LayerNorm folded into the GEMM: a [m, k] bf16 = the RAW rows x, w [n, k] bf16 = W0 diag(gamma),
    colsum [n] f32 = row sums of the rounded w, bias [n] f32 = W0 beta + b0, rowstats [m, 2] f32 = (mean, rstd) of the fp32 rows
    (vit_row_stats).  -> [m, n] bf16 (out: a row-strided view is fine).  act: none | gelu."""
    if a.dtype != torch.bfloat16 or w.dtype != torch.bfloat16:
        raise TypeError("gemm_bf16_lnfold: a and w must be bfloat16")
    a = _rows16(a, "a")
    w = _rows16(w, "w")
    m, k = a.shape
    n = w.shape[0]
    if w.shape[1] != k:
        raise ValueError("gemm_bf16_lnfold: a is %s but w is %s" % (tuple(a.shape), tuple(w.shape)))
    colsum = _req(colsum, torch.float32, "colsum", 1)
    bias = _req(bias, torch.float32, "bias", 1)
    rowstats = _req(rowstats, torch.float32, "rowstats", 2)
    if colsum.shape[0] != n or bias.shape[0] != n or tuple(rowstats.shape) != (m, 2):
        raise ValueError("gemm_bf16_lnfold: colsum / bias need %d entries, rowstats [%d, 2]" % (n, m))
    if out is None:
        out = torch.empty(m, n, dtype=torch.bfloat16, device=a.device)
    elif out.shape != (m, n) or out.stride(1) != 1 or out.dtype != torch.bfloat16:
        raise ValueError("gemm_bf16_lnfold: bad out buffer")
    check(_ffi.load().snf_gemm_bf16_lnfold(_p(a), a.stride(0), _p(w), w.stride(0), _p(colsum), _p(bias), _p(rowstats), m, n, k,
                                           ACT_CODES[act], _p(out), out.stride(0), _stream()), "snf_gemm_bf16_lnfold")
    return out


def gemm_bf16_resid_(x, a, w, bias, x_bf16, stats_part):
    """x [m, n] f32 IN PLACE  x += a @ w.T + bias; x_bf16 [m, n] bf16 receives the new x rounded; stats_part [m, n / 32, 2] f32 the
    (sum, sum of squares) of every 32-column group of the new row (vit_row_stats(part=...) turns them into (mean, rstd))."""
    if a.dtype != torch.bfloat16 or w.dtype != torch.bfloat16:
        raise TypeError("gemm_bf16_resid_: a and w must be bfloat16")
    a = _rows16(a, "a")
    w = _rows16(w, "w")
    m, k = a.shape
    n = w.shape[0]
    x = _req(x, torch.float32, "x", 2)
    bias = _req(bias, torch.float32, "bias", 1)
    if w.shape[1] != k or tuple(x.shape) != (m, n) or tuple(x_bf16.shape) != (m, n) or x_bf16.dtype != torch.bfloat16 \
            or not x_bf16.is_contiguous() or tuple(stats_part.shape) != (m, n // 32, 2) or stats_part.dtype != torch.float32 \
            or not stats_part.is_contiguous() or bias.shape[0] != n:
        raise ValueError("gemm_bf16_resid_: inconsistent shapes")
    check(_ffi.load().snf_gemm_bf16_resid(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), m, n, k, _p(x), x.stride(0), _p(x_bf16),
                                          x_bf16.stride(0), _p(stats_part), _stream()), "snf_gemm_bf16_resid")
    return x


def vit_row_stats(x=None, part=None, d=None, eps=1e-6, want_bf16=False):
    """(mean, rstd) [n, 2] f32 of LayerNorm(eps) over rows of width d: from the fp32 rows x [n, d] (want_bf16: also their bf16 copy) or
    from the moment pairs part [n, slots, 2] of gemm_bf16_resid_.  Returns (stats, bf16 copy or None)."""
    lib = _ffi.load()
    if part is not None:
        part = _req(part, torch.float32, "part", 3)
        n, slots = part.shape[0], part.shape[1]
        stats = torch.empty(n, 2, dtype=torch.float32, device=part.device)
        check(lib.snf_vit_row_stats(None, 0, _p(part), slots, n, int(d), float(eps), _p(stats), None, 0, _stream()), "snf_vit_row_stats")
        return stats, None
    x = _req(x, torch.float32, "x", 2)
    n, d = x.shape
    stats = torch.empty(n, 2, dtype=torch.float32, device=x.device)
    xb = torch.empty(n, d, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    check(lib.snf_vit_row_stats(_p(x), x.stride(0), None, 0, n, d, float(eps), _p(stats), _p(xb), d if want_bf16 else 0, _stream()),
          "snf_vit_row_stats")
    return stats, xb


GEMM_HL = True        # one-pass fp32-class GEMM kernel (snf_gemm_hl_bf16) where the shape fills the chip with 256 x 256 tiles


def gemm_x3(a_img, w_img, bias=None, act="none", out_dtype=torch.float32, out=None, split3=False, hl_out=False, resid=None):
    """fp32-class act(A W^T + bias) from the [hi | hi | lo] / [Wh | Wl | Wh] images as ONE bf16 GEMM over the 3 k concatenated
    columns (gemm_bf16): the form for shapes the one-pass kernel (gemm_hl) does not cover.  resid [m, n] f32 (fp32 output only): added
    in the epilogue (snf_gemm_bf16_resid_f32)."""
    if resid is None:
        return gemm_bf16(a_img, w_img, bias, act, out_dtype, out, split3=split3, hl_out=hl_out)
    if split3 or hl_out or out_dtype != torch.float32 or a_img.dtype != torch.bfloat16 or w_img.dtype != torch.bfloat16:
        raise ValueError("gemm_x3: resid needs a plain fp32 output")
    a_img, w_img = _rows16(a_img, "a"), _rows16(w_img, "w")
    m, k = a_img.shape
    n = w_img.shape[0]
    resid = _rows16(_req(resid, torch.float32, "resid", 2), "resid")
    if bias is not None:
        bias = _req(bias, torch.float32, "bias", 1)
    if w_img.shape[1] != k or tuple(resid.shape) != (m, n):
        raise ValueError("gemm_x3: inconsistent shapes")
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=a_img.device)
    check(_ffi.load().snf_gemm_bf16_resid_f32(_p(a_img), a_img.stride(0), _p(w_img), w_img.stride(0), _p(bias), _p(resid), resid.stride(0),
                                              m, n, k, ACT_CODES[act], _p(out), out.stride(0), 0, _stream()), "snf_gemm_bf16_resid_f32")
    return out


GEMM_TN = True          # the weight-gradient contractions of the training step on snf_gemm_tn_f32; False: batched library GEMMs


def gemm_tn_supported(n, p, q, *images):
    """Shapes / operands snf_gemm_tn_f32 takes: whole 32-row steps, 8-column granules, bf16 images with 16-byte aligned rows, an
    output large enough to be worth the matrix cores."""
    if not GEMM_TN or n < 1024 or n % 32 or p % 8 or q % 8 or p * q < 65536:
        return False
    return all(t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0
               and t.shape[0] == n and 33 * t.stride(0) < 2 ** 31 for t in images)


def gemm_tn(a_img, b_img, p, q, a_planes=(0, -1), b_planes=(0, -1), out=None, hl=False):
    """a^T b over the rows of two bf16 images -> [p, q] f32 (snf_gemm_tn_f32): a = the p columns at column a_planes[0] of a_img [n, .],
    b = the q columns at b_planes[0] of b_img [n, .]; with lo planes (a_planes[1], b_planes[1] >= 0: split images [hi | hi | lo]) the
    product is fp32-class (hi hi + hi lo + lo hi), otherwise one bf16 product.  hl: both images are interleaved ones ([hi(32) | lo(32)]
    per 32 columns; a_planes[0] / b_planes[0] = the IMAGE column where the operand starts, p % 32 == q % 32 == 0).  The contraction runs
    over the BAG axis: the weight gradients of the training step."""
    n = a_img.shape[0]
    if not gemm_tn_supported(n, p, q, a_img, b_img) or (hl and (p % 32 or q % 32 or a_planes[0] % 64 or b_planes[0] % 64)):
        raise ValueError("gemm_tn: shape n=%d p=%d q=%d / operands outside the kernel's domain" % (n, p, q))
    if out is None:
        out = torch.empty(p, q, dtype=torch.float32, device=a_img.device)
    lib = _ffi.load()
    nb = int(lib.snf_gemm_tn_ws_bytes(n, p, q))
    ws = _ws(nb, a_img.device)
    check(lib.snf_gemm_tn_f32(_p(a_img), a_img.stride(0), a_planes[0], a_planes[1], _p(b_img), b_img.stride(0), b_planes[0], b_planes[1],
                              1 if hl else 0, n, p, q, _p(out), out.stride(0), _p(ws), nb, _stream()), "snf_gemm_tn_f32")
    return out


def hl_eligible(m, n, k):
    """Shapes the one-pass fp32-class GEMM takes: 256 x 256 tiles have to fill the chip (>= 0.7 tiles per CU), k % 32 == 0."""
    if not GEMM_HL or k % 32 or n % 8 or n < 256 or k < 32 or m * 2 * k >= 2 ** 31 or n * 2 * k >= 2 ** 31:
        return False
    cus = _ffi.load().snf_device_cu_count()
    return ((m + 255) // 256) * ((n + 255) // 256) * 10 >= cus * 7


def split_hl_rows(x):
    """x [m, k] f32 (row-strided views allowed, k % 32 == 0) -> its interleaved split image [m, 2 k] bf16: every 32 columns as
    [hi(32) | lo(32)] (snf_split_hl_f32) -- the A operand of gemm_hl."""
    if x.dtype != torch.float32:
        raise TypeError("split_hl_rows: x must be float32")
    x = _rows16(x, "x")
    m, k = x.shape
    if k % 32:
        raise ValueError("split_hl_rows: k = %d is not a multiple of 32" % k)
    out = torch.empty(m, 2 * k, dtype=torch.bfloat16, device=x.device)
    check(_ffi.load().snf_split_hl_f32(_p(x), x.stride(0), m, k, _p(out), _stream()), "snf_split_hl_f32")
    return out


def split_hl_weight(w):
    """W [n, k] f32 -> its interleaved split image [n, 2 k] bf16 (the W operand of gemm_hl)."""
    w = w.detach().float()
    n, k = w.shape
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return torch.stack([hi.view(n, k // 32, 32), lo.view(n, k // 32, 32)], dim=2).reshape(n, 2 * k).contiguous()


def layernorm_rows_hl(x, gamma, beta, eps=1e-5, slot=None, patch_rows=None):
    """LayerNorm over rows (as layernorm_rows) written as the interleaved split image [n, 2 d] (d % 32 == 0)."""
    x = _req(x, torch.float32, "x", 2)
    n, d = x.shape
    if gamma is not None:
        gamma = _req(gamma, torch.float32, "gamma", 1)
    if beta is not None:
        beta = _req(beta, torch.float32, "beta", 1)
    if slot is not None:
        slot = _req(slot, torch.int32, "slot", 1)
        patch_rows = _req(patch_rows, torch.float32, "patch_rows", 2)
    out = torch.empty(n, 2 * d, dtype=torch.bfloat16, device=x.device)
    check(_ffi.load().snf_layernorm_rows_hl_f32(_p(x), n, d, _p(slot), _p(patch_rows), _p(gamma), _p(beta), float(eps), _p(out),
                                                _stream()), "snf_layernorm_rows_hl_f32")
    return out


def layernorm_rows_hl_patch_(img, rows, x, addend=None, gamma=None, beta=None, eps=1e-5):
    """LayerNorm of the rows of (x [k, d] + addend [k, d]) written over rows `rows` [k] int64 of the hl image img [n, 2 d] (in place)."""
    x = _req(x, torch.float32, "x", 2)
    k, d = x.shape
    rows = _req(rows, torch.int64, "rows", 1)
    if addend is not None:
        addend = _req(addend, torch.float32, "addend", 2)
    if img.dtype != torch.bfloat16 or img.dim() != 2 or img.shape[1] != 2 * d or not img.is_contiguous() or rows.shape[0] != k \
            or (addend is not None and tuple(addend.shape) != (k, d)):
        raise ValueError("layernorm_rows_hl_patch_: img %s rows %s x %s" % (tuple(img.shape), tuple(rows.shape), tuple(x.shape)))
    if gamma is not None:
        gamma = _req(gamma, torch.float32, "gamma", 1)
    if beta is not None:
        beta = _req(beta, torch.float32, "beta", 1)
    check(_ffi.load().snf_layernorm_rows_hl_patch_f32(_p(x), _p(addend), k, d, _p(rows), _p(gamma), _p(beta), float(eps), _p(img),
                                                      _stream()), "snf_layernorm_rows_hl_patch_f32")
    return img


def gemm_hl(a_hl, w_hl, bias=None, act="none", out_dtype=torch.float32, out=None, hl_out=False, resid=None):
    """fp32-class act(A W^T + bias) in ONE pass over the interleaved split images a_hl [m, 2 k], w_hl [n, 2 k] (split_hl_rows /
    layernorm_rows_hl / a previous call with hl_out=True; split_hl_weight): every product hi hi + hi lo + lo hi, fp32 accumulate.
    Returns [m, n] (out_dtype) or, with hl_out, the hl image [m, 2 n] of the result.  resid [m, n] f32 (fp32 output only) is added
    after the activation, in the epilogue."""
    if a_hl.dtype != torch.bfloat16 or w_hl.dtype != torch.bfloat16:
        raise TypeError("gemm_hl: operands must be bfloat16 hl images")
    a_hl = _rows16(a_hl, "a_hl")
    w_hl = _rows16(w_hl, "w_hl")
    m, k2 = a_hl.shape
    n = w_hl.shape[0]
    if w_hl.shape[1] != k2 or k2 % 64:
        raise ValueError("gemm_hl: images are %s and %s" % (tuple(a_hl.shape), tuple(w_hl.shape)))
    if bias is not None:
        bias = _req(bias, torch.float32, "bias", 1)
    if hl_out:
        if out is None:
            out = torch.empty(m, 2 * n, dtype=torch.bfloat16, device=a_hl.device)
        odt = DT_BF16_HL
    else:
        if out is None:
            out = torch.empty(m, n, dtype=out_dtype, device=a_hl.device)
        odt = DT_F32 if out.dtype == torch.float32 else DT_BF16
    ldr = 0
    if resid is not None:
        if hl_out or out.dtype != torch.float32 or resid.dtype != torch.float32 or tuple(resid.shape) != (m, n):
            raise ValueError("gemm_hl: resid needs an fp32 [m, n] tensor and an fp32 output")
        resid = _rows16(resid, "resid")
        ldr = resid.stride(0)
    lib = _ffi.load()
    ws, ws_bytes = None, 0
    if GEMM_HL_SPLITK:
        ws_bytes = int(lib.snf_gemm_hl_ws_bytes(m, n, k2 // 2))
        if ws_bytes:
            ws = _hl_workspace(a_hl.device, ws_bytes)
    check(lib.snf_gemm_hl_ws_bf16(_p(a_hl), a_hl.stride(0), _p(w_hl), w_hl.stride(0), _p(bias), _p(resid), ldr, m, n, k2 // 2,
                                  ACT_CODES[act], _p(out), out.stride(0), odt, _p(ws), ws_bytes, _stream()), "snf_gemm_hl_ws_bf16")
    return out


def gemm_hl_gated(a_hl, w_hl, gate_hl):
    """The hl image [m, 2 n] of (A W^T) o [gate > 0] in one pass (snf_gemm_hl_gated_bf16): the FFN input gradient behind its ReLU; gate_hl
    [m, 2 n] = the activation's own hl image (its hi values decide)."""
    a_hl, w_hl, gate_hl = _rows16(a_hl, "a_hl"), _rows16(w_hl, "w_hl"), _rows16(gate_hl, "gate_hl")
    if a_hl.dtype != torch.bfloat16 or w_hl.dtype != torch.bfloat16 or gate_hl.dtype != torch.bfloat16:
        raise TypeError("gemm_hl_gated: operands must be bfloat16 hl images")
    m, k2 = a_hl.shape
    n = w_hl.shape[0]
    if w_hl.shape[1] != k2 or k2 % 64 or n % 32 or tuple(gate_hl.shape) != (m, 2 * n):
        raise ValueError("gemm_hl_gated: inconsistent shapes")
    out = torch.empty(m, 2 * n, dtype=torch.bfloat16, device=a_hl.device)
    check(_ffi.load().snf_gemm_hl_gated_bf16(_p(a_hl), a_hl.stride(0), _p(w_hl), w_hl.stride(0), _p(gate_hl), gate_hl.stride(0), m, n,
                                             k2 // 2, _p(out), out.stride(0), _stream()), "snf_gemm_hl_gated_bf16")
    return out


def hl_colsum(img):
    """Column sums [k] f32 of the matrix whose hl image is img [m, 2 k] (k <= 4096): one colsum_fused pass over the image, hi + lo per column."""
    s, _ = colsum_fused(img)
    return s.view(-1, 2, 32).sum(1).reshape(-1)


GEMM_HL_SPLITK = True   # split-K of the last, partly filled round of tiles (snf_gemm_hl_ws_bf16); False: plain tile walk


def _hl_workspace(device, nbytes):
    """Scratch of one split-K call (tickets + partial-tile slabs), taken from the caching allocator PER CALL: the call zeroes its own
    tickets, so nothing has to survive between calls.  Inside a HIP-graph capture the buffer belongs to the graph's private pool and
    stays valid for every replay; concurrent calls on different streams get different buffers (stream-ordered allocator)."""
    return _ws(nbytes, device)


def split3_weight(w, colscale=None):
    """W [n, k] f32 (x diag(colscale), in fp32, if given) -> W3 [n, 3 k] bf16 = [Wh | Wl | Wh]: with an activation image [hi | hi | lo] (layernorm_rows_split3,
    gemm_bf16(split3=True)) one bf16 GEMM over the tripled K axis computes hi Wh^T + hi Wl^T + lo Wh^T, i.e. the fp32
    product to ~2^-17 relative per term with fp32 accumulation."""
    w = w.detach().float()
    if colscale is not None:
        colscale = colscale.detach().float().contiguous()
    if w.is_cuda and w.dim() == 2 and w.shape[1] % 8 == 0 and w.stride(1) == 1 and w.stride(0) % 4 == 0 and w.data_ptr() % 16 == 0:
        out = torch.empty(w.shape[0], 3 * w.shape[1], dtype=torch.bfloat16, device=w.device)
        check(_ffi.load().snf_split3_weight_f32(_p(w), w.stride(0), w.shape[0], w.shape[1], _p(colscale), _p(out), _stream()),
              "snf_split3_weight_f32")
        return out
    if colscale is not None:
        w = w * colscale
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo, hi], dim=1).contiguous()


def split3_rows(x):
    """x [m, k] f32 (row-strided views allowed) -> [m, 3 k] bf16 = [hi | hi | lo] (snf_split3_f32)."""
    if x.dtype != torch.float32:
        raise TypeError("split3_rows: x must be float32")
    x = _rows16(x, "x")
    m, k = x.shape
    if k % 8:
        hi = x.to(torch.bfloat16)
        lo = (x - hi.float()).to(torch.bfloat16)
        return torch.cat([hi, hi, lo], dim=1).contiguous()
    out = torch.empty(m, 3 * k, dtype=torch.bfloat16, device=x.device)
    check(_ffi.load().snf_split3_f32(_p(x), x.stride(0), m, k, _p(out), _stream()), "snf_split3_f32")
    return out


def split3_colsum(x, gate=None, out=None, col=0, want_colsum=True):
    """One pass over a gradient matrix x [m, k] f32 (row-strided views allowed): its split image [hi | hi | lo] (operand of the next
    fp32-class GEMM and of the weight-gradient contractions) and its column sums (the bias gradient); gate [m, k] bf16 (row-strided:
    the hi plane of the activation's own split image) zeroes the elements behind a closed ReLU first.  out / col: write into columns
    col .. col + k of each plane of a wider image out [m, 3 w] ([dQ | dV] share one).  Returns (image, colsum [k] or None).  See
    snf_split3_colsum_f32."""
    if x.dtype != torch.float32:
        raise TypeError("split3_colsum: x must be float32")
    x = _rows16(x, "x")
    m, k = x.shape
    if k % 8 or k > 8192:
        raise ValueError("split3_colsum: k=%d must be a multiple of 8 and <= 8192" % k)
    ldg = 0
    if gate is not None:
        if gate.dtype != torch.bfloat16 or tuple(gate.shape) != (m, k):
            raise ValueError("split3_colsum: gate must be bfloat16 of the shape of x")
        gate = _rows16(gate, "gate")
        ldg = gate.stride(0)
    if out is None:
        out = torch.empty(m, 3 * k, dtype=torch.bfloat16, device=x.device)
        col = 0
    elif (out.dtype != torch.bfloat16 or out.dim() != 2 or not out.is_contiguous() or out.shape[0] != m or out.shape[1] % 3
          or col % 8 or col + k > out.shape[1] // 3):
        raise ValueError("split3_colsum: bad out image")
    plane = out.shape[1] // 3
    lib = _ffi.load()
    part = torch.empty(lib.snf_colsum_blocks(m), k, dtype=torch.float32, device=x.device) if want_colsum else None
    dst = out if col == 0 else out[:, col:]
    check(lib.snf_split3_colsum_f32(_p(x), x.stride(0), m, k, _p(gate), ldg, _p(dst), out.stride(0), plane, _p(part), _stream()),
          "snf_split3_colsum_f32")
    return out, (part.sum(0) if part is not None else None)


def split_hl_colsum(x, gate_hl=None, out=None, col=0, want_colsum=True):
    """split3_colsum writing the INTERLEAVED image ([hi(32) | lo(32)] per 32 columns, [m, 2 k]: operand of gemm_hl and of gemm_tn(hl=True));
    gate_hl [m, 2 k] bf16: the activation's own hl image (its hi values decide).  out / col: write the k columns at TRUE column col of
    a wider image out [m, 2 w].  Returns (image, colsum [k] or None).  snf_split_hl_colsum_f32."""
    if x.dtype != torch.float32:
        raise TypeError("split_hl_colsum: x must be float32")
    x = _rows16(x, "x")
    m, k = x.shape
    if k % 32 or k > 8192:
        raise ValueError("split_hl_colsum: k=%d must be a multiple of 32 and <= 8192" % k)
    ldg = 0
    if gate_hl is not None:
        if gate_hl.dtype != torch.bfloat16 or tuple(gate_hl.shape) != (m, 2 * k):
            raise ValueError("split_hl_colsum: gate must be the bfloat16 hl image of a matrix of the shape of x")
        gate_hl = _rows16(gate_hl, "gate")
        ldg = gate_hl.stride(0)
    if out is None:
        out = torch.empty(m, 2 * k, dtype=torch.bfloat16, device=x.device)
        col = 0
    elif (out.dtype != torch.bfloat16 or out.dim() != 2 or not out.is_contiguous() or out.shape[0] != m or out.shape[1] % 64
          or col % 32 or 2 * (col + k) > out.shape[1]):
        raise ValueError("split_hl_colsum: bad out image")
    lib = _ffi.load()
    part = torch.empty(lib.snf_colsum_blocks(m), k, dtype=torch.float32, device=x.device) if want_colsum else None
    dst = out if col == 0 else out[:, 2 * col:]
    check(lib.snf_split_hl_colsum_f32(_p(x), x.stride(0), m, k, _p(gate_hl), ldg, _p(dst), out.stride(0), _p(part), _stream()),
          "snf_split_hl_colsum_f32")
    return out, (part.sum(0) if part is not None else None)


def linear_rows_x3_supported(r, c, k):
    return 1 <= r <= 8192 and c >= 1 and k >= 16 and k % 16 == 0


def linear_rows_x3(x, w, bias=None, out_dtype=torch.float32):
    """x [r, k] f32 @ w [c, k]^T f32 + bias for a few hundred rows (the K selected rows of a bag), fp32-class: split-bf16 x3 on the
    matrix cores with the operands split in registers (snf_linear_rows_x3_f32) -> [r, c] f32 or bf16."""
    if x.dtype != torch.float32 or w.dtype != torch.float32:
        raise TypeError("linear_rows_x3: x and w must be float32")
    x = _rows16(x, "x")
    w = _rows16(w, "w")
    r, k = x.shape
    c = w.shape[0]
    if w.shape[1] != k or not linear_rows_x3_supported(r, c, k) or out_dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("linear_rows_x3: x %s w %s outside the kernel (r <= 8192, k %% 16 == 0)" % (tuple(x.shape), tuple(w.shape)))
    if bias is not None:
        bias = _req(bias, torch.float32, "bias", 1)
    out = torch.empty(r, c, dtype=out_dtype, device=x.device)
    check(_ffi.load().snf_linear_rows_x3_f32(_p(x), x.stride(0), _p(w), w.stride(0), _p(bias), r, c, k, _p(out), c,
                                             DT_F32 if out_dtype == torch.float32 else DT_BF16, _stream()), "snf_linear_rows_x3_f32")
    return out


def linear_rows_x3_resid(x, w, bias, resid):
    """(delta, resid + delta) with delta = x [r, k] @ w [c, k]^T + bias, fp32-class, in ONE launch (snf_linear_rows_x3_resid_f32):
    the output projection of the K selected rows and x_sel = xs + delta (snuffy.py:205, 108)."""
    if x.dtype != torch.float32 or w.dtype != torch.float32:
        raise TypeError("linear_rows_x3_resid: x and w must be float32")
    x, w = _rows16(x, "x"), _rows16(w, "w")
    r, k = x.shape
    c = w.shape[0]
    resid = _req(resid, torch.float32, "resid", 2)
    if w.shape[1] != k or not linear_rows_x3_supported(r, c, k) or tuple(resid.shape) != (r, c):
        raise ValueError("linear_rows_x3_resid: x %s w %s resid %s outside the kernel" % (tuple(x.shape), tuple(w.shape), tuple(resid.shape)))
    if bias is not None:
        bias = _req(bias, torch.float32, "bias", 1)
    out = torch.empty(r, c, dtype=torch.float32, device=x.device)
    out2 = torch.empty(r, c, dtype=torch.float32, device=x.device)
    check(_ffi.load().snf_linear_rows_x3_resid_f32(_p(x), x.stride(0), _p(w), w.stride(0), _p(bias), _p(resid), resid.stride(0), r, c, k,
                                                   _p(out), c, _p(out2), c, _stream()), "snf_linear_rows_x3_resid_f32")
    return out, out2


def gemm_x3_supported(m, n, k):
    return gemm_supported(m, n, 3 * k) and m * 3 * k < 2 ** 31 and n * 3 * k < 2 ** 31


def layernorm_rows_split3(x, gamma, beta, eps=1e-5, slot=None, patch_rows=None):
    """LayerNorm over rows (as layernorm_rows) written as the split bf16 image [n, 3 d]."""
    x = _req(x, torch.float32, "x", 2)
    n, d = x.shape
    if gamma is not None:
        gamma = _req(gamma, torch.float32, "gamma", 1)
    if beta is not None:
        beta = _req(beta, torch.float32, "beta", 1)
    if slot is not None:
        slot = _req(slot, torch.int32, "slot", 1)
        patch_rows = _req(patch_rows, torch.float32, "patch_rows", 2)
    out = torch.empty(n, 3 * d, dtype=torch.bfloat16, device=x.device)
    check(_ffi.load().snf_layernorm_rows_split3_f32(_p(x), n, d, _p(slot), _p(patch_rows), _p(gamma), _p(beta), float(eps),
                                                    _p(out), _stream()), "snf_layernorm_rows_split3_f32")
    return out


# ----------------------------------------------------------------------------------------------------------------------
# ViT extractor ops (K12-K14)
# ----------------------------------------------------------------------------------------------------------------------
def vit_patchify(img, patch, out_dtype=torch.float32):
    """im2col of the patch-embedding conv: img [B, C, H, W] f32 -> [B * P, C * patch * patch]."""
    img = _req(img, torch.float32, "img", 4)
    b, c, h, w = img.shape
    if h % patch or w % patch:
        raise ValueError("image %dx%d is not a multiple of the patch size %d" % (h, w, patch))
    cols = torch.empty(b * (h // patch) * (w // patch), c * patch * patch, dtype=out_dtype, device=img.device)
    dt = DT_F32 if out_dtype == torch.float32 else DT_BF16
    check(_ffi.load().snf_vit_patchify(_p(img), b, c, h, w, patch, _p(cols), dt, _stream()), "snf_vit_patchify")
    return cols


def vit_assemble_tokens(patch_emb, cls_token, pos_embed, b):
    """[cls ; patches] + pos -> tokens [B * (P + 1), D] f32.  patch_emb [B * P, D] f32 or bf16."""
    if patch_emb.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("patch_emb must be float32 or bfloat16")
    patch_emb = _req(patch_emb, patch_emb.dtype, "patch_emb", 2)
    d = patch_emb.shape[1]
    p = patch_emb.shape[0] // b
    cls_token = _req(cls_token.reshape(-1), torch.float32, "cls_token", 1)
    pos_embed = _req(pos_embed.reshape(-1, d), torch.float32, "pos_embed", 2)
    if pos_embed.shape[0] != p + 1:
        raise ValueError("pos_embed has %d rows, need %d" % (pos_embed.shape[0], p + 1))
    tokens = torch.empty(b * (p + 1), d, dtype=torch.float32, device=patch_emb.device)
    dt = DT_F32 if patch_emb.dtype == torch.float32 else DT_BF16
    check(_ffi.load().snf_vit_assemble_tokens(_p(patch_emb), dt, _p(cls_token), _p(pos_embed), b, p, d, _p(tokens),
                                              _stream()), "snf_vit_assemble_tokens")
    return tokens


def vit_residual_ln_(x, add1=None, add2=None, scale2=1.0, gamma=None, beta=None, eps=1e-6, want_ln=True, want_x_bf16=False):
    """x += add1 + scale2 * add2 in place (bf16 addends); returns (LayerNorm(x) bf16 or None, bf16 copy of x or None)."""
    x = _req(x, torch.float32, "x", 2)
    n, d = x.shape
    for nm, t in (("add1", add1), ("add2", add2)):
        if t is not None:
            _req(t, torch.bfloat16, nm, 2)
            if not t.is_contiguous() or t.shape != x.shape:
                raise ValueError("%s must be a contiguous bf16 tensor of x's shape" % nm)
    ln = torch.empty(n, d, dtype=torch.bfloat16, device=x.device) if want_ln else None
    xb = torch.empty(n, d, dtype=torch.bfloat16, device=x.device) if want_x_bf16 else None
    if want_ln:
        gamma = _req(gamma, torch.float32, "gamma", 1)
        beta = _req(beta, torch.float32, "beta", 1)
    check(_ffi.load().snf_vit_residual_ln(_p(x), n, d, _p(add1), _p(add2), float(scale2), _p(gamma), _p(beta), float(eps),
                                          _p(ln), _p(xb), _stream()), "snf_vit_residual_ln")
    return ln, xb


def vit_attention(qkv, b, t, heads, scale=None, need_attn=False, arithmetic="exact", hl_out=False):
    """Multi-head self-attention on the qkv Linear output [B*T, 3*D].  fp32 -> exact kernel (+ optional attn [B,h,T,T]), or with
    arithmetic="x3" (and no attn asked for, dk == 64) the MFMA kernel in split-bf16 x3 products; bf16 -> MFMA kernel (dk == 64,
    T <= VIT_MFMA_MAX_T; keys in LDS chunks above T = 256)."""
    d3 = qkv.shape[1]
    d = d3 // 3
    dk = d // heads
    scale = dk ** -0.5 if scale is None else scale
    lib = _ffi.load()
    if qkv.dtype == torch.float32:
        qkv = _req(qkv, torch.float32, "qkv", 2)
        if arithmetic == "x3" and not need_attn and vit_mfma_attention_supported(t, dk):
            # hl_out: the result leaves as its interleaved hl image [b t, 2 d] bf16 (the operand of the proj GEMM), never as fp32
            out = (torch.empty(b * t, 2 * d, dtype=torch.bfloat16, device=qkv.device) if hl_out else
                   torch.empty(b * t, d, dtype=torch.float32, device=qkv.device))
            check(lib.snf_vit_attention_x3_f32(_p(qkv), b, t, heads, dk, float(scale), _p(out), DT_BF16_HL if hl_out else DT_F32, _stream()),
                  "snf_vit_attention_x3_f32")
            return out, None
        if hl_out:
            raise ValueError("vit_attention: hl_out needs arithmetic='x3' (no attn, dk == 64)")
        out = torch.empty(b * t, d, dtype=torch.float32, device=qkv.device)
        attn = torch.empty(b, heads, t, t, dtype=torch.float32, device=qkv.device) if need_attn else None
        check(lib.snf_vit_attention_f32(_p(qkv), b, t, heads, dk, float(scale), _p(out), _p(attn), _stream()),
              "snf_vit_attention_f32")
        return out, attn
    qkv = _req(qkv, torch.bfloat16, "qkv", 2)
    if need_attn:
        raise ValueError("vit_attention: the MFMA kernel does not materialise attn; use the fp32 path")
    out = torch.empty(b * t, d, dtype=torch.bfloat16, device=qkv.device)
    check(lib.snf_vit_attention_mfma(_p(qkv), b, t, heads, dk, float(scale), _p(out), _stream()), "snf_vit_attention_mfma")
    return out, None


VIT_MFMA_MAX_T = 4096    # SNF_VIT_MFMA_MAX_T of include/snuffy_hip.h


def vit_mfma_attention_supported(t, dk):
    return dk == 64 and t <= VIT_MFMA_MAX_T
