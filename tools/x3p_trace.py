"""Development: s_memtime anatomy of the pipelined attention kernel (build with SNF_EXTRA_DEFS=X3P_TRACE)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import _ffi, ops  # noqa: E402

WG = int(sys.argv[1]) if len(sys.argv) > 1 else 0
KBW = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n, k, h, dk = 32768, 200, 6, 128
d = h * dk
lib = _ffi.load()
lib.snf_debug_attn_trace.argtypes = [ctypes.c_void_p]
lib.snf_debug_attn_trace.restype = None
lib.snf_debug_attn_trace_wg.argtypes = [ctypes.c_int]
lib.snf_debug_attn_trace_wg.restype = None
lib.snf_debug_x3p_kbw(KBW)
qv = torch.randn(n, 2 * d, device="cuda")
img = ops.split_hl_rows(qv)
kp = torch.randn(k, d, device="cuda")
for _ in range(3):
    ops.sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], kp, h)
buf = torch.zeros(8 * 64 * 8 + 8 * 64 * 16, dtype=torch.int64, device="cuda")
lib.snf_debug_attn_trace_wg(WG)
lib.snf_debug_attn_trace(ctypes.c_void_p(buf.data_ptr()))
ops.sparse_attn_fwd_x3_hl(img[:, :2 * d], img[:, 2 * d:], kp, h)
torch.cuda.synchronize()
lib.snf_debug_attn_trace(None)
t2 = buf.cpu()[8 * 64 * 8:].view(8, 64, 16)
t = buf.cpu()[:8 * 64 * 8].view(8, 64, 8)
names = ["top", "dma issued", "first half", "second half", "vm wait", "barrier"]
for w in range(8):
    if t[w].abs().sum() == 0:
        continue
    t0 = int(t[w, 0, 0]) if int(t[w, 0, 0]) else int(t[w][t[w] > 0].min())
    print("wave", w)
    for it in range(0, 28):
        row = t[w, it]
        if row[:6].abs().sum() == 0:
            continue
        base = int(row[0])
        print("  it %2d  start %8d | " % (it - 1, base - t0) + "  ".join("%s +%d" % (names[k], int(row[k]) - base) for k in range(1, 6) if int(row[k])))
        if it in (6, 7) and int(t2[w, it, 0]):
            print("        fine: " + " ".join("%d" % (int(t2[w, it, k]) - base) for k in range(12)))
