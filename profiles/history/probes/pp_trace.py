#!/usr/bin/env python3
"""Dev: s_memtime anatomy of the ping-pong attention kernel (workgroup 0, one wave of each group): per barrier interval the
time a wave worked (release -> arrival) and waited (arrival -> release)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import _ffi, ops
dev = torch.device("cuda")
N, D, h, K = 32768, 768, 6, 200
g = torch.Generator().manual_seed(1)
kp = torch.randn(K, D, generator=g).to(dev).to(torch.bfloat16)
qv = torch.randn(N, 2 * D, generator=g).to(dev).to(torch.bfloat16)
buf = torch.zeros(4096, dtype=torch.int64, device=dev)
lib = _ffi.load()
lib.snf_debug_attn_trace.argtypes = [ctypes.c_void_p]
lib.snf_debug_attn_trace_wg(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for _ in range(3):
    ops.sparse_attn_fwd_mfma(qv[:, :D], qv[:, D:], kp, N, h)
buf.zero_()
lib.snf_debug_attn_trace(ctypes.c_void_p(buf.data_ptr()))
ops.sparse_attn_fwd_mfma(qv[:, :D], qv[:, D:], kp, N, h)
torch.cuda.synchronize()
lib.snf_debug_attn_trace(ctypes.c_void_p(0))
st = buf.cpu().numpy()[:512].reshape(2, 256)
t0 = st[0][st[0] > 0][0]
names_x = ["Kp", "G1(first)+Q", "S+G2o", "C", "G2", ]
for grp in range(2):
    s = st[grp][st[grp] > 0]
    print(f"group {'XY'[grp]}: {len(s)} stamps, first +{s[0]-t0}")
    for i in range(0, min(len(s) - 2, 60), 2):
        work = s[i] - (s[i - 1] if i else s[i]); wait = s[i + 1] - s[i]
        print(f"   barrier {i//2:2d}: worked {work:6d}  waited {wait:6d}")
    print("   total", s[-1] - s[0])
