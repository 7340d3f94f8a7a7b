import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import ops
for (m, n, k) in ((32768, 768, 3072), (100000, 1536, 768), (100000, 768, 3072), (100000, 3072, 768)):
    a = ops.split_hl_rows(torch.randn(m, k, device="cuda"))
    w = ops.split_hl_weight(torch.randn(n, k, device="cuda") / k ** 0.5)
    res = torch.randn(m, n, device="cuda")
    out = torch.empty(m, n, device="cuda")
    for flag in (False, True, False, True):
        ops.GEMM_HL_SPLITK = flag
        for _ in range(10):
            ops.gemm_hl(a, w, out=out, resid=res)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            ops.gemm_hl(a, w, out=out, resid=res)
        e1.record()
        torch.cuda.synchronize()
        print("m=%d n=%d k=%d splitk=%s: %.1f us" % (m, n, k, flag, e0.elapsed_time(e1) * 5))
