"""The fp32-class weight-gradient contraction of the training step (autograd._tn3) at config B's shapes: the three batched library
products plus the sum of their partials, by row-chunk count, against snf_gemm_tn_f32 on the same images (and, for one bf16 plane each,
against the bf16 chain's _tn_mm).   python tools/tn_chunks_bench.py [n]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import autograd as AG  # noqa: E402
from snuffy_amd import ops  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(1)
    for p, q, what in ((768, 3072, "dW2  = dz^T hid"), (3072, 768, "dW1' = dhid^T xhat"), (1536, 768, "dWqv' = [dQ|dV]^T xhat")):
        a3 = torch.randn(n, 3 * p, device=dev, generator=g).to(torch.bfloat16)
        b3 = torch.randn(n, 3 * q, device=dev, generator=g).to(torch.bfloat16)
        ops.GEMM_TN = False                       # the batched library products of rounds 2 - 5 ...
        ref = AG._tn3(a3, b3, p, q, chunks=8)
        line = []
        for chunks in (2, 4, 8, 16, 32):
            out = AG._tn3(a3, b3, p, q, chunks=chunks)
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            line.append("chunks %2d %7.1f us (rel diff %.1e)" % (chunks, timed(lambda: AG._tn3(a3, b3, p, q, chunks=chunks)), err))
        print("%-24s n=%d p=%d q=%d : %s" % (what, n, p, q, " | ".join(line)))
        ops.GEMM_TN = True                        # ... against the contraction kernel
        out = ops.gemm_tn(a3, b3, p, q, (p, 2 * p), (q, 2 * q))
        print("%-24s snf_gemm_tn_f32 x3     : %7.1f us (rel diff %.1e)" % ("", timed(lambda: ops.gemm_tn(a3, b3, p, q, (p, 2 * p), (q, 2 * q))),
                                                                      ((out - ref).abs().max() / ref.abs().max()).item()))
        a, b = a3[:, :p].contiguous(), b3[:, :q].contiguous()
        print("%-24s snf_gemm_tn_f32 bf16   : %7.1f us" % ("", timed(lambda: ops.gemm_tn(a, b, p, q))))
        line = ["chunks %2d %7.1f us" % (c, timed(lambda: AG._tn_mm(a, b, chunks=c))) for c in (2, 4, 8, 16)]
        print("%-24s bf16 (_tn_mm)        : %s" % ("", " | ".join(line)))


if __name__ == "__main__":
    main()
