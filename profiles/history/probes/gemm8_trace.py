#!/usr/bin/env python3
"""In-kernel anatomy of the quadrant-phase GEMM: s_memtime stamps of workgroup 0 (waves 0 and 4) around every barrier.
  here (no GPU):  python tools/gemm8_trace.py build      -> snuffy_amd/build/variants/lib_g8trace.so (gemm8.hip with -DSNF_GEMM_TRACE)
  GPU box:        SNUFFY_HIP_LIB=snuffy_amd/build/variants/lib_g8trace.so python tools/gemm8_trace.py run m n k"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VAR = os.path.join(ROOT, "snuffy_amd", "build", "variants")


def build(name="g8trace", defs=("-DSNF_GEMM_TRACE",)):
    from snuffy_amd import build as B
    B.build_lib()
    os.makedirs(VAR, exist_ok=True)
    obj = os.path.join(VAR, "gemm8_%s.o" % name)
    subprocess.run([B._hipcc()] + B.FLAGS + list(defs) + ["-c", os.path.join(B.CSRC, "gemm8.hip"), "-o", obj], check=True)
    objs = [os.path.join(B.OBJDIR, os.path.basename(s)[:-4] + ".o") for s in B.sources() if not s.endswith("gemm8.hip")] + [obj]
    subprocess.run([B._hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", os.path.join(VAR, "lib_%s.so" % name)] + objs, check=True)
    print("built", os.path.join(VAR, "lib_%s.so" % name))


def build_ablations():
    for name, defs in (("g8_stageonly_linear", ("-DG8_NOMFMA", "-DG8_NOREAD", "-DG8_LINEAR")), ("g8_linear", ("-DG8_LINEAR",)),
                       ("g8_nomfma", ("-DG8_NOMFMA",)), ("g8_nostage", ("-DG8_NOSTAGE",)), ("g8_noread", ("-DG8_NOREAD",)),
                       ("g8_mfmaonly", ("-DG8_NOSTAGE", "-DG8_NOREAD")), ("g8_stageonly", ("-DG8_NOMFMA", "-DG8_NOREAD"))):
        build(name, defs)


def run(m, n, k):
    import torch
    from snuffy_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(n, generator=g).to(dev)
    tr = torch.zeros(1600, dtype=torch.int64, device=dev)
    for _ in range(3):
        ops.gemm_bf16(a, w, b, "none", tile_n=512)
    os.environ["SNF_GEMM_TRACE_PTR"] = str(tr.data_ptr())
    ops.gemm_bf16(a, w, b, "none", tile_n=512)
    torch.cuda.synchronize()
    t = tr.cpu().numpy().reshape(2, 800)
    # per phase 5 stamps: [issue done (reads, stage, vmcnt wait)] [before barrier 1 (lgkm waited)] [after barrier 1] [before barrier 2
    # (MFMA burst issued)] [after barrier 2]
    names = ["L issue+vmcnt", "lgkm wait", "barrier 1", "MFMA burst", "barrier 2"]
    for grp in range(2):
        st = t[grp][t[grp] > 0]
        st = st[2:]            # the prologue barrier's two stamps
        nph = (len(st) - 1) // 5
        print("group %d: %d phases traced; ticks per segment, K tiles 2..5 (phases 0-3 each)" % (grp, nph))
        for kt in range(2, 6):
            for ph in range(4):
                i = (kt * 4 + ph) * 5
                if i + 5 >= len(st):
                    break
                prev = st[i - 1] if i > 0 else st[0]
                seg = [int(st[i] - prev)] + [int(st[i + j + 1] - st[i + j]) for j in range(4)]
                print("   kt %d ph %d: " % (kt, ph) + "  ".join("%s %4d" % (nm, v) for nm, v in zip(names, seg)))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "ablations":
        build_ablations()
    else:
        run(*[int(v) for v in sys.argv[2:5]])
