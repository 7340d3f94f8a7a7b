#!/usr/bin/env python3
"""Dev builds of libsnuffy_hip.so with pieces of the one-pass x3 GEMM step compiled out (timing ablations; results are wrong).
  here:     python tools/gemm_x3_ablate.py build
  GPU box:  bash tools/gemm_x3_ablate.sh"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VAR = os.path.join(ROOT, "snuffy_amd", "build", "variants")
VARIANTS = {"x3_nomfma": ["-DX3_NOMFMA"], "x3_noread": ["-DX3_NOREAD"], "x3_nostage": ["-DX3_NOSTAGE"], "x3_nostore": ["-DX3_NOSTORE"],
            "x3_stageonly": ["-DX3_NOMFMA", "-DX3_NOREAD", "-DX3_NOSTORE"], "x3_mfmaonly": ["-DX3_NOSTAGE", "-DX3_NOREAD", "-DX3_NOSTORE"],
            "x3_nostore_nomfma": ["-DX3_NOSTORE", "-DX3_NOMFMA"]}

if __name__ == "__main__":
    from snuffy_amd import build as B
    B.build_lib()
    os.makedirs(VAR, exist_ok=True)
    only = sys.argv[2:]
    for name, defs in VARIANTS.items():
        if only and name not in only:
            continue
        obj = os.path.join(VAR, "gemm_%s.o" % name)
        subprocess.run([B._hipcc()] + B.FLAGS + defs + ["-c", os.path.join(B.CSRC, "gemm.hip"), "-o", obj], check=True)
        objs = [os.path.join(B.OBJDIR, os.path.basename(s)[:-4] + ".o") for s in B.sources() if not s.endswith("/gemm.hip")] + [obj]
        subprocess.run([B._hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", os.path.join(VAR, "lib_%s.so" % name)] + objs,
                       check=True)
        print("built", name)
