#!/usr/bin/env python3
"""Scratch (register-spill) bytes per lane of every kernel in the built objects (snuffy_amd/build/*.o), read from the AMDGPU metadata of
the embedded gfx950 code objects -- no recompilation.  Round 5 found 21 spilled registers in the headline attention instantiation this way
(PMC traffic x1.37 instead of x1.29); tests/test_build_no_spills.py keeps the hot-path kernels at zero.
usage: python tools/scan_spills.py [substring of a kernel name ...]"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(objdir=None):
    """[(object, mangled kernel name, scratch bytes per lane, spilled VGPRs, VGPRs)] of every kernel in the build directory."""
    objdir = objdir or os.path.join(ROOT, "snuffy_amd", "build")
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        raise RuntimeError("llvm-objdump / llvm-readelf not found under %s" % LLVM)
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(objdir, "*.o"))):
            local = os.path.join(tmp, os.path.basename(obj))
            shutil.copy(obj, local)
            subprocess.run([objdump, "--offloading", local], capture_output=True, text=True, cwd=tmp)
            for co in glob.glob(local + ".*amdgcn*"):
                notes = subprocess.run([readelf, "--notes", co], capture_output=True, text=True).stdout
                for blk in notes.split("  - .agpr_count")[1:]:
                    name = re.search(r"\.name:\s+(\S+)", blk)
                    scr = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
                    spill = re.search(r"\.vgpr_spill_count:\s+(\d+)", blk)
                    vg = re.search(r"\.vgpr_count:\s+(\d+)", blk)
                    if name and scr:
                        out.append((os.path.basename(obj), name.group(1), int(scr.group(1)), int(spill.group(1)) if spill else 0,
                                    int(vg.group(1)) if vg else 0))
    return out


def demangle(names):
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not filt:
        return list(names)
    return subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()


def main():
    ks = kernels()
    pretty = demangle([k[1] for k in ks])
    pats = sys.argv[1:]
    n = 0
    for (obj, _, scr, spill, vg), name in zip(ks, pretty):
        if pats and not any(p in name for p in pats):
            continue
        if scr or pats:
            print("%-34s scratch %4d B  spilled %3d  vgpr %3d  %s" % (obj, scr, spill, vg, name[:150]))
            n += 1
    print("%d kernels listed of %d" % (n, len(ks)))


if __name__ == "__main__":
    main()
