"""Build libsnuffy_hip.so (hand-written HIP for gfx950) in-tree with hipcc.  No torch, no cmake: plain C ABI."""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsnuffy_hip.so")
OBJDIR = os.path.join(HERE, "build")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result"]
# per-file extras.  The attention kernels never see NaN operands in their max-reductions; without this flag every fmaxf
# costs two extra canonicalising v_max_f32 x,x,x (IEEE sNaN quieting) and the DPP move cannot fold into the max.
# (-inf is still honoured: the padded-key masking relies on exp2(-inf) == 0.)
# The SLP vectoriser is off for the attention kernel: it pairs the running softmax sums into v_pk_add_f32 and then needs
# three v_mov per four probabilities to re-pair them for the explicit v_pk_mul / v_cvt_pk of the normalisation.
EXTRA_FLAGS = {"sparse_attn_mfma.hip": ["-fno-honor-nans", "-fno-slp-vectorize"],
               "sparse_attn_mfma_dk64.hip": ["-fno-honor-nans", "-fno-slp-vectorize"],
               "sparse_attn_mfma_varlen.hip": ["-fno-honor-nans", "-fno-slp-vectorize"],
               "sparse_attn_mfma_varlen_dk64.hip": ["-fno-honor-nans", "-fno-slp-vectorize"], "vit.hip": ["-fno-honor-nans"],
               "sparse_attn_x3.hip": ["-fno-honor-nans"],
               "sparse_attn_x3p.hip": ["-fno-honor-nans", "-fno-slp-vectorize"],
               "sparse_attn_x3p_k2.hip": ["-fno-honor-nans", "-fno-slp-vectorize"],
               "sparse_attn_x3p_dk64.hip": ["-fno-honor-nans", "-fno-slp-vectorize"]}


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libsnuffy_hip.so")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


_INC = None


def _deps_mtime(src=None):
    """Newest modification time among the headers `src` includes (directly or through other headers of csrc/); src=None: among
    all headers.  The public header include/snuffy_hip.h reaches every source through common.h."""
    import re
    global _INC
    if _INC is None:
        _INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)
    pub = os.path.join(os.path.dirname(HERE), "include", "snuffy_hip.h")
    if src is None:
        deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [pub]
        return max(os.path.getmtime(d) for d in deps)
    seen, todo = set(), [src]
    while todo:
        f = todo.pop()
        try:
            with open(f) as fh:
                text = fh.read()
        except OSError:
            continue
        for inc in _INC.findall(text):
            path = os.path.normpath(os.path.join(os.path.dirname(f), inc))
            if path not in seen and os.path.exists(path):
                seen.add(path)
                todo.append(path)
    return max([os.path.getmtime(d) for d in seen] + [0.0])


def _compile(src, force, hdr_mtime):
    obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
            and os.path.getmtime(obj) >= _deps_mtime(src)):
        return obj, False
    cmd = [_hipcc()] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
    if os.environ.get("SNF_ATTN_DEV"):   # development: only the config-B attention variants (7 key blocks)
        cmd.insert(1, "-DSNF_ATTN_DEV")
    for d in os.environ.get("SNF_EXTRA_DEFS", "").split():   # development: timing ablations (-DX3P_ABL_...), never in a shipped build
        cmd.insert(1, "-D" + d)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj, True


def build_lib(force=False, verbose=False, jobs=4):
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hdr = _deps_mtime()
    srcs = sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        res = list(ex.map(lambda s: _compile(s, force, hdr), srcs))
    objs = [o for o, _ in res]
    rebuilt = any(c for _, c in res)
    if rebuilt or not os.path.exists(LIB):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv, verbose=True)
