"""Development probe: is gemm_hl_kernel bound per CU or chip-wide?  Times the config-B FFN-in projection (1536 tiles of 256 x 256)
walked by SNF_GEMM_HL_GRID workgroups (one per CU) and samples clock / power through rocm-smi while a long loop runs.
usage (GPU box): for g in 256 192 128 64; do SNF_GEMM_HL_GRID=$g python tools/gemm_hl_occupancy_probe.py; done"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import ops  # noqa: E402

ops.GEMM_HL_SPLITK = False
m, n, k = 32768, 3072, 768
a = ops.split_hl_rows(torch.randn(m, k, device="cuda"))
w = ops.split_hl_weight(torch.randn(n, k, device="cuda") / k ** 0.5)
out = torch.empty(m, n, device="cuda")
for _ in range(20):
    ops.gemm_hl(a, w, out=out)
torch.cuda.synchronize()
samples = []
stop = False


def sample():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append(r.strip().split("\n")[-1])
        except Exception as e:  # noqa: BLE001
            samples.append(str(e))
        time.sleep(0.2)


th = threading.Thread(target=sample)
th.start()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 4000
e0.record()
for _ in range(reps):
    ops.gemm_hl(a, w, out=out)
e1.record()
torch.cuda.synchronize()
stop = True
th.join()
us = e0.elapsed_time(e1) * 1e3 / reps
print("grid %s: %.1f us per launch, %.2f PF/s issued" % (os.environ.get("SNF_GEMM_HL_GRID", "all"), us, 3 * 2.0 * m * n * k / us / 1e9))
for s in samples[2:8]:
    print("   ", s[:200])
