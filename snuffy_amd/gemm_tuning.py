"""Solution selection for the library GEMMs (hipBLASLt / rocBLAS through torch), via PyTorch's TunableOp.

Measured on MI355X (ROCm 7.2, PyTorch 2.10): for the aggregator's projection shapes (M = 32768, K = 768 / 3072) the
library's default heuristic is already within noise of its best kernel -- a table changes nothing for the eval forward;
the training step's backward shapes gain 5 % (176 -> 185 slides/s, bench.py --mode train --gemm-table).  For the ViT
extractor's skinny-K shapes (M = batch * 197, K = 384) it is not: fc1 + GELU epilogue 297 -> 216 us, proj 97 -> 55 us,
fc2 165 -> 129 us per layer at batch 512, ViT-S/16 + adapter 37.6 k -> 43.1 k img/s.  ``snuffy_amd/tuning/gemm_gfx950.csv``
holds the selections for batch 512; other batch sizes are tuned online on first use (``tune_missing=True``, a few seconds
per new shape, results appended to the table).  The table carries validators (PyTorch / ROCm / hipBLASLt / rocBLAS build,
gfx arch): TunableOp ignores it when they do not match the running stack, so a stale table cannot select a wrong kernel.

Opt-in only (``bench.py --gemm-table``, ``tools/bench_vit.py --gemm-table``, ``compute_feats.py --tune_gemms 1``): nothing
changes on import.
"""
import os

import torch

DEFAULT_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "gemm_gfx950.csv")


def use_pretuned_gemms(path=None, tune_missing=False):
    """Apply the solution table ``path`` (default: the shipped one).  ``tune_missing=True`` additionally times every
    unseen GEMM shape once and records the winner in the table.  Returns True when TunableOp was switched on."""
    path = path or DEFAULT_TABLE
    if not torch.cuda.is_available():
        return False
    if not tune_missing and not os.path.exists(path):
        return False
    tun = torch.cuda.tunable
    tun.enable(True)
    tun.set_filename(path, insert_device_ordinal=False)
    tun.tuning_enable(bool(tune_missing))
    if os.path.exists(path):
        tun.read_file(path)
    return True
