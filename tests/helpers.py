"""Shared test helpers (golden-vector loading)."""
import glob
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_case(path, dtype=torch.float32):
    z = np.load(path, allow_pickle=False)
    sd = {k[3:]: torch.from_numpy(z[k]).to(dtype) for k in z.files if k.startswith("sd.")}
    return z, sd


def forced_sel(z, n_layers):
    """Selection per layer as the reference made it: top (same for every layer) + that layer's random draw."""
    top = torch.from_numpy(z["top"].astype(np.int64)) if "top" in z.files else None
    n_rnd = int(z["n_rnd"])
    sels = []
    for l in range(n_layers):
        if n_rnd:
            sels.append(torch.cat([top, torch.from_numpy(z[f"rnd{l}"].astype(np.int64))]))
        else:
            sels.append(top)
    return sels


class ReplayRNG:
    """Stands in for np.random in the oracle: replays MT19937 from a seed (same stream as np.random.seed(seed))."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)

    def permutation(self, n):
        return self.rs.permutation(n)


def build_amd_milnet(D, h, act, big_lambda, r, depth, C=1, mlp=4, enc_drop=0.0):
    """Same construction sequence as the reference's train.Snuffy._get_milnet (train.py:861-890)."""
    from snuffy_amd import snuffy
    return snuffy.build_milnet(D, h, act, big_lambda, r, depth, C, mlp, enc_drop)


def rel_err(a, b):
    """max |a-b| / max |b| (normalised max error)."""
    a = a.double()
    b = b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
