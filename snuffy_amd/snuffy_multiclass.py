"""MI355X-native drop-in for the reference's ``snuffy_multiclass.py`` (batched / multi-class Snuffy).

Same module API as the reference (snuffy_multiclass.py:60-253): ``EncoderLayer(size, self_attn, feed_forward, num_class,
dropout, big_lambda, random_patch_share)`` with ``forward(x, c, current_layer)``, ``BClassifier`` with ``feats_size`` /
``num_class`` attributes, batches B >= 1, C >= 1 classes.  Everything that is shape-identical to the binary model is
re-used from ``snuffy_amd.snuffy`` (same state-dict keys); only the selection differs (snuffy_multiclass.py:130-171):

  per batch row: top ceil(Lambda(1-r)) indices PER CLASS (descending score), flattened row-major over (rank, class),
  torch.unique (ascending); ref_dim = min over rows of the unique count, then min(ref_dim, N - ref_dim); keep the LOWEST
  ref_dim unique indices; draw ref_dim random indices from the complement of ALL uniques of that row (np.random.choice
  on the global numpy RNG, row after row); K = 2 * ref_dim.

The per-class top-k runs on the exact HIP selector (strided column of c), the layer math on the same fused kernels as
the binary model, one bag row at a time.  Training goes through the same autograd functions as the binary model (every
parameter gradient checked against autograd through the CPU oracle, tests/test_gpu_train.py).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import functional as SF
from . import ops
from .snuffy import (FCLayer, IClassifier, MILNet, MultiHeadedAttention, PositionwiseFeedForward,  # noqa: F401
                     RuntimeConfig, SublayerConnection, _share_config, attention, clones)

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class EncoderLayer(nn.Module):
    "Per-class top-Lambda + equal-size random set, sparse attention on the selected rows, feed forward."

    def __init__(self, size, self_attn, feed_forward, num_class, dropout, big_lambda, random_patch_share):
        super(EncoderLayer, self).__init__()
        self.self_attn = self_attn
        self.feed_forward = feed_forward
        self.sublayer = clones(SublayerConnection(size, dropout), 2)
        self.size = size
        self.big_lambda = big_lambda
        self.random_patch_share = random_patch_share
        self.top_big_lambda_share = 1.0 - random_patch_share
        self.num_classes = num_class
        self.last_selection = None
        self.cfg = RuntimeConfig()
        _share_config(self, self.cfg)

    def select(self, c):
        """c [B, N, C] on the GPU -> (topk [B, ref_dim] int64, rnd [B, ref_dim] int64) as the reference builds them."""
        b, n, ncls = c.shape
        k1 = min(math.ceil(self.big_lambda * self.top_big_lambda_share), n)
        uniq = []
        for i in range(b):
            cols = [ops.topk(c[i, :, cc], k1) for cc in range(ncls)]            # exact selector on a strided column
            flat = torch.stack(cols, dim=1).reshape(-1)                           # (rank, class) row-major = reference
            uniq.append(torch.unique(flat))                                       # ascending
        ref_dim = min(int(u.numel()) for u in uniq)
        ref_dim = min(ref_dim, n - ref_dim)
        topk = torch.stack([u[:ref_dim] for u in uniq]) if ref_dim > 0 else \
            torch.zeros(b, 0, dtype=torch.int64, device=c.device)
        rnd = torch.zeros(b, ref_dim, dtype=torch.int64)
        for i in range(b):
            mask = np.ones(n, dtype=bool)
            mask[uniq[i].cpu().numpy()] = False                                   # device->host, as .tolist() in the reference
            remaining = np.nonzero(mask)[0]
            rnd[i] = torch.from_numpy(np.random.choice(remaining, ref_dim, replace=False).astype(np.int64))
        return topk, rnd.to(c.device)

    def run(self, x, c, need_attn=True):
        """x [B, N, D] -> (list of Parts per row, A [B, h, N, K] or None)."""
        topk, rnd = self.select(c)
        self.last_selection = (topk, rnd)
        sel = torch.cat((topk, rnd), dim=1)
        parts, attns = [], []
        for i in range(x.shape[0]):
            p, a = SF.encoder_layer(x[i].contiguous(), sel[i].contiguous(), self, need_attn, self.cfg.compute)
            parts.append(p)
            attns.append(a)
        attn = torch.cat(attns, dim=0) if need_attn else None
        return parts, attn

    def forward(self, x, c, current_layer):
        xb = _as_batch(x)
        parts, attn = self.run(xb, c, self.cfg.return_attention)
        return torch.stack([SF.materialize(p) for p in parts]), attn


def _as_batch(x):
    if x.dim() != 3:
        raise ValueError("expected [B, N, D], got %s" % (tuple(x.shape),))
    if not x.is_cuda:
        from ._ffi import SnuffyHipError
        raise SnuffyHipError("input must be a GPU tensor: snuffy_amd has no CPU fallback")
    return x.float().contiguous()


class Encoder(nn.Module):
    "Stack of N layers followed by LayerNorm.  Reference snuffy_multiclass.py:75-89."

    def __init__(self, layer, N):
        super(Encoder, self).__init__()
        self.layers = clones(layer, N)
        self.norm = nn.LayerNorm(layer.size)
        self.cfg = RuntimeConfig().bind_stack(self.layers)
        _share_config(self, self.cfg)

    def run_layers(self, x, c):
        parts, attn = None, None
        n_layers = len(self.layers)
        for li, layer in enumerate(self.layers):
            if parts is not None:
                x = torch.stack([SF.materialize(p) for p in parts])
            parts, attn = layer.run(x, c, need_attn=(li == n_layers - 1) and self.cfg.return_attention)
        return parts, attn

    def forward(self, x, c):
        parts, attn = self.run_layers(_as_batch(x), c)
        z = torch.stack([SF.layer_norm(SF.materialize(p), self.norm) for p in parts])
        return z, attn


class BClassifier(nn.Module):
    "Reference snuffy_multiclass.py:60-72."

    def __init__(self, encoder, num_classes, input_size: int):
        super(BClassifier, self).__init__()
        self.encoder = encoder
        self.linear = nn.Linear(input_size, num_classes)
        self.feats_size = input_size
        self.num_class = num_classes
        self.cfg = RuntimeConfig().bind_stack(getattr(encoder, "layers", None))
        _share_config(self, self.cfg)

    def configure(self, precision=None, return_attention=None):
        self.cfg.set(precision, return_attention)
        return self

    def forward(self, x, c):
        "x [B, N, D], c [B, N, C] -> (logits [B, C], A [B, h, N, K])"
        parts, attn = self.encoder.run_layers(_as_batch(x), c.float())
        logits = torch.stack([SF.head(p, self.encoder.norm, self.linear) for p in parts])
        return logits, attn
