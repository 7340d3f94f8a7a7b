"""MI355X-native drop-in for the reference's ``snuffy.py`` module API (binary / single-logit Snuffy MIL model).

Same class names, constructor signatures, forward signatures, return tuples and state-dict keys as the reference
(reference snuffy.py:34-238), so ``train.py`` / ``roi.py`` style callers switch by changing the import.
Underneath, every op on the hot path is a hand-written HIP kernel reached through the C ABI of
``include/snuffy_hip.h``; the dense projections run on the hand-written MFMA GEMM (``snf_gemm_bf16``: every fp32-class
projection, small bags) or the library GEMM where that is faster (the three config-B-sized bf16 projections).

Differences that are deliberate and documented:
  * top-Lambda tie order is defined (descending score, ascending index) where torch.sort leaves it unspecified;
  * ``BClassifier`` / ``EncoderLayer`` run a fused pipeline (no x.clone(), LayerNorm reads the K patched rows in place,
    final LayerNorm + mean + head in one pass); results equal the unfused composition;
  * ``configure(precision=..., return_attention=...)`` selects fp32 (reference-class numerics) or bf16-MFMA arithmetic,
    and lets a caller that discards ``A`` (train.py:830 does) skip materialising the [1,h,N,K] tensor;
  * inputs must live on the GPU: there is no CPU fallback.
"""
import copy
import math

import numpy as np
import torch
import torch.nn as nn

from . import functional as SF
from . import ops

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")  # reference snuffy.py:31


class RuntimeConfig:
    """Execution knobs shared by the modules of one MILNet (not part of the state dict)."""

    def __init__(self, precision="fp32", return_attention=True):
        self.stack = None
        self.sampler = "reference"
        self._device_sampler = None
        self._draws = 0
        self.set(precision, return_attention)

    def set_sampler(self, sampler):
        """Who draws the random patch share (snuffy.py:136-147): "reference" (default) = np.random.choice on the host, the
        reference's own MT19937 stream, bit-exact parity (a device -> host copy per layer, no graph capture); "device" = Philox keys
        + top-k on the GPU (csrc/sampler.hip): the same distribution from another stream, no host sync, graph-capturable."""
        if sampler not in ("reference", "device"):
            raise ValueError("sampler must be 'reference' or 'device', got %r" % (sampler,))
        self.sampler = sampler

    def device_sampler(self, device):
        if self._device_sampler is None or self._device_sampler.state.device != device:
            self._device_sampler = ops.DeviceSampler(device)
        return self._device_sampler

    def bind_stack(self, layers):
        """The encoder stack this configuration drives (its depth decides `compute`)."""
        self.stack = layers
        return self

    @property
    def compute(self):
        """The arithmetic the kernels run in.  ``precision="bf16"`` is north_star's 1e-2 class, and it holds it for the benchmark
        model: ONE encoder layer.  Behind a stack of layers the last layer's K-way softmax amplifies every re-rounding of the
        activations before it (the reference's depth-5 fixture: |dA| = 0.10 in bf16, 1.5e-4 fp32-class), so stacks deeper than
        one layer -- roi.py's depth-5 model, reference roi.py:318-339 -- run the fp32-class kernels whatever `precision` says
        (SF.BF16_DEEP_STACKS = "bf16" restores the literal setting for experiments)."""
        if (self.precision == "bf16" and self.stack is not None and len(self.stack) > 1 and SF.BF16_DEEP_STACKS != "bf16"):
            if not getattr(self, "_warned_deep_bf16", False):
                import warnings
                warnings.warn("snuffy_amd: precision='bf16' on a stack of %d encoder layers runs the fp32-class kernels (the 1e-2 class "
                              "does not survive a stack of softmaxes; RuntimeConfig.compute == 'fp32').  Timings and errors of this "
                              "model are fp32-class figures." % len(self.stack), RuntimeWarning, stacklevel=3)
                self._warned_deep_bf16 = True
            return "fp32"
        return self.precision

    def set(self, precision=None, return_attention=None):
        if precision is not None:
            if precision not in ("fp32", "bf16"):
                raise ValueError("precision must be 'fp32' or 'bf16', got %r" % (precision,))
            self.precision = precision
        if return_attention is not None:
            self.return_attention = bool(return_attention)


def _share_config(module, cfg):
    for m in module.modules():
        if hasattr(m, "cfg"):
            m.cfg = cfg


class FCLayer(nn.Module):
    """Critic: one Linear(in_size, out_size) over every patch.  Reference snuffy.py:34-41."""

    def __init__(self, in_size, out_size=1):
        super(FCLayer, self).__init__()
        self.fc = nn.Sequential(nn.Linear(in_size, out_size))

    def forward(self, feats):
        lin = self.fc[0]
        x = SF.critic_scores(feats, lin.weight, lin.bias)
        return feats, x


class IClassifier(nn.Module):
    """feature_extractor + Linear head, returns (feats, c).  Reference snuffy.py:44-54."""

    def __init__(self, feature_extractor, feature_size, output_class):
        super(IClassifier, self).__init__()
        self.feature_extractor = feature_extractor
        self.fc = nn.Linear(feature_size, output_class)

    def forward(self, x):
        feats = self.feature_extractor(x)
        feats = feats.view(feats.shape[0], -1)
        c = SF.critic_scores(feats, self.fc.weight, self.fc.bias)
        return feats, c


def clones(module, N):
    "Produce N identical layers (independent deep copies).  Reference snuffy.py:57-59."
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


class BClassifier(nn.Module):
    """Encoder stack + mean-pool + Linear.  Reference snuffy.py:62-71."""

    def __init__(self, encoder, num_classes, input_size: int):
        super(BClassifier, self).__init__()
        self.encoder = encoder
        self.linear = nn.Linear(input_size, num_classes)
        self.cfg = RuntimeConfig().bind_stack(getattr(encoder, "layers", None))
        _share_config(self, self.cfg)

    def configure(self, precision=None, return_attention=None):
        self.cfg.set(precision, return_attention)
        return self

    def forward(self, x, c):
        """x [1, N, D], c [1, N, 1] -> (logits [1, C], A [1, h, N, K])."""
        x2, c1 = SF.check_bag(x, c)
        z_parts, attn = self.encoder.run_layers(x2, c1)
        logits = SF.head(z_parts, self.encoder.norm, self.linear)
        return logits.view(1, -1), attn


class Encoder(nn.Module):
    "Core encoder is a stack of N layers followed by LayerNorm.  Reference snuffy.py:74-86."

    def __init__(self, layer, N):
        super(Encoder, self).__init__()
        self.layers = clones(layer, N)
        self.norm = nn.LayerNorm(layer.size)
        self.cfg = RuntimeConfig().bind_stack(self.layers)
        _share_config(self, self.cfg)

    def run_layers(self, x2, c1):
        """Fused layer stack on x2 [N, D], c1 [N].  Returns (pending-z description of the last layer, A)."""
        top = None
        attn = None
        parts = None
        n_layers = len(self.layers)
        if self.cfg.sampler == "device" and any(l.random_patch_share > 0 for l in self.layers):
            self.cfg.device_sampler(x2.device).advance()     # a fresh Philox offset per forward (captured: per replay)
        for li, layer in enumerate(self.layers):
            if parts is not None:
                x2 = SF.materialize(parts)
            if top is None:
                top = SF.select_top(c1, layer.big_lambda, layer.top_big_lambda_share, x2.shape[0])
            parts, attn = layer.run(x2, c1, top, need_attn=(li == n_layers - 1) and self.cfg.return_attention,
                                    last=li == n_layers - 1, layer_index=li)
        return parts, attn

    def forward(self, x, c):
        x2, c1 = SF.check_bag(x, c)
        parts, attn = self.run_layers(x2, c1)
        z = SF.materialize(parts)
        zn = SF.layer_norm(z, self.norm)
        return zn.unsqueeze(0), attn


class SublayerConnection(nn.Module):
    """Residual connection with the norm first.  Reference snuffy.py:89-110."""

    def __init__(self, size, dropout):
        super(SublayerConnection, self).__init__()
        self.norm = nn.LayerNorm(size)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, sublayer, c, top_big_lambda_indices, random_indices, mode):
        if mode == 'attn':
            x2 = SF.as_2d(x)
            idx = top_big_lambda_indices if random_indices is None else torch.cat(
                (top_big_lambda_indices, random_indices))
            top_big_lambda = SF.gather(x2, idx).unsqueeze(0)
            multiheadedattn = sublayer(SF.layer_norm(x2, self.norm).unsqueeze(0))
            return top_big_lambda + self.dropout(multiheadedattn[0]), multiheadedattn[1]
        elif mode == 'ff':
            x2 = SF.as_2d(x)
            return x + self.dropout(sublayer(SF.layer_norm(x2, self.norm).unsqueeze(0)))


class EncoderLayer(nn.Module):
    "Top-Lambda (+random) selection, sparse self-attention on the selected rows, feed forward.  snuffy.py:113-157."

    def __init__(self, size, self_attn, feed_forward, dropout, big_lambda, random_patch_share):
        super(EncoderLayer, self).__init__()
        self.self_attn = self_attn
        self.feed_forward = feed_forward
        self.sublayer = clones(SublayerConnection(size, dropout), 2)
        self.size = size
        self.big_lambda = big_lambda
        self.random_patch_share = random_patch_share
        self.top_big_lambda_share = 1.0 - random_patch_share
        self.last_selection = None
        self.cfg = RuntimeConfig()
        _share_config(self, self.cfg)

    def select(self, c1, n, top=None, layer_index=0):
        """Selected row indices S = top ++ random (snuffy.py:128-147). The random part follows the reference exactly:
        np.random.choice on the global numpy RNG over the ascending complement of `top`."""
        if top is None:
            top = SF.select_top(c1, self.big_lambda, self.top_big_lambda_share, n)
        k2 = min(int(self.big_lambda * self.random_patch_share),
                 max(0, n - math.ceil(self.big_lambda * self.top_big_lambda_share)))
        if k2 == 0:
            return top, None
        if self.cfg.sampler == "device":                     # opt-in fast mode: no host round trip, graph-capturable
            return top, self.cfg.device_sampler(top.device).draw(n, k2, top, layer=layer_index)
        mask = np.ones(n, dtype=bool)
        mask[top.cpu().numpy()] = False                      # device->host sync, as .tolist() in snuffy.py:136
        remaining = np.nonzero(mask)[0]
        rnd = np.random.choice(remaining, k2, replace=False)
        return top, torch.from_numpy(rnd.astype(np.int64)).to(top.device)

    def run(self, x2, c1, top=None, need_attn=True, last=False, layer_index=0):
        top, rnd = self.select(c1, x2.shape[0], top, layer_index)
        self.last_selection = (top, rnd)                    # inspection hook (tests / heat-maps)
        sel = top if rnd is None else torch.cat((top, rnd))
        return SF.encoder_layer(x2, sel, self, need_attn, self.cfg.compute, last=last)

    def forward(self, x, c):
        "x [1, N, D], c [1, N, 1] -> (z [1, N, D], A [1, h, N, K])"
        x2, c1 = SF.check_bag(x, c)
        if self.cfg.sampler == "device" and self.random_patch_share > 0:
            self.cfg.device_sampler(x2.device).advance()     # a layer called on its own (not through Encoder.run_layers): fresh rows per call
        parts, attn = self.run(x2, c1, None, self.cfg.return_attention)
        return SF.materialize(parts).unsqueeze(0), attn


def attention(query, key, value, dropout=None):
    """'Scaled Dot Product Attention' with the transposed pooling of the reference (snuffy.py:160-168).

    query, value [1, h, N, dk]; key [1, h, K, dk] -> (p_attn^T value [1, h, K, dk], p_attn [1, h, N, K])."""
    return SF.attention_4d(query, key, value, dropout)


class MultiHeadedAttention(nn.Module):
    "Reference snuffy.py:171-205."

    def __init__(self, h, d_model, dropout=0.1):
        super(MultiHeadedAttention, self).__init__()
        assert d_model % h == 0
        self.d_big_lambda = d_model // h
        self.h = h
        self.linears = clones(nn.Linear(d_model, d_model), 4)
        self.attn = None
        self.dropout = nn.Dropout(p=dropout)
        self.cfg = RuntimeConfig()

    def forward(self, query, key, value):
        "query = value = LN(x) [1, N, D]; key = selected raw rows [1, K, D] -> (out [1, K, D], P [1, h, N, K])"
        out, self.attn = SF.mha_forward(self, SF.as_2d(query), SF.as_2d(key), SF.as_2d(value), True,
                                        self.cfg.precision)
        return out.unsqueeze(0), self.attn


class PositionwiseFeedForward(nn.Module):
    "FFN: w_2(dropout(activation(w_1(x)))).  Reference snuffy.py:208-225."

    def __init__(self, d_model, d_ff, activation, dropout=0.1):
        super(PositionwiseFeedForward, self).__init__()
        self.w_1 = nn.Linear(d_model, d_ff)
        self.w_2 = nn.Linear(d_ff, d_model)
        self.dropout = nn.Dropout(dropout)
        if activation not in SF.ACTIVATIONS:
            raise KeyError(activation)                      # same failure as the reference's dictionary lookup
        self.activation_name = activation
        self.cfg = RuntimeConfig()

    def forward(self, x):
        return SF.ffn_forward(self, SF.as_2d(x), self.cfg.precision).view(x.shape)


class MILNet(nn.Module):
    "Reference snuffy.py:228-238."

    def __init__(self, i_classifier, b_classifier):
        super(MILNet, self).__init__()
        self.i_classifier = i_classifier
        self.b_classifier = b_classifier

    def configure(self, precision=None, return_attention=None, graph_max_patches=None, sampler=None):
        """graph_max_patches (opt-in, default off): inference forwards of bags with at most that many patches are captured
        into HIP graphs and replayed -- a bag is ~25 kernel launches, ~0.24 ms of host-side issue, more than the GPU time of
        a bag of <= 8k patches and, on a slow host, of larger ones.  Same kernels, bit-identical results.  Small bags
        (<= 32 MB) are captured once per shape behind a static input buffer; larger ones are bound to their own buffer the
        second time the same tensor comes in (a dataset resident in HBM), so nothing is copied.  Only for a selection that
        stays on the device (random_patch_share == 0, or sampler="device"), outside autograd.
        sampler: "reference" (default: the random patch share is np.random.choice on the host, the reference's MT19937 draws bit
        for bit) or "device" (opt-in fast mode: Philox keys + top-k on the GPU, same distribution, no host sync; a captured graph
        draws fresh rows on every replay).  "device" applies to the one-bag-per-forward path; forward_bags (packed.py) draws the
        random share of a packed batch with the reference's host draws whatever this says, and is not graph-captured then."""
        self.b_classifier.configure(precision, return_attention)
        if sampler is not None:
            self.b_classifier.cfg.set_sampler(sampler)       # RuntimeConfig.set_sampler: "reference" (parity, default) | "device"
        if graph_max_patches is not None:
            self._graph_max_patches = int(graph_max_patches)
            self._graphs, self._graph_seen, self._graph_pool = {}, set(), None
        return self

    def forward(self, x):
        if self._graph_ok(x):
            return self._forward_graph(x)
        return self._forward_eager(x)

    def _forward_eager(self, x):
        for layer in self.b_classifier.encoder.layers[:1]:
            layer._xhat_offer = None        # a normalised copy left by an earlier forward is never valid for this bag
            layer._xn3_offer = None
        feats, classes = self._critic(x)
        prediction_bag, A = self.b_classifier(feats, classes)
        return classes, prediction_bag, A

    _GRAPH_SHAPES = 1024          # captured graphs kept per model (oldest dropped first); they share one memory pool
    _GRAPH_COPY_BYTES = 32 << 20  # bags up to this size go through a static input buffer (one copy per forward)

    def _graph_ok(self, x):
        lim = getattr(self, "_graph_max_patches", 0)
        if lim <= 0 or self.training or torch.is_grad_enabled():
            return False
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 3 and x.shape[0] == 1 and x.dtype == torch.float32
                and x.is_contiguous() and 0 < x.shape[1] <= lim and type(self.i_classifier) is FCLayer):
            return False
        if self.b_classifier.cfg.return_attention and x.numel() * 4 > self._GRAPH_COPY_BYTES:
            return False   # the [1, h, N, K] attention tensor would be cloned out of the graph's pool on every forward
        # only the binary model's selection can be captured, and only when it stays on the device: the deterministic selection, or
        # the random share drawn by the device sampler (the multiclass layer and the reference's numpy draws synchronise with the host)
        on_device = self.b_classifier.cfg.sampler == "device"
        return all(type(l) is EncoderLayer and (l.random_patch_share == 0 or on_device) for l in self.b_classifier.encoder.layers)

    def _weights_signature(self):
        """Changes whenever a parameter is written in place (optimizer step -- fused ones included, SF.param_key --, load_state_dict), replaced (.to(), .half(),
        load_state_dict(assign=True), module.weight = ...) or a selection knob of a layer is reassigned.  Not seen: edits
        through ``p.data`` (they do not bump the version counter) -- call ``invalidate()`` after those."""
        plist = list(self.parameters())
        knobs = tuple((l.big_lambda, l.random_patch_share) for l in self.b_classifier.encoder.layers)
        # module-level arithmetic switches are baked into a capture as well
        knobs += (SF.FP32_GEMM, SF.FP32_ATTENTION, SF.X3_HL_ATTENTION, SF.X3_HL_KPFRAG, SF.X3_HL_KPFRAG_GATHER, SF.FP32_SHARED_NORM, ops.GEMM_HL_SPLITK,
                  SF.BF16_DEEP_STACKS, self.b_classifier.cfg.sampler)
        return tuple(SF.param_key(p) for p in plist), knobs

    def invalidate(self):
        """Forget everything derived from the current weights: captured graphs and the folded bf16 weights of every layer."""
        if getattr(self, "_graph_max_patches", 0):
            self._graphs.clear()
            self._graph_seen.clear()
            self._graph_sig = None
        for layer in self.b_classifier.encoder.layers:
            SF.invalidate_folded(layer)
        SF.drop_param_caches(self)        # split images cached on the parameters themselves (fp32-class training / key projections)
        return self

    _GRAPH_STATE = ("_graphs", "_graph_seen", "_graph_pool", "_graph_sig", "_packed_bags", "_capture_stream")

    def __deepcopy__(self, memo):
        """Captured HIP graphs belong to this instance's buffers: a copy starts without them."""
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in self._GRAPH_STATE:
                new.__dict__[k] = copy.deepcopy(v, memo)
        if getattr(self, "_graph_max_patches", 0):
            new._graphs, new._graph_seen, new._graph_pool = {}, set(), None
        # the copy draws its own rows: a cloned {seed, offset} record would make both models select identical random shares
        new.b_classifier.cfg._device_sampler = None
        return new

    def _forward_graph(self, x):
        # a captured graph has the folded bf16 weights of its capture baked in: new weights -> new graphs
        sig = self._weights_signature()
        if sig != getattr(self, "_graph_sig", None):
            self._graphs.clear()
            self._graph_seen.clear()
            self._graph_sig = sig
        cfg = self.b_classifier.cfg
        small = x.numel() * 4 <= self._GRAPH_COPY_BYTES
        key = (tuple(x.shape), x.device, cfg.precision, cfg.return_attention) + (() if small else (x.data_ptr(),))
        ent = self._graphs.get(key)
        if ent is None:
            if not small and key not in self._graph_seen:   # bind a graph to a large buffer only once it comes back
                if len(self._graph_seen) > 8192:
                    self._graph_seen.clear()
                self._graph_seen.add(key)
                return self._forward_eager(x)
            try:
                static_x = torch.empty_like(x).copy_(x) if small else x
                cur = torch.cuda.current_stream()
                side = getattr(self, "_capture_stream", None)
                if side is None or side.device != x.device:
                    side = self._capture_stream = torch.cuda.Stream(x.device)   # warm-up AND capture run on this stream
                side.wait_stream(cur)
                with torch.cuda.stream(side):   # warm-up off the capture: kernel attributes, library handles, weight folds
                    sel_state = ops.fresh_selector(x.device)   # this graph's own selector state (its address is baked in)
                    for _ in range(2):
                        self._forward_eager(static_x)
                cur.wait_stream(side)
                if self._graph_pool is None:
                    self._graph_pool = torch.cuda.graph_pool_handle()
                graph = torch.cuda.CUDAGraph()
                graph._snf_selector = sel_state
                # thread_local: a helper thread of the process (e.g. the RCCL watchdog) may touch the runtime meanwhile
                with torch.cuda.graph(graph, pool=self._graph_pool, stream=side, capture_error_mode="thread_local"):
                    out = self._forward_eager(static_x)
            except Exception as exc:   # capture not possible here: stay on the eager path for good, and say so once
                import warnings
                warnings.warn("snuffy_amd: HIP-graph capture of the inference forward failed (%s: %s); graph replay is "
                              "disabled for this model, every kernel is issued from Python" % (type(exc).__name__, exc),
                              RuntimeWarning, stacklevel=2)
                self._graph_max_patches = 0
                torch.cuda.synchronize()
                return self._forward_eager(x)
            if len(self._graphs) >= self._GRAPH_SHAPES:
                self._graphs.pop(next(iter(self._graphs)))
            ent = self._graphs[key] = (graph, static_x, out)
        graph, static_x, out = ent
        if small:
            static_x.copy_(x)
        graph.replay()
        # the graphs share one pool: the outputs are only valid until the next replay, so hand out copies
        return tuple(o.clone() if isinstance(o, torch.Tensor) else o for o in out)

    # ---- many bags per launch (varlen path, implementation in packed.py) -------------------------------------------------
    def forward_bags(self, bags):
        """``[self(x) for x in bags]`` with the bags' rows packed into ONE set of launches (bags of <= 8 k patches are bound by the
        ~25 launches per bag, not by the GPU).  Same selections as the per-bag forwards bit for bit (random share included), logits
        within fp32 / bf16 rounding of the per-bag forward (and of another batch composition: the projections choose their kernel by
        the packed row count; top-k, attention and head kernels are composition-independent bit for bit); whatever cannot be
        packed takes the per-bag loop.  See packed.forward_bags."""
        from . import packed
        return packed.forward_bags(self, bags)

    def forward_packed(self, x_cat, packed_bags, ragged=False):
        """forward_bags() on rows that are already packed: x_cat [T, D] fp32, packed_bags = ops.PackedBags(sizes, device)."""
        from . import packed
        return packed.forward_packed(self, x_cat, packed_bags, ragged)

    def _pack_groups(self, bags):
        from . import packed
        return packed.pack_groups(self, list(bags))

    def _packable(self, bags):
        from . import packed
        return packed.packable(self, bags)

    def _critic(self, x):
        """i_classifier(x); in the bf16 inference path of a plain FCLayer critic the same pass over the bag also produces
        the normalised input of the first encoder layer (snf_critic_ln_f32) -- identical values, one HBM read less."""
        ic = self.i_classifier
        cfg = getattr(self.b_classifier, "cfg", None)
        if (type(ic) is FCLayer and cfg is not None and cfg.compute == "bf16" and not torch.is_grad_enabled()
                and isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 3 and x.shape[0] == 1
                and len(self.b_classifier.encoder.layers) > 0):
            lin = ic.fc[0]
            eps = self.b_classifier.encoder.layers[0].sublayer[0].norm.eps
            return x, SF.critic_scores_with_xhat(x, lin.weight, lin.bias, eps, self.b_classifier.encoder.layers[0])
        if (type(ic) is FCLayer and cfg is not None and cfg.compute == "fp32" and not torch.is_grad_enabled()
                and isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 3 and x.shape[0] == 1
                and len(self.b_classifier.encoder.layers) > 0 and type(self.b_classifier.encoder.layers[0]) is EncoderLayer):
            # fp32-class inference on large bags: the critic pass also emits LayerNorm_0(x) as the image of the Q|V projection
            lin = ic.fc[0]
            return x, SF.critic_scores_with_hl(x, lin.weight, lin.bias, self.b_classifier.encoder.layers[0])
        if (type(ic) is FCLayer and cfg is not None and cfg.compute == "bf16" and torch.is_grad_enabled()
                and isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 3 and x.shape[0] == 1 and not x.requires_grad
                and x.dtype == torch.float32 and x.is_contiguous() and len(self.b_classifier.encoder.layers) > 0):
            # bf16 training: same one-pass critic, with autograd (the fused first-layer chain consumes the normalised copy)
            from . import autograd as SA
            lin = ic.fc[0]
            layer0 = self.b_classifier.encoder.layers[0]
            if SA.fused_layer0_shape_ok(layer0, x.shape[1], x.shape[2]):
                s = SA.critic_train(x[0], lin.weight, lin.bias, layer0, layer0.sublayer[0].norm.eps)
                return x, s.view(1, x.shape[1], -1)
        return ic(x)


def build_milnet(feats_size, num_heads, activation="relu", big_lambda=200, random_patch_share=0.0, depth=1, num_classes=1,
                 mlp_multiplier=4, encoder_dropout=0.0):
    """MILNet assembled in the order of the reference's train.Snuffy._get_milnet (train.py:861-890): critic FCLayer,
    MultiHeadedAttention (its dropout left at the default 0.1, as train.py:866-869 does), PositionwiseFeedForward,
    `depth` deep-copied EncoderLayers, BClassifier.  Weights are whatever the constructors leave (callers initialise)."""
    i_classifier = FCLayer(in_size=feats_size, out_size=num_classes)
    attn = MultiHeadedAttention(num_heads, feats_size)
    ff = PositionwiseFeedForward(feats_size, feats_size * mlp_multiplier, activation, encoder_dropout)
    layer = EncoderLayer(feats_size, copy.deepcopy(attn), copy.deepcopy(ff), encoder_dropout, big_lambda, random_patch_share)
    return MILNet(i_classifier, BClassifier(Encoder(layer, depth), num_classes, feats_size))
