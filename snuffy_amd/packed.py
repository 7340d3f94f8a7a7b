"""Many bags per launch: the packed ("varlen") inference forward behind ``MILNet.forward_bags`` /
``forward_packed``.  The reference has no counterpart -- it runs one bag per forward (train.py:468-473 batch_size 1, snuffy.py:130-131
indexes with a 1-D tensor); this is the same per-bag arithmetic over rows packed into one tensor (DESIGN.md section 4 "Varlen path").
Functions take the MILNet as their first argument; ``snuffy.MILNet`` exposes them as methods."""
import math

import numpy as np
import torch

from . import functional as SF
from . import snuffy

RAGGED_MAX_ROWS = 4096      # longest bag the ragged (exact fp32, one workgroup per bag and head) attention is meant for


def pack_groups(net, bags):
    """How forward_bags() packs `bags`: a list of (bag indices, ragged flag) groups, or None (nothing can be packed).

    The packed path covers inference of the binary model with a plain one-logit FCLayer critic.  "uniform" groups (ragged =
    False): every bag has at least Lambda patches, so all select the same K rows, and the head width is one the MFMA kernels
    take -- varlen forms of the bf16 / fp32-class attention kernels.  "ragged" groups: bags shorter than Lambda (they select
    ALL their rows, snuffy.py:129) or head widths outside the MFMA kernels (the MIL benchmark sets: D = 166 / 230, h = 2) --
    exact-fp32 ragged attention, bags of at most _RAGGED_MAX_ROWS patches.  With a random share the draws of the reference
    must stay in bag order, so only one uniform group over all bags is formed."""
    cfg = net.b_classifier.cfg
    layers = list(net.b_classifier.encoder.layers)
    if (len(bags) < 2 or net.training or torch.is_grad_enabled() or type(net.i_classifier) is not snuffy.FCLayer or not layers
            or not all(type(l) is snuffy.EncoderLayer for l in layers) or cfg.precision not in ("fp32", "bf16")):
        return None
    if net.i_classifier.fc[0].weight.shape[0] != 1:
        return None
    x0 = bags[0]
    for x in bags:
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.device == x0.device
                and ((x.dim() == 3 and x.shape[0] == 1) or x.dim() == 2) and x.shape[-1] == x0.shape[-1]
                and x.shape[-2] >= 1):
            return None
    d = x0.shape[-1]
    h = layers[0].self_attn.h
    if d % h:
        return None
    kb = None
    for l in layers:
        k1 = math.ceil(l.big_lambda * l.top_big_lambda_share)
        k2 = int(l.big_lambda * l.random_patch_share)
        if k1 < 1 or (kb is not None and (k1, k2) != kb) or l.self_attn.h != h:
            return None
        kb = (k1, k2)
    k1, k2 = kb
    if cfg.compute == "bf16" and any(l.sublayer[0].norm.eps != l.sublayer[1].norm.eps for l in layers):
        return None
    sizes = [x.shape[-2] for x in bags]
    uniform_dims = (d % 4 == 0 and SF.ops.varlen_attn_supported(cfg.compute, k1 + k2, d // h)
                    and (cfg.compute == "bf16" or SF.FP32_ATTENTION == "x3"))
    if k2 > 0:
        ok = (uniform_dims and min(sizes) >= k1 + k2 and max(sizes) <= 65536 and (k1 + k2) * len(bags) <= (1 << 20)
              and sum(sizes) <= getattr(net, "_PACK_MAX_ROWS", PACK_MAX_ROWS))      # one group only: the draws must stay in bag order
        return [(list(range(len(bags))), False)] if ok else None
    uni = [i for i, n in enumerate(sizes) if uniform_dims and k1 <= n <= 65536]
    rest = [i for i in range(len(bags)) if i not in set(uni)]
    rag = [i for i in rest if sizes[i] <= getattr(net, "_RAGGED_MAX_ROWS", RAGGED_MAX_ROWS)]
    if rag and not SF.ops.ragged_attn_supported(min(k1, max(sizes[i] for i in rag)), d // h):
        rag = []
    groups = []
    for g, r in ((uni, False), (rag, True)):
        groups += [(c, r) for c in chunk_rows(net, g, sizes) if len(c) >= 2]
    return groups or None


PACK_MAX_ROWS = 196608      # rows of one packed launch set: the GEMMs address their [T, 3F] images with 32-bit element offsets
PACKED_GRAPHS = 32          # captured batch compositions kept per model (oldest dropped first)


def chunk_rows(net, idx, sizes):
    """Split a group into consecutive chunks of at most _PACK_MAX_ROWS packed rows."""
    chunks, cur, rows = [], [], 0
    for i in idx:
        if cur and rows + sizes[i] > getattr(net, "_PACK_MAX_ROWS", PACK_MAX_ROWS):
            chunks.append(cur)
            cur, rows = [], 0
        cur.append(i)
        rows += sizes[i]
    if cur:
        chunks.append(cur)
    return chunks


def packable(net, bags):
    """True when forward_bags() runs ALL of `bags` as one uniform packed batch."""
    groups = pack_groups(net, list(bags))
    return bool(groups) and len(groups) == 1 and not groups[0][1] and len(groups[0][0]) == len(bags)


def forward_bags(net, bags):
    """``[net(x) for x in bags]`` with the bags' rows packed into ONE set of launches (bags of <= 8 k patches
    are launch-latency bound -- ~25 launches per bag whatever its size).  bags: sequence of [1, N_b, D] (or [N_b, D]) fp32 GPU
    tensors.  Returns the list of (classes [1, N_b, 1], prediction_bag [1, C], A [1, h, N_b, K_b] or None) tuples the per-bag
    forwards return: same selections (bit-exact, random share included: the numpy draws are made bag by bag in the order the
    per-bag forwards make them).  At kernel level (top-k, attention, head) a bag's result does not depend on what it is packed
    with, bit for bit; the projections pick their kernel by the PACKED row count, so a bag's logits move by fp32 / bf16 rounding
    with the batch composition (metrics computed from packed evaluation carry that rounding).  Against the per-bag forwards the
    top-k and head kernels are bit-identical, the attention sums its partial tiles in another order, and the projections
    run over the packed rows (a library / tile choice that depends on the row count): logits move by fp32 / bf16 rounding.
    Bags that select different numbers of rows (shorter than Lambda) or whose head width the MFMA kernels do not take are
    packed as a second, "ragged" group (pack_groups); whatever cannot be packed (training, multiclass critic, ...) takes
    the per-bag loop."""
    bags = list(bags)
    groups = pack_groups(net, bags)
    if not groups:
        return [net(x) for x in bags]
    out = [None] * len(bags)
    lim = getattr(net, "_graph_max_patches", 0)
    graph_ok = lim > 0 and all(l.random_patch_share == 0 for l in net.b_classifier.encoder.layers)
    for idx, ragged in groups:
        sizes = tuple(bags[i].shape[-2] for i in idx)
        rows = [bags[i].reshape(-1, bags[i].shape[-1]) for i in idx]
        if graph_ok and max(sizes) <= lim:
            res = forward_bags_graph(net, rows, sizes, ragged)
        else:
            packed = packed_cache(net, sizes, rows[0].device)
            res = split_packed(forward_packed_raw(net, torch.cat(rows), packed, ragged), packed)
        for i, r in zip(idx, res):
            out[i] = r
    for i, x in enumerate(bags):
        if out[i] is None:
            out[i] = net(x)
    return out


def packed_cache(net, sizes, device):
    """PackedBags (offsets + launch plans) of the most recent batch compositions."""
    cache = net.__dict__.setdefault("_packed_bags", {})
    key = (sizes, str(device))
    pk = cache.get(key)
    if pk is None:
        if len(cache) >= 64:
            cache.pop(next(iter(cache)))
        pk = cache[key] = SF.ops.PackedBags(sizes, device)
    return pk


def forward_bags_graph(net, rows, sizes, ragged=False):
    """forward_bags() as ONE captured HIP graph per batch composition (configure(graph_max_patches=...), deterministic
    selection only): the bags are copied into the graph's static packed buffer and the ~30 launches replay without the host."""
    sig = net._weights_signature()
    if sig != getattr(net, "_graph_sig", None):
        net._graphs.clear()
        net._graph_seen.clear()
        net._graph_sig = sig
    cfg = net.b_classifier.cfg
    dev = rows[0].device
    key = ("bags", sizes, bool(ragged), dev, cfg.precision, cfg.return_attention)
    ent = net._graphs.get(key)
    if ent is None:
        packed = packed_cache(net, sizes, dev)
        if key not in net._graph_seen:
            # a composition is captured only when it comes back (a capture costs three extra forwards and pins a [T, D] buffer):
            # a loader that never repeats a batch stays on the eager path
            if len(net._graph_seen) > 8192:
                net._graph_seen.clear()
            net._graph_seen.add(key)
            return split_packed(forward_packed_raw(net, torch.cat(rows), packed, ragged), packed)
        static_x = torch.cat(rows)
        try:
            cur = torch.cuda.current_stream()
            side = getattr(net, "_capture_stream", None)
            if side is None or side.device != dev:
                side = net._capture_stream = torch.cuda.Stream(dev)      # warm-up AND capture run on this stream
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                sel_state = SF.ops.fresh_selector(dev)                   # this graph's own selector state
                for _ in range(2):
                    forward_packed_raw(net, static_x, packed, ragged)
            cur.wait_stream(side)
            if net._graph_pool is None:
                net._graph_pool = torch.cuda.graph_pool_handle()
            graph = torch.cuda.CUDAGraph()
            graph._snf_selector = sel_state
            with torch.cuda.graph(graph, pool=net._graph_pool, stream=side, capture_error_mode="thread_local"):
                out = forward_packed_raw(net, static_x, packed, ragged)
        except Exception as exc:
            import warnings
            warnings.warn("snuffy_amd: HIP-graph capture of the packed inference forward failed (%s: %s); graph replay is "
                          "disabled for this model" % (type(exc).__name__, exc), RuntimeWarning, stacklevel=2)
            net._graph_max_patches = 0
            torch.cuda.synchronize()
            return split_packed(forward_packed_raw(net, torch.cat(rows), packed, ragged), packed)
        if len(net._graphs) >= net._GRAPH_SHAPES:
            net._graphs.pop(next(iter(net._graphs)))
        held = [k for k in net._graphs if k and k[0] == "bags"]
        if len(held) >= PACKED_GRAPHS:                 # every packed graph pins its own packed input buffer
            net._graphs.pop(held[0])
        ent = net._graphs[key] = (graph, static_x, out, packed)
    graph, static_x, out, packed = ent
    torch.cat(rows, out=static_x)
    graph.replay()
    # the graphs share one pool: outputs are valid until the next replay -- three copies per batch, then per-bag views
    return split_packed(tuple(o.clone() if isinstance(o, torch.Tensor) else o for o in out), packed)


def forward_packed(net, x_cat, packed, ragged=False):
    """forward_bags() on rows that are already packed: x_cat [T, D] fp32, packed = ops.PackedBags(sizes, device) (keep it
    between calls with the same bag sizes: it caches the launch plans).  ragged: the bags select different numbers of rows
    (some are shorter than Lambda) or the head width is outside the MFMA kernels -- deterministic selection only."""
    return split_packed(forward_packed_raw(net, x_cat, packed, ragged), packed)


def split_packed(raw, packed):
    s, logits, attn, kbs = raw
    out = []
    for b, n in enumerate(packed.sizes):
        lo = int(packed.host[b])
        a_b = None
        if attn is not None:
            a_b = attn[:, :, lo:lo + n, :] if kbs is None else attn[:, :, lo:lo + n, :kbs[b]]
        out.append((s[lo:lo + n].view(1, n, -1), logits[b].view(1, -1), a_b))
    return out


def forward_packed_raw(net, x_cat, packed, ragged=False):
    """(critic scores [T, 1], logits [B, C], A [1, h, T, K] or None, per-bag key counts or None) over the packed rows."""
    enc = net.b_classifier.encoder
    cfg = net.b_classifier.cfg
    layers = list(enc.layers)
    lin = net.i_classifier.fc[0]
    x_cat = SF.as_2d(x_cat)
    for layer in layers[:1]:
        layer._xhat_offer = None
        layer._xn3_offer = None
    if cfg.compute == "bf16":
        eps = layers[0].sublayer[0].norm.eps
        s, xhat = SF.ops.critic_ln(x_cat, lin.weight, lin.bias, eps)            # same kernel as the per-bag critic pass
        layers[0]._xhat_offer = (x_cat.data_ptr(), tuple(x_cat.shape), x_cat._version, float(eps), xhat)
    else:
        # fp32-class: where the layer takes the one-pass GEMMs, the critic pass also leaves LayerNorm_0's image (one read of the rows)
        s = SF.critic_scores_with_hl(x_cat, lin.weight, lin.bias, layers[0])
    c1 = s.reshape(-1)
    l0 = layers[0]
    k1 = math.ceil(l0.big_lambda * l0.top_big_lambda_share)
    k2 = int(l0.big_lambda * l0.random_patch_share)
    top = SF.ops.topk_segmented(c1, packed, k1)                                   # [B, k1] inside each bag
    first = packed.dev[:-1].unsqueeze(1)
    rnd = rag = None
    if ragged:
        if k2 > 0:
            raise SF.SnuffyHipError("ragged packed bags support the deterministic selection only (random_patch_share == 0)")
        # a bag shorter than Lambda selects all of its rows (snuffy.py:129): per-bag key counts, selected rows concatenated
        rag = packed.ragged([min(k1, n) for n in packed.sizes])
        pos, base = rag.flat_index(k1)
        sel_ragged = top.reshape(-1)[pos] + base
    elif k2 > 0:
        # the reference's draws (snuffy.py:134-143), in the order the per-bag forwards consume the global numpy stream:
        # bag by bag, and inside a bag layer by layer (every layer draws from the complement of the same `top`)
        top_h = top.cpu().numpy()
        draws = np.empty((len(layers), packed.bags, k2), dtype=np.int64)
        for b, n in enumerate(packed.sizes):
            mask = np.ones(n, dtype=bool)
            mask[top_h[b]] = False
            remaining = np.nonzero(mask)[0]
            for li in range(len(layers)):
                draws[li, b] = np.random.choice(remaining, k2, replace=False)
        rnd = torch.from_numpy(draws).to(top.device)                             # [layers, B, k2]
    parts = attn = None
    x2 = x_cat
    for li, layer in enumerate(layers):
        if parts is not None:
            x2 = SF.materialize(parts)
        layer.last_selection = None
        layer.last_selection_bags = (top, None if rnd is None else rnd[li])     # ragged: entries >= K_b of a row are padding
        if ragged:
            sel = sel_ragged
        else:
            sel_local = top if rnd is None else torch.cat((top, rnd[li]), dim=1)  # [B, K]: top ++ random, as snuffy.py:145
            sel = (sel_local + first).reshape(-1)
        parts, attn = SF.encoder_layer(x2, sel, layer, (li == len(layers) - 1) and cfg.return_attention, cfg.compute,
                                       packed=packed, ragged=rag)
    logits = SF.head(parts, enc.norm, net.b_classifier.linear, packed=packed)   # [B, C]
    return s, logits, attn, (rag.kbs if rag is not None else None)

