"""CPU ORACLE for the Snuffy sparse-attention MIL aggregator -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file.
The product (``snuffy_amd``) never imports anything under ``oracle/``.

This is a restatement, written from the math, of the reference's algorithm (jafarinia/snuffy @ 2024-10-22):

  * FCLayer critic                         snuffy.py:34-41
  * top-Lambda (+ random) patch selection  snuffy.py:126-147
  * SublayerConnection('attn')             snuffy.py:100-108
  * MultiHeadedAttention / attention()     snuffy.py:160-205
  * scatter of the K updated rows          snuffy.py:152-155
  * SublayerConnection('ff') + FFN         snuffy.py:109-110, 208-225
  * Encoder final LayerNorm, mean-pool head snuffy.py:82-86, 68-71
  * SmallWeightTrainer._run_model loss     train.py:828-846, 913-916
  * dropout_patches                        utils.py:244-250
  * multiclass selection                   snuffy_multiclass.py:130-171

It runs the same op sequence as the reference on PyTorch-CPU (so that it is also a fair ``cpu_baseline``
"port"), functionally over a plain state-dict (reference key names).

PARITY PIN: checked against golden vectors produced by importing the unmodified reference in the build
container (tests/golden/make_golden.py -> tests/golden/f*.npz; tests/test_oracle_golden.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

ACTIVATIONS = {
    "relu": F.relu,
    "gelu": F.gelu,                                   # nn.GELU() default = erf form (snuffy.py:218)
    "leakyrelu": lambda t: F.leaky_relu(t, 0.01),     # nn.LeakyReLU() default slope (snuffy.py:219)
    "selu": F.selu,
}


def k_split(big_lambda, random_patch_share, n):
    """k1 (top share) and k2 (random share) exactly as python floats evaluate them (snuffy.py:124,129,137-140)."""
    top_share = 1.0 - random_patch_share
    k1 = min(math.ceil(big_lambda * top_share), n)
    k2 = min(int(big_lambda * random_patch_share), max(0, n - math.ceil(big_lambda * top_share)))
    return k1, k2


def topk_desc_stable(c, k):
    """Indices of the k largest scores; descending score, ties broken by ascending index.

    The reference calls torch.sort(c, 1, descending=True) (snuffy.py:128), which is not stable: its tie order is
    implementation-defined.  The build's rule is the stable order.
    """
    c = torch.as_tensor(c).reshape(-1)
    return torch.sort(c, dim=0, descending=True, stable=True)[1][:k].to(torch.int64)


def select_indices(c, big_lambda, random_patch_share, rng=np.random):
    """top-Lambda + random selection of one layer (snuffy.py:128-147). Consumes rng like np.random.choice does.

    np.random.choice(rem, k2, replace=False) == rem[rng.permutation(len(rem))[:k2]]  (SURVEY.md 8a-7).
    Returns (top[int64 k1], rnd[int64 k2] or None).
    """
    c = c.reshape(-1)
    n = c.numel()
    k1, k2 = k_split(big_lambda, random_patch_share, n)
    top = topk_desc_stable(c, k1)
    rnd = None
    if k2 != 0:
        mask = np.ones(n, dtype=bool)
        mask[top.numpy()] = False
        rem = np.nonzero(mask)[0]                       # ascending, == sorted(set(range(n)) - set(top))
        rnd = torch.from_numpy(rem[rng.permutation(len(rem))[:k2]].astype(np.int64))
    return top, rnd


def layer_norm(x, w, b, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def sparse_attention(q, kp, v, h):
    """attention() of snuffy.py:160-168 on already-projected tensors.

    q, v: [N, D]; kp: [K, D]. Returns (O [K, D] heads concatenated, P [h, N, K]).
    P_a = softmax_K(Q_a Kp_a^T / sqrt(dk));  O_a = P_a^T V_a.
    """
    n, d = q.shape
    k = kp.shape[0]
    dk = d // h
    qh = q.view(n, h, dk).transpose(0, 1)               # [h, N, dk]
    kh = kp.view(k, h, dk).transpose(0, 1)              # [h, K, dk]
    vh = v.view(n, h, dk).transpose(0, 1)
    scores = torch.matmul(qh, kh.transpose(-2, -1)) / math.sqrt(dk)
    p = scores.softmax(dim=-1)                           # [h, N, K]
    o = torch.matmul(p.transpose(-2, -1), vh)            # [h, K, dk]
    return o.transpose(0, 1).contiguous().view(k, d), p


def encoder_layer(x, c, sd, pre, h, act, big_lambda, r, rng=np.random, forced_sel=None):
    """One EncoderLayer (snuffy.py:113-157), eval mode.  x: [N, D]; c: [N].  Returns (z [N,D], P [h,N,K], S)."""
    if forced_sel is not None:
        sel = forced_sel
    else:
        top, rnd = select_indices(c, big_lambda, r, rng)
        sel = top if rnd is None else torch.cat([top, rnd])
    xs = x.index_select(0, sel)                                                       # snuffy.py:131,145-147
    xn = layer_norm(x, sd[pre + "sublayer.0.norm.weight"], sd[pre + "sublayer.0.norm.bias"])   # snuffy.py:107
    lin = pre + "self_attn.linears."
    q = F.linear(xn, sd[lin + "0.weight"], sd[lin + "0.bias"])                        # snuffy.py:187-190
    kp = F.linear(xs, sd[lin + "1.weight"], sd[lin + "1.bias"])
    v = F.linear(xn, sd[lin + "2.weight"], sd[lin + "2.bias"])
    o, p = sparse_attention(q, kp, v, h)
    o = F.linear(o, sd[lin + "3.weight"], sd[lin + "3.bias"])                         # snuffy.py:205
    x_sel = xs + o                                                                    # snuffy.py:108
    y = x.clone()
    y[sel] = x_sel                                                                    # snuffy.py:154-155
    yn = layer_norm(y, sd[pre + "sublayer.1.norm.weight"], sd[pre + "sublayer.1.norm.bias"])
    ff = pre + "feed_forward."
    hid = ACTIVATIONS[act](F.linear(yn, sd[ff + "w_1.weight"], sd[ff + "w_1.bias"]))  # snuffy.py:224-225
    z = y + F.linear(hid, sd[ff + "w_2.weight"], sd[ff + "w_2.bias"])                 # snuffy.py:110
    return z, p, sel


def bclassifier_forward(x, c, sd, h, act, big_lambda, r, depth, rng=np.random, forced_sel=None):
    """BClassifier(x, c) (snuffy.py:62-71) with x [N,D], c [N].  Returns (logits [C], P_last [h,N,K], [S per layer])."""
    sels = []
    p = None
    for l in range(depth):
        fs = None if forced_sel is None else forced_sel[l]
        x, p, sel = encoder_layer(x, c, sd, f"b_classifier.encoder.layers.{l}.", h, act, big_lambda, r, rng, fs)
        sels.append(sel)
    xn = layer_norm(x, sd["b_classifier.encoder.norm.weight"], sd["b_classifier.encoder.norm.bias"])  # snuffy.py:86
    logits = F.linear(xn.mean(dim=0), sd["b_classifier.linear.weight"], sd["b_classifier.linear.bias"])  # :71
    return logits, p, sels


def milnet_forward(x, sd, h, act, big_lambda, r, depth, rng=np.random, forced_sel=None):
    """MILNet.forward (snuffy.py:228-238) for one bag x [N, D] (binary model: C == 1).

    Returns (classes [N, C], logits [C], P_last [h, N, K], [S per layer]).
    """
    classes = F.linear(x, sd["i_classifier.fc.0.weight"], sd["i_classifier.fc.0.bias"])   # snuffy.py:39-41
    logits, p, sels = bclassifier_forward(x, classes[:, 0], sd, h, act, big_lambda, r, depth, rng, forced_sel)
    return classes, logits, p, sels


def run_model(x, y, sd, w, h, act, big_lambda, r, depth, rng=np.random, pos_weight=None, forced_sel=None):
    """SmallWeightTrainer._run_model + Snuffy._run_model (train.py:828-846, 913-916).

    Returns (bag_prediction (tensor scalar), loss, sigmoid(c).view(-1,1), classes, logits).
    """
    classes, logits, _, _ = milnet_forward(x, sd, h, act, big_lambda, r, depth, rng, forced_sel)
    max_pred = classes.max(dim=0)[0]                                   # train.py:831-834 (ins_prediction [1,N,C])
    pw = None if pos_weight is None else torch.as_tensor(pos_weight, dtype=x.dtype)
    bag_loss = F.binary_cross_entropy_with_logits(logits.view(1, -1), y.view(1, -1), pos_weight=pw)
    max_loss = F.binary_cross_entropy_with_logits(max_pred.view(1, -1), y.view(1, -1), pos_weight=pw)
    loss = w * bag_loss + (1 - w) * max_loss
    with torch.no_grad():
        bag_pred = ((1 - w) * torch.sigmoid(max_pred) + w * torch.sigmoid(logits)).squeeze()
    return bag_pred, loss, torch.sigmoid(classes.detach().view(-1, 1)), classes, logits


def dropout_patches(feats, p, rng=np.random):
    """utils.py:244-250: two permutation draws on the global RNG, always permutes rows (even at p=0)."""
    n = feats.shape[0]
    idx = rng.permutation(n)[: int(n * (1 - p))]            # == np.random.choice(np.arange(n), int(n(1-p)), False)
    sampled = np.take(feats, idx, axis=0)
    pad_idx = rng.permutation(sampled.shape[0])[: int(n * p)]
    return np.concatenate((sampled, np.take(sampled, pad_idx, axis=0)), axis=0)


# ----------------------------------------------------------------------------------------------------------------
# multiclass selection (snuffy_multiclass.py:130-171)
# ----------------------------------------------------------------------------------------------------------------
def select_indices_multiclass(c, big_lambda, random_patch_share, rng=np.random):
    """c: [B, N, C].  Returns (topk [B, ref_dim] int64, rnd [B, ref_dim] int64).

    Per batch row: top ceil(Lambda(1-r)) per class (sorted descending per class column), flattened row-major
    over (rank, class), torch.unique (ascending); ref_dim = min(min_b |uniq_b|, N - that); keep the LOWEST ref_dim
    unique indices; draw ref_dim random from the complement of ALL uniques of that row.
    """
    b, n, _ = c.shape
    k1 = math.ceil(big_lambda * (1.0 - random_patch_share))
    order = torch.sort(c, dim=1, descending=True, stable=True)[1][:, :k1, :].flatten(1)
    uniq = [torch.unique(order[i]) for i in range(b)]
    ref_dim = min(len(u) for u in uniq)
    ref_dim = min(ref_dim, n - ref_dim)
    topk = torch.stack([u[:ref_dim] for u in uniq]).to(torch.int64) if ref_dim > 0 else torch.zeros(b, 0, dtype=torch.int64)
    rnd = torch.zeros(b, ref_dim, dtype=torch.int64)
    for i in range(b):
        mask = np.ones(n, dtype=bool)
        mask[uniq[i].numpy()] = False
        rem = np.nonzero(mask)[0]
        if ref_dim > len(rem):
            raise ValueError("Cannot take a larger sample than population when 'replace=False'")
        rnd[i] = torch.from_numpy(rem[rng.permutation(len(rem))[:ref_dim]].astype(np.int64))
    return topk, rnd


def milnet_forward_multiclass(x, sd, h, act, big_lambda, r, depth, rng=np.random):
    """snuffy_multiclass.MILNet.forward for x [B, N, D].  Returns (classes [B,N,C], logits [B,C], P [B,h,N,K])."""
    classes = F.linear(x, sd["i_classifier.fc.0.weight"], sd["i_classifier.fc.0.bias"])
    b = x.shape[0]
    p_all = None
    for l in range(depth):
        pre = f"b_classifier.encoder.layers.{l}."
        topk, rnd = select_indices_multiclass(classes, big_lambda, r, rng)
        sel = torch.cat([topk, rnd], dim=1)
        outs, ps = [], []
        for i in range(b):
            z, p, _ = encoder_layer(x[i], None, sd, pre, h, act, big_lambda, r, rng, forced_sel=sel[i])
            outs.append(z)
            ps.append(p)
        x = torch.stack(outs)
        p_all = torch.stack(ps)
    xn = layer_norm(x, sd["b_classifier.encoder.norm.weight"], sd["b_classifier.encoder.norm.bias"])
    logits = F.linear(xn.mean(dim=1), sd["b_classifier.linear.weight"], sd["b_classifier.linear.bias"])
    return classes, logits, p_all
