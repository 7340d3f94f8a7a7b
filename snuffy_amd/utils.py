"""Host-side bag loader pieces of the hot path (reference utils.py:138-250, 469-507) plus the weight-init / optimizer
registries train.py looks up (utils.py:69-135).  numpy / pandas only -- no GPU code here."""
import os

import numpy as np
import torch
import torch.nn as nn


def _linear_init(fn):
    def init(m):
        if isinstance(m, (nn.Linear, nn.Conv1d)):
            fn(m.weight)
            nn.init.zeros_(m.bias)
    return init


# reference utils.py:69-130.  NB: its 'orthogonal' entry really calls trunc_normal_ (utils.py:114-120) -- kept.
WEIGHT_INITS = {
    'xavier_normal': _linear_init(nn.init.xavier_normal_),
    'xavier_uniform': _linear_init(nn.init.xavier_uniform_),
    'kaiming_normal': _linear_init(nn.init.kaiming_normal_),
    'kaiming_uniform': _linear_init(nn.init.kaiming_uniform_),
    'trunc_normal': _linear_init(nn.init.trunc_normal_),
    'orthogonal': _linear_init(nn.init.trunc_normal_),
}

OPTIMIZERS = {'adam': torch.optim.Adam, 'adamw': torch.optim.AdamW}        # reference utils.py:132-135


def dropout_patches(feats, p):
    """Reference utils.py:244-250: resample rows without replacement, pad back with duplicates.

    Always permutes the rows (even at p = 0) and always consumes TWO draws of the global numpy RNG, exactly like the
    reference's two np.random.choice(..., replace=False) calls."""
    n = feats.shape[0]
    idx = np.random.choice(np.arange(n), int(n * (1 - p)), replace=False)
    sampled = np.take(feats, idx, axis=0)
    pad_idx = np.random.choice(np.arange(sampled.shape[0]), int(n * p), replace=False)
    return np.concatenate((sampled, np.take(sampled, pad_idx, axis=0)), axis=0)


def get_bag_feats(bag_row, args):
    """One bag from its feature CSV (reference utils.py:138-183).  bag_row = (csv_path, label).

    Returns (label [num_classes] f32, feats [N, D] f32, feats_labels or None, positions or None); rows are shuffled with
    sklearn.utils.shuffle like the reference (global numpy RNG)."""
    path = bag_row.iloc[0] if hasattr(bag_row, "iloc") else bag_row[0]
    raw_label = bag_row.iloc[1] if hasattr(bag_row, "iloc") else bag_row[1]
    path = path.replace("datasets/Camelyon16", "embeddings/camelyon16/official/")
    side = path + '.npz'
    if os.path.exists(side) and os.path.getmtime(side) >= os.path.getmtime(path):
        # binary twin written by compute_feats.write_bag_csv(sidecar=True): same parsed values, no text parsing.
        # sklearn.utils.shuffle(df) draws ONE np.random.shuffle of arange(n) from the global RNG; so does this.
        z = np.load(side, allow_pickle=True)
        feats = z['feats']
        idx = np.arange(feats.shape[0])
        np.random.shuffle(idx)
        feats = feats[idx]
        has_patch_labels = 'label' in z.files and 'position' in z.files
        feats_labels = z['label'][idx] if has_patch_labels else None
        positions = list(z['position'][idx]) if has_patch_labels else None
    else:
        import pandas as pd
        from sklearn.utils import shuffle
        df = pd.read_csv(path)
        has_patch_labels = 'position' in df and 'label' in df
        df = shuffle(df).reset_index(drop=True)
        feats = df.drop(columns=['label', 'position']) if has_patch_labels else df
        feats = feats.to_numpy().astype('float32')
        feats_labels = df['label'].to_numpy() if has_patch_labels else None
        positions = list(df['position']) if has_patch_labels else None
    label = np.zeros(args.num_classes)
    if args.num_classes == 1:
        label[0] = raw_label
    elif int(raw_label) <= len(label) - 1:
        label[int(raw_label)] = 1
    return label.astype('float32'), feats, feats_labels, positions


def load_data(bags_df, args):
    """(labels, feats, feat_labels, positions, names) tuple of lists (reference utils.py:186-241), single process."""
    labels, feats, flabels, positions, names = [], [], [], [], []
    have = True
    for i in range(len(bags_df)):
        row = bags_df.iloc[i]
        lab, f, fl, pos = get_bag_feats(row, args)
        labels.append(lab)
        feats.append(f)
        have = have and fl is not None
        if have:
            flabels.append(fl)
            positions.append(pos)
        names.append(str(row.iloc[0]).split('/')[-1].split('.')[0])
    if not have:
        flabels, positions = None, None
    return labels, feats, flabels, positions, names


def compute_pos_weight(labels):
    """Weighted-BCE factor for the unbalanced MIL datasets (reference utils.py:499-507)."""
    pos = 0
    for label in labels:
        pos = pos + np.clip(label, 0, 1)
    return (len(labels) - pos) / pos


def stage_bags(all_feats, device, l2norm=False):
    """Upload a list of [N_i, D] numpy bags into HBM once (the reference re-uploads every bag every epoch,
    train.py:255-256).  Returns a list of [1, N_i, D] fp32 device tensors."""
    out = []
    for f in all_feats:
        f = np.asarray(f, dtype=np.float32)
        if l2norm:
            f = f / np.linalg.norm(f, axis=1, keepdims=True)
        out.append(torch.from_numpy(np.ascontiguousarray(f)).unsqueeze(0).to(device, non_blocking=True))
    return out
