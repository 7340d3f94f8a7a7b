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


def dropout_patches_device(feats, p):
    """dropout_patches for a bag already resident in HBM ([1, N, D] or [N, D] device tensor): the same two draws on the
    global numpy RNG as the host version (so the RNG stream and the resulting row order are identical), one device gather."""
    x = feats[0] if feats.dim() == 3 else feats
    n = x.shape[0]
    idx = np.random.choice(np.arange(n), int(n * (1 - p)), replace=False)
    pad_idx = np.random.choice(np.arange(idx.shape[0]), int(n * p), replace=False)
    rows = np.concatenate((idx, idx[pad_idx]))
    out = x.index_select(0, torch.from_numpy(rows.astype(np.int64)).to(x.device))
    return out.unsqueeze(0) if feats.dim() == 3 else out


def optimal_thresh(fpr, tpr, thresholds, p=0):
    """Threshold minimising (fpr - tpr) (reference utils.py:291-294)."""
    loss = (fpr - tpr) - p * tpr / (fpr + tpr + 1)
    idx = np.argmin(loss, axis=0)
    return fpr[idx], tpr[idx], thresholds[idx]


def multi_label_roc(labels, predictions, num_classes, for_feats=False):
    """Per-class AUC, ROC thresholds and the optimal threshold (reference utils.py:253-276)."""
    from sklearn.metrics import roc_auc_score, roc_curve
    thresholds, thresholds_optimal, aucs = [], [], []
    if len(predictions.shape) == 1 and not for_feats:
        predictions = predictions[:, None]
    for c in range(num_classes):
        label = labels if for_feats else labels[:, c]
        prediction = predictions if for_feats else predictions[:, c]
        fpr, tpr, threshold = roc_curve(label, prediction, pos_label=1)
        _, _, threshold_optimal = optimal_thresh(fpr, tpr, threshold)
        aucs.append(roc_auc_score(label, prediction))
        thresholds.append(threshold)
        thresholds_optimal.append(threshold_optimal)
    return aucs, thresholds, thresholds_optimal


def five_scores(bag_labels, bag_predictions):
    """(accuracy at the optimal threshold, AUC) of one class (reference utils.py:279-288)."""
    from sklearn.metrics import roc_auc_score, roc_curve
    fpr, tpr, threshold = roc_curve(bag_labels, bag_predictions, pos_label=1)
    _, _, threshold_optimal = optimal_thresh(fpr, tpr, threshold)
    auc_value = roc_auc_score(bag_labels, bag_predictions)
    hard = (np.array(bag_predictions) >= threshold_optimal).astype(int)
    accuracy = 1 - np.count_nonzero(np.array(bag_labels).astype(int) - hard) / len(bag_labels)
    return accuracy, auc_value


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


def convert_dsmil_mil_dataset_format_to_our_format(bag_ins_list, args):
    """[[bag_label, [instance vectors ...]], ...] -> (labels, feats, None, None) (reference utils.py:425-451): labels clipped to
    {0, 1} as float arrays of shape [1], feats = the first feats_size columns of the stacked instances."""
    all_labels, all_feats = [], []
    for bag_label, bag_vector in bag_ins_list:
        all_labels.append(np.expand_dims(np.array(int(np.clip(bag_label, 0, 1)), dtype=float), axis=0))
        all_feats.append(np.stack(bag_vector)[:, 0:args.feats_size])
    return all_labels, all_feats, None, None


def cross_validation_set(bag_ins_list, num_folds, current_fold, valid_ratio):
    """Fold split of the classical MIL datasets (reference utils.py:454-466): chunks of len // num_folds bags (a remainder
    becomes an extra chunk), the current chunk is the test split, the rest is cut into train | valid."""
    import itertools
    from copy import deepcopy
    csv_list = deepcopy(bag_ins_list)
    n = int(len(csv_list) / num_folds)
    chunked = [csv_list[i:i + n] for i in range(0, len(csv_list), n)]
    test_list = chunked.pop(current_fold)
    train_valid_list = list(itertools.chain.from_iterable(chunked))
    cut = int(len(train_valid_list) * (1 - valid_ratio))
    return train_valid_list[0:cut], train_valid_list[cut:], test_list


MIL_DATASET_FILES = {'musk1': ('Musk', 'musk1norm'), 'musk2': ('Musk', 'musk2norm'), 'elephant': ('Elephant', 'data_100x100'),
                     'fox': ('Fox', 'data_100x100'), 'tiger': ('Tiger', 'data_100x100')}


def load_mil_data(args, mil_datasets_base_path='./datasets/mil_dataset'):
    """(train, valid, test) tuples of the classical MIL benchmarks -- MUSK1/2, Elephant, Fox, Tiger -- from the pickled
    bag list '<name>_<folds>folds_<ratio>split.pkl' (reference utils.py:469-496; BASELINE config 1 = MUSK bags)."""
    import pickle
    folder, stem = MIL_DATASET_FILES[args.dataset]
    fname = f'{stem}_{args.cv_num_folds}folds_{args.cv_valid_ratio}split.pkl'
    with open(os.path.join(mil_datasets_base_path, folder, fname), 'rb') as f:
        bag_ins_list = pickle.load(f)
    tr, va, te = cross_validation_set(bag_ins_list, args.cv_num_folds, args.cv_current_fold, args.cv_valid_ratio)
    return tuple(convert_dsmil_mil_dataset_format_to_our_format(part, args) for part in (tr, va, te))


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
