"""MI355X-native mirror of the reference's MIL trainer for the Snuffy path (train.py: Trainer / SmallWeightTrainer /
Snuffy / Runner hooks), plus the bag-parallel multi-GPU step the reference does not have.

Kept from the reference (same names, same meaning):
  get_args_parser            the hot-path flags of train.py:54-135 with identical names and defaults
  Trainer._get_milnet/_get_criterion/_get_optimizer/_get_scheduler/_load_init_weights/_run_model/
          _after_run_model_in_training_mode/train/valid
  SmallWeightTrainer         single_weight_parameter (init 0.5, clamped to [0,1], own lr multiplier), loss mix train.py:828-846
  Snuffy                     _get_milnet exactly as train.py:861-911

Deliberately different:
  * bags can be staged in HBM once (utils.stage_bags) instead of a host->device copy per bag per epoch;
  * per-bag scalars (loss, prediction) stay on the device and are read back once per epoch, not three times per bag;
  * wandb / FROC / ECE / ROC plumbing is out of scope; AUC is computed with sklearn when present;
  * world_size > 1: bags are sharded rank::world, gradients are averaged with ONE flat all-reduce per optimizer step
    (RCCL over xGMI) -> effective batch = world_size bags; world_size == 1 is exactly the reference's step.
"""
import argparse
import copy
import os

import numpy as np
import torch
import torch.nn as nn
from torch import optim

from . import autograd as SA
from . import balance
from . import functional as SF
from . import snuffy, snuffy_multiclass
from .utils import (OPTIMIZERS, WEIGHT_INITS, compute_pos_weight, dropout_patches, dropout_patches_device,
                    multi_label_roc)

MIL_DATASETS = ('musk1', 'musk2', 'elephant', 'fox', 'tiger')
device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


def get_args_parser():
    """Hot-path subset of reference train.py:54-135 (same flag names / defaults)."""
    p = argparse.ArgumentParser(description='Train the Snuffy MIL aggregator on precomputed patch features (MI355X)')
    p.add_argument('--num_classes', default=1, type=int)
    p.add_argument('--feats_size', default=512, type=int)
    p.add_argument('--lr', default=2e-4, type=float)
    p.add_argument('--num_epochs', default=200, type=int)
    p.add_argument('--weight_decay', default=5e-3, type=float)
    p.add_argument('--eta_min', default=5e-06, type=float)
    p.add_argument('--dataset', default='camelyon16', type=str)
    p.add_argument('--dropout_patch', default=0, type=float)
    p.add_argument('--weight_init__weight_init_i__weight_init_b',
                   default=['xavier_normal', 'xavier_normal', 'xavier_normal'])
    p.add_argument('--optimizer', default='adam', type=str, choices=['adam', 'adamw', 'sgd'])
    p.add_argument('--scheduler', default='cosine', type=str, choices=['cosinewarmup', 'cosine'])
    p.add_argument('--arch', default='snuffy', type=str)
    p.add_argument('--soft_average', default=0, choices=[0, 1], type=int)
    p.add_argument('--single_weight__lr_multiplier', default=0.1, type=float)
    p.add_argument('--num_heads', default=6, type=int)
    p.add_argument('--big_lambda', default=200, type=int)
    p.add_argument('--random_patch_share', default=0.0, type=float)
    p.add_argument('--mlp_multiplier', default=4, type=int)
    p.add_argument('--encoder_dropout', default=0.0, type=float)
    p.add_argument('--activation', default='relu', type=str)
    p.add_argument('--clip_grad', default=None, type=float)
    p.add_argument('--depth', default=1, type=int)
    p.add_argument('--betas', default=[0.5, 0.9])
    p.add_argument('--l2normed_embeddings', default=0, type=int)
    p.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'], help='eval-forward arithmetic (snuffy_amd)')
    p.add_argument('--eval_bags_per_launch', default=1, type=int,
                   help='snuffy_amd: the evaluation loops pack up to this many SMALL bags (<= --eval_pack_max_patches patches each) '
                        'into one set of launches (MILNet.forward_bags: same selections, same draws of the random share; '
                        'projections over the packed rows, so logits can move by rounding).  1 = one bag per forward, as the reference')
    p.add_argument('--eval_pack_max_patches', default=16384, type=int)
    p.add_argument('--balance_lengths', default=1, type=int, choices=[0, 1],
                   help='snuffy_amd, several ranks only: 1 = the bags of one training step are neighbours in patch count and the '
                        'evaluation bags are dealt longest-first (balance.py; SURVEY 8e); 0 = positions r::W of the shuffle')
    return p


# ----------------------------------------------------------------------------------------------------------------------
# bag-parallel gradient exchange (new)
# ----------------------------------------------------------------------------------------------------------------------
def _fused_step(params):
    """Adam / AdamW as ONE kernel over all parameters (torch's fused implementation: the same update rule as the default
    multi-tensor one, ~50 us instead of ~150 us for the 7 M parameters of a D = 768 model) when they live on the GPU."""
    return {"fused": True} if all(p.is_cuda and p.is_floating_point() for p in params) else {}


class FlatGradAllReduce:
    """One flat fp32 buffer over all trainable parameters; ONE all-reduce (sum, then / world) per optimizer step.

    dist is torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests) or None for a single process.
    """

    def __init__(self, params, dist=None, world_size=1):
        self.params = [p for p in params if p.requires_grad]
        self.dist = dist if world_size > 1 else None
        self.world_size = world_size
        self.flat = None

    def __call__(self):
        if self.dist is None:
            return
        if self.flat is None:
            n = sum(p.numel() for p in self.params)
            self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        self.dist.all_reduce(self.flat, op=self.dist.ReduceOp.SUM)
        self.flat.div_(self.world_size)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = self.flat[off:off + n].view_as(p).clone()
            else:
                p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n


class CosineWarmupScheduler(torch.optim.lr_scheduler.LambdaLR):
    """Linear warm-up then cosine decay of the learning-rate factor.  The reference takes this class from the third-party
    package lightly (requirements.txt: lightly==1.4.8, train.py:19,189-194), which is not vendored; restated from that
    release's published rule: factor = (epoch + 1) / warmup while epoch < warmup, afterwards
    end - (end - 1) * (cos(pi * step / (max_steps - 1)) + 1) / 2 with step = epoch - warmup, max_steps = max - warmup,
    end = 0.001 (parity unpinned: no fixture of the package's output exists here)."""

    def __init__(self, optimizer, warmup_epochs, max_epochs, last_epoch=-1, end_value=0.001):
        self.warmup_epochs, self.max_epochs, self.end_value = warmup_epochs, max_epochs, end_value
        super().__init__(optimizer, lr_lambda=self.scale_lr, last_epoch=last_epoch)

    def scale_lr(self, epoch):
        if epoch < self.warmup_epochs:
            return (epoch + 1) / self.warmup_epochs
        step, max_steps = epoch - self.warmup_epochs, self.max_epochs - self.warmup_epochs
        if max_steps <= 1 or step >= max_steps:
            return self.end_value
        return self.end_value - (self.end_value - 1.0) * (np.cos(np.pi * step / (max_steps - 1)) + 1) / 2


class Trainer:
    def __init__(self, args, dist=None, rank=0, world_size=1):
        self.args = args
        self.dist, self.rank, self.world_size = dist, rank, world_size
        self.milnet = self._get_milnet()
        self._load_init_weights()
        self._sync_replicas()
        self._criterion_is_set = False
        self.criterion = self._get_criterion()
        self.optimizer = self._get_optimizer()
        self.scheduler = self._get_scheduler()
        self._grad_sync = FlatGradAllReduce(self._trainable(), dist, world_size)

    def _sync_replicas(self):
        """world_size > 1: every replica starts from rank 0's weights (the ranks may have been seeded differently)."""
        if self.dist is None or self.world_size <= 1:
            return
        with torch.no_grad():
            for t in self._replicated_tensors():
                self.dist.broadcast(t, src=0)

    def _replicated_tensors(self):
        return [p.data for p in self.milnet.parameters()] + [b.data for b in self.milnet.buffers()]

    def _epoch_order(self, num_bags, cur_epoch):
        """Bag visiting order of one epoch.  One process: sklearn.utils.shuffle on the global numpy RNG, as the reference
        (train.py:231-236).  Several ranks: the per-bag draws (dropout_patches, the random patch share) depend on each rank's
        own bags, so the global RNG streams diverge after the first step; the order therefore comes from rank 0 (same draw on
        the global RNG there) and is broadcast -- every bag of the epoch is visited exactly once across the ranks."""
        from sklearn.utils import shuffle
        if self.dist is None or self.world_size <= 1:
            return shuffle(np.arange(num_bags))
        order = torch.zeros(num_bags, dtype=torch.int64)
        if self.rank == 0:
            order = torch.from_numpy(np.ascontiguousarray(shuffle(np.arange(num_bags))).astype(np.int64))
        dev = device if self.dist.get_backend() == "nccl" else torch.device("cpu")
        order = order.to(dev)
        self.dist.broadcast(order, src=0)
        return order.cpu().numpy()

    # -- hooks (reference names) --------------------------------------------------------------------------------------
    def _get_milnet(self) -> nn.Module:
        raise NotImplementedError

    def _trainable(self):
        return list(self.milnet.parameters())

    def _get_criterion(self):
        self._criterion_is_set = self.args.dataset not in MIL_DATASETS        # train.py:158-164
        return nn.BCEWithLogitsLoss()

    def _get_optimizer(self) -> optim.Optimizer:
        try:
            cls = OPTIMIZERS[self.args.optimizer]
        except KeyError:
            raise Exception(f'Optimizer not found. Given: {self.args.optimizer}, Have: {OPTIMIZERS.keys()}')
        return cls(params=self.milnet.parameters(), lr=self.args.lr, betas=(self.args.betas[0], self.args.betas[1]),
                   weight_decay=self.args.weight_decay, **_fused_step(self.milnet.parameters()))

    def _get_scheduler(self):
        if self.args.scheduler == 'cosine':
            return torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=self.args.num_epochs,
                                                              eta_min=self.args.eta_min)
        if self.args.scheduler == 'cosinewarmup':                                # train.py:189-194
            return CosineWarmupScheduler(self.optimizer, warmup_epochs=int(self.args.num_epochs / 20),
                                         max_epochs=self.args.num_epochs)
        print('Scheduler set to None')
        return None

    def _load_init_weights(self):
        names = self.args.weight_init__weight_init_i__weight_init_b
        try:
            init_i, init_b = WEIGHT_INITS[names[1]], WEIGHT_INITS[names[2]]
        except KeyError:
            if names[0] is not None:
                raise Exception(f'Weight init not found. Given: {names[0]}, Have: {WEIGHT_INITS.keys()} ')
            return
        self.milnet.i_classifier.apply(init_i)
        self.milnet.b_classifier.apply(init_b)

    def _run_model(self, bag_feats, bag_label):
        raise NotImplementedError

    def _after_run_model_in_training_mode(self, step, num_bags, batch_idx):
        self._grad_sync()                                                       # no-op unless world_size > 1
        if self.args.clip_grad is not None:
            torch.nn.utils.clip_grad_norm_(self.milnet.parameters(), max_norm=self.args.clip_grad)
        self.optimizer.step()
        SF.after_optimizer_step()
        self.optimizer.zero_grad()

    # -- epoch loops -------------------------------------------------------------------------------------------------
    def _bag_to_device(self, feats):
        if torch.is_tensor(feats):
            x = feats if feats.dim() == 3 else feats.unsqueeze(0)
            return x.to(device)
        return torch.from_numpy(np.ascontiguousarray(np.asarray(feats, dtype=np.float32))).unsqueeze(0).to(device)

    def train(self, data, cur_epoch):
        """One epoch over (labels, feats, feat_labels, positions) (reference train.py:223-293).

        feats entries are [N_i, D] numpy arrays (as utils.load_data returns them) or [1, N_i, D] device tensors
        (utils.stage_bags).  Bags are visited in a seeded shuffle; rank r takes positions r::world_size."""
        self.milnet.train()
        labels, feats = data[0], data[1]
        order = self._epoch_order(len(labels), cur_epoch)                       # consumes the global numpy RNG once
        if not self._criterion_is_set:
            pw = torch.tensor(compute_pos_weight(labels), device=device, dtype=torch.float32)
            self.criterion = nn.BCEWithLogitsLoss(pw)
            self._criterion_is_set = True
        num_bags = len(order)
        steps = (num_bags + self.world_size - 1) // self.world_size
        if self.world_size > 1 and int(getattr(self.args, 'balance_lengths', 1)):
            # a step ends in ONE all-reduce and costs its LONGEST bag: the ranks' bags of a step are neighbours in length (SURVEY 8e;
            # balance.step_groups -- every bag once per epoch, the groups in the shuffle's order; identical on every rank)
            visit = balance.step_groups(balance.bag_lengths(feats), self.world_size, order)
        else:
            visit = np.asarray(order)[np.arange(steps * self.world_size) % num_bags]   # wrap: every rank steps every time
        losses, preds, seen = [], [], []
        for s in range(steps):
            i = int(visit[s * self.world_size + self.rank])
            f = feats[i]
            if not torch.is_tensor(f):
                if self.args.l2normed_embeddings == 1:
                    f = f / np.linalg.norm(f, axis=1, keepdims=True)
                f = dropout_patches(f, self.args.dropout_patch)
            else:
                # resident bag: the SAME two draws on the global numpy RNG, the row gather on the device
                f = dropout_patches_device(f, self.args.dropout_patch)
            bag_feats = self._bag_to_device(f)
            bag_label = torch.as_tensor(np.asarray(labels[i], dtype=np.float32).reshape(1, -1), device=device)
            bag_prediction, loss, _ = self._run_model(bag_feats, bag_label)
            loss.backward()
            self._after_run_model_in_training_mode(step=num_bags * (cur_epoch - 1) + s, num_bags=num_bags, batch_idx=s)
            losses.append(loss.detach())
            preds.append(bag_prediction)
            seen.append(i)
        total = torch.stack(losses).sum().item()                                # ONE device->host read per epoch
        preds = torch.stack([p.reshape(-1) for p in preds]).cpu().numpy()
        lab = np.array([labels[i] for i in seen])
        res = {'epoch_train_loss': total / max(1, len(losses)), 'predictions': preds, 'labels': lab, 'visited': seen}
        res.update(self._epoch_metrics(lab, preds, 'epoch_train'))
        return res

    def _calc_metrics(self, labels, predictions, predefined_thresholds_optimal=None):
        """accuracy / AUC / optimal thresholds of one epoch (reference train.py:475-506)."""
        assert len(labels) == len(predictions), \
            f"Number of predictions ({len(predictions)}) and labels ({len(labels)}) do not match"
        num_bags = len(labels)
        labels = np.array(labels)
        predictions = np.array(predictions)
        auc_scores, _, thresholds_optimal = multi_label_roc(labels, predictions, self.args.num_classes)
        if predefined_thresholds_optimal is not None:
            thresholds_optimal = predefined_thresholds_optimal
        if self.args.num_classes == 1:
            # the reference holds one scalar per bag here (np.squeeze of [n, 1] labels): [n, 1] predictions are taken as well
            predictions = (predictions >= thresholds_optimal[0]).astype(predictions.dtype).reshape(num_bags)
            labels = labels.reshape(num_bags)
        else:
            predictions = predictions.copy()
            for i in range(self.args.num_classes):
                predictions[:, i] = (predictions[:, i] >= thresholds_optimal[i]).astype(predictions.dtype)
        bag_score = sum(bool(np.array_equal(labels[i], predictions[i])) for i in range(num_bags))
        return bag_score / num_bags, auc_scores, thresholds_optimal

    def _epoch_metrics(self, lab, preds, prefix):
        lab = np.asarray(lab).reshape(len(preds), -1)
        try:
            if all(len(np.unique(lab[:, c])) > 1 for c in range(lab.shape[1])):
                acc, aucs, thr = self._calc_metrics(lab, np.asarray(preds, dtype=np.float64).reshape(lab.shape))
                return {prefix + '_accuracy': acc, prefix + '_aucs': aucs, prefix + '_thresholds_optimal': thr}
        except ImportError:
            pass
        return {}

    @torch.no_grad()
    def valid(self, data, predefined_thresholds_optimal=None):
        """Per-bag evaluation loop (reference train.py:295-360): mean loss, predictions, labels, accuracy / AUCs / optimal
        thresholds.  world_size > 1: rank r evaluates bags r::world_size, ONE all_gather of the (loss, prediction) rows
        (SURVEY 8e); every rank returns the full result in the original bag order."""
        self.milnet.eval()
        labels, feats = data[0], data[1]
        num_bags = len(labels)
        if self.world_size > 1 and int(getattr(self.args, 'balance_lengths', 1)):
            # no collective in the data path: longest-processing-time-first over the bags' patch counts
            shares = balance.lpt_assignment(balance.bag_lengths(feats), self.world_size)
        else:
            shares = [list(range(r, num_bags, self.world_size)) for r in range(self.world_size)]
        mine = shares[self.rank]
        losses, preds = [], []

        def load(i):
            f = feats[i]
            if not torch.is_tensor(f) and self.args.l2normed_embeddings == 1:
                f = f / np.linalg.norm(f, axis=1, keepdims=True)
            return (self._bag_to_device(f),
                    torch.as_tensor(np.asarray(labels[i], dtype=np.float32).reshape(1, -1), device=device))

        per_launch = int(getattr(self.args, 'eval_bags_per_launch', 1) or 1)
        pack_max = int(getattr(self.args, 'eval_pack_max_patches', 16384))
        # a subclass that overrides _run_model with the reference's two-argument signature keeps the one-bag-per-forward loop
        import inspect
        can_pack = (per_launch > 1 and hasattr(self.milnet, 'forward_bags') and hasattr(self, '_outputs_to_loss')
                    and 'outputs' in inspect.signature(self._run_model).parameters)
        pos = 0
        while pos < len(mine):
            chunk = [load(i) for i in mine[pos:pos + (per_launch if can_pack else 1)]]
            pos += len(chunk)
            if len(chunk) > 1 and all(b.shape[-2] <= pack_max for b, _ in chunk):
                # small bags: one set of launches for the chunk (forward_bags falls back to the per-bag loop by itself when the
                # chunk cannot be packed); the loss / prediction arithmetic per bag is the per-bag path's
                outs = self.milnet.forward_bags([b for b, _ in chunk])
                results = [self._run_model(b, y, outputs=o) for (b, y), o in zip(chunk, outs)]
            else:
                results = [self._run_model(b, y) for b, y in chunk]
            for bag_prediction, loss, _ in results:
                losses.append(loss.reshape(1).float())
                preds.append(bag_prediction.reshape(-1).float())
        ncls = int(np.asarray(labels[0]).size)
        rows = torch.cat([torch.cat(losses).view(-1, 1), torch.stack(preds)], dim=1) if mine else \
            torch.zeros(0, 1 + ncls, device=device)
        if self.dist is not None and self.world_size > 1:
            per = max(len(sh) for sh in shares)
            pad = torch.zeros(per, rows.shape[1], device=rows.device, dtype=rows.dtype)
            pad[:rows.shape[0]] = rows
            if self.dist.get_backend() != "nccl":
                pad = pad.cpu()
            gathered = [torch.empty_like(pad) for _ in range(self.world_size)]
            self.dist.all_gather(gathered, pad)
            full = torch.zeros(num_bags, rows.shape[1], dtype=rows.dtype)
            for r in range(self.world_size):
                idx = shares[r]
                full[idx] = gathered[r][:len(idx)].cpu()
            rows = full
        rows = rows.cpu().numpy()
        preds = rows[:, 1:]
        lab = np.array(labels).reshape(num_bags, -1)
        res = {'epoch_valid_loss': float(rows[:, 0].mean()), 'predictions': preds, 'labels': lab}
        if predefined_thresholds_optimal is not None:
            acc, aucs, thr = self._calc_metrics(lab, preds.astype(np.float64), predefined_thresholds_optimal)
            res.update({'epoch_valid_accuracy': acc, 'epoch_valid_aucs': aucs, 'epoch_valid_thresholds_optimal': thr})
        else:
            res.update(self._epoch_metrics(lab, preds, 'epoch_valid'))
        return res


class SmallWeightTrainer(Trainer):
    """Bag loss and max-instance loss mixed by one scalar w (reference train.py:797-858)."""

    def __init__(self, args, dist=None, rank=0, world_size=1):
        self.args = args
        self.single_weight_parameter = self._get_single_weight_parameter()
        super().__init__(args, dist, rank, world_size)

    def _get_single_weight_parameter(self):
        w = torch.tensor(0.5, requires_grad=bool(self.args.soft_average), device=device)
        w.data.clamp_(0, 1)
        return w

    def _trainable(self):
        extra = [self.single_weight_parameter] if self.single_weight_parameter.requires_grad else []
        return extra + list(self.milnet.parameters())

    def _replicated_tensors(self):
        return [self.single_weight_parameter.data] + super()._replicated_tensors()

    def _get_optimizer(self) -> optim.Optimizer:
        try:
            cls = OPTIMIZERS[self.args.optimizer]
        except KeyError:
            raise Exception(f'Optimizer not found. Given: {self.args.optimizer}, Have: {OPTIMIZERS.keys()}')
        return cls(params=[{'params': self.single_weight_parameter,
                            'lr': self.args.lr * self.args.single_weight__lr_multiplier},
                           {'params': self.milnet.parameters()}],
                   lr=self.args.lr, betas=(self.args.betas[0], self.args.betas[1]), weight_decay=self.args.weight_decay,
                   **_fused_step([self.single_weight_parameter] + list(self.milnet.parameters())))

    _outputs_to_loss = True      # _run_model accepts the model outputs of a packed forward (Trainer.valid)

    def _run_model(self, bag_feats, bag_label, outputs=None):
        ins_prediction, bag_prediction, _ = self.milnet(bag_feats) if outputs is None else outputs
        if torch.is_grad_enabled() and ins_prediction.requires_grad:
            # training: max over the instances, both BCE terms, their mix and the bag prediction in one launch (autograd.MilLossFn)
            fused = SA.mil_loss(ins_prediction, bag_prediction, bag_label, self.single_weight_parameter, self.criterion)
            if fused is not None:
                return fused[1].squeeze(), fused[0], ins_prediction
        max_prediction, _ = torch.max(ins_prediction, 0 if ins_prediction.dim() == 2 else 1)
        bag_loss = self.criterion(bag_prediction.view(1, -1), bag_label.view(1, -1))
        max_loss = self.criterion(max_prediction.view(1, -1), bag_label.view(1, -1))
        w = self.single_weight_parameter
        loss = w * bag_loss + (1 - w) * max_loss
        with torch.no_grad():                      # stays on the device: the reference's .cpu().numpy() sync is deferred
            bag_pred = ((1 - w) * torch.sigmoid(max_prediction) + w * torch.sigmoid(bag_prediction)).squeeze()
        return bag_pred, loss, ins_prediction

    def _after_run_model_in_training_mode(self, step, num_bags, batch_idx):
        super()._after_run_model_in_training_mode(step, num_bags, batch_idx)
        self.single_weight_parameter.data.clamp_(0, 1)

    def __str__(self):
        return f'Single_Weight__sa{self.args.soft_average}'


# per-tensor initialisers of the _get_milnet hooks (train.py:892-899)
_INIT_FUNCS = {'trunc_normal': nn.init.trunc_normal_, 'kaiming_uniform': nn.init.kaiming_uniform_,
               'kaiming_normal': nn.init.kaiming_normal_, 'xavier_uniform': nn.init.xavier_uniform_,
               'xavier_normal': nn.init.xavier_normal_, 'orthogonal': nn.init.orthogonal_}


class Snuffy(SmallWeightTrainer):
    def _get_milnet(self) -> nn.Module:
        """Same construction + init order as reference train.py:861-911."""
        a = self.args
        i_classifier = snuffy.FCLayer(in_size=a.feats_size, out_size=a.num_classes).to(device)
        c = copy.deepcopy
        attn = snuffy.MultiHeadedAttention(a.num_heads, a.feats_size).to(device)
        ff = snuffy.PositionwiseFeedForward(a.feats_size, a.feats_size * a.mlp_multiplier, a.activation,
                                            a.encoder_dropout).to(device)
        b_classifier = snuffy.BClassifier(
            snuffy.Encoder(snuffy.EncoderLayer(a.feats_size, c(attn), c(ff), a.encoder_dropout, a.big_lambda,
                                               a.random_patch_share), a.depth),
            a.num_classes, a.feats_size).to(device)
        milnet = snuffy.MILNet(i_classifier, b_classifier).to(device)
        names = a.weight_init__weight_init_i__weight_init_b
        for init_name, module_name in [(names[1], 'i_classifier'), (names[2], 'b_classifier')]:
            fn = _INIT_FUNCS.get(init_name)
            for name, p in milnet.named_parameters():
                if p.dim() > 1 and name.split(".")[0] == module_name:
                    fn(p)
        milnet.configure(precision=getattr(a, 'precision', 'fp32'), return_attention=False)   # A is discarded, train.py:830
        return milnet

    def _run_model(self, bag_feats, bag_label, outputs=None):
        bag_prediction, loss, ins_prediction = super()._run_model(bag_feats, bag_label, outputs)
        return bag_prediction, loss, torch.sigmoid(ins_prediction.view(-1, 1))

    def __str__(self):
        return f'Snuffy_k{self.args.big_lambda}_sa{self.args.soft_average}_depth{self.args.depth}'


class SnuffyMulticlass(SmallWeightTrainer):
    def _get_milnet(self) -> nn.Module:
        """Same construction + init order as reference train.py:922-972 (multi-class / batched model)."""
        a = self.args
        smc = snuffy_multiclass
        i_classifier = smc.FCLayer(in_size=a.feats_size, out_size=a.num_classes).to(device)
        c = copy.deepcopy
        attn = smc.MultiHeadedAttention(a.num_heads, a.feats_size).to(device)
        ff = smc.PositionwiseFeedForward(a.feats_size, a.feats_size * a.mlp_multiplier, a.activation).to(device)
        b_classifier = smc.BClassifier(
            smc.Encoder(smc.EncoderLayer(a.feats_size, c(attn), c(ff), a.num_classes, a.encoder_dropout, a.big_lambda,
                                         a.random_patch_share), a.depth),
            a.num_classes, a.feats_size).to(device)
        milnet = smc.MILNet(i_classifier, b_classifier).to(device)
        names = a.weight_init__weight_init_i__weight_init_b
        for init_name, module_name in [(names[1], 'i_classifier'), (names[2], 'b_classifier')]:
            fn = _INIT_FUNCS.get(init_name)
            for name, p in milnet.named_parameters():
                if p.dim() > 1 and name.split(".")[0] == module_name:
                    fn(p)
        milnet.b_classifier.configure(precision=getattr(a, 'precision', 'fp32'), return_attention=False)
        return milnet

    def _run_model(self, bag_feats, bag_label, outputs=None):
        bag_prediction, loss, ins_prediction = super()._run_model(bag_feats, bag_label, outputs)
        return bag_prediction, loss, torch.sigmoid(ins_prediction.view(-1, 1))

    def __str__(self):
        return f'Snuffy_Multiclass_k{self.args.big_lambda}_sa{self.args.soft_average}_depth{self.args.depth}'


ARCH_REGISTRY = {'snuffy': Snuffy, 'snuffy_multiclass': SnuffyMulticlass}


# ----------------------------------------------------------------------------------------------------------------------
# Runner: the epoch loop around a trainer (reference train.py:523-794) and the CLI entry (train.py:985-1039)
# ----------------------------------------------------------------------------------------------------------------------
SAVE_PATH = 'runs'


class Runner:
    """Epoch loop of the reference's Runner: validation before the first epoch, per epoch train / valid / scheduler step, the
    model's state dict + thresholds + single_weight_parameter saved every epoch under the reference's file names
    (`{epoch}.pth`, `thresholds_{epoch}.txt`, `single_weight_parameter_{epoch}`), best-AUC bookkeeping, the test pass on the
    best-AUC and the last epoch with their stored thresholds, and the clean-up of the other epochs' files.  wandb / FROC / ECE
    plumbing is out of scope.  `data` = (train, valid, test) tuples as utils.load_data / load_mil_data return them; MIL
    datasets are loaded here when it is None."""

    def __init__(self, args, trainer, data=None, save_path=None, log=print):
        self.args, self.trainer, self.log = args, trainer, log
        self.save_path = save_path or os.path.join(SAVE_PATH, args.dataset, str(trainer))
        if trainer.rank == 0:
            os.makedirs(self.save_path, exist_ok=True)
        if data is None:
            if args.dataset not in MIL_DATASETS:
                raise Exception(f'Runner: pass data=(train, valid, test) for dataset {args.dataset!r} (utils.load_data on the '
                                f'split dataframes), or use one of {MIL_DATASETS}')
            from .utils import load_mil_data
            data = load_mil_data(args)
        self.train_data, self.valid_data, self.test_data = data
        log(f'Num Bags (Train: {len(self.train_data[0])}) (Valid: {len(self.valid_data[0])}) (Test: {len(self.test_data[0])})')

    def _paths(self, epoch):
        return (os.path.join(self.save_path, f'{epoch}.pth'), os.path.join(self.save_path, f'thresholds_{epoch}.txt'),
                os.path.join(self.save_path, f'single_weight_parameter_{epoch}'))

    def _save_epoch_model(self, thresholds_optimal, epoch, auc, report_prefix=None):
        if self.trainer.rank != 0:                     # replicas are identical: one writer
            return
        import json
        model_path, log_path, w_path = self._paths(epoch)
        torch.save(self.trainer.milnet.state_dict(), model_path)
        with open(log_path, 'w') as f:
            json.dump({'auc': float(auc), 'thresholds_optimal': str([float(v) for v in np.asarray(thresholds_optimal, dtype=np.float64).reshape(-1)]),
                       'feats_thresholds_optimal': None}, f)
        if hasattr(self.trainer, 'single_weight_parameter'):
            torch.save(self.trainer.single_weight_parameter, w_path)
        if report_prefix:
            self.log(f'\t[{report_prefix}] model saved at: {model_path} threshold: {thresholds_optimal}')

    def _load_epoch_model(self, epoch):
        import ast
        import json
        model_path, log_path, w_path = self._paths(epoch)
        self.trainer.milnet.load_state_dict(torch.load(model_path, map_location=device), strict=True)
        with open(log_path) as f:
            rec = json.load(f)
        thresholds = np.asarray(ast.literal_eval(rec['thresholds_optimal']), dtype=np.float32)
        if hasattr(self.trainer, 'single_weight_parameter') and os.path.exists(w_path):
            self.trainer.single_weight_parameter = torch.load(w_path, map_location=device)
        return thresholds

    def run_train(self):
        import json
        import time
        best_auc, best_auc_epochs = 0, []
        self.initial_metrics = self.trainer.valid(self.valid_data)              # train.py:611-618
        for epoch in range(1, self.args.num_epochs + 1):
            t0 = time.time()
            tm = self.trainer.train(self.train_data, epoch)
            vm = self.trainer.valid(self.valid_data)
            aucs = vm.get('epoch_valid_aucs', [0.0])
            thr = vm.get('epoch_valid_thresholds_optimal', [0.5] * self.args.num_classes)
            self.log('Epoch [%d/%d] time %.1fs train loss: %.4f test loss: %.4f, thresholds_optimal: %s, accuracy: %.4f, AUC: %s'
                     % (epoch, self.args.num_epochs, time.time() - t0, tm['epoch_train_loss'], vm['epoch_valid_loss'], thr,
                        vm.get('epoch_valid_accuracy', float('nan')),
                        '|'.join('class-{0}>>{1:.4f}'.format(*k) for k in enumerate(aucs))))
            if self.trainer.scheduler is not None:
                self.trainer.scheduler.step()
            current_auc = aucs[0]
            prefix = ''
            if current_auc >= best_auc:
                prefix = '[best auc]'
                if current_auc > best_auc:
                    best_auc_epochs = []
                best_auc = current_auc
                best_auc_epochs.append(epoch)
            self._save_epoch_model(thr, epoch, current_auc, report_prefix=prefix)
        if self.trainer.rank == 0:
            with open(os.path.join(self.save_path, 'train_metrics.json'), 'w') as f:
                json.dump({'best_auc': float(best_auc), 'best_auc_epochs': best_auc_epochs}, f)
        return [min(best_auc_epochs, default=None)]

    def run_test(self, best_auc_epochs):
        """Test pass (train.py:752-778) on the earliest best-AUC epoch and the last epoch with THEIR validation thresholds."""
        out = {}
        for epoch, tag in ((min([e for e in best_auc_epochs if e is not None], default=None), 'best_auc'),
                           (self.args.num_epochs, 'last_epoch')):
            if epoch is None:
                continue
            if self.trainer.dist is not None and self.trainer.world_size > 1:
                self.trainer.dist.barrier()                                     # rank 0 wrote the files
            thr = self._load_epoch_model(epoch)
            res = self.trainer.valid(self.test_data, predefined_thresholds_optimal=thr)
            out[tag] = {k.replace('epoch_valid', tag): v for k, v in res.items()}
            self.log(f'[{tag}] epoch {epoch}: test loss {res["epoch_valid_loss"]:.4f} accuracy '
                     f'{res.get("epoch_valid_accuracy", float("nan")):.4f}')
        return out

    def clean_up(self, best_auc_epochs):
        """Keep the best-AUC and the last epoch's files (train.py:780-794)."""
        if self.trainer.rank != 0:
            return
        keep = {e for e in best_auc_epochs if e is not None} | {self.args.num_epochs}
        for epoch in range(1, self.args.num_epochs + 1):
            if epoch not in keep:
                for path in self._paths(epoch):
                    if os.path.exists(path):
                        os.remove(path)

    def run(self):
        best = self.run_train()
        res = self.run_test(best)
        self.clean_up(best)
        return res


def validate_args(args):
    """train.py:985-1001: MIL datasets fix the feature width."""
    args.soft_average = bool(args.soft_average)
    feats = {'musk1': 166, 'musk2': 166, 'elephant': 230}
    if args.dataset in feats:
        args.feats_size = feats[args.dataset]
        print(f'Setting feats_size to {args.feats_size} for {args.dataset}')
    return args


def main(argv=None, data=None):
    """CLI entry (train.py:1004-1039): parse, build the trainer from the architecture registry, run.  List-valued flags may come
    as strings (wandb sweeps) and are literal-eval'd like the reference does.  One process per GPU under torch.distributed.run:
    the bag-parallel trainer picks RANK / WORLD_SIZE up from the environment."""
    import ast
    parser = get_args_parser()
    parser.add_argument('--cv_num_folds', default=10, type=int)
    parser.add_argument('--cv_current_fold', default=0, type=int)
    parser.add_argument('--cv_valid_ratio', default=0.2, type=float)
    parser.add_argument('--save_path', default=None, type=str)
    args = validate_args(parser.parse_args(argv))
    if isinstance(args.betas, str):
        args.betas = ast.literal_eval(args.betas)
    if isinstance(args.weight_init__weight_init_i__weight_init_b, str):
        args.weight_init__weight_init_i__weight_init_b = ast.literal_eval(args.weight_init__weight_init_i__weight_init_b)
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl' if torch.cuda.is_available() else 'gloo', rank=rank, world_size=world)
    try:
        trainer = ARCH_REGISTRY[args.arch](args, dist, rank, world)
    except KeyError:
        raise Exception(f'Invalid Architecture: {args.arch} | Choose from: {ARCH_REGISTRY.keys()}')
    res = Runner(args, trainer, data=data, save_path=args.save_path).run()
    if dist is not None:
        dist.destroy_process_group()
    return res


class BagParallelStepper:
    """bench.py's training step: one resident bag per rank, backward, flat-gradient all-reduce, AdamW (train.py defaults)."""

    def __init__(self, milnet, world_size=1, dist=None, device=None, lr=2e-4, betas=(0.5, 0.9), weight_decay=5e-3,
                 precision="fp32"):
        self.milnet = milnet.train()
        self.milnet.configure(precision=precision, return_attention=False)
        self.w = torch.tensor(0.5, device=device)
        self.criterion = nn.BCEWithLogitsLoss()
        self.optimizer = torch.optim.AdamW(self.milnet.parameters(), lr=lr, betas=betas, weight_decay=weight_decay,
                                           **_fused_step(self.milnet.parameters()))
        self.sync = FlatGradAllReduce(self.milnet.parameters(), dist, world_size)

    def step(self, bag, label):
        ins, logits, _ = self.milnet(bag)
        fused = SA.mil_loss(ins, logits, label, self.w, self.criterion)         # the trainer's loss head (_run_model), one launch each way
        if fused is not None:
            loss = fused[0]
        else:
            max_pred, _ = torch.max(ins, 1)
            loss = self.w * self.criterion(logits.view(1, -1), label.view(1, -1)) + \
                (1 - self.w) * self.criterion(max_pred.view(1, -1), label.view(1, -1))
        loss.backward()
        self.sync()
        self.optimizer.step()
        SF.after_optimizer_step()
        self.optimizer.zero_grad()           # set_to_none: the next backward assigns the gradients, no fill + add per parameter
        return loss.detach()


if __name__ == '__main__':
    main()
