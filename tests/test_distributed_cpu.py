"""world_size-2 gloo test (CPU) of the bag-parallel gradient exchange: after FlatGradAllReduce every rank holds the MEAN
of the per-rank gradients, i.e. exactly what one process accumulating both bags and dividing by 2 would hold."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 1))


def _bag(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(5, 8, generator=g), torch.tensor([[float(rank)]])


def _loss(model, x, y):
    return torch.nn.functional.binary_cross_entropy_with_logits(model(x).mean(0, keepdim=True), y)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from snuffy_amd.train import FlatGradAllReduce
    model = _model()
    w = torch.tensor(0.5, requires_grad=True)
    x, y = _bag(rank)
    (w * _loss(model, x, y)).backward()
    sync = FlatGradAllReduce([w] + list(model.parameters()), dist, world)
    sync()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    opt.step()
    out[rank] = [w.grad.clone()] + [p.grad.clone() for p in model.parameters()] + [p.detach().clone() for p in model.parameters()]
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_world2_equals_accumulate_and_average():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    # single-process reference: accumulate both bags, divide by world
    model = _model()
    w = torch.tensor(0.5, requires_grad=True)
    for r in range(world):
        x, y = _bag(r)
        (w * _loss(model, x, y) / world).backward()
    ref = [w.grad] + [p.grad for p in model.parameters()]
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    opt.step()
    ref += [p.detach() for p in model.parameters()]
    for r in range(world):
        got = out[r]
        assert len(got) == len(ref)
        for a, b in zip(got, ref):
            assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()
    for a, b in zip(out[0], out[1]):                      # ranks stay bit-identical -> replicas never drift
        assert torch.equal(a, b)


def test_world1_is_a_no_op():
    from snuffy_amd.train import FlatGradAllReduce
    model = _model()
    x, y = _bag(0)
    _loss(model, x, y).backward()
    before = [p.grad.clone() for p in model.parameters()]
    FlatGradAllReduce(model.parameters(), None, 1)()
    for a, p in zip(before, model.parameters()):
        assert torch.equal(a, p.grad)


# ----------------------------------------------------------------------------------------------------------------------
# the trainer's epoch loop under world_size 2 (gloo): bag partition, replica sync, sharded validation
# ----------------------------------------------------------------------------------------------------------------------
class _TinyMIL(torch.nn.Module):
    """CPU stand-in with MILNet's return convention (ins [1, N, 1], logits [1, 1], A): the product model has no CPU path;
    what is under test here is the model-agnostic Trainer loop."""

    def __init__(self):
        super().__init__()
        self.i_classifier = torch.nn.Linear(6, 1)
        self.b_classifier = torch.nn.Linear(6, 1)

    def forward(self, x):
        return self.i_classifier(x), self.b_classifier(x.mean(dim=1)), None


def _toy_bags(n_bags=11):
    import numpy as np
    g = np.random.RandomState(0)
    labels = [np.array([float(i % 2)], dtype=np.float32) for i in range(n_bags)]
    feats = [g.randn(int(g.randint(3, 30)), 6).astype(np.float32) + labels[i][0] for i in range(n_bags)]
    return labels, feats, None, None


def _trainer_worker(rank, world, port, out, balance_lengths=1):
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from snuffy_amd import train as T
    T.device = torch.device("cpu")

    class Tiny(T.SmallWeightTrainer):
        def _get_milnet(self):
            return _TinyMIL()

    args = T.get_args_parser().parse_args(["--optimizer", "adamw", "--num_epochs", "4", "--lr", "1e-2", "--dropout_patch",
                                           "0.1", "--soft_average", "1", "--balance_lengths", str(balance_lengths)])
    args.weight_init__weight_init_i__weight_init_b = [None, None, None]
    torch.manual_seed(100 + rank)            # replicas are built from DIFFERENT seeds: the trainer must sync them
    np.random.seed(7 + rank)                 # and the global numpy RNGs differ from the start
    tr = Tiny(args, dist=dist, rank=rank, world_size=world)
    data = _toy_bags()
    visited = []
    for epoch in (1, 2, 3):
        visited.append(list(tr.train(data, epoch)["visited"]))
    res = tr.valid(data)
    out[rank] = dict(visited=visited, weights=[p.detach().clone() for p in tr.milnet.parameters()],
                     w=tr.single_weight_parameter.detach().clone(), preds=res["predictions"], loss=res["epoch_valid_loss"])
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("balance_lengths", [1, 0])
def test_trainer_epoch_partition_replica_sync_and_sharded_valid_world2(balance_lengths):
    """balance_lengths = 1 (default, round 6): the two bags of a step are neighbours in patch count, the validation bags are dealt
    longest-first; 0: positions r::W of the shuffle.  Either way every bag is visited once per epoch and the replicas stay identical."""
    import numpy as np
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_trainer_worker, args=(world, _free_port(), out, balance_lengths), nprocs=world, join=True)
    n_bags = 11
    if balance_lengths:
        lens = np.array([f.shape[0] for f in _toy_bags()[1]])
        for e in range(3):   # step s: no third bag's length lies strictly between the two ranks' bags (but for the wrap-around pad)
            pads = 0
            for a, b in zip(out[0]["visited"][e], out[1]["visited"][e]):
                lo, hi = sorted((lens[a], lens[b]))
                between = int(((lens > lo) & (lens < hi)).sum())
                pads += between > 0
            assert pads <= 1
    for e in range(3):
        seen = out[0]["visited"][e] + out[1]["visited"][e]
        assert len(out[0]["visited"][e]) == len(out[1]["visited"][e]) == (n_bags + 1) // 2
        assert set(seen) == set(range(n_bags))                         # every bag of the epoch, across the ranks
        assert len(seen) - len(set(seen)) == (world - n_bags % world) % world   # only the wrap-around pad repeats
    for a, b in zip(out[0]["weights"], out[1]["weights"]):             # replicas identical after three epochs
        assert torch.equal(a, b)
    assert torch.equal(out[0]["w"], out[1]["w"])
    # sharded validation: both ranks hold the full, ordered result; it equals a single-process pass with the same weights
    assert np.array_equal(out[0]["preds"], out[1]["preds"]) and out[0]["loss"] == out[1]["loss"]
    model = _TinyMIL()
    with torch.no_grad():
        for p, w in zip(model.parameters(), out[0]["weights"]):
            p.copy_(w)
        labels, feats, _, _ = _toy_bags()
        w = out[0]["w"]
        ref = []
        for f in feats:
            ins, logit, _ = model(torch.from_numpy(f).unsqueeze(0))
            ref.append(float((1 - w) * torch.sigmoid(ins.max()) + w * torch.sigmoid(logit.squeeze())))
    assert np.allclose(out[0]["preds"][:, 0], np.array(ref), atol=1e-6)
