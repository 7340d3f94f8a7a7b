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
