"""Worker of tests/test_gpu_configs.py::test_bag_parallel_two_ranks_on_rccl (launched by torch.distributed.run, one rank per GPU).
Rank r: the REAL model, BagParallelStepper at world 2 on the "nccl" (= RCCL) backend, steps on bag r of two; then a sharded
Snuffy.valid over five bags.  Rank 0 writes what the single-process reference needs to a file."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
    dev = torch.device("cuda", torch.cuda.current_device())
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from snuffy_amd import train as T
    from snuffy_amd.train import BagParallelStepper, Snuffy
    T.device = dev
    from snuffy_amd import snuffy as S
    S.device = dev
    D = 128
    args = T.get_args_parser().parse_args([])
    args.feats_size, args.optimizer, args.num_epochs, args.num_heads, args.big_lambda = D, "adamw", 2, 2, 32
    torch.manual_seed(1 + rank)                           # different seeds: the trainer must broadcast rank 0's weights
    tr = Snuffy(args, dist=dist, rank=rank, world_size=world)
    for m in tr.milnet.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    start = {k: v.detach().clone().cpu() for k, v in tr.milnet.state_dict().items()}
    g = torch.Generator().manual_seed(2)
    bags = [torch.randn(1, n, D, generator=g) for n in (700, 1500)]
    labels = [torch.tensor([1.0]), torch.tensor([0.0])]
    st = BagParallelStepper(tr.milnet, world_size=world, dist=dist, device=dev, lr=args.lr, betas=tuple(args.betas),
                            weight_decay=args.weight_decay, precision="fp32")
    st.step(bags[rank].to(dev), labels[rank].to(dev))
    after = {k: v.detach().clone().cpu() for k, v in tr.milnet.state_dict().items()}
    # sharded validation (longest-first assignment, one all_gather): every rank returns the full, ordered result
    vg = torch.Generator().manual_seed(5)
    vfeats = [torch.randn(n, D, generator=vg).numpy() for n in (300, 2000, 900, 1200, 450)]
    vlabels = [np.array([float(i % 2)], dtype=np.float32) for i in range(5)]
    res = tr.valid((vlabels, vfeats, None, None))
    gathered = [None] * world
    dist.all_gather_object(gathered, dict(after=after, preds=res["predictions"], loss=res["epoch_valid_loss"]))
    if rank == 0:
        torch.save(dict(start=start, ranks=gathered, bags=bags, labels=labels, vfeats=vfeats, vlabels=vlabels), out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
