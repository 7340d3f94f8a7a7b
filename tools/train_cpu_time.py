"""Is the training step bound by the host?  Time to ENQUEUE a step (no synchronisation inside the loop) next to the time the GPU
needs for it, config B, one rank.   python tools/train_cpu_time.py [fp32|bf16]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from snuffy_amd.train import BagParallelStepper  # noqa: E402


def main():
    precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    dev = torch.device("cuda:0")
    net = bench.build_net(768, 6, 200, precision, dev)
    st = BagParallelStepper(net, world_size=1, dist=None, device=dev, precision=precision)
    g = torch.Generator().manual_seed(1)
    bags = [torch.randn(1, 32768, 768, generator=g).to(dev) for _ in range(4)]
    lab = [torch.tensor([float(i % 2)], device=dev) for i in range(4)]
    for i in range(8):
        st.step(bags[i % 4], lab[i % 4])
    torch.cuda.synchronize()
    n = 40
    t0 = time.perf_counter()
    for i in range(n):
        st.step(bags[i % 4], lab[i % 4])
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    # host time of a step with the GPU kept idle-free: synchronise BEFORE each step so that nothing queues up, time the enqueue only
    host = 0.0
    for i in range(10):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        st.step(bags[i % 4], lab[i % 4])
        host += time.perf_counter() - t1
    print("%s: %d steps enqueued in %.2f ms/step, finished in %.2f ms/step; host time of one step with an empty queue %.2f ms"
          % (precision, n, t_enq / n * 1e3, t_all / n * 1e3, host / 10 * 1e3))


if __name__ == "__main__":
    main()
