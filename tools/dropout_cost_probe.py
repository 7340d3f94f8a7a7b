import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from snuffy_amd.train import BagParallelStepper
dev = torch.device("cuda:0")
for p in (0.1, 0.0):
    net = bench.build_net(768, 6, 200, "fp32", dev)
    for l in net.b_classifier.encoder.layers:
        l.self_attn.dropout.p = p
    st = BagParallelStepper(net, world_size=1, dist=None, device=dev, precision="fp32")
    g = torch.Generator().manual_seed(1)
    bags = [torch.randn(1, 32768, 768, generator=g).to(dev) for _ in range(4)]
    lab = [torch.tensor([float(i % 2)], device=dev) for i in range(4)]
    for i in range(10): st.step(bags[i % 4], lab[i % 4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(40): st.step(bags[i % 4], lab[i % 4])
    torch.cuda.synchronize()
    print("attention dropout p=%.1f: %.3f ms/step" % (p, (time.perf_counter() - t0) / 40 * 1e3))
