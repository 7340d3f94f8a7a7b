#!/usr/bin/env python3
"""One packed configuration in a loop (for rocprofv3): python tools/varlen_one.py N D precision B [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_net  # noqa: E402

N, D, prec, B = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 50
dev = torch.device("cuda", 0)
net = build_net(D, 6, 200, prec, dev).eval()
g = torch.Generator().manual_seed(1)
bags = [torch.randn(1, N, D, generator=g).to(dev) for _ in range(B)]
with torch.no_grad():
    for _ in range(steps):
        net.forward_bags(bags)
torch.cuda.synchronize()
