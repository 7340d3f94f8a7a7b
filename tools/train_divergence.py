"""How far does fp32-CLASS training (split-bf16 x3 products on the matrix cores, ~1e-5 per product) drift from plain-fp32 training
(fp32 library GEMMs, exact VALU attention) over a run?  VERDICT r3 #5: 200 optimizer steps on a fixed synthetic set, same
initial weights, same bag order, same (deterministic) selection rule; records both loss curves and the divergence of the
final weights.   python tools/train_divergence.py [steps] > profiles/r04_train_divergence.md"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snuffy_amd import functional as SF  # noqa: E402
from snuffy_amd.snuffy import build_milnet  # noqa: E402
from snuffy_amd.train import BagParallelStepper  # noqa: E402

DEV = torch.device("cuda")
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N, D, H, LAM, NBAGS = 8192, 384, 6, 200, 16


def run(plain):
    SF.FP32_GEMM = "library" if plain else "x3"
    SF.FP32_ATTENTION = "exact" if plain else "x3"
    torch.manual_seed(0)
    net = build_milnet(D, H, "relu", LAM, 0.0, 1).to(DEV)
    for m in net.modules():          # attention dropout (reference default 0.1) off: the two arithmetics draw their masks through
        if isinstance(m, torch.nn.Dropout):   # different code paths, and the question here is the arithmetic, not the mask stream
            m.p = 0.0
    g = torch.Generator().manual_seed(42)
    bags = [torch.randn(1, N, D, generator=g).to(DEV) for _ in range(NBAGS)]
    # a learnable signal: positive bags carry a shifted cluster of 2 % of their patches
    labels = []
    for i, b in enumerate(bags):
        y = float(i % 2)
        if y:
            b[0, : N // 50] += 0.75
        labels.append(torch.tensor([y], device=DEV))
    st = BagParallelStepper(net, world_size=1, dist=None, device=DEV, precision="fp32")
    losses = []
    for s in range(STEPS):
        losses.append(float(st.step(bags[s % NBAGS], labels[s % NBAGS])))
    return losses, {k: v.detach().double().cpu() for k, v in net.state_dict().items()}


la, wa = run(plain=False)
lb, wb = run(plain=True)
lc, wc = run(plain=False)          # the fp32-class run again: run-to-run reproducibility of the arithmetic under test
print("# fp32-class vs plain-fp32 training, %d AdamW steps (lr 2e-4), %d synthetic bags of %d x %d, Lambda %d, deterministic selection, dropout off\n"
      % (STEPS, NBAGS, N, D, LAM))
print("| step | loss fp32-class | loss plain fp32 | abs diff |\n|---|---|---|---|")
for s in list(range(0, STEPS, max(1, STEPS // 20))) + [STEPS - 1]:
    print("| %d | %.6f | %.6f | %.2e |" % (s, la[s], lb[s], abs(la[s] - lb[s])))
d = [abs(x - y) for x, y in zip(la, lb)]
print("\nmax |loss difference| over the run: %.3e (step %d); mean %.3e; final-epoch mean loss %.5f vs %.5f"
      % (max(d), d.index(max(d)), sum(d) / len(d), sum(la[-NBAGS:]) / NBAGS, sum(lb[-NBAGS:]) / NBAGS))
print("\n| parameter | max |w| | max |dw| fp32-class vs plain | relative to max |w| | moved since init (plain) |\n|---|---|---|---|---|")
torch.manual_seed(0)
init = {k: v.detach().double().cpu() for k, v in build_milnet(D, H, "relu", LAM, 0.0, 1).state_dict().items()}
worst = 0.0
for k in wa:
    if wa[k].numel() == 0 or not wa[k].dtype.is_floating_point:
        continue
    dw = (wa[k] - wb[k]).abs().max().item()
    mx = max(wb[k].abs().max().item(), 1e-30)
    mv = (wb[k] - init[k]).abs().max().item()
    worst = max(worst, dw / mx)
    print("| %s | %.3e | %.3e | %.2e | %.3e |" % (k, mx, dw, dw / mx, mv))
print("\nworst relative weight divergence after %d steps: %.2e" % (STEPS, worst))
same = all(torch.equal(wa[k], wc[k]) for k in wa) and la == lc
print("fp32-class run repeated: %s" % ("bit-identical losses and weights" if same else "NOT bit-identical (max |dw| %.2e)"
                                       % max((wa[k] - wc[k]).abs().max().item() for k in wa if wa[k].numel())))
