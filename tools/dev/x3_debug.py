import torch, sys
sys.path.insert(0, '.')
from snuffy_amd import functional as SF, ops
from tests.helpers import build_amd_milnet
DEV = 'cuda'
torch.manual_seed(0)
n, d = 8192, 384
for act, depth in (("relu", 1), ("gelu", 1), ("gelu", 2)):
    net = build_amd_milnet(d, 6, act, 200, 0.0, depth).to(DEV).eval()
    net.configure(precision="fp32")
    x = torch.randn(1, n, d, device=DEV) * 0.7
    outs = {}
    with torch.no_grad():
        for mode in ("x3", "library"):
            SF.FP32_GEMM = mode
            x2 = x[0]
            c = net.i_classifier(x)[1] if False else None
            feats, classes = net._critic(x)
            x2c, c1 = SF.check_bag(feats, classes)
            enc = net.b_classifier.encoder
            parts, attn = enc.run_layers(x2c, c1)
            z = SF.materialize(parts)
            logits = SF.head(parts, enc.norm, net.b_classifier.linear)
            outs[mode] = (z.clone(), attn.clone(), logits.clone())
    z3, a3, l3 = outs["x3"]; z0, a0, l0 = outs["library"]
    print(act, depth, "z maxdiff", (z3 - z0).abs().max().item(), "z scale", z0.abs().max().item(),
          "mean diff", (z3 - z0).mean().item(), "attn diff", (a3 - a0).abs().max().item(), "logits", l3.flatten().tolist(), l0.flatten().tolist())
    dz = (z3 - z0)
    print("   col-mean diff max", dz.mean(0).abs().max().item(), " row with max diff", dz.abs().max(1).values.argmax().item())
