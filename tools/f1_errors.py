import sys, numpy as np, torch
sys.path.insert(0,'.')
from tests.helpers import build_amd_milnet, golden_files, load_case
for path in golden_files("f1_"):
    z, sd = load_case(path)
    N, D, h, lam, depth, seed = [int(v) for v in z["cfg"]]
    out=[]
    for precision in ("fp32","bf16"):
        net = build_amd_milnet(D, h, str(z["act"]), lam, float(z["r"]), depth); net.load_state_dict(sd, strict=True)
        net = net.to("cuda").eval().configure(precision=precision, return_attention=True)
        x = torch.from_numpy(z["x"]).to("cuda"); np.random.seed(seed)
        with torch.no_grad(): c, l, A = net(x)
        out.append((float(np.abs(l.cpu().numpy()-z["logits"]).max()), float(np.abs(A.cpu().numpy()-z["A"]).max()) if "A" in z else -1))
    print(path.split('/')[-1], "depth",depth,"K",min(lam,N), "fp32 dlogit %.2e dA %.2e | bf16 dlogit %.2e dA %.2e"%(out[0]+out[1]))
