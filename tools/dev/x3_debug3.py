import torch, sys
import torch.nn.functional as F
sys.path.insert(0, '.')
from snuffy_amd import functional as SF, ops
from tests.helpers import build_amd_milnet
DEV = 'cuda'
torch.manual_seed(0)
n, d, h = 8192, 384, 6
net = build_amd_milnet(d, h, "relu", 200, 0.0, 1).to(DEV).eval()
net.configure(precision="fp32")
x = torch.randn(1, n, d, device=DEV) * 0.7
with torch.no_grad():
    feats, classes = net._critic(x)
    x2, c1 = SF.check_bag(feats, classes)
    layer = net.b_classifier.encoder.layers[0]
    sel = layer.select(c1, n)[0]
    print("sel", sel.shape, sel.dtype, sel[:5].tolist())
    mha, ff = layer.self_attn, layer.feed_forward
    n0, n1 = layer.sublayer[0].norm, layer.sublayer[1].norm
    lq, lk, lv, lo = mha.linears
    fw = SF._split_weights(layer)
    xs, slot = ops.gather_slot_map(x2, sel)
    kp = F.linear(xs, lk.weight, lk.bias)
    xn = ops.layernorm_rows(x2, n0.weight, n0.bias, n0.eps)
    xn3 = ops.layernorm_rows_split3(x2, n0.weight, n0.bias, n0.eps)
    print("xn3", (xn3.float() - ops.split3_rows(xn).float()).abs().max().item())
    q0 = F.linear(xn, lq.weight, lq.bias); v0 = F.linear(xn, lv.weight, lv.bias)
    qv = ops.gemm_bf16(xn3, fw["wqv"], fw["bqv"], out_dtype=torch.float32)
    print("q", (qv[:, :d] - q0).abs().max().item(), "v", (qv[:, d:] - v0).abs().max().item())
    qd = xn.double() @ lq.weight.double().t() + lq.bias.double()
    print("q0 vs f64", (q0.double() - qd).abs().max().item(), "q3 vs f64", (qv[:, :d].double() - qd).abs().max().item())
    o3, a3, _ = ops.sparse_attn_fwd_x3(qv[:, :d], qv[:, d:], kp, h, need_attn=True)
    o0, a0, _ = ops.sparse_attn_fwd_x3(q0, v0, kp, h, need_attn=True)
    oe, ae, _ = ops.sparse_attn_fwd(q0, kp, v0, h, need_attn=True)
    print("attn x3(q3) vs x3(q0)", (a3 - a0).abs().max().item(), "x3(q0) vs exact", (a0 - ae).abs().max().item(), "o", (o3 - o0).abs().max().item(), (o0 - oe).abs().max().item())
