import torch, sys
sys.path.insert(0, '.')
from snuffy_amd import ops
DEV = 'cuda'
torch.manual_seed(0)
n, d, h, k = 8192, 384, 6, 200
qv = torch.randn(n, 2 * d, device=DEV)
kp = torch.randn(k, d, device=DEV)
q, v = qv[:, :d], qv[:, d:]
o1, a1, _ = ops.sparse_attn_fwd_x3(q, v, kp, h, need_attn=True)
o2, a2, _ = ops.sparse_attn_fwd_x3(q.contiguous(), v.contiguous(), kp, h, need_attn=True)
print("strided vs contiguous: out", (o1 - o2).abs().max().item(), "attn", (a1 - a2).abs().max().item())
# GEMM x3 at this shape vs fp64
a = torch.randn(n, d, device=DEV)
w = torch.randn(2 * d, d, device=DEV) / d ** 0.5
b = torch.randn(2 * d, device=DEV)
ref = (a.double() @ w.double().t() + b.double())
g = torch.randn(d, device=DEV); bt = torch.randn(d, device=DEV)
a3 = ops.split3_rows(a)
out = ops.gemm_bf16(a3, ops.split3_weight(w), b, "none", torch.float32)
print("gemm x3 err", (out.double() - ref).abs().max().item())
ln = ops.layernorm_rows(a, g, bt, 1e-5)
ln3 = ops.layernorm_rows_split3(a, g, bt, 1e-5)
print("ln3 vs split(ln)", (ln3.float() - ops.split3_rows(ln).float()).abs().max().item())
out2 = ops.gemm_bf16(ln3, ops.split3_weight(w), b, "none", torch.float32)
ref2 = ln.double() @ w.double().t() + b.double()
print("gemm(ln3) err", (out2.double() - ref2).abs().max().item())
