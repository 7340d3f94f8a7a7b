import torch
DEV='cuda'
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/it*1e3
n,h,dk,k=32768,6,128,200
d=h*dk
ds=torch.randn(h,n,k,device=DEV,dtype=torch.bfloat16)
qv=torch.randn(n,2*d,device=DEV,dtype=torch.bfloat16)
q=qv[:,:d]
def base():
    qh=q.view(n,h,dk).transpose(0,1)
    return torch.bmm(ds.transpose(1,2),qh).float().transpose(0,1).reshape(k,d)
ref=base()
print("bmm base %.1f us"%t(base))
for c in (4,8,16):
    def chunked():
        dsv=ds.view(h,c,n//c,k).transpose(2,3)            # [h,c,k,n/c]
        qh=q.view(c,n//c,h,dk).permute(2,0,1,3)           # [h,c,n/c,dk]
        return torch.matmul(dsv,qh).sum(1,dtype=torch.float32).transpose(0,1).reshape(k,d)
    out=chunked()
    print("chunk %2d  %.1f us  maxdiff %.3g (scale %.3g)"%(c,t(chunked),(out-ref).abs().max().item(),ref.abs().max().item()))
