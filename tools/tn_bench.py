import torch, time
DEV='cuda'
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/it*1e3
n=32768
for (m_, k_) in ((3072,768),(768,3072),(1536,768)):
    a=torch.randn(n,m_,device=DEV,dtype=torch.bfloat16); b=torch.randn(n,k_,device=DEV,dtype=torch.bfloat16)
    ref=None
    print("dW [%d,%d] K=%d"%(m_,k_,n))
    us=t(lambda: torch.mm(a.t(), b)); print("   mm TN            %7.1f us  %6.1f TF"%(us, 2*n*m_*k_/us/1e6))
    for c in (2,4,8,16):
        av=a.view(c,n//c,m_).transpose(1,2); bv=b.view(c,n//c,k_)
        us=t(lambda: torch.bmm(av,bv).sum(0,dtype=torch.float32)); print("   bmm split %2d + sum %7.1f us  %6.1f TF"%(c,us, 2*n*m_*k_/us/1e6))
    try:
        us=t(lambda: torch.mm(a.t(), b, out_dtype=torch.float32)); print("   mm TN f32 out    %7.1f us"%us)
    except Exception as ex:
        print("   out_dtype unsupported:", str(ex)[:80])
    at=a.t().contiguous()
    us=t(lambda: a.t().contiguous()); print("   transpose a      %7.1f us"%us)
    bt=b.t().contiguous()
    us=t(lambda: torch.mm(at, bt.t())); print("   mm NT (pre-transposed both K-major) %7.1f us"%us)
