import sys, torch
sys.path.insert(0, '.')
from snuffy_amd import ops
def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/it*1e3
M=100864
for (n,k,act) in ((64,384,"relu"),(384,64,"none"),(32,384,"relu"),(384,32,"none")):
    a=torch.randn(M,k,device='cuda',dtype=torch.bfloat16); w=torch.randn(n,k,device='cuda',dtype=torch.bfloat16); b=torch.randn(n,device='cuda'); bb=b.to(torch.bfloat16)
    if act=="relu": lib=lambda: torch._addmm_activation(bb, a, w.t())
    else: lib=lambda: torch.addmm(bb, a, w.t())
    line="n=%d k=%d %s: library %.1f us"%(n,k,act,t(lib))
    if ops.gemm_supported(M,n,k):
        for tn in (128,256):
            try: line+=" | ours tile %d %.1f us"%(tn,t(lambda: ops.gemm_bf16(a,w,b,act,tile_n=tn)))
            except Exception as ex: line+=" | ours %d err"%tn
    print(line)
