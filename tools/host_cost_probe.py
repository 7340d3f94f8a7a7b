"""Host time per launch of a few op wrappers (no synchronisation inside the loop): where a host-bound training step spends its 2 ms.
   python tools/host_cost_probe.py   (round 6, MI355X box: torch.add 4.9 us, torch.mm bf16 18.9, ops.layernorm_rows 7.2, ops.colsum_fused 13.1, raw ctypes call 5.1)"""
import time, torch, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from snuffy_amd import ops, _ffi
dev='cuda:0'
x=torch.randn(256,768,device=dev); w=torch.randn(768,768,device=dev); b=torch.randn(768,device=dev)
xb=x.to(torch.bfloat16); wb=w.to(torch.bfloat16)
def t(fn,n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    dt=(time.perf_counter()-t0)/n*1e6
    torch.cuda.synchronize(); return dt
print('torch.add            %.1f us' % t(lambda: x+x))
print('torch.mm bf16        %.1f us' % t(lambda: torch.mm(xb,wb)))
print('F.linear f32         %.1f us' % t(lambda: torch.nn.functional.linear(x,w,b)))
print('ops.layernorm_rows   %.1f us' % t(lambda: ops.layernorm_rows(x,None,None,1e-5)))
print('ops.colsum_fused     %.1f us' % t(lambda: ops.colsum_fused(x)))
print('ops.split3_rows      %.1f us' % t(lambda: ops.split3_rows(x)))
lib=_ffi.load()
out=torch.empty(256,3*768,dtype=torch.bfloat16,device=dev)
s=torch.cuda.current_stream().cuda_stream
print('raw ctypes split3    %.1f us' % t(lambda: lib.snf_split3_f32(x.data_ptr(), 768, 256, 768, out.data_ptr(), s)))
print('torch.empty          %.1f us' % t(lambda: torch.empty(256,3*768,dtype=torch.bfloat16,device=dev)))
print('current_stream       %.1f us' % t(lambda: torch.cuda.current_stream().cuda_stream))
