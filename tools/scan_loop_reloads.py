#!/usr/bin/env python3
"""Kernels that re-read spilled registers INSIDE an MFMA loop, and how many of those reloads are followed by s_waitcnt vmcnt(0) -- which also
drains whatever operand load was issued just before, so the loads of the next tile go out one memory round trip after the other (round 5:
gemm_hl_kernel, bwd_dq_dv_lds_kernel and the dropout instantiation of sparse_attn_x3_kernel lost 3 - 10 % to exactly this; the cure was to
make the spilled address terms cheaper to recompute than to keep -- an opaque copy of the lane index inside the address computation).
Compiles every attention / GEMM translation unit to ISA with the build's flags (a few minutes) and walks the loops.
usage: python tools/scan_loop_reloads.py"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snuffy_amd import build as B
res=[]
for src in sorted(glob.glob(os.path.join(ROOT, 'snuffy_amd', 'csrc', '*.hip'))):
    base=os.path.basename(src)
    if base in ('core.hip','sampler.hip','tiles.hip','vit.hip','rowops.hip'): continue
    flags=[f for f in B.FLAGS if f not in ('-fPIC',)]+B.EXTRA_FLAGS.get(base,[])
    out='/tmp/ls_%s.s'%base
    subprocess.run([B._hipcc()]+flags+['-S','--cuda-device-only',src,'-o',out],capture_output=True,text=True)
    if not os.path.exists(out): continue
    s=open(out).read()
    for m in re.finditer(r'^(_Z\w+):\s*; @', s, re.M):
        name=m.group(1); i=m.end(); j=s.find('.Lfunc_end', i)
        body=s[i:j].split('\n')
        sl=[n for n,l in enumerate(body) if 'scratch_load' in l]
        if not sl: continue
        labels={}
        for n,l in enumerate(body):
            mm=re.match(r'^(\.LBB\d+_\d+):', l)
            if mm: labels[mm.group(1)]=n
        loops=[]
        for n,l in enumerate(body):
            mm=re.search(r's_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
            if mm:
                t=mm.group(1) or mm.group(2)
                if t in labels and labels[t]<n: loops.append((labels[t],n))
        mf=[n for n,l in enumerate(body) if 'v_mfma' in l]
        # loops that contain at least one mfma
        mloops=[lp for lp in loops if any(lp[0]<x<lp[1] for x in mf)]
        inl=[n for n in sl if any(a<n<b for a,b in mloops)]
        # of those, followed within 2 lines by vmcnt(0)
        ser=[n for n in inl if any('vmcnt(0)' in body[k] for k in range(n+1,min(n+4,len(body))))]
        if inl: res.append((base,name,len(sl),len(inl),len(ser)))
filt='/usr/bin/c++filt'
for base,name,a,b,c in res:
    d=subprocess.run([filt,name],capture_output=True,text=True).stdout.strip()
    print("%-32s loads %3d in-mfma-loop %3d serialised %3d  %s"%(base,a,b,c,d[:120]))
