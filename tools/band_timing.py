"""debug build only (-DSFA_BAND_TIMING in sor.hip): where the waves of window 0's bands spend their cycles in one batched solve"""
import sys, ctypes
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, slowflow_amd as sfa
from synth import sor_system
B=int(sys.argv[1]) if len(sys.argv)>1 else 32
ctx=sfa.Context(0); K=30; W,H=1024,436
rng=np.random.default_rng(0); s=sor_system(rng,W,H)
planes=[np.ascontiguousarray(s[k]) for k in ("du","dv","a11","a12","a22","b1","b2","sh","sv")]
sb=sfa.SorBatch(ctx,W,H,B)
for b in range(B): sb.upload(b,*planes)
for _ in range(3): sb.run(K,1.9)
ctx.sync()
out=(ctypes.c_ulonglong*(16*16*6))()
rc=sfa.lib().sfa_debug_band_timing(out)
a=np.array(out,dtype=np.uint64).reshape(16,16,6).astype(np.int64)
t0=a[:8,:10,0].min()
for b in range(8):
    row=[]
    for w in range(10):
        beg,end,up,down,above,first=a[b,w]
        run=end-first
        row.append(f"w{w}: start {int(first-t0)//1000:5d}k run {int(run)//1000:5d}k up {100*up/run:3.0f}% down {100*down/run:3.0f}% above {100*above/run:3.0f}%")
    print("band",b); print("   "+"\n   ".join(row))
