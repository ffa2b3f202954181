import sys, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, slowflow_amd as sfa, oracle as orc
from synth import sor_system, copy_sys
ctx=sfa.Context(0); O=orc.Oracle()
def c_(x): return np.ascontiguousarray(x)
for F in (2,3):
    os.environ["SFA_SOR_BAND"]=str(F)
    for (w,h,K) in [(130,30,12),(67,98,6),(130,98,12),(130,98,6),(300,70,30),(64,200,6)]:
        if K%F: continue
        rng=np.random.default_rng(1)
        s=sor_system(rng,w,h)
        sb=sfa.SorBatch(ctx,w,h,1)
        sb.upload(0,*[c_(s[k]) for k in ("du","dv","a11","a12","a22","b1","b2","sh","sv")])
        sb.run(K,1.9)
        du,dv=sb.download(0)
        a=copy_sys(s)
        O.sor(a["du"],a["dv"],a["a11"],a["a12"],a["a22"],a["b1"],a["b2"],a["sh"],a["sv"],w,K,1.9)
        bad=(a["du"][:,:w]!=du[:,:w])|(a["dv"][:,:w]!=dv[:,:w])
        rows=np.where(bad.any(axis=1))[0]; cols=np.where(bad.any(axis=0))[0]
        print(f"F={F} {w}x{h} K={K} NB={(h+K-1+63)//64}: bad={bad.sum()} rows={rows[:6]}..{rows[-3:] if len(rows) else ''} cols={cols[:6]}", flush=True)
        sb.close()
