import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, oracle as orc, slowflow_amd as sfa
from synth import sor_system, copy_sys
w,h,K=[int(x) for x in sys.argv[1:4]]
o=orc.Oracle(); ctx=sfa.Context(0)
rng=np.random.default_rng(1)
s0=sor_system(rng,w,h)
a=copy_sys(s0)
o.sor(a["du"],a["dv"],a["a11"],a["a12"],a["a22"],a["b1"],a["b2"],a["sh"],a["sv"],w,K,1.9)
b={k:np.ascontiguousarray(v).copy() for k,v in s0.items()}
ctx.sor_coupled(b["du"],b["dv"],b["a11"],b["a12"],b["a22"],b["b1"],b["b2"],b["sh"],b["sv"],w,K,1.9)
bad=(a["du"][:,:w]!=b["du"][:,:w])
print("mismatch rows:", [(r,int(bad[r].sum()), int(np.argmax(bad[r]))) for r in range(h) if bad[r].any()][:20])
print("nan count", int(np.isnan(b["du"][:,:w]).sum()))
r=[r for r in range(h) if bad[r].any()]
if r:
    r=r[0]; c=int(np.argmax(bad[r])); print("first", r, c, a["du"][r,c], b["du"][r,c], "prev col", a["du"][r,max(c-1,0)], b["du"][r,max(c-1,0)])
