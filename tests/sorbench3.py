import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, slowflow_amd as sfa
from synth import sor_system
ctx=sfa.Context(0)
W,H,B,K=[int(x) for x in sys.argv[1:5]]
rng=np.random.default_rng(0)
s=sor_system(rng,W,H)
planes=[np.ascontiguousarray(s[k]) for k in ("du","dv","a11","a12","a22","b1","b2","sh","sv")]
sb=sfa.SorBatch(ctx,W,H,B)
for b in range(B): sb.upload(b,*planes)
for _ in range(3): sb.run(K,1.9)
ctx.sync()
print("done")
