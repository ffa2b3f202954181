import sys, json
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, slowflow_amd as sfa
from synth import sor_system
W,H,K=1024,436,30
ctx=sfa.Context(0)
rng=np.random.default_rng(0)
s=sor_system(rng,W,H)
planes=[np.ascontiguousarray(s[k]) for k in ("du","dv","a11","a12","a22","b1","b2","sh","sv")]
for B in [int(x) for x in sys.argv[1:]]:
    sb=sfa.SorBatch(ctx,W,H,B)
    for b in range(B): sb.upload(b,*planes)
    sb.run(K,1.9); ctx.sync()
    ctx.profile_enable(True)
    for _ in range(10): sb.run(K,1.9)
    n,ms,by=ctx.profile_read(); ctx.profile_enable(False)
    per=ms/n
    print(f"batch {B:3d}: {per*1e3:8.1f} us/launch  {per*1e3/B:7.1f} us/solve  {by/(ms*1e-3)/1e9:8.1f} GB/s  frac {by/(ms*1e-3)/1e9/8000:.3f}", flush=True)
    sb.close()
