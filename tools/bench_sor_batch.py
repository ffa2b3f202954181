import sys
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, slowflow_amd as sfa
from synth import sor_system
ctx=sfa.Context(0)
K=int(os.environ.get('SOR_K','30'))
W,H=1024,436
for B in [int(x) for x in sys.argv[1:]] or [32,64,128]:
    rng=np.random.default_rng(0)
    s=sor_system(rng,W,H)
    planes=[np.ascontiguousarray(s[k]) for k in ("du","dv","a11","a12","a22","b1","b2","sh","sv")]
    sb=sfa.SorBatch(ctx,W,H,B)
    for b in range(B): sb.upload(b,*planes)
    sb.run(K,1.9); ctx.sync()
    ctx.profile_enable(True)
    for _ in range(10): sb.run(K,1.9)
    n,ms,by=ctx.profile_read(); ctx.profile_enable(False)
    per=ms/n
    print(f"K={K} {W}x{H} batch {B:3d}: {per*1e3:8.1f} us/launch  {per*1e3/B:6.1f} us/solve  {by/n/(per*1e-3)/1e9:7.0f} GB/s alg", flush=True)
    sb.close()
