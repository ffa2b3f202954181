"""SOR only, 1024x436 (or W H from the environment), K = 30: launch time per batch size and solver shape; every shape's result is compared bit for bit
with the first one's.  usage: bench_sor_chain.py "1 2 4 8 16 32" "0 1 2 5 3"   (batches, SFA_SOR_CHAIN ids; 0 = the band / task kernels)"""
import sys
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, slowflow_amd as sfa
from synth import sor_system
ctx=sfa.Context(0)
K=int(os.environ.get('SOR_K','30'))
W,H=int(os.environ.get('SOR_W','1024')),int(os.environ.get('SOR_H','436'))
batches=[int(x) for x in (sys.argv[1] if len(sys.argv)>1 else "1 4 16").split()]
shapes=[int(x) for x in (sys.argv[2] if len(sys.argv)>2 else "0 1 2").split()]
rng=np.random.default_rng(0)
s=sor_system(rng,W,H)
s["du"][:, :W] = rng.uniform(-.2, .2, (H, W)); s["dv"][:, :W] = rng.uniform(-.2, .2, (H, W))
planes=[np.ascontiguousarray(s[k]) for k in ("du","dv","a11","a12","a22","b1","b2","sh","sv")]
for B in batches:
    ref=None
    for sh in shapes:
        sfa.debug_set("SFA_SOR_CHAIN", sh)
        sb=sfa.SorBatch(ctx,W,H,B)
        for b in range(B): sb.upload(b,*planes)
        sb.run(K,1.9); ctx.sync()
        out=sb.download(B-1)
        if ref is None: ref=out
        same=bool(np.array_equal(out[0][:, :W],ref[0][:, :W]) and np.array_equal(out[1][:, :W],ref[1][:, :W]))
        ts=[]
        for rep in range(3):
            for b in range(B): sb.upload(b,*planes)
            ctx.profile_enable(True)
            sb.run(K,1.9)
            n,ms,by=ctx.profile_read(); ctx.profile_enable(False)
            ts.append(ms/n)
        per=min(ts)
        print(f"K={K} {W}x{H} batch {B:3d} chain {sh}: {per*1e3:8.1f} us/launch  {per*1e3/B:7.1f} us/solve  (runs {[round(t*1e3) for t in ts]})  same_bits={same}", flush=True)
        sb.close()
