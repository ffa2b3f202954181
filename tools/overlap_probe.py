"""Does an HBM-bound streaming kernel run beside the solver for free?  Stream A: the real solver (SorBatch, 64 windows of 1024x436, K = 30) launch after launch;
stream B: a plain streaming kernel (torch: c = a + b over 3 x 256 MiB) launch after launch.  Each alone, then both at once from two host threads."""
import sys, os, time, threading
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, torch, slowflow_amd as sfa
from synth import sor_system
B=int(sys.argv[1]) if len(sys.argv)>1 else 64
ctx=sfa.Context(0)
W,H,K=1024,436,30
rng=np.random.default_rng(0); s=sor_system(rng,W,H)
planes=[np.ascontiguousarray(s[k]) for k in ("du","dv","a11","a12","a22","b1","b2","sh","sv")]
sb=sfa.SorBatch(ctx,W,H,B)
for b in range(B): sb.upload(b,*planes)
n=64*1024*1024
a=torch.rand(n,device='cuda'); b_=torch.rand(n,device='cuda'); c=torch.empty(n,device='cuda')
st=torch.cuda.Stream()
done={}
def solver(N):
    t=time.perf_counter()
    for _ in range(N): sb.run(K,1.9)
    ctx.sync(); done['solver']=time.perf_counter()-t
def stream(N):
    t=time.perf_counter()
    with torch.cuda.stream(st):
        for _ in range(N): torch.add(a,b_,out=c)
    st.synchronize(); done['stream']=time.perf_counter()-t
solver(3); stream(3)
NS,NT=40,400
t0=time.perf_counter(); solver(NS); ts=time.perf_counter()-t0
t0=time.perf_counter(); stream(NT); tt=time.perf_counter()-t0
print(f"alone: solver {ts/NS*1e3:.3f} ms per launch; streaming {tt/NT*1e3:.3f} ms per launch = {3*4*n/(tt/NT)/1e12:.2f} TB/s")
# both: as many streaming launches as fit the solver's alone time x 2
t0=time.perf_counter()
th=[threading.Thread(target=solver,args=(NS,)),threading.Thread(target=stream,args=(NT,))]
for t in th: t.start()
for t in th: t.join()
tb=time.perf_counter()-t0
print(f"   solver thread done after {done['solver']*1e3:.1f} ms, streaming thread after {done['stream']*1e3:.1f} ms")
print(f"both at once: {tb*1e3:.1f} ms against {ts*1e3:.1f} + {tt*1e3:.1f} = {(ts+tt)*1e3:.1f} ms one after the other (max of the two alone: {max(ts,tt)*1e3:.1f})")
