"""sustained-load determinism check: two lockstep groups on two streams refine the bench workload N times; after every pass the flow of a few windows of each
group must equal the first pass bit for bit (a stale hand-over in the band pipeline would show as a changed bit or a bounded-wait error).
usage: soak.py [passes=100] [batch per group=64] [cfg]      (cfg: the driver's default schedule -- alternations with the occlusion cut, break thresholds, passengers)"""
import sys, os, threading, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, slowflow_amd as sfa, bench
N=int(sys.argv[1]) if len(sys.argv)>1 else 100
B=int(sys.argv[2]) if len(sys.argv)>2 else 64
ctxs=[sfa.Context(0) for _ in range(2)]
p=bench.bench_params()
if len(sys.argv)>3 and sys.argv[3]=='cfg':
    p=sfa.default_params(); p.S=bench.S; p.layers=bench.LAYERS; p.hbit=0
wins=[bench.synth_window(b) for b in range(4)]
avg,std=ctxs[0].normalize([f for w in wins for f in w], bench.W)
for k in range(3): p.norm_avg[k]=float("%g"%avg[k]); p.norm_std[k]=float("%g"%std[k])
jobs=[sfa.Job(c,p,bench.W,bench.H,B) for c in ctxs]
for g,job in enumerate(jobs):
    for b in range(B): job.upload(b,wins[(b+g)%4])
probe=sorted(set(min(b,B-1) for b in (0,1,B//2,B-1)))
ref=[None,None]; bad=[0,0]; err=[None,None]
def work(g):
    try:
        for it in range(N):
            jobs[g].run(); ctxs[g].sync()
            cur=[jobs[g].download(b)[:2] for b in probe]
            if it==0: ref[g]=cur
            else:
                for (a0,a1),(b0,b1) in zip(ref[g],cur):
                    if not (np.array_equal(a0,b0) and np.array_equal(a1,b1)): bad[g]+=1
    except Exception as e:
        err[g]=repr(e)
t0=time.perf_counter()
th=[threading.Thread(target=work,args=(g,)) for g in range(2)]
for t in th: t.start()
for t in th: t.join()
print(f"{N} passes x 2 groups of {B} windows in {time.perf_counter()-t0:.1f} s: mismatching downloads {bad}, errors {err}", flush=True)
for j in jobs: j.close()
for c in ctxs: c.close()
sys.exit(1 if (sum(bad) or any(err)) else 0)
