"""config 4 per-GPU load: cfgs/slow_flow.cfg schedule (S=3, 5 levels, 10 alternations x 10 outer x 30 sweeps, occlusion reasoning,
thresholds 1e-5) on 16 windows of 1024x436 (64 jets x 2 directions over 8 GPUs)"""
import sys, time
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, slowflow_amd as sfa, bench
B=int(sys.argv[1]) if len(sys.argv)>1 else 16
ctx=sfa.Context(0)
windows=[bench.synth_window(b, n=5) for b in range(min(B,4))]
allf=[f for w in windows for f in w]
avg,std=ctx.normalize(allf,bench.W)
p=sfa.default_params()
p.S=3; p.layers=5; p.hbit=0
p.rho[0]=1; p.rho[1]=1; p.omega[0]=0; p.omega[1]=2
for k in range(3): p.norm_avg[k]=float("%g"%avg[k]); p.norm_std[k]=float("%g"%std[k])
print("alter",p.niter_alter,"outer",p.niter_outer,"occ",p.occlusion_reasoning,"thres",p.thres_outer,p.thres_inner, flush=True)
NS=int(sys.argv[2]) if len(sys.argv)>2 else 1                 # lockstep groups on separate streams (B windows in total)
import threading
ctxs=[ctx]+[sfa.Context(0) for _ in range(NS-1)]
BL=B//NS
jobs=[sfa.Job(c,p,bench.W,bench.H,BL) for c in ctxs]
for g,job in enumerate(jobs):
    for b in range(BL): job.upload(b,windows[(g*BL+b)%len(windows)])
def run_all():
    def work(g): jobs[g].run(); ctxs[g].sync()
    th=[threading.Thread(target=work,args=(g,)) for g in range(NS)]
    for t in th: t.start()
    for t in th: t.join()
t0=time.perf_counter(); run_all(); t1=time.perf_counter()
print(f"first run {1e3*(t1-t0):.1f} ms", flush=True)
t0=time.perf_counter(); run_all(); t1=time.perf_counter()
print(f"batch {B} in {NS} group(s): {1e3*(t1-t0):.1f} ms per run = {1e3*(t1-t0)/B:.1f} ms per window", flush=True)
wx,wy,ch=jobs[0].download(0)
print("median flow", np.median(wx[:,:bench.W]), np.median(wy[:,:bench.W]), "change", ch)
