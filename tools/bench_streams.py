"""two (or more) jobs on separate contexts/streams, driven by host threads: do their kernels fill each other's idle CUs?"""
import sys, time, threading
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, slowflow_amd as sfa, bench
B=int(sys.argv[1]) if len(sys.argv)>1 else 32
NS=int(sys.argv[2]) if len(sys.argv)>2 else 2
ctxs=[sfa.Context(0) for _ in range(NS)]
windows=[bench.synth_window(b) for b in range(4)]
allf=[f for w in windows for f in w]
avg,std=ctxs[0].normalize(allf,bench.W)
p=bench.bench_params()
if os.environ.get('SOLVER_K'): p.niter_solver=int(os.environ['SOLVER_K'])   # experiment: K=15 band workgroups (77 KB LDS) can share a CU with the assembly's
for k in range(3): p.norm_avg[k]=float("%g"%avg[k]); p.norm_std[k]=float("%g"%std[k])
jobs=[sfa.Job(c,p,bench.W,bench.H,B) for c in ctxs]
for j in jobs:
    for b in range(B): j.upload(b,windows[b%4])
DELAY=float(sys.argv[3])*1e-3 if len(sys.argv)>3 else 0.0
def run(i,n):
    if i: time.sleep(DELAY*i)
    for _ in range(n): jobs[i].run()
    ctxs[i].sync()
for i in range(NS): run(i,1)
t0=time.perf_counter()
th=[threading.Thread(target=run,args=(i,3)) for i in range(NS)]
for t in th: t.start()
for t in th: t.join()
dt=(time.perf_counter()-t0)/3*1e3
print(f"{NS} streams x batch {B}: {dt:.2f} ms per step of {NS*B} windows = {dt/(NS*B):.3f} ms/window", flush=True)

for j in jobs: j.close()
for c in ctxs: c.close()
