"""cost of the occlusion step: whole-path step with niter_alter=2, occlusion reasoning on vs off"""
import sys, time
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, slowflow_amd as sfa, bench
B=int(sys.argv[1]) if len(sys.argv)>1 else 32
ctx=sfa.Context(0)
windows=[bench.synth_window(b) for b in range(min(B,4))]
allf=[f for w in windows for f in w]
avg,std=ctx.normalize(allf,bench.W)
for occ in (0,1):
    p=bench.bench_params(); p.niter_alter=2; p.occlusion_reasoning=occ
    for k in range(3): p.norm_avg[k]=float("%g"%avg[k]); p.norm_std[k]=float("%g"%std[k])
    job=sfa.Job(ctx,p,bench.W,bench.H,B)
    for b in range(B): job.upload(b,windows[b%len(windows)])
    job.run(); ctx.sync()
    t0=time.perf_counter()
    for _ in range(3): job.run()
    ctx.sync()
    dt=(time.perf_counter()-t0)/3*1e3
    print(f"batch {B} occlusion_reasoning={occ}: {dt:.2f} ms/step", flush=True)
    if occ:
        wx,wy,ch=job.download(0)
    job.close()
