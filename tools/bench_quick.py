"""quick whole-path timing (no SOR-only section, no cpu baseline): prints ms/step"""
import sys, time
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, slowflow_amd as sfa, bench
B=int(sys.argv[1]) if len(sys.argv)>1 else 32
ctx=sfa.Context(0); p=bench.bench_params()
windows=[bench.synth_window(b) for b in range(min(B,4))]
allf=[f for w in windows for f in w]
avg,std=ctx.normalize(allf,bench.W)
for k in range(3): p.norm_avg[k]=float("%g"%avg[k]); p.norm_std[k]=float("%g"%std[k])
job=sfa.Job(ctx,p,bench.W,bench.H,B)
for b in range(B): job.upload(b,windows[b%len(windows)])
job.run(); ctx.sync()
t0=time.perf_counter()
for _ in range(3): job.run()
ctx.sync()
print(f"batch {B}: {(time.perf_counter()-t0)/3*1e3:.2f} ms/step", flush=True)
