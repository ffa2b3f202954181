"""BASELINE config 5 (2048x2048, 6 levels, Lorentzian penalties, 5 outer x 30 sweeps) on one stream: a warm-up and one measured refinement of `batch` windows; meant
to run under rocprofv3 (profiles/collect_config5.sh) for the per-kernel HBM report.  usage: bench_config5_step.py [batch]"""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,os.path.join(ROOT,'tools'))
import numpy as np, slowflow_amd as sfa, bench
from bench_configs import cfg5
B=int(sys.argv[1]) if len(sys.argv)>1 else 32
w=h=2048
ctx=sfa.Context(0)
wins=[bench.synth_window(b,w=w,h=h,n=3) for b in range(2)]
avg,std=ctx.normalize([f for wd in wins for f in wd],w)
p=cfg5(sfa.default_params())
for k in range(3): p.norm_avg[k]=float("%g"%avg[k]); p.norm_std[k]=float("%g"%std[k])
job=sfa.Job(ctx,p,w,h,B)
for b in range(B): job.upload(b,wins[b%2])
job.run(); ctx.sync()
t0=time.perf_counter(); job.run(); ctx.sync(); dt=time.perf_counter()-t0
print(f"config 5, batch {B}: {dt*1e3:.1f} ms per refinement, {dt*1e3/B:.2f} ms per window, {job.mpix_iters()/dt:.0f} Mpix*iters/s, {job.device_bytes()/B/1e9:.2f} GB per window", flush=True)
