"""one resident frame window of the bench configuration, refined a few times on one stream (what bench.py reports as latency_one_window_ms); meant for
rocprofv3 --kernel-trace + tools/trace_gaps.py.  usage: bench_one_window.py [runs]"""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, slowflow_amd as sfa, bench
runs=int(sys.argv[1]) if len(sys.argv)>1 else 4
ctx=sfa.Context(0); win=bench.synth_window(1)
avg,std=ctx.normalize(win,bench.W); p=bench.bench_params()
for k in range(3): p.norm_avg[k]=float("%g"%avg[k]); p.norm_std[k]=float("%g"%std[k])
job=sfa.Job(ctx,p,bench.W,bench.H,1); job.upload(0,win)
job.run(); ctx.sync()
for i in range(runs):
    t0=time.perf_counter(); job.run(); ctx.sync(); print(f"run {i}: {(time.perf_counter()-t0)*1e3:.3f} ms", flush=True)
