"""per-kernel GPU time of one bench step (one stream, batch windows) from HIP events around job.run with rocprof-free timing is not available: use rocprofv3 stats instead.
usage: rocprofv3 --kernel-trace --stats -- python tools/bench_kernels.py [batch]"""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, slowflow_amd as sfa, bench
B=int(sys.argv[1]) if len(sys.argv)>1 else 64
ctx=sfa.Context(0)
p=bench.bench_params()
wins=[bench.synth_window(b) for b in range(min(B,4))]
avg,std=ctx.normalize([f for w in wins for f in w], bench.W)
for k in range(3): p.norm_avg[k]=float("%g"%avg[k]); p.norm_std[k]=float("%g"%std[k])
job=sfa.Job(ctx,p,bench.W,bench.H,B)
for b in range(B): job.upload(b,wins[b%len(wins)])
for _ in range(4): job.run()
ctx.sync()
