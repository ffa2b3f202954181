"""latency of the drop-in entry point (sfa_variational: one window per call, host buffers in and out)"""
import sys, time
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, slowflow_amd as sfa, bench
ctx=sfa.Context(0)
win=bench.synth_window(0)
avg,std=ctx.normalize(win,bench.W)
p=bench.bench_params()
for k in range(3): p.norm_avg[k]=float("%g"%avg[k]); p.norm_std[k]=float("%g"%std[k])
stride=sfa.stride_of(bench.W)
for i in range(4):
    wx,wy=np.zeros((bench.H,stride),np.float32),np.zeros((bench.H,stride),np.float32)
    t0=time.perf_counter(); ctx.variational(p,wx,wy,win,bench.W); dt=time.perf_counter()-t0
    print(f"call {i}: {dt*1e3:.1f} ms  median flow {np.median(wx[:,:bench.W]):.3f}", flush=True)
ctx.close()
