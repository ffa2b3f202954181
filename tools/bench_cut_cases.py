"""time of one exact two-label cut (sfa_grid_cut, one window) on the test suite's cost patterns at 1024x436"""
import sys, time
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, slowflow_amd as sfa
from test_gpu_parity import _cut_case
ctx=sfa.Context(0)
w,h=1024,436
for kind in ("blobs","stripes","noise"):
    for alpha in (0.1,0.5,2.0):
        rng=np.random.default_rng(1)
        d0,d1=_cut_case(rng,w,h,kind)
        d0=np.ascontiguousarray(d0); d1=np.ascontiguousarray(d1)
        ctx.grid_cut(d0,d1,alpha,w)
        t0=time.perf_counter(); occ=ctx.grid_cut(d0,d1,alpha,w); dt=time.perf_counter()-t0
        print(f"{kind:8s} alpha {alpha}: {dt*1e3:8.1f} ms  label+1 share {(occ[:,:w]>0).mean():.3f}", flush=True)
