"""Throughput of the other BASELINE configurations (they are parity cases, not bench lines; this is for the record):
   config 3 stand-in 2560x1440 with the cfgs/slow_flow.cfg schedule, config 4's per-GPU share (16 windows of 1024x436, same schedule),
   config 5 2048x2048, 6 levels, Lorentzian.  Synthetic windows as in bench.py."""
import sys, time
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, slowflow_amd as sfa, bench

def cfg_schedule(p):            # cfgs/slow_flow.cfg: S=3, rho 1/1, omega 0/2, 5 levels, 10 alternations x 10 outer x 30 sweeps, occlusion reasoning, thresholds 1e-5
    p.S=3; p.layers=5; p.hbit=0; p.rho[0]=1; p.rho[1]=1; p.omega[0]=0; p.omega[1]=2
    return p

def run(name, w, h, nframes, batch, setup):
    ctx=sfa.Context(0)
    windows=[bench.synth_window(b, w=w, h=h, n=nframes) for b in range(min(batch,2))]
    avg,std=ctx.normalize([f for wd in windows for f in wd], w)
    p=setup(sfa.default_params())
    for k in range(3): p.norm_avg[k]=float("%g"%avg[k]); p.norm_std[k]=float("%g"%std[k])
    job=sfa.Job(ctx,p,w,h,batch)
    for b in range(batch): job.upload(b,windows[b%len(windows)])
    job.run(); ctx.sync()
    t0=time.perf_counter(); job.run(); ctx.sync(); dt=time.perf_counter()-t0
    wx,wy,ch=job.download(0)
    print(f"{name}: {w}x{h}, {nframes} frames, batch {batch}: {dt*1e3:.1f} ms per run, {dt*1e3/batch:.2f} ms per window, scheduled {job.mpix_iters()/dt:.0f} Mpix*iters/s "
          f"(median flow {np.median(wx[:,:w]):.2f},{np.median(wy[:,:w]):.2f})", flush=True)
    job.close(); ctx.close()

def cfg5(p):
    p.S=2; p.layers=6; p.niter_alter=1; p.niter_outer=5; p.niter_inner=1; p.niter_solver=30; p.thres_outer=0; p.thres_inner=0; p.occlusion_reasoning=0; p.hbit=0
    p.rho[0]=1; p.omega[0]=0
    for pen in (p.robust_color, p.robust_grad, p.robust_reg): pen.id=2; pen.eps=0.05
    return p

if __name__ == "__main__":
    only = sys.argv[1:] or ["4", "3", "5"]
    if "4" in only:
        run("config 4 share (cfg schedule, thresholds may stop iterations early)", 1024, 436, 5, 16, cfg_schedule)
    if "3" in only:
        run("config 3 stand-in (cfg schedule)", 2560, 1440, 5, 4, cfg_schedule)
        run("config 3 stand-in (cfg schedule)", 2560, 1440, 5, 8, cfg_schedule)
    if "5" in only:
        run("config 5 (2048x2048, 6 levels, Lorentzian, 5 outer x 30 sweeps)", 2048, 2048, 3, 8, cfg5)
        run("config 5 (2048x2048, 6 levels, Lorentzian, 5 outer x 30 sweeps)", 2048, 2048, 3, 32, cfg5)
