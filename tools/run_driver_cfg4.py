"""BASELINE config 4 through the drop-in driver on the GPUs of this box: cfgs/slow_flow.cfg schedule (S=3, 5 layers, 10 alternations x 10 outer
x 30 sweeps, occlusion reasoning) over JETS consecutive high-speed frame pairs of a synthetic 1024x436 sequence (PPM files), forward and
backward: ./slow_flow cfg -> .flo files.  Reports the driver's wall time including file I/O.  usage: run_driver_cfg4.py [jets=64] [outdir]"""
import os, sys, time, subprocess, tempfile, json
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np
from scipy.ndimage import gaussian_filter
JETS=int(sys.argv[1]) if len(sys.argv)>1 else 64
out=sys.argv[2] if len(sys.argv)>2 else tempfile.mkdtemp(prefix="sfa_cfg4_")
W,H,S=1024,436,3
steps=S-1; nframes=1+(JETS+2)*steps
rng=np.random.default_rng(0); pad=96
base=gaussian_filter(rng.uniform(0,1,size=(3,H+2*pad,W+2*pad)),sigma=(0,2.0,2.0),mode="nearest")
base=(base-base.min())/(base.max()-base.min())*255.0
yy,xx=np.mgrid[0:H,0:W].astype(np.float64)
fu=0.5+0.15*np.sin(2*np.pi*yy/H); fv=0.15*np.cos(2*np.pi*xx/W)      # high-speed camera: sub-pixel motion per frame
t0=time.perf_counter()
os.makedirs(out,exist_ok=True)
for t in range(nframes):
    sx,sy=xx-t*fu+pad,yy-t*fv+pad
    x0,y0=np.floor(sx).astype(int),np.floor(sy).astype(int); ax,ay=sx-x0,sy-y0
    img=np.empty((H,W,3),np.uint8)
    for c in range(3):
        b=base[c]
        img[...,c]=np.clip(np.round(b[y0,x0]*(1-ax)*(1-ay)+b[y0,x0+1]*ax*(1-ay)+b[y0+1,x0]*(1-ax)*ay+b[y0+1,x0+1]*ax*ay),0,255)
    with open(os.path.join(out,"f_%04d.ppm"%(100-steps+t)),"wb") as f:
        f.write(b"P6\n%d %d\n255\n"%(W,H)); f.write(img.tobytes())
print(f"{nframes} frames written in {time.perf_counter()-t0:.1f} s", flush=True)
ref=open(os.path.join(ROOT,"cfgs","slow_flow_amd.cfg")).read() if os.path.exists(os.path.join(ROOT,"cfgs","slow_flow_amd.cfg")) else ""
cfg=os.path.join(out,"run.cfg")
with open(cfg,"w") as f:
    f.write("file\t%s/f_%%04i.ppm\noutput\t%s/out\nJets\t%d\nstart\t100\nmax_fps\t200\n16bit\t0\nraw\t0\nscale\t1.0\ndeep_matching\t0\n"%(out,out,JETS))
    # the reference's cfgs/slow_flow.cfg solver section
    f.write("slow_flow_S\t3\nslow_flow_layers\t5\nslow_flow_p_scale\t0.9\nslow_flow_niter_alter\t10\nslow_flow_niter_outer\t10\nslow_flow_niter_inner\t1\n"
            "slow_flow_niter_solver\t30\nslow_flow_sor_omega\t1.9\nslow_flow_occlusion_reasoning\t1\nslow_flow_occlusion_penalty\t0.1\nslow_flow_occlusion_alpha\t0.1\n"
            "slow_flow_rho_0\t1\nslow_flow_rho_1\t1\nslow_flow_omega_0\t0\nslow_flow_omega_1\t2\nslow_flow_alpha\t4.0\nslow_flow_gamma\t6.0\nslow_flow_delta\t1.0\n"
            "slow_flow_thres_outer\t1e-5\nslow_flow_thres_inner\t1e-5\nslow_flow_output_occlusions\t0\n")
    for key, env in (("gpu_batch", "SFA_DRV_BATCH"), ("gpu_streams", "SFA_DRV_STREAMS")):      # experiments: the driver's lockstep batch / groups per GPU
        if os.environ.get(env): f.write("%s\t%s\n" % (key, os.environ[env]))
t0=time.perf_counter()
r=subprocess.run([os.path.join(ROOT,"slowflow_amd","host","slow_flow"),cfg,"-overwrite"],capture_output=True,text=True)
dt=time.perf_counter()-t0
print(r.stdout[-600:]); print(r.stderr[-400:])
assert r.returncode==0
tj=json.load(open(os.path.join(out,"out","timings.json")))
print(f"driver: {len(tj)} windows ({JETS} jets x 2 directions) in {dt:.2f} s wall = {1e3*dt/len(tj):.1f} ms per window incl. frame ingest and .flo / .png output", flush=True)
try:
    rj=json.load(open(os.path.join(out,"out","run.json")))
    print("run.json:", json.dumps(rj))
except Exception as e:
    print("no run.json", e)
