"""debug: batch vs single under the cfg schedule"""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, slowflow_amd as sfa
sys.path.insert(0, ROOT)
from bench import synth_window
W,H=int(sys.argv[1]) if len(sys.argv)>1 else 1024, int(sys.argv[2]) if len(sys.argv)>2 else 436
NB=int(sys.argv[3]) if len(sys.argv)>3 else 8
ctx=sfa.Context(0)
wins=[synth_window(10+b, W, H, 5) for b in range(NB)]
avg,std=ctx.normalize([f for w in wins for f in w], W)
def params(occ, thr, alter=3, outer=4):
    p=sfa.default_params()
    p.S=3; p.layers=3; p.niter_alter=alter; p.niter_outer=outer; p.occlusion_reasoning=occ; p.thres_outer=thr; p.thres_inner=thr
    p.hbit=0; p.rho[0]=1; p.rho[1]=1; p.omega[0]=0; p.omega[1]=2; p.occlusion_penalty=0.1; p.occlusion_alpha=0.1
    for k in range(3):
        p.norm_avg[k]=float("%g"%avg[k]); p.norm_std[k]=float("%g"%std[k])
    return p
for occ in (0,1):
    for thr in (0.0, 1e-3):
        p=params(occ,thr)
        job=sfa.Job(ctx,p,W,H,NB)
        for b in range(NB): job.upload(b,wins[b])
        job.run()
        bat=[job.download(b) for b in range(NB)]
        bocc=[job.download_occlusions(b) for b in range(NB)]
        job.close()
        nd=0
        for b in range(NB):
            j1=sfa.Job(ctx,p,W,H,1); j1.upload(0,wins[b]); j1.run(); s=j1.download(0); so=j1.download_occlusions(0); j1.close()
            d=max(np.abs(s[0]-bat[b][0]).max(), np.abs(s[1]-bat[b][1]).max())
            nd+= d>0
            if d>0: print("  occ",occ,"thr",thr,"window",b,"maxdiff",d,"labels differ",(so!=bocc[b]).sum(), "chg",s[2],bat[b][2])
        print("occ",occ,"thr",thr,"windows differing:",nd,flush=True)
