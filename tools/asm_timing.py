"""debug build only (kernels.hip with -DSFA_ASM_TIMING, library given by SFA_LIB): wave-cycles of k_assemble_images by phase, summed over all waves of the launches of a few bench steps"""
import sys, ctypes, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, slowflow_amd as sfa, bench
B=int(sys.argv[1]) if len(sys.argv)>1 else 64
ctx=sfa.Context(0); p=bench.bench_params()
wins=[bench.synth_window(b) for b in range(min(B,4))]
avg,std=ctx.normalize([f for w in wins for f in w], bench.W)
for k in range(3): p.norm_avg[k]=float("%g"%avg[k]); p.norm_std[k]=float("%g"%std[k])
job=sfa.Job(ctx,p,bench.W,bench.H,B)
for b in range(B): job.upload(b,wins[b%len(wins)])
job.run(); ctx.sync()
out=(ctypes.c_ulonglong*16)()
L=sfa.lib(); L.sfa_debug_asm_timing(out,1)
job.run(); ctx.sync()
L.sfa_debug_asm_timing(out,0)
a=np.array(out,dtype=np.float64)[:14]
names=["per-pixel terms","wait: planes free","DMA issue","DMA vmcnt wait","wait: DMA barrier","convert","wait: before stage 1","stage 1","wait: after stage 1","wait: epilogue","epilogue per pixel","wait: tile","diagonal stores","prologue"]
tot=a.sum()
# residency check: 25 launches per run; a CU holds 2 blocks of 8 waves, so a launch of T seconds offers 256 CUs x 16 waves x T x clock wave-cycles
import time
t0=time.perf_counter(); job.run(); ctx.sync(); T=time.perf_counter()-t0
LV=[(1024,436),(921,392),(828,352),(745,316),(670,284)]
blocks=sum(((w+63)//64)*((h+7)//8) for w,h in LV)*5*B
print(f"sum of wave cycles (1 block in 61 sampled) {tot:.4g}; blocks {blocks}; per wave {61*tot/(blocks*8):.0f} cycles; whole run {T*1e3:.1f} ms (all kernels)")
for n,v in sorted(zip(names,a),key=lambda t:-t[1]): print(f"{n:24s} {100*v/tot:5.1f} %")
