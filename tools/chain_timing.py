"""debug build only (-DSFA_CHAIN_TIMING, SFA_LIB=that build): where the waves of window 0's first workgroups spend their cycles in one solve by
sor_chain.hip.  usage: SFA_LIB=... SFA_SOR_CHAIN=<shape> chain_timing.py [batch]"""
import sys, ctypes
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, slowflow_amd as sfa
from synth import sor_system
B=int(sys.argv[1]) if len(sys.argv)>1 else 1
ctx=sfa.Context(0); K=30; W,H=1024,436
rng=np.random.default_rng(0); s=sor_system(rng,W,H)
planes=[np.ascontiguousarray(s[k]) for k in ("du","dv","a11","a12","a22","b1","b2","sh","sv")]
sb=sfa.SorBatch(ctx,W,H,B)
for b in range(B): sb.upload(b,*planes)
for _ in range(3):
    for b in range(B): sb.upload(b,*planes)
    ctx.profile_enable(True); sb.run(K,1.9); n,ms,by=ctx.profile_read(); ctx.profile_enable(False)
print(f"shape {os.environ.get('SFA_SOR_CHAIN')} batch {B}: launch {ms/n*1e3:.0f} us")
out=(ctypes.c_ulonglong*(64*16*16))()
rc=sfa.lib().sfa_debug_chain_timing(out)
a=np.array(out,dtype=np.uint64).reshape(64,16,16).astype(np.int64)
ins=[(wg,w) for wg in range(64) for w in range(16) if a[wg,w,12]==0x10]
if not ins:     # a build with an idle IN wave: compute waves only
    cw=[(wg,w) for wg in range(64) for w in range(16) if a[wg,w,12]==0 and a[wg,w,7]>0]
    per=[(a[wg,w,2]-a[wg,w,1])/a[wg,w,7] for wg,w in cw]; bar=[a[wg,w,4]/a[wg,w,7] for wg,w in cw]
    print(f"compute waves only ({len(cw)} recorded): cycles per chunk median {np.median(per):.0f} (min {min(per):.0f}, max {max(per):.0f}), at the barrier {np.median(bar):.0f}")
    sys.exit(0)
outs={wg:w for wg in range(64) for w in range(16) if a[wg,w,12]==0x20}
cw=[(wg,w) for wg in range(64) for w in range(16) if a[wg,w,12]==0 and a[wg,w,7]>0]
per=[(a[wg,w,2]-a[wg,w,1])/a[wg,w,7] for wg,w in cw]; bar=[a[wg,w,4]/a[wg,w,7] for wg,w in cw]
print(f"compute waves ({len(cw)} recorded): cycles per chunk median {np.median(per):.0f} (min {min(per):.0f}, max {max(per):.0f}), of which at the barrier {np.median(bar):.0f}")
print("workgroup (band,group): cycles per interval | IN: at barrier, blocked on producers (intervals) | OUT: at barrier, publish wait | stage 0: per chunk, at barrier")
for wg,w in sorted(ins, key=lambda t:(a[t[0],t[1],9],a[t[0],t[1],10]))[:48]:
    r=a[wg,w]; NI=r[11]; tot=r[1]-r[0]
    o=a[wg,outs[wg]] if wg in outs else None
    c=a[wg,1]
    print(f"  ({r[9]},{r[10]:2d}) {tot/NI:6.0f} | {r[2]/NI:6.0f} {r[6]/NI:6.0f} ({r[8]:3d} of {NI}) | " + (f"{o[2]/NI:6.0f} {o[4]/NI:6.0f}" if o is not None else "   -") + f" | {(c[2]-c[1])/max(c[7],1):6.0f} {c[4]/max(c[7],1):6.0f}")
print("all compute waves of the first workgroups: (band,k0) cycles per chunk | at barrier | LDS-read wait | LDS-write drain | rest")
for wg,w in sorted(ins, key=lambda t:(a[t[0],t[1],9],a[t[0],t[1],10]))[:8]:
    for cwv in range(1,16):
        r=a[wg,cwv]
        if r[12]!=0 or r[7]==0: continue
        n=r[7]; tot=(r[2]-r[1])/n
        print(f"  ({r[5]},{r[6]:2d}) {tot:6.0f} | {r[4]/n:6.0f} | {r[8]/n:6.0f} | {r[9]/n:6.0f} | {tot-(r[4]+r[8]+r[9])/n:6.0f}")
print("placement: workgroup (band,group): XCC, SE, CU | SIMD of each wave (IN, stages..., OUT)")
seen={}
for wg,w in sorted(ins, key=lambda t:(a[t[0],t[1],9],a[t[0],t[1],10])):
    r=a[wg,w]; hw=int(r[13]); xcc=int(r[14])&15
    cu=(hw>>8)&15; sh=(hw>>12)&1; se=(hw>>13)&7
    simds=[(int(a[wg,k,13])>>4)&3 for k in range(16) if a[wg,k,13]!=0]
    key=(xcc,se,sh,cu); seen.setdefault(key,[]).append((int(r[9]),int(r[10])))
    print(f"  ({r[9]},{r[10]:2d}): xcc {xcc} se {se} sh {sh} cu {cu:2d} | simd {simds}")
print("CUs holding more than one recorded workgroup:", {k:v for k,v in seen.items() if len(v)>1})
print("timeline (s_memrealtime, 100 MHz): workgroup (band,group): start .. end in us after the first start")
rt0=min(a[wg,w,3] for wg,w in ins)
for wg,w in sorted(ins, key=lambda t:(a[t[0],t[1],9],a[t[0],t[1],10])):
    r=a[wg,w]
    print(f"  ({r[9]},{r[10]:2d}) {(r[3]-rt0)/100:8.1f} .. {(r[4]-rt0)/100:8.1f}   ({(r[4]-r[3])/100:7.1f} us, {(r[4]-r[3])/100/r[11]:.2f} us per interval)")
