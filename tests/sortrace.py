import sys, ctypes as C
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, slowflow_amd as sfa
from synth import sor_system
ctx=sfa.Context(0)
K=30; W,H=1024,436
B=int(sys.argv[1]) if len(sys.argv)>1 else 32
F=int(__import__('os').environ.get('SFA_SOR_BAND','3'))
NW=K//F; NB=(H+K-1+63)//64
rng=np.random.default_rng(0)
s=sor_system(rng,W,H)
planes=[np.ascontiguousarray(s[k]) for k in ("du","dv","a11","a12","a22","b1","b2","sh","sv")]
sb=sfa.SorBatch(ctx,W,H,B)
for b in range(B): sb.upload(b,*planes)
for _ in range(3): sb.run(K,1.9)
ctx.sync()
n=B*NB*NW*4
out=np.zeros(n,np.uint64)
L=sfa.lib()
L.sfa_debug_sor_trace.argtypes=[C.c_void_p,C.c_void_p,C.c_int]
r=L.sfa_debug_sor_trace(sb.h_, out.ctypes.data, n)
assert r==n, r
t=out.reshape(NB,B,NW,4).astype(np.int64)
t0=t[...,0].min()
us=lambda x:(x-t0)/100.0
print("band: start(min..max over jobs)  prologue-done  end   blocked(mean)  [wave 0 | last wave]")
for b in range(NB):
    for w in (0,NW-1):
        print(f" b{b} w{w:2d}: start {us(t[b,:,w,0]).min():7.1f}..{us(t[b,:,w,0]).max():7.1f}  ready {us(t[b,:,w,1]).min():7.1f}..{us(t[b,:,w,1]).max():7.1f}  end {us(t[b,:,w,2]).min():7.1f}..{us(t[b,:,w,2]).max():7.1f}  run {((t[b,:,w,2]-t[b,:,w,1])/100).mean():7.1f}  blocked {t[b,:,w,3].mean():6.1f}")
