"""Rough VGPR liveness of one kernel in a hipcc -S listing (linear scan, control flow ignored): live registers at every s_barrier / s_setprio, and with an
instruction index as third argument (+ any fourth) the definition and next use of every register live there.  It is what showed ~45 registers of hoisted LDS
addresses held across k_assemble_images' term loop (round 4).
usage: isa_live.py file.s kernel_substring [index [x]]"""
import re,sys
lines=open(sys.argv[1]).read().split('\n')
pat=sys.argv[2]
start=next(i for i,l in enumerate(lines) if re.match(r"^_Z\S*:",l) and pat in l)
end=next(i for i in range(start,len(lines)) if lines[i].startswith(".Lfunc_end"))
ins=[]
for l in lines[start:end]:
    t=l.split(';')[0].strip()
    if not t or t.startswith('.') and not t.startswith('.LBB'): continue
    ins.append(t)
def regs(tok):
    out=[]
    for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b",tok):
        if m.group(1): out+=list(range(int(m.group(1)),int(m.group(2))+1))
        else: out.append(int(m.group(3)))
    return out
acc=[]  # (idx, defs, uses)
for i,t in enumerate(ins):
    if t.endswith(':'): acc.append((set(),set())); continue
    parts=t.split(None,1)
    op=parts[0]; ops=parts[1].split(',') if len(parts)>1 else []
    ops=[o.strip() for o in ops]
    d=set();u=set()
    if op.startswith(('ds_write','global_store','buffer_store','scratch_store','ds_add','global_atomic','buffer_atomic','global_load_lds')) or op.startswith('v_cmp') or op.startswith('s_'):
        for o in ops: u|=set(regs(o))
    elif op.startswith(('v_','ds_read','global_load','buffer_load','scratch_load')):
        if ops: d|=set(regs(ops[0]))
        for o in ops[1:]: u|=set(regs(o))
        if op.startswith(('v_mac','v_fmac','v_dot')) : u|=d
        if 'dst_unused:UNUSED_PRESERVE' in t or op.endswith('_dpp') or 'sdwa' in op: u|=d   # conservative
    acc.append((d,u))
# live at each index: backward linear scan
live=set(); liv=[None]*len(ins)
for i in range(len(ins)-1,-1,-1):
    d,u=acc[i]
    live=(live-d)|u
    liv[i]=set(live)
marks=[i for i,t in enumerate(ins) if t.startswith(('s_barrier','s_setprio'))]
for i in marks: print(i, ins[i], len(liv[i]))
if len(sys.argv)>3:
    i=int(sys.argv[3]); print(sorted(liv[i]))
if len(sys.argv)>4:
    i=int(sys.argv[3])
    for r in sorted(liv[i]):
        # next use
        nu=next((k for k in range(i,len(ins)) if r in acc[k][1]),None)
        # previous def
        pd=next((k for k in range(i,-1,-1) if r in acc[k][0]),None)
        print(f"v{r}: def@{pd} {ins[pd] if pd is not None else ''}   || use@{nu} {ins[nu] if nu is not None else ''}")
# the pressure profile: live registers every 100 instructions and the peak
pk=max(range(len(ins)),key=lambda i:len(liv[i]))
print("peak", len(liv[pk]), "at", pk, ins[pk])
print(" ".join(f"{i}:{len(liv[i])}" for i in range(0,len(ins),100)))
