"""List the loops of one kernel in a hipcc -S listing with their instruction mix (VALU / SALU / branch / LDS / VMEM / waitcnt).
usage: isa_loops.py file.s kernel_name_substring [min_instructions]"""
import re, sys
src, pat = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
end = max(i for i in range(start, len(lines)) if lines[i].strip().startswith(".amdhsa_kernel")) if False else end
# the function may have several s_endpgm: take up to the .section / .rodata marker
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
labels, ins = {}, []
for i in range(start, end):
    l = lines[i].split(";")[0].rstrip()
    m = re.match(r"^(\.LBB\S+):", l)
    if m: labels[m.group(1)] = len(ins); continue
    t = l.strip()
    if not t or t.startswith(".") or t.endswith(":"): continue
    ins.append(t)
def kind(t):
    op = t.split()[0]
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_nop") or op.startswith("s_sleep"): return "nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    if op.startswith("v_"): return "valu"
    return "other"
print(f"{pat}: {len(ins)} instructions")
for idx, t in enumerate(ins):
    op = t.split()[0]
    if op.startswith(("s_cbranch", "s_branch")):
        tgt = t.split()[-1]
        if tgt in labels and labels[tgt] <= idx and idx - labels[tgt] >= minlen:
            body = ins[labels[tgt]:idx + 1]
            c = {}
            for b in body: c[kind(b)] = c.get(kind(b), 0) + 1
            pk = sum(1 for b in body if b.startswith("v_pk_"))
            dpp = sum(1 for b in body if "dpp" in b or "row_" in b or "wave_shr" in b)
            print(f"loop {tgt}: {len(body)} instr  {c}  v_pk={pk} dpp={dpp}")
