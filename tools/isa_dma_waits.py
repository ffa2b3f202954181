"""Build-time check of k_assemble_images: no `s_waitcnt vmcnt` may sit between the first and the last `global_load_lds` of a staging round (the DMA pieces
are issued untracked; a compiler-made wait between them -- a register of the address arithmetic with a load pending -- exposes a memory round trip per piece).
usage: isa_dma_waits.py [file.s [kernel substring]]      without a listing, kernels.hip is compiled to one with the product's flags; exit code 1 if any
instance has such a wait"""
import os, re, subprocess, sys, tempfile
if len(sys.argv) > 1:
    lines = open(sys.argv[1]).read().split("\n")
else:
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slowflow_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "kernels.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
                        "-Wno-unused-function", "-S", "--cuda-device-only", os.path.join(csrc, "kernels.hip"), "-o", out], check=True, capture_output=True)
        lines = open(out).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else "k_assemble_images"
bad = 0
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l]
for st in starts:
    end = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
    ins = [l.split(";")[0].strip() for l in lines[st:end]]
    ins = [t for t in ins if t and not t.startswith(".")]          # labels and directives dropped: program order of the listing
    # an issue phase = global_load_lds instructions less than GAP instructions apart, plus the LEAD instructions in front of the first one (the address
    # arithmetic of the piece, where both waits of round 4 sat); basic-block placement does not matter to this window
    GAP, LEAD = 40, 14
    dma = [i for i, t in enumerate(ins) if t.startswith("global_load_lds")]
    waits, prev = [], None
    for i in dma:
        lo = prev + 1 if prev is not None and i - prev < GAP else max(0, i - LEAD)
        waits += [f"{k}: {ins[k]}" for k in range(lo, i) if ins[k].startswith("s_waitcnt") and "vmcnt" in ins[k]]
        prev = i
    print(f"{lines[st][:70]:70s} {len(dma):3d} DMA instructions, vmcnt waits between them: {len(waits)}")
    for w in waits: print("      ", w)
    # the run-time instances (FAST = 0: run-time penalties, channel weights; 2-6 spill reloads -- vector-memory instructions themselves -- at the kernel's 80
    # registers) are reported, not counted: no configuration of BASELINE.json runs them
    if "ELi0ELb" not in lines[st]: bad += len(waits)
print("no wait inside any DMA issue phase" if not bad else f"{bad} waits inside DMA issue phases")
sys.exit(1 if bad else 0)
