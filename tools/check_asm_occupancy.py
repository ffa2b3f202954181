"""Build-time check (run by tests/test_abi.py): the folded instances of k_assemble_images keep what three blocks per CU need on gfx950 -- at most 80 registers
(512 / 6 waves per SIMD, allocated in eights), at most a third of the 160 KB of LDS, and no spills where du = dv = 0 (the instances of every BASELINE configuration) -- in the compiler's own resource report of kernels.hip built with the
product's flags.  A toolchain bump or a source change that costs the third block shows up here, not as a 10 % slower launch on the GPU.
usage: check_asm_occupancy.py        exit code 1 on a violation"""
import os, re, subprocess, sys, tempfile
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slowflow_amd", "csrc")
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
                        "-Wno-unused-function", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(csrc, "kernels.hip"),
                        "-o", os.path.join(d, "k.o")], capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stderr[-2000:]); sys.exit(2)
rep = r.stderr
bad = 0
cur = None
vals = {}
def flush():
    global bad
    if cur and "k_assemble_images" in cur:
        m = re.search(r"k_assemble_imagesILi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELi(\d)ELb([01])E", cur)
        fast = int(m.group(5)) if m else -1
        v, sp, lds, occ = vals.get("VGPRs"), vals.get("VGPRs Spill"), vals.get("LDS Size [bytes/block]"), vals.get("Occupancy [waves/SIMD]")
        zuv = int(m.group(4)) if m else 0
        # no spills in the instances BASELINE's configurations run (one inner iteration: ZUV); the others may spill a register or two -- a reload is a vector-memory
        # instruction next to the untracked DMA, so it is reported
        ok = fast == 0 or (v <= 80 and lds * 3 <= 160 * 1024 and occ >= 6 and (sp == 0 or not zuv))
        print(f"{'ok ' if ok else 'BAD'} k_assemble_images<TY {m.group(1)}, {m.group(2)} threads, ZUV {m.group(4)}, FAST {fast}, XT {m.group(6)}>: {v} VGPRs, {sp} spilled, {lds} B LDS, "
              f"{occ} waves per SIMD" + ("  (run-time instance: reported only)" if fast == 0 else ""))
        if not ok: bad += 1
for line in rep.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        flush(); cur = m.group(1); vals = {}; continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
    if m: vals[m.group(1).strip()] = int(m.group(2))
flush()
print("three blocks per CU for every folded instance" if not bad else f"{bad} instances lost the third block")
sys.exit(1 if bad else 0)
