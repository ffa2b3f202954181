#!/usr/bin/env python3
"""Condense the rocprofv3 output of profiles/collect.sh (under gpurun_out/) into the files kept in profiles/:
   <tag>_kernel_stats.csv   per-kernel totals of the bench command (rocprofv3 --kernel-trace --stats)
   <tag>_sor_by_level.csv   SOR kernel: dispatches / average duration per pyramid level
   <tag>_traffic.json       HBM-side bytes per SOR launch: 2 x FETCH_SIZE (gfx950 tallies 128-B reads at 64 B) + WRITE_SIZE
usage: python profiles/summarize.py r01 [windows_per_launch]     (bench.py --batch 64 --streams 2 launches 32 windows at a time)"""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = os.path.join(root, "gpurun_out")
out = os.path.join(root, "profiles")

# provenance: the HEAD the collection was started from (profiles/collect.sh writes it next to the raw files, with the hash of the library that
# ran) must be the HEAD of this tree, and the kernels' sources must not have changed since; both go into every JSON written here
import hashlib
import subprocess
prov = {"collected_at_head": None, "library_sha256": None}
try:
    lines = open(os.path.join(go, f"{tag}_head.txt")).read().split()
    prov["collected_at_head"], prov["library_sha256"] = lines[0], (lines[1] if len(lines) > 1 else None)
except OSError:
    pass
try:
    head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "slowflow_amd/csrc", "include"], capture_output=True, text=True).stdout.strip()
    prov["summarized_at_head"] = head or "(on the GPU box: no .git there; the raw counter files are too large to travel back)"
    prov["kernel_sources_modified_since"] = bool(dirty) if head else None
    lib = os.path.join(root, "slowflow_amd", "libslowflow_amd.so")
    if os.path.exists(lib):
        prov["library_sha256_here"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()
    if head and "--force" not in sys.argv:
        assert prov["collected_at_head"] in (None, "unknown") or head.startswith(prov["collected_at_head"]) or prov["collected_at_head"].startswith(head), \
            f"profiles of {tag} were collected at {prov['collected_at_head']}, this tree is at {head}: collect again (or --force)"
        assert not dirty, "kernel sources differ from HEAD: commit first, then collect and summarise (or --force)\n" + dirty
except FileNotFoundError:
    pass
json.dump(prov, open(os.path.join(out, f"{tag}_provenance.json"), "w"), indent=1)


def find(d, suffix):
    for base, _, files in os.walk(os.path.join(go, d)):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(base, f)
    raise FileNotFoundError((d, suffix))


shutil.copy(find(f"{tag}_stats", "kernel_stats.csv"), os.path.join(out, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(go, f"{tag}_stats.json"), os.path.join(out, f"{tag}_bench_under_rocprof.json"))

try:                                  # the one-stream run of the same workload (collect.sh pass 6)
    shutil.copy(find(f"{tag}_stats1", "kernel_stats.csv"), os.path.join(out, f"{tag}_kernel_stats_one_stream.csv"))
except FileNotFoundError:
    pass

# SOR kernel per level from the trace
rows = list(csv.DictReader(open(find(f"{tag}_stats", "kernel_trace.csv"))))
acc = collections.defaultdict(list)
for r in rows:
    if "k_sor_band" in r["Kernel_Name"] or "k_sor_solve" in r["Kernel_Name"] or "k_sor_chain" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(os.path.join(out, f"{tag}_sor_by_level.csv"), "w") as f:
    f.write("kernel,grid_threads,dispatches,avg_us,min_us,max_us\n")
    for (k, g), v in sorted(acc.items(), key=lambda kv: -kv[0][1]):
        f.write(f"\"{k}\",{g},{len(v)},{sum(v) / len(v):.1f},{min(v):.1f},{max(v):.1f}\n")


def pmc(d, counter):
    rows = list(csv.DictReader(open(find(d, "counter_collection.csv"))))
    per = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return per


# the two dominant kernels ALONE (pass 6: one stream, nothing else on the GPU), by launch size: grid = windows x tiles (assembly) / workgroups (solver)
try:
    rows1 = list(csv.DictReader(open(find(f"{tag}_stats1", "kernel_trace.csv"))))
    acc1 = collections.defaultdict(list)
    for r in rows1:
        k = r["Kernel_Name"].split("(")[0]
        if "k_assemble_images" in k or "k_sor_chain" in k:
            acc1[(k, int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(os.path.join(out, f"{tag}_alone_by_level.csv"), "w") as f:
        f.write("kernel,grid_threads,dispatches,avg_us,min_us,max_us\n")
        for (k, g), v in sorted(acc1.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
            f.write(f"\"{k}\",{g},{len(v)},{sum(v) / len(v):.1f},{min(v):.1f},{max(v):.1f}\n")
except FileNotFoundError:
    pass

fetch, write = pmc(f"{tag}_fetch", "FETCH_SIZE"), pmc(f"{tag}_write", "WRITE_SIZE")
res = {"batch": batch, "batch_note": "windows per SOR launch (bench.py: --batch / --streams)", "unit_note": "FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B by rocprofv3; x2 on FETCH_SIZE per MI355X_MICROARCH.md (gfx950)",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("void sfa::") and not k.startswith("sfa::"):
        continue
    fz = fetch.get(k, [])
    wz = write.get(k, [])
    fb = 2 * 1024 * sum(fz) / max(len(fz), 1)
    wb = 1024 * sum(wz) / max(len(wz), 1)
    res["kernels"][k] = {"dispatches": len(fz), "fetch_bytes_per_launch_x2": round(fb), "write_bytes_per_launch": round(wb), "traffic_bytes_per_launch": round(fb + wb)}
# the solver kernel of the path: the k_sor_* kernel that moves the most bytes over the run (the lone-window sections launch another shape a few times)
sor = sorted((k for k in res["kernels"] if "k_sor_" in k and "prepare" not in k and "finish" not in k),
             key=lambda k: -res["kernels"][k]["traffic_bytes_per_launch"] * res["kernels"][k]["dispatches"])
if sor:
    res["traffic_bytes_per_launch"] = res["kernels"][sor[0]]["traffic_bytes_per_launch"]
    res["kernel"] = sor[0]
res["provenance"] = prov
json.dump(res, open(os.path.join(out, f"{tag}_traffic.json"), "w"), indent=1)
# SQ passes (when collected): per kernel the share of SIMD time with a VALU instruction active, the LDS bank-conflict share and the instruction mix
try:
    sq = {}
    for d in (f"{tag}_sq1", f"{tag}_sq2"):
        for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
            k = r["Kernel_Name"].split("(")[0]
            sq.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT",
             "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "GRBM_GUI_ACTIVE"]
    sqj = {"note": "means per dispatch; SQ_*_CYCLES / ACTIVE / WAIT counters are in quad-cycles summed over waves (MI355X_MICROARCH.md); valu_active = 4 * SQ_ACTIVE_INST_VALU / "
                   "(1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs); lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE", "kernels": {}}
    with open(os.path.join(out, f"{tag}_sq_by_kernel.csv"), "w") as f:
        f.write("kernel,dispatches," + ",".join(names) + ",valu_active_frac,lds_conflict_frac,wave_active_frac,wave_wait_frac\n")
        for k, v in sorted(sq.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
            if "sfa::" not in k:
                continue
            m = {n: (sum(v[n]) / len(v[n]) if v.get(n) else 0.0) for n in names}
            cyc = m["GRBM_GUI_ACTIVE"] / 8.0
            valu = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (1024.0 * cyc) if cyc else 0.0
            conf = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"] if m["SQ_LDS_IDX_ACTIVE"] else 0.0
            act = m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"] if m["SQ_WAVE_CYCLES"] else 0.0
            wait = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"] if m["SQ_WAVE_CYCLES"] else 0.0
            n = len(v.get("SQ_WAVE_CYCLES", v.get("SQ_WAVES", [])))
            f.write('"%s",%d,' % (k, n) + ",".join("%.0f" % m[x] for x in names) + ",%.4f,%.4f,%.4f,%.4f\n" % (valu, conf, act, wait))
            sqj["kernels"][k] = {"valu_active_frac": round(valu, 4), "lds_conflict_frac": round(conf, 4), "wave_active_frac": round(act, 4), "wave_wait_frac": round(wait, 4)}
    sor_k = [k for k in sqj["kernels"] if k == res.get("kernel")] or [k for k in sqj["kernels"] if "k_sor_band" in k]
    if sor_k:
        sqj["valu_busy_frac"] = sqj["kernels"][sor_k[0]]["valu_active_frac"]
        sqj["kernel"] = sor_k[0]
    # the data-term assembly kernel: VALU instructions one pixel spends per data term = wave instructions x 64 lanes / (pixels x terms) of a launch
    # (bench configuration: 5 levels, 2 terms, `batch` windows per launch; the mean over the launches of all levels against the mean level size)
    # per-wave instruction counts (SQ_INSTS_* / SQ_WAVES summed over all dispatches of the kernel: independent of how many windows a dispatch carried)
    def per_wave(k, name):
        return sum(sq[k][name]) / sum(sq[k]["SQ_WAVES"]) if sq[k].get(name) and sq[k].get("SQ_WAVES") and sum(sq[k]["SQ_WAVES"]) else None
    if sor_k and per_wave(sor_k[0], "SQ_INSTS_VALU"):
        sqj["sor_valu_inst_per_wave"] = round(per_wave(sor_k[0], "SQ_INSTS_VALU"), 1)
        sqj["sor_lds_inst_per_wave"] = round(per_wave(sor_k[0], "SQ_INSTS_LDS") or 0, 1)
    asm_all = [k for k in sq if "k_assemble_images" in k and sq[k].get("SQ_INSTS_VALU") and sq[k].get("SQ_WAVES")]
    if asm_all:
        sqj["assemble_valu_inst_per_wave"] = round(sum(sum(sq[k]["SQ_INSTS_VALU"]) for k in asm_all) / sum(sum(sq[k]["SQ_WAVES"]) for k in asm_all), 1)
        sqj["assemble_lds_inst_per_wave"] = round(sum(sum(sq[k]["SQ_INSTS_LDS"]) for k in asm_all) / sum(sum(sq[k]["SQ_WAVES"]) for k in asm_all), 1)
        sqj["assemble_per_wave_note"] = "a wave = 64 pixels of a 64 x 8 tile x all data terms of the level (2 in the bench configuration), staging and epilogue included"
    asm_k = [k for k in sq if "k_assemble_images" in k and sq[k].get("SQ_INSTS_VALU")]
    if asm_k:
        LV = [(1024, 436), (921, 392), (828, 352), (745, 316), (670, 284)]
        px = sum(w * h for w, h in LV) / len(LV)
        inst = sum(sum(sq[k]["SQ_INSTS_VALU"]) for k in asm_k) / sum(len(sq[k]["SQ_INSTS_VALU"]) for k in asm_k)
        sqj["assemble_valu_inst_per_pixel_term"] = round(inst * 64.0 / (px * batch * 2), 1)
        sqj["assemble_note"] = "SQ_INSTS_VALU (wave instructions, mean per dispatch of both k_assemble_images instances) x 64 / (mean level pixels x %d windows x 2 terms)" % batch
    sqj["provenance"] = prov
    json.dump(sqj, open(os.path.join(out, f"{tag}_sq.json"), "w"), indent=1)
    print("SQ:", {k[-40:]: v for k, v in list(sqj["kernels"].items())[:4]})
except FileNotFoundError:
    print("no SQ passes under gpurun_out/ for", tag)
print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))
for k, v in res["kernels"].items():
    print(f"{k[:60]:60s} n={v['dispatches']:5d} fetch(x2) {v['fetch_bytes_per_launch_x2'] / 1e6:10.1f} MB  write {v['write_bytes_per_launch'] / 1e6:10.1f} MB")
