"""Per-kernel HBM report of one config-5 refinement from the four rocprofv3 passes of profiles/collect_config5.sh.  usage: config5_report.py <tag> <batch>"""
import csv, os, sys, collections
tag, B = sys.argv[1], int(sys.argv[2])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = os.path.join(root, "gpurun_out")
PEAK = 8000.0
def find(d, suffix):
    for base, _, files in os.walk(os.path.join(go, d)):
        for f in files:
            if f.endswith(suffix): return os.path.join(base, f)
    raise FileNotFoundError((d, suffix))
def base(k): return k.split("(")[0].replace("void ", "").replace("sfa::", "")
dur = collections.defaultdict(list)
for r in csv.DictReader(open(find(f"{tag}_c5_stats", "kernel_trace.csv"))):
    dur[base(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
def pmc(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
        acc[base(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
fe, wr, sq = pmc(f"{tag}_c5_fetch"), pmc(f"{tag}_c5_write"), pmc(f"{tag}_c5_sq")
# algorithmic bytes per pixel of the launch's level (fp32 planes; S = 2: two data terms) for the kernels where the figure is defined
ALG = {"k_warp_jobs": ("2 flow + 2 warps x (3 gather + 3 store + 1 mask) x 4 B", 64.0), "k_smoothness_tiled": ("uu, vv, dpsis in; sh, sv out", 20.0),
       "k_warp_smooth": ("the warps (64 B) + dpsis in; sh, sv out, the flow read once", 76.0),
       "k_update_outer_x": ("du, dv, wx, wy in; uu, vv, wx, wy out", 32.0), "k_assemble_images": ("3 frames x 3 ch + 2 masks + occ, uu, vv, sh, sv in; 40 B of solver operands out", 104.0),
       "k_sor_chain": ("fused 30-sweep solve: 2 x 16 B operands + 8 B iterate in + 8 B out (SURVEY 8(d) per-sweep model: 1332 B)", 48.0), "k_dpsis": ("3 ch in, 1 plane out", 16.0)}
plain = open(os.path.join(go, f"{tag}_c5_plain.txt")).read().strip().splitlines()[-1]
print(f"BASELINE configs[4]: 2048x2048 synthetic sequence, 6 pyramid levels (2048 1843 1658 1492 1342 1207), Lorentzian penalties (eps 0.05), S = 2, 5 outer x 30 sweeps per level,")
print(f"{B} windows in lockstep on ONE stream of one MI355X (kernels run alone: the durations are not stretched by a second group).  {plain}")
print("Per kernel over the warm-up + the measured refinement: launches, mean duration, HBM-side bytes per launch from the counters (2 x FETCH_SIZE + WRITE_SIZE, KiB units x 1024; separate passes),")
print(f"achieved GB/s = counter bytes / mean duration, its fraction of the {PEAK:.0f} GB/s peak, and SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles) (a quad-cycle per wave instruction: about")
print("twice the issue-slot use).  `algorithmic`: bytes per pixel the kernel has to move (mean over the launches of all levels is what the counters show; the per-pixel figure is level independent).\n")
print(f"{'kernel':44s} {'launches':>8s} {'mean us':>10s} {'share':>6s} {'fetch MB':>10s} {'write MB':>10s} {'GB/s':>8s} {'of peak':>8s} {'VALU act':>9s}  algorithmic")
tot = sum(sum(v) for v in dur.values())
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / tot < 0.002: continue
    f = 2 * 1024 * (sum(fe[k]["FETCH_SIZE"]) / max(len(fe[k]["FETCH_SIZE"]), 1)) if k in fe else 0.0
    w = 1024 * (sum(wr[k]["WRITE_SIZE"]) / max(len(wr[k]["WRITE_SIZE"]), 1)) if k in wr else 0.0
    us = sum(v) / len(v)
    gbs = (f + w) / (us * 1e-6) / 1e9 if us > 0 else 0.0
    va = ""
    if k in sq and sq[k].get("GRBM_GUI_ACTIVE"):
        cyc = sum(sq[k]["GRBM_GUI_ACTIVE"]) / len(sq[k]["GRBM_GUI_ACTIVE"]) / 8.0
        va = "%.3f" % (4.0 * (sum(sq[k]["SQ_ACTIVE_INST_VALU"]) / len(sq[k]["SQ_ACTIVE_INST_VALU"])) / (1024.0 * cyc)) if cyc else ""
    alg = next((f"{b:.0f} B/px ({t})" for n, (t, b) in ALG.items() if n in k), "")
    print(f"{k[:44]:44s} {len(v):8d} {us:10.1f} {100 * sum(v) / tot:5.1f}% {f / 1e6:10.1f} {w / 1e6:10.1f} {gbs:8.0f} {gbs / PEAK:8.3f} {va:>9s}  {alg}")
