"""per-call breakdown of the occlusion cut from a rocprofv3 --kernel-trace database (rocpd sqlite): python tools/cut_profile.py <results.db>"""
import sqlite3, re, collections, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end, grid_x, grid_y, grid_z from kernels order by start"))
tot = collections.Counter(); cnt = collections.Counter()
calls = []; cc = None
for n, s, e, gx, gy, gz in rows:
    short = re.sub(r"\(.*", "", n).replace("sfa::", "").replace("void ", "")
    tot[short] += e - s; cnt[short] += 1
    if short == "k_cut_count":
        cc = {"t0": s, "k": collections.Counter(), "n": collections.Counter(), "grid": None}; calls.append(cc)
    if cc is not None and short.startswith("k_cut"):
        cc["k"][short] += e - s; cc["n"][short] += 1; cc["t1"] = e
        if short == "k_cut_init": cc["grid"] = (gx, gy, gz)
    if short == "k_cut_labels": cc = None
all_t = sum(tot.values())
print("kernel time %.1f ms" % (all_t / 1e6))
for k, v in tot.most_common(16):
    print("  %-50s %6d launches %8.2f ms %5.1f %%  avg %.1f us" % (k[:50], cnt[k], v / 1e6, 100 * v / all_t, v / cnt[k] / 1e3))
bylev = collections.defaultdict(list)
for c in calls: bylev[c["grid"]].append(c)
for g, cs in sorted(bylev.items()):
    n = len(cs)
    keys = sorted({k for c in cs for k in c["k"]})
    print("grid", g, "calls", n, " wall %.2f ms / call, kernels %.2f ms / call" % (sum(c["t1"] - c["t0"] for c in cs) / n / 1e6, sum(sum(c["k"].values()) for c in cs) / n / 1e6))
    for k in keys:
        print("     %-22s %6.1f launches / call  %7.3f ms / call" % (k, sum(c["n"][k] for c in cs) / n, sum(c["k"][k] for c in cs) / n / 1e6))
