"""Σ kernel durations against the wall time of a kernel trace (rocprofv3 --kernel-trace csv): how much of a stream's time is launch gaps.
usage: trace_gaps.py <kernel_trace.csv> [skip_first_fraction]   -- the trace of tools/bench_single.py (one window, one stream); the first part (warm-up run) is skipped"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
gaps = []
prev_end = int(rows[0]["End_Timestamp"])
for r in rows[1:]:
    s = int(r["Start_Timestamp"])
    gaps.append(max(0, s - prev_end)); prev_end = max(prev_end, int(r["End_Timestamp"]))
print(f"{len(rows)} launches, wall {(t1 - t0) / 1e6:.3f} ms, sum of kernel durations {busy / 1e6:.3f} ms, idle between kernels {sum(gaps) / 1e6:.3f} ms "
      f"(mean gap {sum(gaps) / max(len(gaps), 1) / 1e3:.2f} us, gaps > 20 us: {sum(1 for g in gaps if g > 20000)})")
per = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:70]
    per[k][0] += 1; per[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, ns) in sorted(per.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"  {k:70s} {n:5d} x {ns / n / 1e3:8.1f} us = {ns / 1e6:7.3f} ms")
