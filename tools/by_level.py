"""per-launch-size durations of the kernels whose name contains a substring, from a rocprofv3 kernel trace (csv): one row per distinct grid size (= pyramid level)
usage: by_level.py <kernel_trace.csv> <substring> [...]"""
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r.get("Kernel_Name", "")
    for sub in sys.argv[2:]:
        if sub in name:
            grid = tuple(int(r.get(k, 0) or 0) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else (int(r.get("Grid_Size", 0)),)
            acc[(sub, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (sub, grid), v in sorted(acc.items(), key=lambda kv: (kv[0][0], -sum(kv[0][1]) if len(kv[0][1]) == 1 else -kv[0][1][0] * kv[0][1][1] * kv[0][1][2])):
    v = sorted(v)
    print("%-22s grid %-22s n %4d  mean %8.1f us  min %8.1f  median %8.1f" % (sub, "x".join(map(str, grid)), len(v), sum(v) / len(v), v[0], v[len(v) // 2]))
