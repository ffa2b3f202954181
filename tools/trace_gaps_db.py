"""GPU idle time inside the occlusion cut from a rocprofv3 --kernel-trace rocpd database: per cut call (k_cut_count .. k_cut_labels) the wall time, the sum of kernel
durations and the idle time between kernels (host round trips: every relabelling batch is followed by a 24-byte read-back), and the same for everything outside the cut.
usage: trace_gaps_db.py <results.db>"""
import sqlite3, re, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
rows = rows[len(rows) // 2:]            # the second (timed) run of tools/bench_cfg4.py
def short(n): return re.sub(r"\(.*", "", n).replace("sfa::", "").replace("void ", "")
in_cut = False
cut_wall = cut_busy = cut_gaps = other_busy = other_gaps = 0
n_cut = n_gap20 = 0
t0 = None
prev_end = rows[0][2]
for n, s, e in rows:
    k = short(n)
    gap = max(0, s - prev_end)
    if k == "k_cut_count":
        in_cut = True; t0 = s; n_cut += 1; other_gaps += gap
    elif in_cut:
        cut_gaps += gap
        if gap > 20000: n_gap20 += 1
    else:
        other_gaps += gap
    if in_cut: cut_busy += e - s
    else: other_busy += e - s
    if k == "k_cut_labels" and in_cut:
        in_cut = False; cut_wall += e - t0
    prev_end = max(prev_end, e)
wall = rows[-1][2] - rows[0][1]
print(f"timed run: wall {wall / 1e6:.1f} ms; {n_cut} cut calls: wall {cut_wall / 1e6:.1f} ms = {100 * cut_wall / wall:.1f} % (kernels {cut_busy / 1e6:.1f} ms, idle between them {cut_gaps / 1e6:.1f} ms, "
      f"gaps > 20 us: {n_gap20}); everything else: kernels {other_busy / 1e6:.1f} ms, idle {other_gaps / 1e6:.1f} ms")
