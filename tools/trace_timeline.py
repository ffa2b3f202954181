"""kernel timeline of the LAST run of a traced process (rocprofv3 --kernel-trace rocpd database): per kernel name the launches, busy time and the idle time in front of
them; with a second argument N the first N rows of the run's timeline.  The run = everything after the last gap of more than 200 us (the host's print between runs).
TAIL_MS=x in the environment: the last x ms of the trace instead (a step of bench.py has no such gap in front of it).
usage: trace_timeline.py <results.db> [rows]"""
import sqlite3, re, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
def short(n): return re.sub(r"\(.*", "", n).replace("sfa::", "").replace("void ", "")[:44]
cut = 0
for i in range(1, len(rows)):
    if rows[i][1] - rows[i - 1][2] > 200000: cut = i
import os
if os.environ.get("TAIL_MS"):
    t_end = rows[-1][2]; cut = next(i for i, r in enumerate(rows) if r[1] >= t_end - float(os.environ["TAIL_MS"]) * 1e6)
rows = rows[cut:]
wall = rows[-1][2] - rows[0][1]
busy = collections.Counter(); gaps = collections.Counter(); cnt = collections.Counter()
prev = rows[0][1]
for n, s, e in rows:
    k = short(n); busy[k] += e - s; gaps[k] += max(0, s - prev); cnt[k] += 1; prev = max(prev, e)
print(f"last run: {len(rows)} launches, wall {wall / 1e3:.1f} us, busy {sum(busy.values()) / 1e3:.1f} us, idle {sum(gaps.values()) / 1e3:.1f} us")
for k, v in busy.most_common():
    print(f"  {k:<46s}{cnt[k]:5d} launches  busy {v / 1e3:9.1f} us ({v / cnt[k] / 1e3:7.1f} each)  idle in front {gaps[k] / 1e3:8.1f} us ({gaps[k] / cnt[k] / 1e3:5.1f} each)")
if len(sys.argv) > 2:
    prev = rows[0][1]
    for n, s, e in rows[:int(sys.argv[2])]:
        print(f"    +{(s - rows[0][1]) / 1e3:9.1f} us  gap {max(0, s - prev) / 1e3:6.1f}  {(e - s) / 1e3:8.1f} us  {short(n)}"); prev = max(prev, e)
