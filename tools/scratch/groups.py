import os, sys, time, threading
ROOT = "/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, slowflow_amd as sfa, bench
TOTAL = int(sys.argv[1]); S_ = int(sys.argv[2])
ctxs = [sfa.Context(0) for _ in range(S_)]
p = sfa.default_params(); p.S = 3; p.layers = bench.LAYERS; p.hbit = 0
p.rho[0] = 1; p.rho[1] = 1; p.omega[0] = 0; p.omega[1] = 2
base = [bench.synth_window(300 + b, n=5) for b in range(4)]
avg, std = ctxs[0].normalize([f for w in base for f in w], bench.W)
for k in range(3):
    p.norm_avg[k] = float("%g" % avg[k]); p.norm_std[k] = float("%g" % std[k])
per = TOTAL // S_
jobs = [sfa.Job(ctxs[g], p, bench.W, bench.H, per) for g in range(S_)]
for g, job in enumerate(jobs):
    for b in range(per): job.upload(b, base[(g * per + b) % 4])
def run_all():
    def work(g): jobs[g].run(); ctxs[g].sync()
    th = [threading.Thread(target=work, args=(g,)) for g in range(S_)]
    for t in th: t.start()
    for t in th: t.join()
run_all()
for var in sys.argv[3:]:
    for kv in var.split(","):
        k, v = kv.split("="); os.environ[k] = v
    t0 = time.perf_counter(); run_all(); sec = time.perf_counter() - t0
    print("%d windows in %d groups, %s: %.3f s, %.2f ms / window" % (TOTAL, S_, var, sec, sec / TOTAL * 1e3), flush=True)
