"""the exact two-label cut on cost planes no sane run produces (NaN, +-Inf, negative, 1e30, denormal): it must end in bounded time with labels in {-1, 0, 1} -- a diverged
window under occlusion reasoning hands the cut such costs.  usage (GPU box): python3 tools/hostile_cut_probe.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, slowflow_amd as sfa
ctx = sfa.Context(0)
rng = np.random.default_rng(3)
worst = 0.0
for (w, h) in ((67, 45), (130, 98), (300, 200)):
    st = sfa.stride_of(w)
    for name, v0, v1 in (("nan all", np.nan, np.nan), ("nan d0", np.nan, 1.0), ("inf d0", np.inf, 1.0), ("-inf d1", 1.0, -np.inf), ("inf both", np.inf, np.inf), ("negative", -1.0, -2.0),
                         ("huge", 1e30, 1e30), ("denormal", 1e-42, 2e-42), ("patches", None, None)):
        d0, d1 = np.zeros((h, st), np.float32), np.zeros((h, st), np.float32)
        d0[:, :w] = rng.uniform(0, 2, (h, w)); d1[:, :w] = rng.uniform(0, 2, (h, w))
        if v0 is None:
            for k, v in enumerate((np.nan, np.inf, -np.inf, 1e30, -1e30, -5.0)):
                d0[3 + 6 * k:6 + 6 * k, 5:w // 2] = v; d1[20 + 3 * k:22 + 3 * k, w // 3:w - 3] = v
        else:
            d0[::2, :w:3] = v0; d1[1::2, 1:w:3] = v1
        for alpha in (0.5, 0.0):
            t0 = time.perf_counter()
            try:
                occ = ctx.grid_cut(d0, d1, alpha, w)
                el = time.perf_counter() - t0
                vals = set(np.unique(occ[:, :w]).tolist())
                print(f"{w}x{h} {name:9s} alpha {alpha}: {el * 1e3:8.1f} ms  labels {sorted(vals)}", flush=True)
            except sfa.SlowflowError as e:
                el = time.perf_counter() - t0
                print(f"{w}x{h} {name:9s} alpha {alpha}: {el * 1e3:8.1f} ms  ERROR {str(e)[:120]}", flush=True)
            worst = max(worst, el)
print("worst %.1f ms" % (worst * 1e3))
