#!/usr/bin/env python3
"""Randomised parity of the HIP path against the oracle (round 6: GPU minutes spent on correctness instead of on a tenth of a per cent).

For `seconds` of wall time, seeded: random frame sizes (off every tile grid: 5 .. 330 columns, 5 .. 260 rows), window widths S = 2 / 3, directions, smoothing methods,
penalty ids, data-term normalisation, inner iterations, sweep counts and relaxation, break thresholds on or off, pyramid depths, channel weights, initial flows and lockstep
batch sizes --

  * `sor`:   a batch of random SPD systems through sfa_sor_batch (every solver shape the library picks by itself for that size and batch) against the raster-order oracle: IEEE ==
  * `level`: one level (compute_one_level) or the whole pyramid (variational) against the oracle on the same frames: max-abs (u, v) <= max(2e-5 [level] / 1e-4 [pyramid],
             3 x the oracle's own sensitivity to one-ulp input noise) -- the bound of tests/test_gpu_parity.py --, and the same windows as a lockstep job bit for bit what they give alone

Every case prints one line; a failure prints the case's parameters (reproducible: the case's seed) and the tool exits 1 at the end.
usage (GPU box): python3 tools/fuzz_parity.py [seconds=240] [seed=0]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle as orc
import slowflow_amd as sfa
from synth import copy_sys, noise_plane, smooth_noise_color, sor_system

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = sfa.Context(0)
o = orc.Oracle()
c_ = lambda a: np.ascontiguousarray(a, dtype=np.float32)
fails, cases = [], 0
t_end = time.time() + budget


def frames_for(rng, w, h, n):
    m = max(8, 2 * n)
    base = smooth_noise_color(rng, w + 2 * m, h + 2 * m, float(rng.uniform(20, 60)))
    dx, dy = int(rng.integers(0, 3)), int(rng.integers(0, 2))
    fr = []
    for k in range(n):
        f = orc.aligned_zeros((3, h, orc.stride_of(w)))
        f[:, :, :w] = base[:, m - dy * k:m - dy * k + h, m - dx * k:m - dx * k + w]
        fr.append(f)
    _, _, af, sf = o.normalize(fr, w)
    return fr, af, sf


def set_params(kw):
    po, ps = o.default_params(), sfa.default_params()
    for p in (po, ps):
        p.niter_alter = 1; p.occlusion_reasoning = 0; p.hbit = 0
        for k, v in kw.items():
            if k in ("rho", "omega", "norm_avg", "norm_std"):
                for i, x in enumerate(v):
                    getattr(p, k)[i] = x
            elif k in ("robust_color", "robust_grad", "robust_reg"):
                getattr(p, k).id, getattr(p, k).eps, getattr(p, k).trunc = v
            else:
                setattr(p, k, v)
    return po, ps


def sensitivity(po, frames, w, h, whole):
    outs = []
    for pert in (False, True):
        fr = []
        r = np.random.default_rng(1)
        for f in frames:
            g = orc.aligned_zeros(f.shape); g[...] = f
            if pert:
                for i in r.integers(0, h * w, 20):
                    y, x = divmod(int(i), w)
                    g[0, y, x] = np.nextafter(g[0, y, x], np.float32(1e9))
            fr.append(g)
        wx, wy = orc.plane(h, orc.stride_of(w)), orc.plane(h, orc.stride_of(w))
        (o.variational if whole else o.compute_one_level)(po, wx, wy, fr, w)
        outs.append((wx, wy))
    return float(max(np.abs(outs[0][0][:, :w] - outs[1][0][:, :w]).max(), np.abs(outs[0][1][:, :w] - outs[1][1][:, :w]).max()))


case_seed = seed0
while time.time() < t_end:
    case_seed += 1
    rng = np.random.default_rng(case_seed)
    kind = "sor" if rng.random() < 0.35 else "level"
    try:
        if kind == "sor":
            w, h = int(rng.integers(2, 331)), int(rng.integers(2, 261))
            K = int(rng.choice([1, 2, 3, 5, 6, 7, 10, 15, 30, 30, 30, 31]))
            nb = int(rng.choice([1, 1, 2, 3, 5, 8, 9, 12, 17, 40]))
            omega = float(rng.choice([1.0, 1.5, 1.9]))
            systems = [sor_system(rng, w, h) for _ in range(min(nb, 3))]
            for s in systems:
                if rng.random() < 0.5:
                    s["du"][:, :w] = rng.uniform(-.2, .2, (h, w)); s["dv"][:, :w] = rng.uniform(-.2, .2, (h, w))
            sb = sfa.SorBatch(ctx, w, h, nb)
            for b in range(nb):
                sb.upload(b, *[c_(systems[b % len(systems)][k]) for k in ("du", "dv", "a11", "a12", "a22", "b1", "b2", "sh", "sv")])
            ctx.profile_enable(True); sb.run(K, omega); kernel = ctx.profile_read_kernels()[3].split(" ")[0]; ctx.profile_enable(False)
            ok = True
            for b in sorted({0, nb - 1, nb // 2}):
                a = copy_sys(systems[b % len(systems)])
                o.sor(a["du"], a["dv"], a["a11"], a["a12"], a["a22"], a["b1"], a["b2"], a["sh"], a["sv"], w, K, omega)
                du, dv = sb.download(b)
                ok = ok and np.array_equal(a["du"][:, :w], du[:, :w]) and np.array_equal(a["dv"][:, :w], dv[:, :w])
            sb.close()
            desc = f"sor {w}x{h} K={K} nb={nb} omega={omega} {kernel}"
            d = tol = 0.0
        else:
            S = int(rng.choice([2, 2, 3]))
            w, h = int(rng.integers(5, 331)), int(rng.integers(5, 261))
            whole = rng.random() < 0.4 and w >= 40 and h >= 40
            pid = lambda: int(rng.choice([1, 1, 1, 2, 3, 4, 0]))
            pen = lambda i: (i, 0.001 if i in (1, 3) else 0.05, 0.5)
            kw = dict(S=S, niter_outer=int(rng.integers(1, 5)), niter_inner=int(rng.choice([1, 1, 2, 3])), niter_solver=int(rng.choice([30, 30, 15, 7, 10])),
                      sor_omega=float(rng.choice([1.9, 1.5])), smoothing=int(rng.choice([1, 1, 0, 2])), dataterm_norm=int(rng.choice([1, 1, 0])),
                      one_direction=int(rng.random() < 0.2), delta=float(rng.choice([1.0, 0.0, 0.5])), gamma=float(rng.choice([6.0, 0.2])), alpha=float(rng.choice([4.0, 1.0])),
                      robust_color=pen(pid()), robust_grad=pen(pid()), robust_reg=pen(pid()),
                      thres_outer=float(rng.choice([0, 0, 2e-3])), thres_inner=float(rng.choice([0, 0, 1e-3])),
                      layers=int(rng.integers(2, 5)) if whole else 1)
            if S == 2:
                kw["rho"] = [1.0]; kw["omega"] = [float(rng.choice([0, 0, 1.0]))]
            else:
                kw["rho"] = [1.0, float(rng.choice([1.0, 0.5, 0.0]))]; kw["omega"] = [float(rng.choice([0, 0.5])), float(rng.choice([2.0, 0.0, 1.0]))]
            fr, af, sf = frames_for(rng, w, h, 2 * S - 1)
            kw["norm_avg"] = af; kw["norm_std"] = sf
            po, ps = set_params(kw)
            chw = [noise_plane(rng, w, h, 0.5, 1.5) for _ in range(3)] if rng.random() < 0.25 else None
            init = (noise_plane(rng, w, h, -1, 1), noise_plane(rng, w, h, -1, 1)) if rng.random() < 0.3 else None
            stride = orc.stride_of(w)
            wxo, wyo = orc.plane(h, stride), orc.plane(h, stride)
            if init is not None:
                wxo[...] = init[0]; wyo[...] = init[1]
            wxg, wyg = c_(wxo).copy(), c_(wyo).copy()
            if whole:
                rc = o.variational(po, wxo, wyo, fr, w, chw)[0]
                ctx.variational(ps, wxg, wyg, [c_(f) for f in fr], w, [c_(x) for x in chw] if chw else None)
            else:
                rc = o.compute_one_level(po, wxo, wyo, fr, w, chw)[0]
                ctx.compute_one_level(ps, wxg, wyg, [c_(f) for f in fr], w, [c_(x) for x in chw] if chw else None)
            base_tol = 1e-4 if whole else 2e-5
            fo = np.isfinite(wxo[:, :w]) & np.isfinite(wyo[:, :w])
            if fo.all():
                d = float(max(np.abs(wxo[:, :w] - wxg[:, :w]).max(), np.abs(wyo[:, :w] - wyg[:, :w]).max()))
                tol = base_tol if d <= base_tol else max(base_tol, 3 * sensitivity(po, fr, w, h, whole))
                ok = rc == 0 and np.isfinite(d) and d <= tol
            else:
                # an ill-posed parameter set: the ORACLE's refinement diverged (singular 2 x 2 blocks: NaN / Inf).  Then the bar is: no fault, and the GPU's field is
                # not a number where the oracle's is not (a diverging field has no meaningful tolerance on its finite part)
                fg = np.isfinite(wxg[:, :w]) & np.isfinite(wyg[:, :w])
                d = float((fo != fg).mean())
                tol = 0.02
                ok = rc == 0 and d <= tol
                desc_extra = " [oracle diverged: %.0f %% non-finite; d = share of pixels whose finiteness differs]" % (100.0 * (1 - fo.mean()))
            # the same window inside a lockstep job: bit for bit what it gives alone
            nb = int(rng.choice([1, 2, 5]))
            if ok and fo.all() and chw is None and init is None and rng.random() < 0.5:
                alone = sfa.Job(ctx, ps, w, h, 1)
                alone.upload(0, [c_(f) for f in fr]); alone.run(); ax, ay, _ = alone.download(0); alone.close()
                job = sfa.Job(ctx, ps, w, h, nb)
                for b in range(nb):
                    job.upload(b, [c_(f) for f in fr])
                job.run()
                for b in range(nb):
                    gx, gy, _ = job.download(b)
                    ok = ok and np.array_equal(gx, ax) and np.array_equal(gy, ay)
                job.close()
                if whole:                                            # ... and the binding's result for the whole pyramid
                    ok = ok and np.array_equal(ax[:, :w], wxg[:, :w]) and np.array_equal(ay[:, :w], wyg[:, :w])
            desc = f"{'pyramid' if whole else 'level'} {w}x{h} " + (desc_extra if not fo.all() else "") + " ".join(f"{k}={v}" for k, v in kw.items() if k not in ("norm_avg", "norm_std")) + f" chw={chw is not None} init={init is not None} job={nb}"
        cases += 1
        print(f"[{case_seed}] {'ok  ' if ok else 'FAIL'} d={d:.3g} tol={tol:.3g} {desc}", flush=True)
        if not ok:
            fails.append((case_seed, desc, d, tol))
    except sfa.SlowflowError as e:
        cases += 1
        print(f"[{case_seed}] ERROR {kind}: {e}", flush=True)
        fails.append((case_seed, "error: " + str(e)[:200], 0, 0))
print(f"{cases} cases, {len(fails)} failures (seeds {seed0 + 1} .. {case_seed})")
for f in fails:
    print("  FAILED", f)
ctx.close()
sys.exit(1 if fails else 0)
