#!/usr/bin/env python3
"""Randomised layouts of the slow_flow driver (round 6): the same synthetic sequence and cfg run (a) the plain way -- one GPU, one window per batch, one stream -- and
(b) under a random layout: `gpus` 1-3 virtual GPUs on the one card (`gpu_oversubscribe 1`: the N-GPU path with its sharded ingest), `gpu_streams` 1-2 workers per GPU,
`gpu_batch` 1-9 windows per lockstep job, `io_threads`, `-threads`.  Every .flo of (b) must equal (a)'s byte for byte (what a window's flow is may not depend on who its
batch mates are, on which GPU holds its frames, or on the order the workers finish), the occlusion maps too, and run.json must account for every window.
(a) itself is tied to the Python binding by tests/test_host.py::test_slow_flow_driver_end_to_end.

usage (GPU box): python3 tools/fuzz_driver.py [seconds=240] [seed=0]"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from synth import smooth_noise_color

EXE = os.path.join(ROOT, "slowflow_amd", "host", "slow_flow")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t_end = time.time() + budget
fails, cases = [], 0


def write_ppm(path, img):
    h, w = img.shape[1:]
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(np.clip(np.round(img), 0, 255).astype(np.uint8).transpose(1, 2, 0).tobytes())


case_seed = seed0
while time.time() < t_end:
    case_seed += 1
    rng = np.random.default_rng(case_seed)
    w, h = int(rng.integers(32, 161)), int(rng.integers(24, 121))
    S = int(rng.choice([2, 2, 3]))
    steps = S - 1
    jets = int(rng.integers(1, 10))
    nframes = 1 + (jets + 2) * steps
    tmp = tempfile.mkdtemp(prefix="sfa_fd_")
    try:
        m = 4 * nframes + 8
        base = smooth_noise_color(rng, w + 2 * m, h + 2 * m, float(rng.uniform(25, 60)))
        dx, dy = int(rng.integers(0, 3)), int(rng.integers(0, 2))
        still_from = int(rng.integers(0, nframes + 3))                         # the tail of the sequence stands still: windows that meet a threshold at once
        for k in range(nframes):
            kk = min(k, still_from)
            write_ppm(os.path.join(tmp, "f_%03d.ppm" % (10 - steps + k)), base[:, m - dy * kk:m - dy * kk + h, m - dx * kk:m - dx * kk + w])
        occ = int(rng.random() < 0.35)
        alter = int(rng.integers(1, 3)) if occ else 1
        thres = float(rng.choice([0, 1e-3, 5e-3]))
        body = ("file\t%s/f_%%03i.ppm\nJets\t%d\nstart\t10\nmax_fps\t200\n16bit\t0\nraw\t0\nscale\t1.0\ndeep_matching\t0\n"
                "slow_flow_S\t%d\nslow_flow_layers\t%d\nslow_flow_niter_alter\t%d\nslow_flow_niter_outer\t%d\nslow_flow_niter_inner\t%d\nslow_flow_occlusion_reasoning\t%d\n"
                "slow_flow_thres_outer\t%g\nslow_flow_thres_inner\t%g\nslow_flow_output_occlusions\t%d\n" % (
                    tmp, jets, S, int(rng.integers(1, 4)), alter, int(rng.integers(1, 6)), int(rng.choice([1, 1, 2])), occ, thres, float(rng.choice([0, thres])), occ))
        if S == 2:
            body += "slow_flow_rho_0\t1\nslow_flow_omega_0\t%g\n" % float(rng.choice([0, 0, 1]))
        layout = dict(gpus=int(rng.integers(1, 4)), gpu_streams=int(rng.integers(1, 3)), gpu_batch=int(rng.choice([1, 2, 3, 4, 5, 9])), io_threads=int(rng.choice([1, 2, 8])))
        runs = {"plain": "gpus\t1\ngpu_streams\t1\ngpu_batch\t1\n",
                "layout": "gpus\t%(gpus)d\ngpu_oversubscribe\t1\ngpu_streams\t%(gpu_streams)d\ngpu_batch\t%(gpu_batch)d\nio_threads\t%(io_threads)d\n" % layout}
        ok, why = True, ""
        for name, extra in runs.items():
            cfg = os.path.join(tmp, name + ".cfg")
            with open(cfg, "w") as f:
                f.write("output\t%s/out_%s\n" % (tmp, name) + body + extra)
            args = [EXE, cfg, "-overwrite"] + (["-threads", str(int(rng.integers(1, 5)))] if name == "layout" and rng.random() < 0.5 else [])
            r = subprocess.run(args, capture_output=True, text=True, timeout=300)
            if r.returncode != 0 or "Done!" not in r.stdout:
                ok, why = False, "%s run failed (rc %d): %s" % (name, r.returncode, (r.stdout + r.stderr)[-300:])
                break
        if ok:
            a, b = os.path.join(tmp, "out_plain"), os.path.join(tmp, "out_layout")
            flo = sorted(f for f in os.listdir(a) if f.endswith(".flo"))
            if len(flo) != 2 * jets or sorted(f for f in os.listdir(b) if f.endswith(".flo")) != flo:
                ok, why = False, "file sets differ: %d / %d .flo for %d jets" % (len(flo), len([f for f in os.listdir(b) if f.endswith('.flo')]), jets)
            for f in flo if ok else []:
                if open(os.path.join(a, f), "rb").read() != open(os.path.join(b, f), "rb").read():
                    ok, why = False, "%s differs" % f
                    break
            if ok and occ:
                for f in sorted(os.listdir(os.path.join(a, "occlusion"))):
                    if open(os.path.join(a, "occlusion", f), "rb").read() != open(os.path.join(b, "occlusion", f), "rb").read():
                        ok, why = False, "occlusion/%s differs" % f
                        break
            if ok:
                rj = json.load(open(os.path.join(b, "run.json")))
                tj = json.load(open(os.path.join(b, "timings.json")))
                if rj["windows"] != 2 * jets or len(tj) != 2 * jets or len(rj["per_gpu"]) != layout["gpus"]:
                    ok, why = False, "run.json / timings.json do not account for the windows: %s" % json.dumps(rj)[:200]
        cases += 1
        print(f"[{case_seed}] {'ok  ' if ok else 'FAIL'} {w}x{h} S={S} jets={jets} occ={occ} alter={alter} thres={thres} still_from={still_from} layout={layout} {why}", flush=True)
        if not ok:
            fails.append((case_seed, why))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
print(f"{cases} cases, {len(fails)} failures (seeds {seed0 + 1} .. {case_seed})")
for f in fails:
    print("  FAILED", f)
sys.exit(1 if fails else 0)
