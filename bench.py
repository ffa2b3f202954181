#!/usr/bin/env python3
"""bench.py -- headline measurement of the variational-refinement hot path on MI355X.

Metric (BASELINE.json): Mpix*solver-iters/s at 1024x436, with the SOR kernel's achieved algorithmic GB/s against
the HBM peak.  A *step* = one complete coarse-to-fine refinement (sfa_job_run: pyramid, per-level warps, derivative
stacks, data-term assembly, SOR solves, flow updates) of one resident batch of synthetic frame windows -- BASELINE
config 2: 1024x436, S=2 (3 frames), 5 pyramid levels, 5 outer x 1 inner x 30 SOR sweeps per level.  Inputs are
uploaded to HBM before the timed region.  value = (sum over all ranks and SOR solves of w*h*K) / wall time / 1e6.

  python bench.py --gpus N --steps K --warmup W [--batch B]
N > 1 runs one rank per GPU over RCCL: either the caller starts the ranks (`python -m torch.distributed.run ... bench.py
--gpus N ...`: RANK / WORLD_SIZE are in the environment) or, when they are not, bench.py starts them itself as a CHILD
torch.distributed.run process before anything touches the GPU and relays its output and exit code (launch_ranks).  Frame
windows shard across ranks with no data-path collective (weak scaling); the only exchange is a gather of per-rank timings.

The JSON line also carries `roofline` (SOR solve kernel: algorithmic bytes / HIP-event duration over the timed
region) and, on rank 0 at N=1, `cpu_baseline` (the reference's own sor_coupled from oracle/_ref, or the oracle port,
timed on one host core over a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import slowflow_amd as sfa  # noqa: E402

W, H, LAYERS, OUTER, INNER, SWEEPS, S = 1024, 436, 5, 5, 1, 30, 2
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
CLOCK_GHZ = 2.4              # peak engine clock the issue-rate and latency-floor figures are quoted at


def assemble_valu_per_pixel_term():
    """VALU instructions one pixel spends per data term in k_assemble_images (staging, filters, term arithmetic, epilogue -- everything the kernel issues),
    from the SQ counter pass of the newest profile round that has it: SQ_INSTS_VALU / SQ_WAVES (both counters sample the same subset of the launch's waves, so
    their ratio is per wave; rounds 1-3 divided the raw instruction count by ALL pixels and reported 440 where a wave really issues ~1950 for its 64 pixels x 2
    terms) / 2 terms.  Returns (instructions, source, share of SIMD time with a VALU instruction in flight when the kernel runs alone)."""
    for tag in ("r06", "r05", "r04", "r03", "r02"):
        try:
            with open(os.path.join(ROOT, "profiles", tag + "_sq.json")) as f:
                j = json.load(f)
            v = j.get("assemble_valu_inst_per_wave")
            v = v / 2.0 if v else None
            if not v and tag == "r03":
                v = 1957.0 / 2.0                         # profiles/r03_sq_by_kernel.csv: 372131681 / 190151
            if v:
                act = None
                try:
                    act = max(x["valu_active_frac"] for k, x in j.get("kernels", {}).items() if "k_assemble_images" in k)
                except (ValueError, KeyError):
                    pass
                return float(v), "profiles/%s_sq.json" % tag, act
        except (OSError, ValueError):
            continue
    return None, None, None


def sor_launch_waves(kernel, nb):
    """waves of the mean SOR launch of the bench workload (5 levels): windows x bands x groups x waves per workgroup, from the chain kernel's shape
    k_sor_chain<FA, NA, FB, NB, ...> (NA + NB compute waves + 2 I/O waves; groups = K / (FA NA + FB NB); bands = ceil((h + K - 1) / 64), sor.hip)"""
    import re
    m = re.search(r"k_sor_chain<\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)", kernel)
    if not m:
        return None
    fa, na, fb, nb_ = (int(x) for x in m.groups())
    kg = fa * na + fb * nb_
    if kg <= 0 or SWEEPS % kg:
        return None
    hs, h = [], H
    w = W
    for _ in range(LAYERS):
        hs.append(h)
        w, h = int(np.floor(np.float32(w) * np.float32(0.9))), int(np.floor(np.float32(h) * np.float32(0.9)))
    bands = np.mean([(hh + SWEEPS - 1 + 63) // 64 for hh in hs])
    fill = 1 if (fa == 1 and nb_ == 0 and na < 8) else 0        # the one-sweep shapes carry a FILL wave (ChainShape::INFILL)
    return nb * bands * (SWEEPS // kg) * (na + nb_ + 2 + fill)


def limiter(kernel, batch, streams):
    """What the solver kernel waits for -- a statement measured for ONE configuration (k_sor_chain<3,5,3,0,...>, 64 windows per launch: what-if builds of
    DESIGN.md 5.1, profiles/r03_chain_whatif.txt) and printed only for it; any other shape or batch gets the plain label."""
    k = kernel.replace(" ", "")
    if "k_sor_chain<2,6,3,1" in k and batch >= 10:
        return ("VALU time of the busiest SIMDs: one 9-wave workgroup per CU (its operand ring fills the LDS), seven compute waves of 2,2,2,2,2,2,3 sweeps -- the first six "
                "two to a SIMD (4 sweeps each), the last beside the two I/O waves; a packed operation takes two passes, so a pair of compute waves keeps its SIMD busy nearly "
                "all of an interval (DESIGN.md 5.1: the six-stage shape of round 4 put 5 sweeps on two of the SIMDs)")
    if "k_sor_chain<3,3,2,3" in k and batch >= 16:
        return ("wave issue rate: one 8-wave workgroup per CU (its operand ring fills the LDS), six compute waves of 3,3,3,2,2,2 sweeps two to a SIMD -- the SIMDs that "
                "carry 5 sweeps set the pace (the 5 x 3 shape of round 3, 6 sweeps on one SIMD, was within 6 % of its compute-only build: DESIGN.md 5.1)")
    if "k_sor_chain<3,5,3,0" in k and batch == 64:
        return ("wave issue rate: one 7-wave workgroup per CU (its operand ring fills the LDS), five compute waves on four SIMDs each issuing one instruction per "
                "4-8 cycles; a build whose I/O waves only walk the barriers needs 0.94 of the launch, one without operand loads 0.97 (DESIGN.md 5.1)")
    if batch <= 8:
        return "dependency latency: W + H + 2K hyperplane steps of ~14 dependent packed operations each, plus the start-up skew of the bands (DESIGN.md 5.1)"
    return "dependency latency / wave issue rate (not measured for this shape and batch size)"


def sor_valu_per_wave(kernel):
    """(VALU instructions per wave of the SOR kernel, waves per workgroup, source) from the newest SQ pass whose kernel shape is the one that ran."""
    def norm(k):
        return k.replace("void ", "").replace("sfa::", "").split("(")[0].replace(" ", "")
    for tag in ("r06", "r05", "r04", "r03"):
        try:
            with open(os.path.join(ROOT, "profiles", tag + "_sq.json")) as f:
                j = json.load(f)
            if j.get("kernel") and norm(j["kernel"]) == norm(kernel) and j.get("sor_valu_inst_per_wave"):
                return float(j["sor_valu_inst_per_wave"]), "profiles/%s_sq.json" % tag
        except (OSError, ValueError):
            continue
    return None, None


def valu_model():
    """profiles/<tag>_valu_model.json of the newest round that has one (tools/valu_time_model.py: the VALU mix of the solver's stages and of the assembly kernel from
    their ISA, priced with the measured pass costs of tools/ubench/valu_rates.hip)"""
    for tag in ("r06",):
        try:
            with open(os.path.join(ROOT, "profiles", tag + "_valu_model.json")) as f:
                return json.load(f), "profiles/%s_valu_model.json" % tag
        except (OSError, ValueError):
            continue
    return None, None


def solver_valu_floor(kernel, nb):
    """VALU-time floor of the mean solver launch of the bench workload (one launch per level and outer iteration, nb windows each): for every level the launch's
    workgroups / 256 CUs x the VALU cycles the BUSIEST SIMD of a workgroup needs over the workgroup's life (its waves' instruction mix x the measured pass costs; a
    SIMD executes one VALU instruction at a time whichever wave it comes from) / clock -- the time the launch would take if that SIMD never waited.  Also the mean
    over the four SIMDs.  Returns (seconds busiest, seconds mean, per-SIMD cycles per step at level 0, source) or None when the model has no such shape."""
    model, src = valu_model()
    if not model:
        return None
    k = kernel.replace(" ", "")
    m = next((v for name, v in model["solver"].items() if k.startswith(name.replace(" ", ""))), None)
    if not m:
        return None
    sh = m["shape"]
    KG, NW, CH, FA, NA, FB = sh["KG"], sh["NW"], sh["CH"], sh["FA"], sh["NA"], sh["FB"]
    if SWEEPS % KG:
        return None
    FMAX, AH = max(FA, FB if sh["NB"] else FA), 2
    LEAD, shift = AH + 1, (2 if KG <= 10 else 4)                          # sor_chain.hip: chain_start_shift (every default shape has an operand ring)
    cyc = {}
    for st in m["stages"]:
        cyc[(st["F"], st["role"].startswith("first"))] = st["valu_cycles_per_step"]

    def stage_cycles(F, first):
        return cyc.get((F, first), cyc.get((F, False)))
    io = {who: v["valu_cycles_per_interval"] for who, v in m["io_waves"].items()}
    ws, hs = sfa.pyramid_sizes(W, H, LAYERS, np.float32(0.9))
    rnd = lambda a, b: (a + b - 1) // b * b
    busiest = mean = 0.0
    per_step0 = None
    for w_, h_ in zip(ws, hs):
        NB = (h_ + SWEEPS - 1 + 63) // 64
        NCH = rnd((w_ + 64 + KG - NW + 2 * CH + FMAX + CH - 1) // CH + shift, 4)          # sor.hip: chunks per stage
        NI = rnd(NCH + LEAD + NW + 2, AH)                                                    # barrier intervals per workgroup
        simd = []
        for roles in m["simd_placement"]:
            c = 0.0
            for r in roles:
                if r in io:
                    c += io[r] * NI
                elif r.startswith("stage"):
                    idx, F = int(r.split()[1]), int(r.split("F=")[1].rstrip(")"))
                    c += stage_cycles(F, idx == 0) * NCH * CH
            simd.append(c)
        if per_step0 is None:
            per_step0 = [round(c / (NCH * CH), 1) for c in simd]
        nwg = nb * NB * (SWEEPS // KG)
        busiest += nwg / 256.0 * max(simd) / (CLOCK_GHZ * 1e9)
        mean += nwg / 256.0 * (sum(simd) / 4.0) / (CLOCK_GHZ * 1e9)
    return busiest / len(ws), mean / len(ws), per_step0, src


def synth_window(seed, w=W, h=H, n=3):
    """Config-2 stand-in: seeded band-limited noise texture moved by a smooth flow field of at most
    3 px per frame, 8-bit quantised; returns n (3,h,stride) fp32 frames."""
    rng = np.random.default_rng(seed)
    stride = sfa.stride_of(w)
    pad = 16
    from scipy.ndimage import gaussian_filter
    base = gaussian_filter(rng.uniform(0, 1, size=(3, h + 2 * pad, w + 2 * pad)), sigma=(0, 2.0, 2.0), mode="nearest")
    base = (base - base.min()) / (base.max() - base.min()) * 255.0
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    fu = 2.0 + 1.0 * np.sin(2 * np.pi * yy / h)          # |flow| <= 3 px
    fv = 1.0 * np.cos(2 * np.pi * xx / w)
    frames = []
    for t in range(n):
        sx, sy = xx - t * fu + pad, yy - t * fv + pad
        x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
        ax, ay = sx - x0, sy - y0
        f = np.zeros((3, h, stride), np.float32)
        for c in range(3):
            b = base[c]
            v = (b[y0, x0] * (1 - ax) * (1 - ay) + b[y0, x0 + 1] * ax * (1 - ay) + b[y0 + 1, x0] * (1 - ax) * ay + b[y0 + 1, x0 + 1] * ax * ay)
            f[c, :, :w] = np.round(v).astype(np.float32)
        frames.append(f)
    return frames


def bench_params():
    p = sfa.default_params()
    p.S = S; p.layers = LAYERS; p.niter_alter = 1; p.niter_outer = OUTER; p.niter_inner = INNER; p.niter_solver = SWEEPS
    p.thres_outer = 0; p.thres_inner = 0            # fixed work: exactly OUTER x SWEEPS per level
    p.occlusion_reasoning = 0; p.hbit = 0
    p.rho[0] = 1; p.omega[0] = 0
    return p


PARITY_WINDOWS = (0, 63, 64, 127)      # both words of the job's window mask, first and last bit of each (sfa_internal.h: WMask)


def reference_statistics(frames, w):
    """normalize()'s statistics (variational_mt.cpp:26-52) evaluated with numpy on the RAW frames: per frame and channel the fp64 sum of I and of the fp32
    product I*I over the valid pixels, each divided by w*h and accumulated in frame order, then mean and sqrt(mean of squares - mean^2) / 255.  The bench's
    frames are 8-bit values, so every partial sum is an integer below 2^53 and the result does not depend on the summation order: the GPU's tree sum, the
    oracle's raster sum and this evaluation must agree to the last bit."""
    avg, sq = np.zeros(3), np.zeros(3)
    for f in frames:
        _, h, _ = f.shape
        v = f[:, :, :w]
        avg += v.sum(axis=(1, 2), dtype=np.float64) / (h * w)
        sq += (v * v).sum(axis=(1, 2), dtype=np.float64) / (h * w)
    avg /= len(frames)
    std = np.sqrt(sq / len(frames) - avg * avg) / 255.0
    return avg, std


def cpu_baseline(windows=None, stats=None, budget_s=12.0):
    """The same workload on the host: the CPU restatement of the whole path (oracle/, kind "port": the reference's own
    Variational_MT needs OpenCV/GCO and cannot be built here) refining frame windows of the bench configuration on ONE core,
    as the reference runs a window single-threaded.  `windows` = [(index, normalised frames)] are the very windows the GPU job
    timed (identical inputs -- the frames as normalize() left them and the sequence statistics published at six
    digits, variational_mt.cpp:71-84); their flow fields come back in out["_flows"] for the caller's parity figure.  Beside it,
    the reference's own compiled sor_coupled (oracle/_ref, solver.c:63) on the metric's 1024x436 x 30 solve."""
    import oracle as orc
    from synth import copy_sys, sor_system
    o = orc.Oracle()
    p = o.default_params()
    p.S = S; p.layers = LAYERS; p.niter_alter = 1; p.niter_outer = OUTER; p.niter_inner = INNER; p.niter_solver = SWEEPS
    p.thres_outer = 0; p.thres_inner = 0; p.occlusion_reasoning = 0; p.hbit = 0
    p.rho[0] = 1; p.omega[0] = 0
    n, t_total, mpix = 0, 0.0, 0.0
    flows = {}
    if windows is None:                                # (stand-alone use: windows of its own)
        windows = []
        for i in range(4):
            fr = []
            for f in synth_window(5000 + i):
                a = orc.aligned_zeros(f.shape)
                a[...] = f
                fr.append(a)
            _, _, af, sf = o.normalize(fr, W)
            windows.append((i, fr, (af, sf)))
    for item in windows:
        idx, frames = item[0], item[1]
        af, sf = item[2] if len(item) > 2 else stats
        fr = []
        for f in frames:
            a = orc.aligned_zeros(f.shape)
            a[...] = f
            fr.append(a)
        for k in range(3):
            p.norm_avg[k] = af[k]; p.norm_std[k] = sf[k]
        wx, wy = orc.plane(H, orc.stride_of(W)), orc.plane(H, orc.stride_of(W))
        t0 = time.perf_counter()
        o.variational(p, wx, wy, fr, W)
        t_total += time.perf_counter() - t0
        flows[idx] = (wx[:, :W].copy(), wy[:, :W].copy())
        n += 1
    ws, hs = sfa.pyramid_sizes(W, H, LAYERS, p.p_scale)
    mpix = n * sum(w_ * h_ for w_, h_ in zip(ws, hs)) * OUTER * INNER * SWEEPS / 1e6
    out = {"value": round(mpix / t_total, 2), "unit": "Mpix*solver-iters/s", "cores": 1, "kind": "port",
           "sample": f"{n} frame window(s) of the bench configuration through the whole path ({t_total:.1f} s, {t_total / n:.2f} s per window): "
                     f"windows {sorted(flows)} of the GPU's timed job, same normalised frames and statistics", "_flows": flows}
    # the solver alone, the reference's own code
    rng = np.random.default_rng(0)
    s0 = sor_system(rng, W, H)
    if orc.ref_available():
        lib, kind = orc.RefLib(), "reference"
    else:
        lib, kind = o, "port"
    k, t_sor = 0, 0.0
    while t_sor < budget_s / 2 and k < 400:
        s = copy_sys(s0)
        t0 = time.perf_counter()
        lib.sor(s["du"], s["dv"], s["a11"], s["a12"], s["a22"], s["b1"], s["b2"], s["sh"], s["sv"], W, SWEEPS, 1.9)
        t_sor += time.perf_counter() - t0
        k += 1
    out["sor_only"] = {"value": round(k * W * H * SWEEPS / 1e6 / t_sor, 2), "unit": "Mpix*solver-iters/s", "cores": 1, "kind": kind,
                       "sample": f"{k} sor_coupled calls, {SWEEPS} sweeps each, {W}x{H}, one thread ({t_sor:.1f} s)"}
    # the same solver on every core this process may use, one independent solve per core at a time -- how the reference uses a node
    # (one window per OpenMP thread, slow_flow.cpp:706).  Separate worker processes that never touch the GPU.
    try:
        import subprocess, sys as _sys
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = max(1, min(cores, 64))
        code = ("import sys,time; sys.path.insert(0,%r); sys.path.insert(0,%r)\n"
                "import numpy as np, oracle as orc\nfrom synth import copy_sys, sor_system\n"
                "lib = orc.RefLib() if orc.ref_available() else orc.Oracle()\n"
                "s0 = sor_system(np.random.default_rng(0), %d, %d)\nk = 0; t_end = time.perf_counter() + 4.0; t0 = time.perf_counter()\n"
                "while time.perf_counter() < t_end:\n"
                "    s = copy_sys(s0); lib.sor(s['du'], s['dv'], s['a11'], s['a12'], s['a22'], s['b1'], s['b2'], s['sh'], s['sv'], %d, %d, 1.9); k += 1\n"
                "print(k, time.perf_counter() - t0)\n") % (ROOT, os.path.join(ROOT, "tests"), W, H, W, SWEEPS)
        procs = [subprocess.Popen([_sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(cores)]
        rate = 0.0
        for pr in procs:
            so, _ = pr.communicate(timeout=120)
            kk, tt = so.split()
            rate += int(kk) * W * H * SWEEPS / 1e6 / float(tt)
        model = ""
        try:
            model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            pass
        out["sor_only_all_cores"] = {"value": round(rate, 1), "unit": "Mpix*solver-iters/s", "cores": cores, "kind": kind, "cpu": model,
                                     "sample": f"{cores} worker processes, each solving {W}x{H} x {SWEEPS} sweeps back to back for 4 s"}
    except Exception as e:                                        # a reported extra, never a reason to lose the bench line
        out["sor_only_all_cores"] = {"error": str(e)[:200]}
    # the WHOLE PATH on every core, one frame window per worker process at a time -- the node-level figure SURVEY.md 8(d)(ii) asks for
    # ("OMP over jets with threads = physical cores", slow_flow.cpp:706): each worker refines distinct windows of the bench configuration
    # with the oracle port for about 8 s
    try:
        import subprocess, sys as _sys
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = max(1, min(cores, 64))
        code = ("import sys,time,os; sys.path.insert(0,%r); sys.path.insert(0,%r)\n"
                "os.environ['OMP_NUM_THREADS']='1'\n"
                "import numpy as np, oracle as orc, bench\n"
                "o = orc.Oracle(); p = o.default_params()\n"
                "p.S = bench.S; p.layers = bench.LAYERS; p.niter_alter = 1; p.niter_outer = bench.OUTER; p.niter_inner = bench.INNER; p.niter_solver = bench.SWEEPS\n"
                "p.thres_outer = 0; p.thres_inner = 0; p.occlusion_reasoning = 0; p.hbit = 0; p.rho[0] = 1; p.omega[0] = 0\n"
                "seed = int(sys.argv[1]); n = 0; t = 0.0\n"
                "while t < 8.0 and n < 4:\n"
                "    fr = []\n"
                "    for f in bench.synth_window(7000 + 10 * seed + n):\n"
                "        a = orc.aligned_zeros(f.shape); a[...] = f; fr.append(a)\n"
                "    _, _, af, sf = o.normalize(fr, bench.W)\n"
                "    for k in range(3): p.norm_avg[k] = af[k]; p.norm_std[k] = sf[k]\n"
                "    wx, wy = orc.plane(bench.H, orc.stride_of(bench.W)), orc.plane(bench.H, orc.stride_of(bench.W))\n"
                "    t0 = time.perf_counter(); o.variational(p, wx, wy, fr, bench.W); t += time.perf_counter() - t0; n += 1\n"
                "print(n, t)\n") % (ROOT, os.path.join(ROOT, "tests"))
        env = dict(os.environ); env["OMP_NUM_THREADS"] = "1"; env["HIP_VISIBLE_DEVICES"] = ""          # the workers never touch the GPU
        t_wall = time.perf_counter()
        procs = [subprocess.Popen([_sys.executable, "-c", code, str(i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for i in range(cores)]
        nwin, rate = 0, 0.0
        per_window = sum(w_ * h_ for w_, h_ in zip(ws, hs)) * OUTER * INNER * SWEEPS / 1e6
        for pr in procs:
            so, _ = pr.communicate(timeout=300)
            kk, tt = so.split()
            nwin += int(kk); rate += int(kk) * per_window / float(tt)
        t_wall = time.perf_counter() - t_wall
        model = ""
        try:
            model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            pass
        out["all_cores"] = {"value": round(rate, 1), "unit": "Mpix*solver-iters/s", "cores": cores, "kind": "port", "cpu": model,
                            "sample": f"{cores} worker processes, {nwin} frame windows of the bench configuration through the whole path in all "
                                      f"({t_wall:.1f} s wall including start-up; rate = sum over workers of windows / their own refinement time)"}
    except Exception as e:
        out["all_cores"] = {"error": str(e)[:200]}
    return out


def hbm_triad_gbs(torch):
    """what this box's HBM delivers to a plain streaming kernel (a = b + s * c on 3 x 1 GiB, torch elementwise): printed beside the
    8 TB/s vendor peak that `roofline.frac` divides by (SURVEY.md 8d asks for both)"""
    n = 1 << 28
    b = torch.ones(n, dtype=torch.float32, device="cuda"); c = torch.ones(n, dtype=torch.float32, device="cuda"); a = torch.empty_like(b)
    for _ in range(2):
        torch.add(b, c, alpha=2.0, out=a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for _ in range(reps):
        torch.add(b, c, alpha=2.0, out=a)
    e1.record(); torch.cuda.synchronize()
    return 3.0 * 4 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def measured_traffic(batch, kernel=None):
    """HBM-side bytes per SOR launch from the PMC passes of profiles/collect.sh (FETCH_SIZE x2 on gfx950 for wide
    reads + WRITE_SIZE, separate passes; MI355X_MICROARCH.md).  PMC counters cannot be read from inside this process,
    so the committed measurement of the same command is reported, and only when it was taken at this batch size.
    Returns (bytes per launch, source, valu_busy) -- valu_busy = SQ_ACTIVE_INST_VALU x 4 cycles / (SIMDs x kernel cycles) of the SOR kernel
    when the SQ pass of the same round is there (profiles/<tag>_sq.json), else None."""
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", tag + "_traffic.json")
        try:
            with open(path) as f:
                t = json.load(f)
            if int(t["batch"]) != int(batch):
                continue
            # a counter file taken from another kernel (shape) says nothing about this run: refuse it
            def norm(k):
                return k.replace("void ", "").replace("sfa::", "").split("(")[0].replace(" ", "")
            if kernel is not None and t.get("kernel") and norm(t["kernel"]) != norm(kernel):
                continue
            valu = None
            try:
                with open(os.path.join(ROOT, "profiles", tag + "_sq.json")) as f:
                    valu = json.load(f).get("valu_busy_frac")
            except (OSError, ValueError):
                pass
            return t["traffic_bytes_per_launch"], "profiles/%s_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, python bench.py --path-only)" % tag, valu
        except (OSError, KeyError, ValueError):
            continue
    return None, None, None


def one_window_latency(ctx, reps=3):
    """the whole path for ONE frame window on one stream, inputs resident: what a single pair (BASELINE config 2 is literally one) costs.
    Returns (ms in the reference order, dict for the labelled red-black mode: its latency and how far its flow is from the reference order's)."""
    win = synth_window(1)
    avg, std = ctx.normalize(win, W)
    out = []
    for order in (0, 1):
        p = bench_params()
        p.sor_order = order
        for k in range(3):
            p.norm_avg[k] = float("%g" % avg[k]); p.norm_std[k] = float("%g" % std[k])
        job = sfa.Job(ctx, p, W, H, 1)
        job.upload(0, win)
        job.run(); ctx.sync()
        # the latency WITHOUT the per-launch events of the library's profiling (two hipEventRecords around each of a run's 50 solver and data-term launches: ~0.3 ms of
        # a lone window's 6.5), then one profiled run for the per-solve figure
        t0 = time.perf_counter()
        for _ in range(reps):
            job.run()
        ctx.sync()
        ms = (time.perf_counter() - t0) / reps * 1e3
        ctx.profile_enable(True)
        job.run(); ctx.sync()
        n, sor_ms, _ = ctx.profile_read()
        ctx.profile_enable(False)
        wx, wy, _ = job.download(0)
        job.close()
        out.append((ms, sor_ms / max(n, 1), wx[:, :W], wy[:, :W]))
    dev = float(max(np.abs(out[0][2] - out[1][2]).max(), np.abs(out[0][3] - out[1][3]).max()))
    rb = {"label": "slow_flow_sor_order red_black: a DIFFERENT ALGORITHM (two-colour sweeps), never the default; does not reproduce the reference",
          "latency_one_window_ms": round(out[1][0], 3), "sor_ms_per_solve_avg_over_levels": round(out[1][1], 4),
          "max_abs_flow_deviation_from_reference_order_px": round(dev, 5), "meets_1e-4_parity": bool(dev <= 1e-4)}
    return out[0][0], out[0][1], rb

def cfg_schedule_sample(ctx, B=16):
    """what the driver's DEFAULT schedule costs per window (slow_flow.cpp:64-128: 10 alternations x 10 outer iterations with the cfg's break
    thresholds and occlusion reasoning on; windows of a lockstep batch then stop at different iterations and ride along as passengers):
    B windows of the bench size on one stream.  An extra beside the fixed-work metric, never part of `value`."""
    p = sfa.default_params()
    p.S = S; p.layers = LAYERS; p.hbit = 0
    wins = [synth_window(900 + b) for b in range(min(B, 4))]
    avg, std = ctx.normalize([f for w in wins for f in w], W)
    for k in range(3):
        p.norm_avg[k] = float("%g" % avg[k]); p.norm_std[k] = float("%g" % std[k])
    job = sfa.Job(ctx, p, W, H, B)
    for b in range(B):
        job.upload(b, wins[b % len(wins)])
    job.run(); ctx.sync()
    t0 = time.perf_counter()
    job.run(); ctx.sync()
    sec = time.perf_counter() - t0
    job.close()
    return {"windows": B, "streams": 1, "seconds_per_window": round(sec / B, 5),
            "schedule": "sfa_params_default: %d alternations x %d outer x %d inner x %d sweeps, thresholds %g / %g, occlusion reasoning %d, %d levels" % (
                p.niter_alter, p.niter_outer, p.niter_inner, p.niter_solver, p.thres_outer, p.thres_inner, p.occlusion_reasoning, p.layers)}


def run_groups(work, n):
    """work(g) for g in range(n) on one host thread each; an exception in any of them is re-raised here after every thread has ended
    (threading.Thread swallows it otherwise, and a failed refinement would still be timed)"""
    import threading
    errs = [None] * n

    def guarded(g):
        try:
            work(g)
        except BaseException as e:                    # re-raised below
            errs[g] = e
    if n == 1:
        guarded(0)
    else:
        th = [threading.Thread(target=guarded, args=(g,)) for g in range(n)]
        for t in th: t.start()
        for t in th: t.join()
    for e in errs:
        if e is not None:
            raise e


def strong_section(ctxs, world, rank, dist, xdev, total_windows, make_params, w, h, n_frames, label):
    """A FIXED set of frame windows partitioned over the ranks (shard.partition: contiguous blocks, no data-path collective) -- STRONG scaling.
    Wall time = max over ranks of one resident refinement of the rank's share (each rank's windows in len(ctxs) lockstep groups).
    Everything that can fail (allocation, upload, the refinement) runs under try on every rank and only sets a flag; the collectives (two barriers, the
    max of the seconds, the min of the flags) are then reached by EVERY rank in the same order, failed or not, so one rank's error can never leave
    the others waiting in a collective."""
    from slowflow_amd import shard
    lo, hi = shard.partition(total_windows, world, rank)
    nloc = hi - lo
    S_ = len(ctxs)
    jobs, err, sec, p = [], None, 0.0, None
    groups = [(nloc * g // S_, nloc * (g + 1) // S_) for g in range(S_)]

    def run_all():
        def work(g):
            if jobs[g] is not None:
                jobs[g].run(); ctxs[g].sync()
        run_groups(work, S_)
    try:
        p = make_params()
        base = [synth_window(300 + b, n=n_frames, w=w, h=h) for b in range(4)]
        avg, std = ctxs[0].normalize([f for w_ in base for f in w_], w)
        for k in range(3):
            p.norm_avg[k] = float("%g" % avg[k]); p.norm_std[k] = float("%g" % std[k])
        jobs = [sfa.Job(c, p, w, h, max(1, b1 - b0)) if b1 > b0 else None for c, (b0, b1) in zip(ctxs, groups)]
        for job, (b0, b1) in zip(jobs, groups):
            for b in range(b1 - b0):
                job.upload(b, base[(lo + b0 + b) % len(base)])
        run_all()                                     # warm-up (workspaces, first-touch)
    except Exception as e:
        err = "rank %d: %s" % (rank, str(e)[:160])
    if dist is not None:
        dist.barrier()
    if err is None:
        try:
            t0 = time.perf_counter()
            run_all()
            sec = time.perf_counter() - t0
        except Exception as e:
            err = "rank %d: %s" % (rank, str(e)[:160])
    if dist is not None:
        dist.barrier()
    sec_max = shard.max_over_ranks(dist, sec, device=xdev)
    failed_ranks = int(round(shard.sum_over_ranks(dist, 0.0 if err is None else 1.0, device=xdev)))
    windows_seen = int(round(shard.sum_over_ranks(dist, float(nloc), device=xdev)))
    per_rank = shard.gather_timings(dist, {rank: sec}, world, device=xdev)
    for job in jobs:
        if job is not None:
            job.close()
    if failed_ranks:
        return {"error": err or "%d other rank(s) failed" % failed_ranks, "failed_ranks": failed_ranks, "n_gpus": world, "workload": label}
    return {"workload": label, "windows_total": total_windows, "windows_this_rank": nloc, "windows_all_ranks": windows_seen, "groups_per_rank": S_, "n_gpus": world,
            "seconds": round(sec_max, 4), "seconds_per_rank": [round(float(v), 4) for v in per_rank],
            "windows_per_second": round(total_windows / sec_max, 2), "scaling": "strong",
            "schedule": "S=%d (%d frames), %d levels, %d alternations x %d outer x %d inner x %d sweeps, thresholds %g / %g, occlusion reasoning %d, penalties %d/%d/%d" % (
                p.S, 2 * p.S - 1, p.layers, p.niter_alter, p.niter_outer, p.niter_inner, p.niter_solver, p.thres_outer, p.thres_inner, p.occlusion_reasoning,
                p.robust_color.id, p.robust_grad.id, p.robust_reg.id)}


def config4_strong(ctxs, dev, world, rank, dist, xdev, total_windows=128):
    """BASELINE config 4 as north_star states it: a FIXED set of 128 frame windows (64 jets x 2 directions, slow_flow.cpp:706) under the cfg's
    schedule (cfgs/slow_flow.cfg: S = 3, 5 levels, 10 alternations x 10 outer x 30 sweeps, occlusion reasoning, break thresholds 1e-5):
    128 windows on one GPU, 16 per GPU on eight."""
    def make():
        p = sfa.default_params()
        p.S = 3; p.layers = LAYERS; p.hbit = 0
        p.rho[0] = 1; p.rho[1] = 1; p.omega[0] = 0; p.omega[1] = 2
        return p
    return strong_section(ctxs, world, rank, dist, xdev, total_windows, make, W, H, 5, "BASELINE configs[3]: cfgs/slow_flow.cfg schedule over 64 jets x 2 directions at 1024x436")


def config5_strong(ctxs, dev, world, rank, dist, xdev, total_windows=32):
    """BASELINE config 5: 2048 x 2048 synthetic sequence, 6 pyramid levels, Lorentzian penalties (data and smoothness), S = 2, fixed work
    (5 outer x 30 sweeps per level): a FIXED set of 32 windows over the ranks (32 on one GPU: what the arena of 1.63 GB per window allows in
    two groups of 16; 4 per GPU on eight)."""
    def make():
        p = bench_params()
        p.layers = 6
        p.robust_color.id = 2; p.robust_grad.id = 2; p.robust_reg.id = 2
        return p
    return strong_section(ctxs, world, rank, dist, xdev, total_windows, make, 2048, 2048, 3, "BASELINE configs[4]: 2048x2048, 6 levels, Lorentzian penalties, 5 outer x 30 sweeps")


def sor_launch(ctx, B, rank, reps=20):
    """one batch size of the metric's own kernel: (launches, ms, algorithmic bytes, kernel shape)"""
    from synth import sor_system
    sb = sfa.SorBatch(ctx, W, H, B)
    rng = np.random.default_rng(7 + rank)
    systems = [sor_system(rng, W, H) for _ in range(min(B, 4))]
    for b in range(B):
        s = systems[b % len(systems)]
        sb.upload(b, *[np.ascontiguousarray(s[k]) for k in ("du", "dv", "a11", "a12", "a22", "b1", "b2", "sh", "sv")])
    sb.run(SWEEPS, 1.9); ctx.sync()
    ctx.profile_enable(True)
    for _ in range(reps):
        sb.run(SWEEPS, 1.9)
    n, ms, by = ctx.profile_read()
    kernel = ctx.profile_read_kernels()[3]
    ctx.profile_enable(False)
    sb.close()
    return n, ms, by, kernel


def sor_only(ctx, B, rank):
    from synth import sor_system
    sb = sfa.SorBatch(ctx, W, H, B)
    rng = np.random.default_rng(7 + rank)
    for b in range(B):
        s = sor_system(rng, W, H)
        sb.upload(b, *[np.ascontiguousarray(s[k]) for k in ("du", "dv", "a11", "a12", "a22", "b1", "b2", "sh", "sv")])
    sb.run(SWEEPS, 1.9); ctx.sync()
    ctx.profile_enable(True)
    reps = 20
    for _ in range(reps):
        sb.run(SWEEPS, 1.9)
    n1, ms1, by1 = ctx.profile_read()
    ctx.profile_enable(False)
    # single-solve latency (batch of one)
    sb1 = sfa.SorBatch(ctx, W, H, 1)
    s = sor_system(rng, W, H)
    sb1.upload(0, *[np.ascontiguousarray(s[k]) for k in ("du", "dv", "a11", "a12", "a22", "b1", "b2", "sh", "sv")])
    sb1.run(SWEEPS, 1.9); ctx.sync()
    ctx.profile_enable(True)
    for _ in range(reps):
        sb1.run(SWEEPS, 1.9)
    n2, ms2, by2 = ctx.profile_read()
    ctx.profile_enable(False)

    sb.close(); sb1.close()
    return n1, ms1, by1, n2, ms2, by2


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_command(n, argv, port):
    """the command line that runs this script as n ranks on one node (the form the round driver itself uses)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(n, argv):
    """`--gpus N` without a torch.distributed environment: start the N ranks as a child process (never exec: nothing in this
    process has touched the GPU, and nothing will) and hand back its exit code; the child's rank 0 prints the JSON line."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")             # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(launch_command(n, argv, free_port()), env=env).returncode


def selftest_launch(args):
    """CPU rehearsal of the N>1 plumbing (tests/test_bench_launch.py): the ranks meet over gloo, run the same barrier / max-over-ranks /
    timing-gather sequence as the real bench around a sleep, and rank 0 prints a line that is labelled as a selftest -- no GPU work,
    no throughput claim."""
    import torch
    import torch.distributed as dist
    from slowflow_amd import shard
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group(backend="gloo")
    d = dist if world > 1 else None
    B = args.batch
    if d is not None:
        d.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * args.steps * (1 + rank))
    elapsed = time.perf_counter() - t0
    if d is not None:
        d.barrier()
    elapsed_max = shard.max_over_ranks(d, elapsed)
    lo, hi = shard.partition(B * world, world, rank)
    ws = shard.gather_timings(d, {i: elapsed / args.steps / B for i in range(lo, hi)}, B * world)
    if rank == 0:
        print(json.dumps({"metric": "selftest (launcher + torch.distributed plumbing on CPU, no GPU work)", "value": None, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 3),
                          "seconds_per_window": {"n": int(ws.size), "nonzero": int((ws > 0).sum())}}), flush=True)
    if d is not None:
        d.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=128, help="frame windows per GPU (fwd/bwd of several jets), refined as --streams lockstep groups")
    ap.add_argument("--streams", type=int, default=1, help="the batch is refined as this many lockstep groups on separate HIP streams, one host thread each "
                    "(the reference drives its windows from OpenMP threads, slow_flow.cpp:706).  Default since round 5: ONE group of 128 windows (a job holds up to 128 "
                    "since the window set became two mask words): the solver's launches then fill whole rounds of the 256 CUs at every pyramid level -- 115.8 ms per "
                    "step against 117.4-117.9 for two groups of 64 on two streams, same box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--path-only", action="store_true", help="skip the SOR-only section (used for the PMC passes: every SOR dispatch then belongs to the path)")
    ap.add_argument("--bench-only", action="store_true", help="--path-only and none of the reported extras either (one-window latency, labelled modes, cfg-schedule sample, triad): "
                                                              "every kernel launch of the process is a launch of the timed workload (the kernel-stats pass of profiles/collect.sh)")
    ap.add_argument("--selftest-launch", action="store_true", help="CPU rehearsal of the multi-rank launch and exchange (gloo); prints a selftest line, never a result")
    ap.add_argument("--no-strong", action="store_true", help="skip the config-4 strong-scaling section (128 windows in total under the cfg schedule)")
    args = ap.parse_args()
    if args.bench_only:
        args.path_only = True

    # N > 1 and nobody started the ranks: do it here, as a child process, before torch / HIP are touched
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.selftest_launch:
        return selftest_launch(args)
    if int(os.environ.get("WORLD_SIZE", "1")) != max(1, args.gpus):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}: start one rank per GPU")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    import torch
    # one rank per GPU over RCCL ("nccl" is RCCL on ROCm).  SFA_BENCH_BACKEND=gloo is the rehearsal of the same code on fewer GPUs than ranks (the
    # ranks then share the visible GPUs and exchange over CPU tensors): it exists to run the N>1 path end to end on a one-GPU box, never for a result
    backend = os.environ.get("SFA_BENCH_BACKEND", "nccl")
    ndev = max(1, torch.cuda.device_count())
    xdev = "cuda" if backend == "nccl" else "cpu"                 # where the tensors of the timing exchange live
    # a process group whenever this process was started as a rank (torch.distributed.run exports WORLD_SIZE), also as the only one: `torch.distributed.run
    # --nproc-per-node 1 bench.py --gpus 1` then walks the same RCCL calls as N = 8 -- init with a device id, barriers, the MAX / SUM all-reduces and the timing
    # gather on device tensors -- which is as much of that leg as a one-GPU box can run (tests/test_bench_launch.py::test_one_rank_over_rccl)
    if world > 1 or all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
        import torch.distributed as dist
        torch.cuda.set_device(local_rank % ndev)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    else:
        torch.cuda.set_device(0)

    import threading
    dev = (local_rank % ndev) if dist is not None else 0
    S = max(1, args.streams)
    B = args.batch
    if B % S:
        raise SystemExit("--batch must be a multiple of --streams")
    BL = B // S                                       # windows per launch (one lockstep group)
    ctxs = [sfa.Context(dev) for _ in range(S)]
    ctx = ctxs[0]
    p = bench_params()
    # one normalisation for the whole sequence, as the driver does (slow_flow.cpp:673)
    windows = [synth_window(1000 * rank + b) for b in range(B)]
    allf = [f for wdw in windows for f in wdw]
    # parity of the timed workload (rank 0, N = 1, with the CPU leg): the windows PARITY_WINDOWS of the timed job are refined by the oracle from the same
    # normalised frames and statistics after the bench, and normalize() itself is checked against a numpy evaluation of the reference's statistics on the raw frames
    want_parity = world == 1 and not args.no_cpu_baseline
    par_idx = [b for b in PARITY_WINDOWS if b < B] if want_parity else []
    raw = {b: [f.copy() for f in windows[b]] for b in par_idx}
    ref_stats = reference_statistics(allf, W) if want_parity else None
    avg, std = ctx.normalize(allf, W)
    for k in range(3):
        p.norm_avg[k] = float("%g" % avg[k]); p.norm_std[k] = float("%g" % std[k])     # 6-digit publish, variational_mt.cpp:71-84
    norm_check = None
    if want_parity:
        worst = 0
        for b in par_idx:
            for fr_raw, fr_gpu in zip(raw[b], windows[b]):
                for k in range(3):
                    want = ((fr_raw[k, :, :W].astype(np.float64) - ref_stats[0][k]) / ref_stats[1][k]).astype(np.float32)      # variational_mt.cpp:64-66
                    worst = max(worst, int(np.abs(want.view(np.int32).astype(np.int64) - fr_gpu[k, :, :W].view(np.int32).astype(np.int64)).max()))
        norm_check = {"statistics_equal_numpy_fp64": bool(all(avg[k] == ref_stats[0][k] and std[k] == ref_stats[1][k] for k in range(3))),
                      "frames_checked": 3 * len(par_idx), "max_ulp_normalised_frames": worst,
                      "note": "normalize() of all %d frames on the GPU against variational_mt.cpp:26-66 evaluated with numpy on the raw 8-bit frames (integer sums: exact in any order)" % len(allf)}
    del raw
    jobs = [sfa.Job(c, p, W, H, BL) for c in ctxs]
    for g, job in enumerate(jobs):
        for b in range(BL):
            job.upload(b, windows[g * BL + b])
    mpix_iters = sum(job.mpix_iters() for job in jobs)

    def barrier():
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def run_all(n):
        """every group runs n passes on its own stream; returns when all streams have drained"""
        def work(g):
            for _ in range(n):
                jobs[g].run()
            ctxs[g].sync()
        run_groups(work, S)

    run_all(args.warmup)
    barrier()
    for c in ctxs:
        c.profile_enable(True)
    t0 = time.perf_counter()
    run_all(args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n_sor, sor_ms, sor_bytes = 0, 0.0, 0.0
    n_asm, asm_ms, asm_px, sor_kernel = 0, 0.0, 0.0, ""
    for c in ctxs:
        n_, ms_, by_ = c.profile_read()
        n_sor += n_; sor_ms += ms_; sor_bytes += by_
        na_, ma_, pa_, sor_kernel = c.profile_read_kernels()
        n_asm += na_; asm_ms += ma_; asm_px += pa_
        c.profile_enable(False)
    # what the timed job computed: the flow fields of the parity windows, as the last timed step left them
    gpu_flows = {}
    for b in par_idx:
        gx, gy, _ = jobs[b // BL].download(b % BL)
        gpu_flows[b] = (gx[:, :W].copy(), gy[:, :W].copy())
    from slowflow_amd import shard
    if dist is not None:
        dist.barrier()
    elapsed_max = shard.max_over_ranks(dist, elapsed, device=xdev)
    # the per-window timings of every rank: the only exchange of the path (a few hundred bytes over RCCL)
    lo, hi = shard.partition(B * world, world, rank)
    window_seconds = shard.gather_timings(dist, {i: elapsed / args.steps / B for i in range(lo, hi)}, B * world, device=xdev)

    # SOR-only: the metric's own kernel at 1024x436, same batch, HIP events on the launch stream
    n1 = ms1 = by1 = n2 = ms2 = by2 = 0
    sor16 = None
    if not args.path_only:
        n1, ms1, by1, n2, ms2, by2 = sor_only(ctx, BL, rank)
        sor16 = sor_launch(ctx, 16, rank)            # what one GPU of an 8-GPU node gets of config 4's 128 windows
    # the strong-scaling sections: every rank walks the same collectives whether its own share worked or not (strong_section), so nothing here may raise
    # between them; the bench's own jobs are closed first -- config 5's 32 windows need the memory
    strong = strong5 = None
    if not args.no_strong and not args.path_only:
        # config 4 runs under break thresholds: two lockstep groups carry fewer passengers than one (1.56 against 1.60 s for the 128 windows on one GPU), which is
        # what the driver does too (gpu_streams 2) -- a second context for this section when the bench itself runs one group
        ctxs4 = ctxs
        if len(ctxs) < 2:
            try:
                ctxs4 = ctxs + [sfa.Context(dev)]
            except Exception:                                      # (nothing may raise between the sections' collectives: one group then)
                ctxs4 = ctxs
        strong = config4_strong(ctxs4, dev, world, rank, dist, xdev)
        for extra in ctxs4[len(ctxs):]:
            extra.close()
        strong5 = config5_strong(ctxs, dev, world, rank, dist, xdev)
    if rank == 0:
        total = mpix_iters * args.steps * world
        value = total / elapsed_max
        achieved = sor_bytes / (sor_ms * 1e-3) / 1e9 if sor_ms > 0 else 0.0
        traffic, traffic_src, valu_busy = measured_traffic(BL, sor_kernel)
        avg_launch_s = sor_ms / max(n_sor, 1) * 1e-3
        # what one launch must move at least: every operand entry read once (SA 16 B + SB 16 B + x 8 B), x written once (8 B)
        compulsory = 48.0 * (sor_bytes / (44.0 * SWEEPS + 12.0)) / max(n_sor, 1)
        out = {
            "metric": "Mpix*solver-iters/s at 1024x436 (whole coarse-to-fine path)", "value": round(value, 1), "unit": "Mpix*solver-iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 1024x436, S=2 (3 frames), 5 pyramid levels, 5 outer x 1 inner x 30 SOR sweeps, "
                                   "symmetric window, modified-L1 penalties, thresholds off",
                       "frame_windows_per_gpu": B, "streams": S, "windows_per_launch": BL, "mpix_iters_per_step_per_gpu": round(mpix_iters, 3), "sor_order": "lexicographic (reference-identical)",
                       "parallelism": f"frame-window data parallel x{world}" + ("" if backend == "nccl" or world == 1 else f" (REHEARSAL over {backend}: ranks share {ndev} GPU(s))")},
            # The solver kernel against the HBM roof.  `achieved` = the bytes the fused solve has to move (every operand entry read once, the iterate read and
            # written once: 48 B per pixel and solve -- DESIGN.md 5.1) / the live launch duration; `frac` = that / peak, a true fraction.  SURVEY 8(d)'s byte model
            # charges 44 K + 12 bytes per pixel because it re-reads the operands in every sweep; a kernel that keeps 15 sweeps of a band in LDS moves far fewer, so
            # that figure exceeds the peak and is carried as `algorithmic_gbs_8d` / `algorithmic_8d_over_peak`, NOT as a fraction of anything.  The kernel is not
            # bandwidth bound at all: `limiter` says what it waits for, `valu_issue_floor_frac` how close it is to that roof.
            # `bound` names the roof the contract prices the kernel against (HBM bandwidth, MFMA not applicable); what the kernel actually waits for is
            # `limited_by` / `limiter` (the two are different statements and are labelled as such)
            "roofline": {"bound": "hbm", "limited_by": "valu_issue" if BL >= 16 else "dependency_latency", "kernel": sor_kernel + " -- the shape the library picked for %d windows per launch" % BL,
                         "achieved": round(compulsory / avg_launch_s / 1e9, 1) if avg_launch_s > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(compulsory / avg_launch_s / 1e9 / HBM_PEAK_GBS, 4) if avg_launch_s > 0 else None, "traffic": traffic,
                         "launches": n_sor, "avg_launch_ms": round(sor_ms / max(n_sor, 1), 4),
                         "algorithmic_bytes_per_launch": round(compulsory),
                         "algorithmic_bytes_note": "48 B per pixel and solve (2 x 16 B operand entries + 8 B iterate in + 8 B out), mean over the launches of all 5 levels",
                         "traffic_source": traffic_src,
                         "hbm_physical": (round(traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS, 4) if traffic and avg_launch_s > 0 else None),
                         "traffic_over_compulsory": (round(traffic / compulsory, 2) if traffic and compulsory > 0 else None),
                         "algorithmic_gbs_8d": round(achieved, 1), "algorithmic_8d_over_peak": round(achieved / HBM_PEAK_GBS, 3),
                         "algorithmic_8d_note": "SURVEY 8(d): (44 K + 12) B per pixel and solve, every sweep re-reading its operands; above the peak because the kernel fuses the sweeps",
                         "limiter": limiter(sor_kernel, BL, S),
                         "valu_active_frac": valu_busy,
                         "steps_critical_level0": W + H + 2 * SWEEPS,
                         "alu_latency_floor_us_per_step": round(150.0 / CLOCK_GHZ / 1e3, 4),
                         "note": "over the timed region (launches of all streams; with streams > 1 a launch shares the GPU with the other groups' kernels, so its duration is longer than alone): sums over every SOR launch (all 5 levels) / sum of HIP-event durations",
                         },
            "sor_share_of_step": round(sor_ms / S / (elapsed * 1e3), 4),
            "seconds_per_window": {"mean": round(float(window_seconds.mean()), 6), "max": round(float(window_seconds.max()), 6), "n": int(window_seconds.size)},
            # how the ranks met (None: a lone process without torch.distributed)
            "distributed": ({"backend": backend, "world_size": world, "exchange_tensors_on": xdev, "collectives": "barrier, all_reduce(MAX), all_reduce(SUM)"} if dist is not None else None),
        }
        # the roof that BINDS: the VALU time of the busiest SIMD.  `bound` names it for the batch that ran; the HBM figures above stay as they are
        vf = solver_valu_floor(sor_kernel, BL) if n_sor else None
        if vf and avg_launch_s > 0:
            out["roofline"]["valu_time_floor_frac"] = round(vf[0] / avg_launch_s, 4)
            out["roofline"]["valu_time_floor_frac_mean_simd"] = round(vf[1] / avg_launch_s, 4)
            out["roofline"]["valu_time_floor_ms"] = round(vf[0] * 1e3, 4)
            out["roofline"]["valu_cycles_per_step_by_simd"] = vf[2]
            out["roofline"]["valu_time_floor_note"] = ("mean over the 5 levels of: workgroups / 256 CUs x VALU cycles of the busiest SIMD of a workgroup (its waves' ISA mix x the measured pass costs: "
                                                        "v_pk_*_f32 and DPP moves 4.4 cycles, plain VALU 2.3; %s, tools/valu_time_model.py) / %.1f GHz, over the live launch duration -- 1.0 = that "
                                                        "SIMD never waits; `_mean_simd`: the same with the mean of the four SIMDs" % (vf[3], CLOCK_GHZ))
            if BL >= 16:
                out["roofline"]["bound"] = "valu"
                out["roofline"]["bound_note"] = ("the roof that binds at %d windows per launch is the VALU time of the busiest SIMD (`valu_time_floor_frac`); achieved / peak / frac / traffic are the "
                                                 "HBM figures the contract asks for (bytes the fused solve has to move / duration / 8 TB/s) and are not what limits the kernel" % BL)
        vpw, vpw_src = sor_valu_per_wave(sor_kernel)
        if vpw and n_sor:
            # VALU issue floor of the solver: instructions per wave (SQ counters) x waves of the mean launch x 2 cycles per wave64 instruction on a SIMD-32
            # (MI355X_MICROARCH.md; a wave alone on its SIMD -- this kernel's regime, one 7-wave workgroup per CU -- issues one per 4-5) / 1024 SIMDs
            waves = sor_launch_waves(sor_kernel, BL) or 0.0
            floor_s = vpw * waves * 2.0 / (1024.0 * CLOCK_GHZ * 1e9)
            out["roofline"]["valu_issue_floor_frac"] = round(floor_s / avg_launch_s, 4)
            out["roofline"]["valu_issue_floor_note"] = "%.0f VALU instructions per wave (%s) x %.0f waves per launch x 2 cycles / (1024 SIMDs x %.1f GHz) / avg_launch" % (vpw, vpw_src, waves, CLOCK_GHZ)

        def sor_entry(batch, n, ms, by, kernel=None):
            steps = W + H + 2 * SWEEPS
            e = {"batch": batch, "avg_launch_ms": round(ms / n, 4), "algorithmic_gbs_8d": round(by / (ms * 1e-3) / 1e9, 1),
                 "hbm_frac_compulsory": round(48.0 * W * H * batch * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 "mpix_iters_per_s": round(W * H * SWEEPS * batch * n / 1e6 / (ms * 1e-3), 1),
                 "us_per_solve": round(ms / n * 1e3 / batch, 1),
                 # the launch against the length of the dependency chain: microseconds per hyperplane step, and how many times the ALU-latency floor that is
                 "us_per_critical_step": round(ms / n * 1e3 / steps, 4),
                 "x_alu_latency_floor": round(ms / n * 1e3 / steps / (150.0 / CLOCK_GHZ / 1e3), 2)}
            if kernel:
                e["kernel"] = kernel
            return e
        if n1:
            out["roofline"]["sor_1024x436_batch"] = sor_entry(BL, n1, ms1, by1)
            out["roofline"]["sor_1024x436_single"] = sor_entry(1, n2, ms2, by2)
        if sor16:
            out["roofline"]["sor_1024x436_batch16"] = sor_entry(16, sor16[0], sor16[1], sor16[2], sor16[3])
        # north_star's own budget: ">= 60 % of the HBM roofline on the SOR inner loop at 1024x436" under SURVEY 8(d)'s byte model (1332 B per pixel and 30-sweep solve)
        # is <= 124 us per solve; per batch size, is it met?
        budget = {"north_star_budget_us_per_solve": 124.0, "byte_model": "SURVEY 8(d): (44 K + 12) B per pixel = 594.7 MB per 1024x436x30 solve; 60 % of 8 TB/s"}
        for key, name in (("sor_1024x436_single", "batch_1"), ("sor_1024x436_batch16", "batch_16"), ("sor_1024x436_batch", "batch_%d" % BL)):
            e = out["roofline"].get(key)
            if e:
                budget[name] = {"us_per_solve": e["us_per_solve"], "meets": bool(e["us_per_solve"] <= 124.0)}
        out["roofline"]["north_star_budget"] = budget
        # second kernel of the step: the data-term assembly (about 40 % of it).  VALU bound: wave instructions issued per second against the chip's
        # issue rate (256 CUs x 4 SIMDs, one wave64 instruction per 2 cycles).  Instructions per pixel and term from the SQ counter pass of the same
        # kernel (profiles/), duration live from HIP events around every launch of the timed region.
        if n_asm:
            ipt, ipt_src, valu_active = assemble_valu_per_pixel_term()
            peak_issue = 256 * 4 * CLOCK_GHZ * 1e9 / 2.0         # a SIMD-32 issues a wave64 VALU instruction over 2 cycles (MI355X_MICROARCH.md): 1229 G/s
            ach = (ipt * asm_px / 64.0) / (asm_ms * 1e-3) if ipt else None
            out["roofline_assemble"] = {"kernel": "k_assemble_images<8,512,6>", "bound": "valu", "launches": n_asm, "avg_launch_ms": round(asm_ms / n_asm, 4),
                                        "pixel_terms_per_launch": round(asm_px / n_asm), "valu_instructions_per_pixel_term": ipt, "source": ipt_src,
                                        "achieved": (round(ach / 1e9, 2) if ach else None), "peak": round(peak_issue / 1e9, 2), "unit": "G wave-instructions/s",
                                        "frac": (round(ach / peak_issue, 4) if ach else None),
                                        # the SIMDs' own view from the same counter pass: SQ_ACTIVE_INST_VALU x 4 / SIMD cycles when the kernel runs alone.  The counter
                                        # charges a quad-cycle (4 cycles) per wave instruction where the SIMD-32 needs 2, so it reads about twice the issue-slot use
                                        # (six waves per SIMD: the raw ratio passes 1 -- carried as a ratio, and halved as the issue-slot use at 2 cycles per instruction)
                                        "valu_time_floor_frac": None,
                                        "valu_quadcycle_ratio_alone": valu_active,
                                        "valu_issue_slot_frac_alone": (round(valu_active / 2.0, 4) if valu_active else None),
                                        "share_of_step": round(asm_ms / S / (elapsed * 1e3), 4),
                                        "note": "launch durations are those inside the timed region, where the other stream's kernels share the GPU (alone: about half)"}
            vm, vm_src = valu_model()
            if vm and ach:
                inst = vm["assemble"].get("k_assemble_images<8,512,6,true,1,true>")      # the instance the bench's first (and only) inner iteration runs: du = dv = 0, cfg defaults folded in
                if inst:
                    cpi = inst["cycles_per_valu_instruction_term_loop"]
                    out["roofline_assemble"]["valu_time_floor_frac"] = round(ach * cpi / (256 * 4 * CLOCK_GHZ * 1e9), 4)
                    out["roofline_assemble"]["valu_time_floor_note"] = ("dynamic VALU instructions (SQ counters) x %.3f cycles per instruction of the term loop's static mix (%s: plain 2.3, "
                                                                        "v_rcp / v_sqrt 4.6; no packed operations in this kernel) / (1024 SIMDs x %.1f GHz) over the live launch durations; "
                                                                        "`frac` prices an instruction at the guide's 2 cycles" % (cpi, vm_src, CLOCK_GHZ))
        if strong is not None:
            out["config4_strong"] = strong
        if strong5 is not None:
            out["config5_strong"] = strong5
        try:
            if args.bench_only:
                raise RuntimeError("--bench-only")
            lat, sor1, rb = one_window_latency(ctx)
            out["latency_one_window_ms"] = round(lat, 3)
            out["latency_one_window_sor_ms_per_solve"] = round(sor1, 4)
            out["labelled_modes"] = {"red_black": rb}
        except Exception as e:                                    # a reported extra
            out["latency_one_window_ms"] = None
        if not args.path_only:
            try:
                out["cfg_schedule_with_thresholds"] = cfg_schedule_sample(ctx)
            except Exception as e:                                # a reported extra
                out["cfg_schedule_with_thresholds"] = None
        try:
            if args.bench_only:
                raise RuntimeError("--bench-only")
            out["roofline"]["measured_triad_gbs"] = round(hbm_triad_gbs(torch), 1)
        except Exception as e:                                    # a measurement aid only
            out["roofline"]["measured_triad_gbs"] = None
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(windows=[(b, windows[b]) for b in par_idx], stats=([p.norm_avg[k] for k in range(3)], [p.norm_std[k] for k in range(3)]))
            cpu_flows = cb.pop("_flows")
            out["cpu_baseline"] = cb
            d = max(max(float(np.abs(cpu_flows[b][0] - gpu_flows[b][0]).max()), float(np.abs(cpu_flows[b][1] - gpu_flows[b][1]).max())) for b in par_idx)
            finite = all(np.isfinite(gpu_flows[b][0]).all() and np.isfinite(gpu_flows[b][1]).all() for b in par_idx)
            out["parity"] = {"windows": list(par_idx), "max_abs_uv": float("%.3g" % d), "tol": 1e-4, "ok": bool(finite and d <= 1e-4),
                             "of": "the flow fields the LAST TIMED STEP left in the timed job (windows of both mask words of the %d-window launch group), downloaded after the timed region" % BL,
                             "against": "oracle.variational (the CPU restatement, pinned to the compiled reference: DESIGN.md 3) on the same normalised frames and 6-digit statistics -- "
                                        "the runs cpu_baseline times",
                             "mean_flow_px": [round(float(np.mean([gpu_flows[b][0].mean() for b in par_idx])), 4), round(float(np.mean([gpu_flows[b][1].mean() for b in par_idx])), 4)],
                             "normalisation": norm_check}
        else:
            out["parity"] = None                                  # (N > 1 or --no-cpu-baseline: no CPU leg in this run; tests/test_gpu_parity.py::test_bench_job_128_windows_against_the_oracle)
        print(json.dumps(out), flush=True)
    for job in jobs:
        job.close()
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.barrier()                                            # nobody tears the group down while rank 0 is still measuring its extras
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
