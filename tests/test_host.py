"""Host side of the drop-in (C++): builds slowflow_amd/host and runs its CPU checks; on a GPU box also drives the
slow_flow binary end to end over a synthetic PPM sequence and compares the .flo files with the Python binding."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "slowflow_amd", "host")


@pytest.fixture(scope="module")
def host_build():
    import slowflow_amd as sfa
    if not os.path.exists(sfa.LIB_PATH):
        sfa.build()
    r = subprocess.run(["make", "-C", HOST], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return HOST


def test_host_mirror_cpu(host_build, tmp_path):
    exe = str(tmp_path / "test_host")
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-pthread", "-I", HOST, os.path.join(ROOT, "tests", "host", "test_host.cpp"),
                        os.path.join(HOST, "libslowflow_host.a"), "-L", os.path.join(ROOT, "slowflow_amd"), "-lslowflow_amd",
                        "-Wl,-rpath," + os.path.join(ROOT, "slowflow_amd"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and "host tests OK" in r.stdout, r.stdout + r.stderr


def test_driver_rejects_out_of_scope_inputs(host_build, tmp_path):
    cfg = tmp_path / "a.cfg"
    cfg.write_text("file\t%s/f_%%03i.ppm\noutput\t%s/out\nJets\t1\nstart\t1\nraw\t0\ndeep_matching\t1\n" % (tmp_path, tmp_path))
    r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg)], capture_output=True, text=True)
    assert r.returncode == 2 and "deep_matching" in r.stderr
    r = subprocess.run([os.path.join(HOST, "slow_flow"), str(tmp_path / "missing.cfg")], capture_output=True, text=True)
    assert r.returncode != 0 and "Couldn't find" in r.stderr


def write_ppm(path, img):          # img: (3,h,w) float 0..255
    h, w = img.shape[1:]
    data = np.clip(np.round(img), 0, 255).astype(np.uint8).transpose(1, 2, 0).tobytes()
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(data)


def read_flo(path):
    with open(path, "rb") as f:
        tag, w, h = struct.unpack("<fii", f.read(12))
        assert tag == 202021.25
        d = np.frombuffer(f.read(), dtype=np.float32).reshape(h, w, 2)
    return d[..., 0], d[..., 1]


@pytest.mark.gpu
@pytest.mark.parametrize("alter,occ", [(1, 0), (3, 1)])
def test_slow_flow_driver_end_to_end(host_build, tmp_path, alter, occ):
    """cfg + PPM frames -> ./slow_flow -> .flo, against the same windows refined through the Python binding; the second
    case alternates with the discrete occlusion step, as cfgs/slow_flow.cfg does by default"""
    import slowflow_amd as sfa
    from synth import texture_frame
    w, h, jets, S = 96, 64, 3, 2
    steps = S - 1
    nframes = 1 + (jets + 2) * steps
    frames = [np.clip(np.round(texture_frame(w, h, k)[:, :, :w]), 0, 255) for k in range(nframes)]
    for k, f in enumerate(frames):
        write_ppm(str(tmp_path / ("f_%03d.ppm" % (10 - steps + k))), f)
    cfg = tmp_path / "run.cfg"
    cfg.write_text(
        "file\t%s/f_%%03i.ppm\noutput\t%s/out\nJets\t%d\nstart\t10\nmax_fps\t200\n16bit\t0\nraw\t0\nscale\t1.0\ndeep_matching\t0\n"
        "slow_flow_S\t%d\nslow_flow_layers\t2\nslow_flow_niter_alter\t%d\nslow_flow_niter_outer\t3\nslow_flow_occlusion_reasoning\t%d\n"
        "slow_flow_thres_outer\t0\nslow_flow_thres_inner\t0\nslow_flow_rho_0\t1\nslow_flow_omega_0\t0\ngpus\t1\ngpu_batch\t4\n" % (tmp_path, tmp_path, jets, S, alter, occ))
    r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg), "-overwrite"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Done!" in r.stdout
    out = tmp_path / "out"
    assert (out / "config.cfg").exists() and (out / "timings.json").exists()
    # the same computation through the Python binding
    ctx = sfa.Context(0)
    stride = sfa.stride_of(w)
    fr = []
    for f in frames:
        a = np.zeros((3, h, stride), np.float32)
        a[:, :, :w] = f
        fr.append(a)
    avg, std = ctx.normalize(fr, w)
    p = sfa.default_params()
    p.S = S; p.layers = 2; p.niter_alter = alter; p.niter_outer = 3; p.occlusion_reasoning = occ; p.thres_outer = 0; p.thres_inner = 0
    p.hbit = 0; p.smoothing = 1; p.rho[0] = 1; p.omega[0] = 0
    for k in range(3):
        p.norm_avg[k] = float("%g" % avg[k]); p.norm_std[k] = float("%g" % std[k])
    for j in range(jets):
        f0 = j * steps
        for back in (False, True):
            win = fr[f0:f0 + 2 * steps + 1] if not back else [fr[nframes - 1 - i] for i in range(nframes - 1 - f0 - 3 * steps, nframes - 1 - f0 - 3 * steps + 2 * steps + 1)]
            wx, wy = np.zeros((h, stride), np.float32), np.zeros((h, stride), np.float32)
            ctx.variational(p, wx, wy, win, w)
            name = "f_%03d%s.flo" % ((10 + f0) if not back else (10 + f0 + steps), "_back" if back else "")
            u, v = read_flo(str(out / name))
            assert np.array_equal(u, wx[:, :w] * steps) and np.array_equal(v, wy[:, :w] * steps), name
    # forward flows recover the synthetic translation (1.5, -0.75) px per frame
    u, v = read_flo(str(out / "f_010.flo"))
    assert abs(np.median(u) - 1.5) < 0.1 and abs(np.median(v) + 0.75) < 0.1
    if occ:                                                       # the occlusion labels of every forward window were written
        for j in range(jets):
            with open(str(out / "occlusion" / ("f_%03d_occ.pgm" % (10 + j * steps))), "rb") as f:
                assert f.readline() == b"P5\n" and f.readline().split() == [b"%d" % w, b"%d" % h]
    # -resume skips what exists
    r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg), "-resume"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.count("already exist") == 2 * jets
    ctx.close()


def write_pgm(path, img):           # img: (h,w) float 0..255
    h, w = img.shape
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (w, h))
        f.write(np.clip(np.round(img), 0, 255).astype(np.uint8).tobytes())


@pytest.mark.gpu
def test_driver_ingest_scale_and_raw(host_build, tmp_path):
    """rank-2 ingest through the driver: (a) scale 0.5 = GaussianBlur(1/sqrt(2*scale)) + resize(fx) on load, against the same
    operators applied through the binding; (b) raw 1 with the reference's own bilinear demosaicer and raw weighting runs and
    recovers the motion of a colour-constant textured mosaic"""
    import slowflow_amd as sfa
    from synth import texture_frame
    w, h, jets, S = 128, 96, 1, 2
    steps = S - 1
    nframes = 1 + (jets + 2) * steps
    frames = [np.clip(np.round(texture_frame(w, h, k)[:, :, :w]), 0, 255) for k in range(nframes)]
    for k, f in enumerate(frames):
        write_ppm(str(tmp_path / ("f_%03d.ppm" % (10 - steps + k))), f)
    common = ("Jets\t%d\nstart\t10\nmax_fps\t200\n16bit\t0\ndeep_matching\t0\nslow_flow_S\t%d\nslow_flow_layers\t2\nslow_flow_niter_alter\t1\n"
              "slow_flow_niter_outer\t3\nslow_flow_occlusion_reasoning\t0\nslow_flow_thres_outer\t0\nslow_flow_thres_inner\t0\nslow_flow_rho_0\t1\n"
              "slow_flow_omega_0\t0\ngpus\t1\ngpu_batch\t4\n" % (jets, S))
    # (a) scale
    cfg = tmp_path / "scale.cfg"
    cfg.write_text("file\t%s/f_%%03i.ppm\noutput\t%s/out_scale\nraw\t0\nscale\t0.5\n" % (tmp_path, tmp_path) + common)
    r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg), "-overwrite"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    ctx = sfa.Context(0)
    sw, sh = w // 2, h // 2
    stride = sfa.stride_of(sw)
    sigma = np.float32(1 / np.sqrt(2 * 0.5))
    fr = []
    for f in frames:
        a = np.zeros((3, sh, stride), np.float32)
        for c in range(3):
            src = np.zeros((h, sfa.stride_of(w)), np.float32)
            src[:, :w] = f[c]
            small, dw = ctx.resize_linear_fx(ctx.gaussian_blur(src, w, float(sigma)), w, 0.5, 0.5)
            assert dw == sw and small.shape == (sh, stride)
            a[c] = small
        fr.append(a)
    avg, std = ctx.normalize(fr, sw)
    p = sfa.default_params()
    p.S = S; p.layers = 2; p.niter_alter = 1; p.niter_outer = 3; p.occlusion_reasoning = 0; p.thres_outer = 0; p.thres_inner = 0
    p.hbit = 0; p.smoothing = 1; p.rho[0] = 1; p.omega[0] = 0
    for k in range(3):
        p.norm_avg[k] = float("%g" % avg[k]); p.norm_std[k] = float("%g" % std[k])
    wx, wy = np.zeros((sh, stride), np.float32), np.zeros((sh, stride), np.float32)
    ctx.variational(p, wx, wy, fr[0:3], sw)
    u, v = read_flo(str(tmp_path / "out_scale" / "f_010.flo"))
    assert u.shape == (sh, sw)
    assert np.array_equal(u, wx[:, :sw] * steps) and np.array_equal(v, wy[:, :sw] * steps)
    assert abs(np.median(u) - 0.75) < 0.1 and abs(np.median(v) + 0.375) < 0.1          # half the resolution, half the motion
    ctx.close()
    # (b) raw: grey mosaics of the same frames (all three channels of texture_frame are sampled by the pattern)
    for k, f in enumerate(frames):
        yy, xx = np.mgrid[0:h, 0:w]
        red = (xx % 2 == 1) & (yy % 2 == 0)
        blue = (xx % 2 == 0) & (yy % 2 == 1)
        write_pgm(str(tmp_path / ("m_%03d.pgm" % (10 - steps + k))), np.where(red, f[0], np.where(blue, f[2], f[1])))
    cfg = tmp_path / "raw.cfg"
    cfg.write_text("file\t%s/m_%%03i.pgm\noutput\t%s/out_raw\nraw\t1\nraw_demosaicing\t0\nraw_red_loc\t1,0\nraw_weight\t2\nscale\t1.0\n" % (tmp_path, tmp_path) + common)
    r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg), "-overwrite"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    u, v = read_flo(str(tmp_path / "out_raw" / "m_010.flo"))
    assert abs(np.median(u) - 1.5) < 0.15 and abs(np.median(v) + 0.75) < 0.15
    cfg.write_text("file\t%s/m_%%03i.pgm\noutput\t%s/out_raw2\nraw\t1\nraw_demosaicing\t1\n" % (tmp_path, tmp_path) + common)
    r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg), "-overwrite"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "raw_demosaicing" in r.stderr
