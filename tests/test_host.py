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


def png_bytes(w, h, ctype, depth, rows, palette=None, filt=None):
    """a PNG written here, independently of png.cpp: `rows` = packed scanline bytes; filt = per-row filter type (None: cycle 0..4)"""
    import zlib
    comps = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, comps * depth // 8)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)

    def paeth(a, b, c):
        p = a + b - c
        pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
        return a if pa <= pb and pa <= pc else (b if pb <= pc else c)
    raw = bytearray()
    prev = bytes(len(rows[0]))
    for y, cur in enumerate(rows):
        ft = (y % 5) if filt is None else filt
        raw.append(ft)
        for i, v in enumerate(cur):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = [0, a, b, (a + b) >> 1, paeth(a, b, c)][ft]
            raw.append((v - pred) & 255)
        prev = cur
    z = zlib.compress(bytes(raw), 9)
    idats = b"".join(chunk(b"IDAT", z[i:i + 97]) for i in range(0, len(z), 97))      # split over many IDAT chunks
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + chunk(b"tEXt", b"Comment\0x")
            + (chunk(b"PLTE", palette) if palette is not None else b"") + idats + chunk(b"IEND", b""))


def write_png_cases(tmp_path):
    """PNG files of every kind the reader claims (png.h) + the samples it must return, for tests/host/test_host.cpp"""
    rng = np.random.default_rng(5)
    w, h = 19, 11
    lines = []

    def emit(name, ctype, depth, rows, want, ch, out_depth, palette=None, filt=None):
        (tmp_path / (name + ".png")).write_bytes(png_bytes(w, h, ctype, depth, rows, palette, filt))
        np.ascontiguousarray(want, dtype="<u2").tofile(str(tmp_path / (name + ".raw")))
        lines.append("%s %d %d %d %d" % (name, w, h, ch, out_depth))
    for depth in (8, 16):
        hi = 256 if depth == 8 else 65536
        dt = ">u1" if depth == 8 else ">u2"
        for ctype, comps in ((0, 1), (2, 3), (4, 2), (6, 4)):
            px = rng.integers(0, hi, size=(h, w, comps))
            px[0, :3] = hi - 1
            smooth = (np.arange(w)[None, :, None] * 3 + np.arange(h)[:, None, None] * 5 + np.arange(comps)[None, None, :]) % hi
            for tag, data in (("n", px), ("s", smooth)):
                rows = [data[y].astype(dt).tobytes() for y in range(h)]
                keep = data[..., :1] if comps <= 2 else data[..., :3]                 # alpha dropped
                emit("c%d_d%d_%s" % (ctype, depth, tag), ctype, depth, rows, keep, keep.shape[2], depth)
    for f in range(5):                                                                # each filter on its own for every row
        px = rng.integers(0, 256, size=(h, w, 3))
        emit("filter%d" % f, 2, 8, [px[y].astype(np.uint8).tobytes() for y in range(h)], px, 3, 8, filt=f)
    pal = rng.integers(0, 256, size=(16, 3)).astype(np.uint8)
    for depth in (1, 2, 4, 8):
        n = min(16, 1 << depth)
        idx = rng.integers(0, n, size=(h, w))
        per = 8 // depth
        rows = []
        for y in range(h):
            line = bytearray((w + per - 1) // per)
            for x in range(w):
                line[x // per] |= int(idx[y, x]) << ((per - 1 - x % per) * depth)
            rows.append(bytes(line))
        emit("pal_d%d" % depth, 3, depth, rows, pal[idx], 3, 8, palette=pal[:n].tobytes())
        if depth < 8:
            emit("grey_d%d" % depth, 0, depth, rows, (idx * 255 // ((1 << depth) - 1))[..., None], 1, 8)
    (tmp_path / "png_cases.txt").write_text("\n".join(lines) + "\n")


def test_host_mirror_cpu(host_build, tmp_path):
    write_png_cases(tmp_path)
    exe = str(tmp_path / "test_host")
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-pthread", "-I", HOST, os.path.join(ROOT, "tests", "host", "test_host.cpp"),
                        os.path.join(HOST, "libslowflow_host.a"), "-L", os.path.join(ROOT, "slowflow_amd"), "-lslowflow_amd", "-lz",
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
    ext = "png" if occ else "ppm"                                 # the second case reads PNG frames (filtered scanlines) and has ground truth
    for k, f in enumerate(frames):
        if occ:
            rows = [f[:, y, :].T.astype(np.uint8).tobytes() for y in range(h)]
            (tmp_path / ("f_%03d.png" % (10 - steps + k))).write_bytes(png_bytes(w, h, 2, 8, rows))
        else:
            write_ppm(str(tmp_path / ("f_%03d.ppm" % (10 - steps + k))), f)
    gt_line = ""
    if occ:
        with open(str(tmp_path / "gt_010.flo"), "wb") as g:       # ground truth for jet 0 only: the synthetic translation
            g.write(struct.pack("<fii", 202021.25, w, h))
            g.write(np.tile(np.array([1.5 * steps, -0.75 * steps], np.float32), w * h).tobytes())
        gt_line = "file_gt\t%s/gt_%%03i.flo\n" % tmp_path
    cfg = tmp_path / "run.cfg"
    cfg.write_text(gt_line +
        "file\t%s/f_%%03i.%s\noutput\t%s/out\nJets\t%d\nstart\t10\nmax_fps\t200\n16bit\t0\nraw\t0\nscale\t1.0\ndeep_matching\t0\n"
        "slow_flow_S\t%d\nslow_flow_layers\t2\nslow_flow_niter_alter\t%d\nslow_flow_niter_outer\t3\nslow_flow_occlusion_reasoning\t%d\n"
        "slow_flow_thres_outer\t0\nslow_flow_thres_inner\t0\nslow_flow_rho_0\t1\nslow_flow_omega_0\t0\ngpus\t1\ngpu_batch\t4\n" % (tmp_path, ext, tmp_path, jets, S, alter, occ))
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
    for j in range(jets):                                         # the colour-coded forward flows: a constant translation = one colour
        with open(str(out / ("frame_%d.png" % (10 + j * steps))), "rb") as f:
            assert f.read(8) == b"\x89PNG\r\n\x1a\n"
    if occ:
        import json
        tj = json.load(open(str(out / "timings.json")))
        fwd0 = [t for t in tj if t["jet"] == 0 and t["direction"] == "forward"][0]
        uu, vv = read_flo(str(out / "f_010.flo"))
        assert abs(fwd0["epe"] - np.mean(np.hypot(uu - 1.5 * steps, vv + 0.75 * steps))) < 1e-4 and 0 <= fwd0["aae"] < 0.2
        assert all("epe" not in t for t in tj if t["jet"] != 0 or t["direction"] != "forward")
        assert (out / "gt" / "flow_00010.png").exists()
        gu, gv = read_flo(str(out / "gt" / "flow_00010.flo"))
        assert np.all(gu == 1.5 * steps) and np.all(gv == -0.75 * steps)
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
