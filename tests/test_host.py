"""Host side of the drop-in (C++): builds slowflow_amd/host and runs its CPU checks; on a GPU box also drives the
slow_flow binary end to end over a synthetic PPM sequence and compares the .flo files with the Python binding."""
import json
import os
import struct
import subprocess
import sys

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


def bayer_numpy(src, rx, ry):
    """utils.cpp:1241-1334 evaluated with numpy: green by the mean of the 4 neighbours at red / blue sites (mirrored borders), red and blue
    through the local green ratio; float operands, double products where the C expression promotes (0.25 * float, float * 0.5 * float)"""
    h, w = src.shape
    xm1 = np.where(np.arange(w) > 0, np.arange(w) - 1, np.arange(w) + 1); xp1 = np.where(np.arange(w) < w - 1, np.arange(w) + 1, np.arange(w) - 1)
    ym1 = np.where(np.arange(h) > 0, np.arange(h) - 1, np.arange(h) + 1); yp1 = np.where(np.arange(h) < h - 1, np.arange(h) + 1, np.arange(h) - 1)
    Y, X = np.mgrid[0:h, 0:w]
    blue_row = (Y + (1 - ry)) % 2 == 0
    green = np.where(blue_row, (X + rx) % 2 == 0, (X + (1 - rx)) % 2 == 0)
    f32 = np.float32
    nb4 = ((src[ym1][:, :] + src[yp1][:, :]).astype(f32) + src[:, xm1]).astype(f32) + src[:, xp1]           # left to right, float
    G = np.where(green, src, (0.25 * nb4.astype(np.float64)).astype(f32)).astype(f32)

    def ratio(ys, xs):
        return (src[np.ix_(ys, xs)] / G[np.ix_(ys, xs)]).astype(f32)
    ar = np.arange
    vert = (ratio(ym1, ar(w)) + ratio(yp1, ar(w))).astype(f32)
    horz = (ratio(ar(h), xm1) + ratio(ar(h), xp1)).astype(f32)
    diag = (((ratio(ym1, xm1) + ratio(ym1, xp1)).astype(f32) + ratio(yp1, xm1)).astype(f32) + ratio(yp1, xp1)).astype(f32)
    g64 = G.astype(np.float64)
    half_v, half_h, quarter_d = (g64 * 0.5 * vert).astype(f32), (g64 * 0.5 * horz).astype(f32), (g64 * 0.25 * diag).astype(f32)
    R = np.where(blue_row, np.where(green, half_v, quarter_d), np.where(green, half_h, src))
    B = np.where(blue_row, np.where(green, half_h, src), np.where(green, half_v, quarter_d))
    return np.stack([R, G, B]).astype(f32)


def raw_weights_numpy(w, h, rx, ry, weight):
    """utils.cpp:1336-1374"""
    weight = min(max(weight, 0.0), 3.0)
    other = np.float32(0.5 * (3 - np.float64(np.float32(weight))))
    Y, X = np.mgrid[0:h, 0:w]
    blue_row = (Y + (1 - ry)) % 2 == 0
    green = np.where(blue_row, ((X + (1 - rx)) % 2 == 0) if ry == 1 else ((X + rx) % 2 == 0), ((X + (1 - rx)) % 2 == 0) if ry == 0 else ((X + rx) % 2 == 0))
    W = np.full((3, h, w), other, np.float32)
    W[1][green] = weight
    W[2][blue_row & ~green] = weight
    W[0][~blue_row & ~green] = weight
    return W


@pytest.mark.parametrize("rx,ry,weight", [(1, 0, 2.0), (0, 0, 1.0), (0, 1, 5.0), (1, 1, 0.5)])
def test_demosaic_and_raw_weights_against_numpy(host_build, tmp_path, rx, ry, weight):
    """the host ingest routines (restated from utils.cpp:1241-1374, which needs OpenCV to build) against a direct numpy evaluation"""
    w, h = 37, 22
    st = ((w + 3) // 4) * 4
    rng = np.random.default_rng(rx + 2 * ry)
    mosaic = np.zeros((h, st), np.float32)
    mosaic[:, :w] = rng.uniform(20, 4000, (h, w)).astype(np.float32)
    mosaic.tofile(str(tmp_path / "bayer_in.bin"))
    (tmp_path / "bayer.txt").write_text("%d %d %d %d %g\n" % (w, h, rx, ry, weight))
    write_png_cases(tmp_path)
    exe = _link_host_test(tmp_path, ["test_host.cpp"], "test_host")
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and "host tests OK" in r.stdout, r.stdout + r.stderr
    rgb = np.fromfile(str(tmp_path / "bayer_rgb.bin"), dtype=np.float32).reshape(3, h, st)[:, :, :w]
    cw = np.fromfile(str(tmp_path / "bayer_w.bin"), dtype=np.float32).reshape(3, h, st)[:, :, :w]
    assert np.array_equal(rgb, bayer_numpy(mosaic[:, :w], rx, ry))
    assert np.array_equal(cw, raw_weights_numpy(w, h, rx, ry, weight))


def bayer_cv8u_numpy(src, rx, ry):
    """cv::cvtColor(CV_Bayer*2RGB) on 8-bit data as the reference uses it (slow_flow.cpp:502-520), written with whole-array shifts: saturating
    round-half-even conversion, (4 neighbours + 2) >> 2 / (4 diagonals + 2) >> 2 at red and blue sites, (2 neighbours + 1) >> 1 at green sites,
    outer rows and columns repeated from their inner neighbours"""
    m = np.clip(np.rint(src.astype(np.float64)), 0, 255).astype(np.int64)
    h, w = m.shape
    c = m[1:-1, 1:-1]
    up, dn, lf, rt = m[:-2, 1:-1], m[2:, 1:-1], m[1:-1, :-2], m[1:-1, 2:]
    cross = (up + dn + lf + rt + 2) >> 2
    diag = (m[:-2, :-2] + m[:-2, 2:] + m[2:, :-2] + m[2:, 2:] + 2) >> 2
    horiz, vert = (lf + rt + 1) >> 1, (up + dn + 1) >> 1
    Y, X = np.mgrid[1:h - 1, 1:w - 1]
    red_row, red_col = (Y - ry) % 2 == 0, (X - rx) % 2 == 0
    site = red_row == red_col
    R = np.where(site, np.where(red_row, c, diag), np.where(red_row, horiz, vert))
    G = np.where(site, cross, c)
    B = np.where(site, np.where(red_row, diag, c), np.where(red_row, vert, horiz))
    out = np.zeros((3, h, w), np.int64)
    out[:, 1:-1, 1:-1] = np.stack([R, G, B])
    out[:, :, 0] = out[:, :, 1]; out[:, :, -1] = out[:, :, -2]
    out[:, 0, :] = out[:, 1, :]; out[:, -1, :] = out[:, -2, :]
    return out.astype(np.float32)


@pytest.mark.parametrize("rx,ry", [(1, 0), (0, 0), (0, 1), (1, 1)])
def test_opencv_style_demosaic_and_tiff_reader(host_build, tmp_path, rx, ry):
    """raw_demosaicing 2 (OpenCV's 8-bit bilinear Bayer conversion, restated: OpenCV is not in the image) against an independent numpy formulation,
    and the TIFF reader (the reference cfg's input format) against Pillow/libtiff on files of every compression and layout it accepts"""
    from PIL import Image
    w, h = 41, 26
    st = ((w + 3) // 4) * 4
    rng = np.random.default_rng(10 + rx + 2 * ry)
    mosaic = np.zeros((h, st), np.float32)
    mosaic[:, :w] = rng.uniform(-20, 300, (h, w)).astype(np.float32)
    mosaic[3, 5:9] = [0.5, 1.5, 2.5, 254.5]                                      # halves round to even
    mosaic.tofile(str(tmp_path / "bayer_in.bin"))
    (tmp_path / "bayer.txt").write_text("%d %d %d %d %g\n" % (w, h, rx, ry, 1.0))
    write_png_cases(tmp_path)
    # TIFF files: 8 / 16 bit grey and RGB, every compression, several strips; a big-endian one assembled by hand
    cases = {}
    g16 = (rng.integers(0, 65536, (53, 70))).astype(np.uint16)
    g16[10:20] = (np.arange(70) * 7)[None, :]                                    # compressible rows
    g8 = (g16 >> 8).astype(np.uint8)
    rgb8 = rng.integers(0, 256, (31, 45, 3)).astype(np.uint8)
    rgb8[5:15] = 77
    for comp in (None, "tiff_lzw", "packbits", "tiff_adobe_deflate"):
        tag = comp or "raw"
        for name, arr in (("g16", g16), ("g8", g8), ("rgb8", rgb8)):
            fn = "%s_%s.tif" % (name, tag)
            kw = {} if comp is None else {"compression": comp}
            Image.fromarray(arr).save(str(tmp_path / fn), **kw)
            cases[fn] = arr
    fn = "g16_lzw_pred.tif"
    Image.fromarray(g16).save(str(tmp_path / fn), compression="tiff_lzw", tiffinfo={317: 2})
    cases[fn] = g16
    fn = "rgb8_lzw_pred_strips.tif"
    Image.fromarray(rgb8).save(str(tmp_path / fn), compression="tiff_lzw", tiffinfo={317: 2, 278: 4})
    cases[fn] = rgb8
    import struct
    be = g16[:7, :9]
    ifd = [(256, 3, 1, 9), (257, 3, 1, 7), (258, 3, 1, 16), (259, 3, 1, 1), (262, 3, 1, 1), (273, 4, 1, 8), (277, 3, 1, 1), (278, 3, 1, 7), (279, 4, 1, be.size * 2)]
    blob = b"MM" + struct.pack(">HI", 42, 8 + be.size * 2) + be.astype(">u2").tobytes() + struct.pack(">H", len(ifd))
    for tag, typ, cnt, val in ifd:
        blob += struct.pack(">HHI", tag, typ, cnt) + (struct.pack(">HH", val, 0) if typ == 3 else struct.pack(">I", val))
    blob += struct.pack(">I", 0)
    (tmp_path / "g16_be.tif").write_bytes(blob)
    cases["g16_be.tif"] = be
    (tmp_path / "tiff_list.txt").write_text("\n".join(cases) + "\n")
    exe = _link_host_test(tmp_path, ["test_host.cpp"], "test_host")
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and "host tests OK" in r.stdout, r.stdout + r.stderr
    got = np.fromfile(str(tmp_path / "bayer_rgb_cv.bin"), dtype=np.float32).reshape(3, h, st)[:, :, :w]
    want = bayer_cv8u_numpy(mosaic[:, :w], rx, ry)
    assert np.array_equal(got, want)
    assert want.min() == 0 and want.max() == 255                                  # the saturation was exercised
    for fn, arr in cases.items():
        ok, tw, th, tc, td = [int(v) for v in (tmp_path / (fn + ".txt")).read_text().split()]
        assert ok == 1, fn
        assert (tw, th, tc, td) == (arr.shape[1], arr.shape[0], 1 if arr.ndim == 2 else 3, 16 if arr.dtype == np.uint16 else 8), fn
        samples = np.fromfile(str(tmp_path / (fn + ".bin")), dtype=np.uint16).reshape(arr.shape)
        assert np.array_equal(samples, arr.astype(np.uint16)), fn
        assert np.array_equal(np.asarray(Image.open(str(tmp_path / fn))).astype(np.uint16), arr.astype(np.uint16)), fn    # Pillow agrees on what the file holds


def test_image_readers_survive_malformed_files(host_build, tmp_path):
    """the driver reads user files: byte-mutated and truncated PNG / TIFF inputs must be rejected or decoded, never read or written out of bounds
    (address + undefined-behaviour sanitizers on the CPU build of the two readers)"""
    from PIL import Image
    rng = np.random.default_rng(3)
    g16 = rng.integers(0, 65536, (37, 50)).astype(np.uint16); g16[5:15] = 7
    rgb = rng.integers(0, 256, (21, 33, 3)).astype(np.uint8); rgb[3:9] = 9
    files = []
    for name, im, kw in (("a.tif", g16, {}), ("b.tif", g16, {"compression": "tiff_lzw", "tiffinfo": {317: 2, 278: 5}}), ("c.tif", rgb, {"compression": "packbits"}),
                         ("d.tif", rgb, {"compression": "tiff_adobe_deflate"}), ("a.png", g16, {}), ("b.png", rgb, {})):
        Image.fromarray(im).save(str(tmp_path / name), **kw)
        files.append(str(tmp_path / name))
    Image.fromarray(rgb).convert("P").save(str(tmp_path / "c.png")); files.append(str(tmp_path / "c.png"))
    exe = str(tmp_path / "fuzz_readers")
    r = subprocess.run(["g++", "-std=c++11", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", HOST,
                        os.path.join(ROOT, "tests", "host", "fuzz_readers.cpp"), os.path.join(HOST, "tiff.cpp"), os.path.join(HOST, "png.cpp"), "-lz", "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "600", str(tmp_path / "mutant.bin")] + files, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "none crashed" in r.stdout, r.stdout + r.stderr[-2000:]


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
    # (dm_scale != 1 was refused until round 5: test_driver_deep_matching_at_half_resolution)
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
            with open(str(out / "occlusion" / ("frame_%d.pgm" % (10 + j * steps))), "rb") as f:
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
def test_driver_refuses_a_frame_of_another_size(host_build, tmp_path):
    """a smaller frame in the middle of the sequence: the uploaders copy the published size out of every frame's buffer, so the frame must be caught before it is
    marked ready -- whichever frame finishes decoding first"""
    from synth import texture_frame
    w, h, jets, steps = 96, 64, 3, 1
    nframes = 1 + (jets + 2) * steps
    for k in range(nframes):
        ww, hh = (80, 48) if k == 3 else (w, h)
        write_ppm(str(tmp_path / ("f_%03d.ppm" % (10 - steps + k))), np.clip(np.round(texture_frame(ww, hh, k)[:, :, :ww]), 0, 255))
    cfg = tmp_path / "run.cfg"
    cfg.write_text("file\t%s/f_%%03i.ppm\noutput\t%s/out\nJets\t%d\nstart\t10\nmax_fps\t200\n16bit\t0\nraw\t0\nscale\t1.0\ndeep_matching\t0\n"
                   "slow_flow_S\t2\nslow_flow_layers\t1\nslow_flow_niter_alter\t1\nslow_flow_niter_outer\t1\nslow_flow_occlusion_reasoning\t0\ngpus\t1\n" % (tmp_path, tmp_path, jets))
    for _ in range(3):                                            # the decode order varies from run to run
        r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg), "-overwrite"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 3 and "different sizes" in r.stderr, r.stdout + r.stderr


def test_driver_refuses_gpu_batch_beyond_a_job(host_build, tmp_path):
    """a lockstep job takes at most 128 windows (two 64-bit words of windows still iterating): the driver says so instead of clamping the key;
    checked before a frame is read, so no GPU is needed"""
    cfg = tmp_path / "run.cfg"
    cfg.write_text("file\t%s/f_%%03i.ppm\noutput\t%s/out\nJets\t1\nstart\t10\nmax_fps\t200\n16bit\t0\nraw\t0\nscale\t1.0\ndeep_matching\t0\nslow_flow_S\t2\ngpu_batch\t129\n" % (tmp_path, tmp_path))
    r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg), "-overwrite"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "gpu_batch 129 is out of range" in r.stderr, r.stdout + r.stderr


@pytest.mark.gpu
def test_driver_multi_gpu_path_rehearsed_on_one_card(host_build, tmp_path):
    """the driver's N-GPU path (a resident, normalised sequence per GPU; `gpu_streams` workers per GPU; windows dealt out by plan_workers) cannot run on
    this one-GPU box as written, so it is rehearsed: `gpu_oversubscribe 1` maps GPU g to device g mod devices.  Three virtual GPUs x 2 workers must
    write exactly the files, bit for bit, that one GPU writes, and name all three in the log."""
    from synth import texture_frame
    w, h, jets, S = 96, 64, 7, 2
    steps = S - 1
    nframes = 1 + (jets + 2) * steps
    for k in range(nframes):
        write_ppm(str(tmp_path / ("f_%03d.ppm" % (10 - steps + k))), np.clip(np.round(texture_frame(w, h, k)[:, :, :w]), 0, 255))
    body = ("file\t%s/f_%%03i.ppm\nJets\t%d\nstart\t10\nmax_fps\t200\n16bit\t0\nraw\t0\nscale\t1.0\ndeep_matching\t0\n"
            "slow_flow_S\t%d\nslow_flow_layers\t2\nslow_flow_niter_alter\t2\nslow_flow_niter_outer\t3\nslow_flow_occlusion_reasoning\t1\n"
            "slow_flow_rho_0\t1\nslow_flow_omega_0\t0\ngpu_batch\t2\n" % (tmp_path, jets, S))
    outs = {}
    for name, extra in (("one", "gpus\t1\ngpu_streams\t1\n"), ("three", "gpus\t3\ngpu_oversubscribe\t1\ngpu_streams\t2\n")):
        cfg = tmp_path / (name + ".cfg")
        cfg.write_text("output\t%s/out_%s\n" % (tmp_path, name) + body + extra)
        r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg), "-overwrite"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "Done!" in r.stdout, r.stdout + r.stderr
        outs[name] = r.stdout
        tj = json.load(open(str(tmp_path / ("out_" + name) / "timings.json")))
        assert len(tj) == 2 * jets
    assert all(("GPU %d" % g) in outs["three"] for g in range(3)), outs["three"]
    flo = sorted(f for f in os.listdir(str(tmp_path / "out_one")) if f.endswith(".flo"))
    assert len(flo) == 2 * jets
    for f in flo:
        assert (tmp_path / "out_one" / f).read_bytes() == (tmp_path / "out_three" / f).read_bytes(), f
    # the ingest is sharded: a GPU is sent the frames of its own windows + the halo it shares with its neighbour, not the sequence; the statistics
    # are formed once from per-frame sums (the same .flo as the one-GPU run, above, says the normalisation is the same bit for bit)
    one, three = (json.load(open(str(tmp_path / ("out_" + n) / "run.json"))) for n in ("one", "three"))
    seq_bytes = nframes * 3 * w * h * 4
    assert one["sequence_bytes"] == seq_bytes and [g["frames"] for g in one["per_gpu"]] == [[0, nframes]] and one["per_gpu"][0]["upload_bytes"] == seq_bytes
    # 14 windows on 6 workers: GPU 0 gets windows 0..3 (jets 0, 1 -> frames 0..4), GPU 1 windows 4..8 (jets 2, 3 and jet 4 forwards -> 2..6), GPU 2 the rest
    # (jet 4 backwards, jets 5, 6 -> 5..9)
    assert [g["frames"] for g in three["per_gpu"]] == [[0, 5], [2, 7], [5, nframes]], three["per_gpu"]
    for g in three["per_gpu"]:
        assert g["upload_bytes"] == (g["frames"][1] - g["frames"][0]) * 3 * w * h * 4 and g["upload_done_s"] <= g["ready_s"] <= g["first_refine_s"]
    assert sum(g["upload_bytes"] for g in three["per_gpu"]) < 3 * seq_bytes * 0.6          # round 4: three whole sequences


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
    # (c) the reference cfg's own input side (cfgs/slow_flow.cfg:5,13-16): TIFF mosaics, raw_demosaicing 2 (OpenCV's 8-bit conversion, restated)
    from PIL import Image
    for k, f in enumerate(frames):
        yy, xx = np.mgrid[0:h, 0:w]
        red = (xx % 2 == 1) & (yy % 2 == 0)
        blue = (xx % 2 == 0) & (yy % 2 == 1)
        Image.fromarray(np.where(red, f[0], np.where(blue, f[2], f[1])).astype(np.uint16)).save(str(tmp_path / ("t_%03d.tif" % (10 - steps + k))), compression="tiff_lzw")
    cfg.write_text("file\t%s/t_%%03i.tif\noutput\t%s/out_tif\nraw\t1\nraw_demosaicing\t2\nraw_red_loc\t1,0\nraw_weight\t1\nscale\t1.0\n" % (tmp_path, tmp_path) + common)
    r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg), "-overwrite"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    u, v = read_flo(str(tmp_path / "out_tif" / "t_010.flo"))
    assert abs(np.median(u) - 1.5) < 0.15 and abs(np.median(v) + 0.75) < 0.15


def _link_host_test(tmp_path, sources, name):
    exe = str(tmp_path / name)
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-pthread", "-I", HOST] + [os.path.join(ROOT, "tests", "host", x) for x in sources] +
                       [os.path.join(HOST, "libslowflow_host.a"), "-L", os.path.join(ROOT, "slowflow_amd"), "-lslowflow_amd", "-lz",
                        "-Wl,-rpath," + os.path.join(ROOT, "slowflow_amd"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_entry_point_tests_compile_and_link(host_build, tmp_path):
    """the GPU tests of the C++ entry points build against the host library and the C-ABI .so here (they run on the GPU box)"""
    _link_host_test(tmp_path, ["test_entry_points.cpp", "test_entry_symbols.cpp"], "test_entry_points")


@pytest.mark.gpu
def test_cpp_entry_points(host_build, tmp_path):
    """north_star's entry points, called as the reference's own caller calls them (slow_flow.cpp:673, :865-888, :1018-1023): C++
    normalize() + Variational_MT::variational (forward with setChannelWeights, backward without) against the same windows through the
    C-ABI binding, bit for bit; the literal symbols sor_coupled (solver.h:11) and variational (variational.h:34) on the golden systems
    against the outputs of the compiled reference (tests/golden) and the oracle."""
    import slowflow_amd as sfa
    from synth import texture_frame
    exe = _link_host_test(tmp_path, ["test_entry_points.cpp", "test_entry_symbols.cpp"], "test_entry_points")
    w, h, S, n = 96, 64, 3, 7
    F = 2 * (S - 1) + 1
    stride = sfa.stride_of(w)
    rng = np.random.default_rng(0)
    frames = []
    for k in range(n):
        a = np.zeros((3, h, stride), np.float32)
        a[:, :, :w] = np.clip(np.round(texture_frame(w, h, k)[:, :, :w]), 0, 255)
        a.tofile(str(tmp_path / ("ep_frame_%d.bin" % k)))
        frames.append(a)
    chw = np.ones((3, h, stride), np.float32)
    chw[:, :, :w] = rng.uniform(0.5, 1.5, (3, h, w)).astype(np.float32)
    chw.tofile(str(tmp_path / "ep_chw.bin"))
    (tmp_path / "ep.cfg").write_text(
        "slow_flow_S\t3\nslow_flow_layers\t3\nslow_flow_p_scale\t0.9\nslow_flow_niter_alter\t2\nslow_flow_niter_outer\t3\nslow_flow_niter_inner\t1\n"
        "slow_flow_niter_solver\t30\nslow_flow_sor_omega\t1.9\nslow_flow_thres_outer\t0\nslow_flow_thres_inner\t0\nslow_flow_occlusion_reasoning\t1\n"
        "slow_flow_occlusion_penalty\t0.1\nslow_flow_occlusion_alpha\t0.1\nslow_flow_rho_0\t1\nslow_flow_rho_1\t1\nslow_flow_omega_0\t0\nslow_flow_omega_1\t2\n"
        "slow_flow_alpha\t4.0\nslow_flow_gamma\t6.0\nslow_flow_delta\t1.0\nslow_flow_smoothing\t1\nslow_flow_dataterm\t1\n16bit\t0\n"
        "slow_flow_robust_color\t1\nslow_flow_robust_color_eps\t0.001\nslow_flow_robust_color_truncation\t0.5\n"
        "slow_flow_robust_reg\t1\nslow_flow_robust_reg_eps\t0.001\nslow_flow_robust_reg_truncation\t0.5\n")
    # the literal symbols' inputs: the golden SOR system and the golden two-frame case (outputs of the compiled reference are committed)
    G = np.load(os.path.join(ROOT, "tests", "golden", "ref_vectors.npz"))
    T = np.load(os.path.join(ROOT, "tests", "golden", "ref_two_frame.npz"))
    sw, sh_ = 67, 45
    names = ["du", "dv", "a11", "a12", "a22", "b1", "b2", "sh", "sv"]
    sysm = {nm: np.ascontiguousarray(G["sor_%dx%d_in_%s" % (sw, sh_, nm)], dtype=np.float32) for nm in names}
    for nm in names:
        sysm[nm].tofile(str(tmp_path / ("sym_sor_%s.bin" % nm)))
    (tmp_path / "sym_sor.txt").write_text("%d %d 30 1.9\n" % (sw, sh_))
    tw, th = (int(v) for v in T["size"])
    for nm, key in (("im1", "im1"), ("im2", "im2"), ("wx", "wx0"), ("wy", "wy0")):
        np.ascontiguousarray(T[key], dtype=np.float32).tofile(str(tmp_path / ("sym_var_%s.bin" % nm)))
    (tmp_path / "sym_var.txt").write_text("%d %d 3.0 0.2 0.0 1.0 5 1 7 1.5\n" % (tw, th))        # the "weights" case of make_golden_2frame.py
    r = subprocess.run([exe, str(tmp_path), str(w), str(h), str(n)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "entry points OK" in r.stdout, r.stdout + r.stderr

    def rd(name, hh=h, st=stride):
        return np.fromfile(str(tmp_path / name), dtype=np.float32).reshape(hh, st)
    # ---- Variational_MT against the binding ---------------------------------------------------------------------------
    ctx = sfa.Context(0)
    fr = [f.copy() for f in frames]
    avg, std = ctx.normalize(fr, w)
    cfg = (tmp_path / "ep_after_normalize.cfg").read_text()
    for k in range(3):
        assert ("slow_flow_img_norm_avg_%d\t%g" % (k + 1, avg[k])) in cfg and ("slow_flow_img_norm_std_%d\t%g" % (k + 1, std[k])) in cfg
    p = sfa.default_params()
    p.S = 3; p.layers = 3; p.niter_alter = 2; p.niter_outer = 3; p.occlusion_reasoning = 1; p.thres_outer = 0; p.thres_inner = 0; p.hbit = 0
    p.rho[0] = 1; p.rho[1] = 1; p.omega[0] = 0; p.omega[1] = 2; p.occlusion_penalty = 0.1; p.occlusion_alpha = 0.1
    for k in range(3):
        p.norm_avg[k] = float("%g" % avg[k]); p.norm_std[k] = float("%g" % std[k])
    for tag, win, cw in (("fwd", fr[0:F], [np.ascontiguousarray(chw[k]) for k in range(3)]), ("bwd", [fr[n - 1 - i] for i in range(n - F, n)], None)):
        wx, wy = np.zeros((h, stride), np.float32), np.zeros((h, stride), np.float32)
        chg, occ = ctx.variational(p, wx, wy, win, w, cw, want_occ=True)
        assert np.array_equal(rd("ep_%s_wx.bin" % tag)[:, :w], wx[:, :w]) and np.array_equal(rd("ep_%s_wy.bin" % tag)[:, :w], wy[:, :w]), tag
        assert np.array_equal(rd("ep_%s_occ.bin" % tag)[:, :w], occ[:, :w]), tag
        c = [float(v) for v in (tmp_path / ("ep_%s_change.txt" % tag)).read_text().split()]
        assert abs(c[0] - chg[0]) <= 1e-7 and abs(c[1] - chg[1]) <= 1e-7
    fx = rd("ep_fwd_wx.bin")[:, :w]
    assert abs(np.median(fx) - 1.5) < 0.15                                     # and it is a flow: the texture moves by (1.5, -0.75) px / frame
    assert not np.array_equal(fx, -rd("ep_bwd_wx.bin")[:, :w])
    ctx.close()
    # ---- sor_coupled (solver.h:11): the committed outputs of the compiled reference's own sor_coupled (tests/golden/make_golden.py), a11 / a12 /
    #      a22 overwritten with the inverted blocks as solver.c:183-188 does ----------------------------------------------------------------
    sst = sysm["du"].shape[1]
    for nm, key in (("du", "K30_du"), ("dv", "K30_dv"), ("a11", "inv_a11"), ("a12", "inv_a12"), ("a22", "inv_a22")):
        assert np.array_equal(rd("sym_sor_out_%s.bin" % nm, sh_, sst)[:, :sw], G["sor_%dx%d_%s" % (sw, sh_, key)][:, :sw]), nm
    # ---- variational (variational.h:34): the committed output of the compiled reference -----------------------------------------
    tst = T["wx0"].shape[1]
    assert np.array_equal(rd("sym_var_out_wx.bin", th, tst)[:, :tw], T["weights_wx"]) and np.array_equal(rd("sym_var_out_wy.bin", th, tst)[:, :tw], T["weights_wy"])


def _write_sequence(out, nframes, first, W, H, seed=0, amp=0.5):
    """synthetic high-speed sequence as PPM files f_%04d.ppm: band-limited texture moving by a smooth sub-pixel flow per frame"""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    pad = 96
    base = gaussian_filter(rng.uniform(0, 1, size=(3, H + 2 * pad, W + 2 * pad)), sigma=(0, 2.0, 2.0), mode="nearest")
    base = (base - base.min()) / (base.max() - base.min()) * 255.0
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    fu = amp + 0.3 * amp * np.sin(2 * np.pi * yy / H)
    fv = 0.3 * amp * np.cos(2 * np.pi * xx / W)
    frames = []
    for t in range(nframes):
        sx, sy = xx - t * fu + pad, yy - t * fv + pad
        x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
        ax, ay = sx - x0, sy - y0
        img = np.empty((3, H, W), np.float64)
        for c in range(3):
            b = base[c]
            img[c] = b[y0, x0] * (1 - ax) * (1 - ay) + b[y0, x0 + 1] * ax * (1 - ay) + b[y0 + 1, x0] * (1 - ax) * ay + b[y0 + 1, x0 + 1] * ax * ay
        write_ppm(os.path.join(out, "f_%04d.ppm" % (first + t)), img)
        frames.append(np.clip(np.round(img), 0, 255).astype(np.float32))
    return frames, fu, fv


CFG4_SOLVER = ("slow_flow_S\t3\nslow_flow_layers\t5\nslow_flow_p_scale\t0.9\nslow_flow_niter_alter\t10\nslow_flow_niter_outer\t10\nslow_flow_niter_inner\t1\n"
               "slow_flow_niter_solver\t30\nslow_flow_sor_omega\t1.9\nslow_flow_occlusion_reasoning\t1\nslow_flow_occlusion_penalty\t0.1\nslow_flow_occlusion_alpha\t0.1\n"
               "slow_flow_rho_0\t1\nslow_flow_rho_1\t1\nslow_flow_omega_0\t0\nslow_flow_omega_1\t2\nslow_flow_alpha\t4.0\nslow_flow_gamma\t6.0\nslow_flow_delta\t1.0\n"
               "slow_flow_thres_outer\t1e-5\nslow_flow_thres_inner\t1e-5\n")


@pytest.mark.gpu
def test_config4_cfg_over_64_jets_1024x436(host_build, tmp_path):
    """BASELINE config 4 on this box's GPU: the reference's cfgs/slow_flow.cfg solver section (S=3, 5 layers, 10 alternations x 10 outer x 30
    sweeps, occlusion reasoning, thresholds 1e-5) over 64 consecutive jets of a 1024x436 sequence, forward and backward, through the
    slow_flow driver (133 PPM frames in; 128 .flo, 64 colour PNG, 64 occlusion maps out).  A sample of the .flo files equals the same
    windows refined one by one through the C-ABI binding bit for bit (windows batched 32 at a time in the driver: lockstep, passengers and
    job reuse included), the sub-pixel motion is recovered, and timings.json accounts for all 128 windows.  run.json shows where the wall
    time went: the refinement (GPU-bound) must be the bulk, ingest and output overlapped around it."""
    import json
    import slowflow_amd as sfa
    W, H, S, JETS = 1024, 436, 3, 64
    steps = S - 1
    nframes = 1 + (JETS + 2) * steps
    seqdir = str(tmp_path)
    frames, fu, fv = _write_sequence(seqdir, nframes, 100 - steps, W, H)
    cfg = tmp_path / "run.cfg"
    cfg.write_text("file\t%s/f_%%04i.ppm\noutput\t%s/out\nJets\t%d\nstart\t100\nmax_fps\t200\n16bit\t0\nraw\t0\nscale\t1.0\ndeep_matching\t0\ngpus\t1\n"
                   "slow_flow_output_occlusions\t1\n" % (seqdir, seqdir, JETS) + CFG4_SOLVER)
    r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg), "-overwrite"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "Done!" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    out = tmp_path / "out"
    tj = json.load(open(str(out / "timings.json")))
    assert len(tj) == 2 * JETS and all(t["gpu"] == 0 and t["seconds"] > 0 for t in tj)
    assert sorted((t["jet"], t["direction"]) for t in tj) == sorted((j, d) for j in range(JETS) for d in ("forward", "backward"))
    for j in range(JETS):
        assert (out / ("f_%04d.flo" % (100 + j * steps))).exists() and (out / ("f_%04d_back.flo" % (100 + j * steps + steps))).exists()
        assert (out / ("frame_%d.png" % (100 + j * steps))).exists() and (out / "occlusion" / ("frame_%d.pgm" % (100 + j * steps))).exists()
    run = json.load(open(str(out / "run.json")))
    assert run["windows"] == 128 and run["refine_seconds"] <= run["total_seconds"]
    print("config 4 through the driver on one GPU: %s" % run)
    # the pipeline hides output behind the refinement and sends every frame to the GPU once: beyond the refinement itself (which includes creating the
    # workers' contexts and jobs) the run may spend 15 % (measured on MI355X: 2.40 s total, 2.20 s refinement, 0.15 s ingest)
    assert run["total_seconds"] <= 1.15 * run["refine_seconds"] + 0.3, run
    # ---- a sample of windows through the binding ------------------------------------------------------------------------------
    ctx = sfa.Context(0)
    stride = sfa.stride_of(W)
    fr = []
    for f in frames:
        a = np.zeros((3, H, stride), np.float32)
        a[:, :, :W] = f
        fr.append(a)
    avg, std = ctx.normalize(fr, W)
    p = sfa.default_params()
    p.S = S; p.layers = 5; p.niter_alter = 10; p.niter_outer = 10; p.occlusion_reasoning = 1; p.thres_outer = 1e-5; p.thres_inner = 1e-5
    p.hbit = 0; p.smoothing = 1; p.rho[0] = 1; p.rho[1] = 1; p.omega[0] = 0; p.omega[1] = 2; p.occlusion_penalty = 0.1; p.occlusion_alpha = 0.1
    for k in range(3):
        p.norm_avg[k] = float("%g" % avg[k]); p.norm_std[k] = float("%g" % std[k])
    for j, back in ((0, False), (0, True), (15, True), (16, False), (31, True), (47, False), (63, False), (63, True)):
        f0 = j * steps
        win = fr[f0:f0 + 2 * steps + 1] if not back else [fr[nframes - 1 - i] for i in range(nframes - 1 - f0 - 3 * steps, nframes - 1 - f0 - 3 * steps + 2 * steps + 1)]
        wx, wy = np.zeros((H, stride), np.float32), np.zeros((H, stride), np.float32)
        ctx.variational(p, wx, wy, win, W)
        name = "f_%04d%s.flo" % ((100 + f0) if not back else (100 + f0 + steps), "_back" if back else "")
        u, v = read_flo(str(out / name))
        assert np.array_equal(u, wx[:, :W] * steps) and np.array_equal(v, wy[:, :W] * steps), name
    ctx.close()
    # the flow of the reference frame over `steps` frames: the synthetic sub-pixel motion, forward and (negated) backward
    u, v = read_flo(str(out / "f_0100.flo"))
    inner = (slice(20, H - 20), slice(20, W - 20))
    assert np.abs(u[inner] - steps * fu[inner]).mean() < 0.05 and np.abs(v[inner] - steps * fv[inner]).mean() < 0.05
    ub, vb = read_flo(str(out / "f_0102_back.flo"))
    assert np.abs(ub[inner] + steps * fu[inner]).mean() < 0.05 and np.abs(vb[inner] + steps * fv[inner]).mean() < 0.05


@pytest.mark.gpu
def test_driver_adaptive_frame_rates_and_alternation_outputs(host_build, tmp_path):
    """the rest of the driver surface (slow_flow.cpp:277-399, :878-884): with adaptiveFR.dat and <sequence>/quantil.dat the run is done twice,
    at the high and the low frame rate, into high_fr/ and low_fr/ (`-fr k` selects one), reading every skip-th frame; with WRITE_FILES
    verbosity the occlusion labels of every alternation are written as tmp/frame_<n>_<alter>.png, the last one being the final estimate."""
    import json
    import slowflow_amd as sfa
    W, H, S, JETS = 96, 64, 2, 2
    steps = S - 1
    seqdir = tmp_path / "seq"
    seqdir.mkdir()
    # hfr_quantil 2 / quantil 1.0 -> hfr_rate 2; keyframes = 200 / 20 = 10 -> while 10 % 2: ok; lfr = min(10, 2 * 4) = 8 -> 8 does not divide 10 -> 10 -> min(10 / 1, 10)
    frames, fu, fv = _write_sequence(str(seqdir), 1 + (JETS + 2) * steps * 10 + 10, 0, W, H, seed=1, amp=0.25)
    (seqdir / "quantil.dat").write_text("1.0\n")
    (tmp_path / "adaptiveFR.dat").write_text("opt_hfr_quantil\t2\nopt_lfr_quantil\t8\nopt_lfr_rate\t4\n")
    cfg = tmp_path / "run.cfg"
    cfg.write_text("file\t%s/f_%%04i.ppm\noutput\t%s/out\nJets\t%d\nstart\t10\nmax_fps\t200\nref_fps\t20\nadaptive\t1\nadaptive_fr_file\t%s/adaptiveFR.dat\n"
                   "16bit\t0\nraw\t0\nscale\t1.0\ndeep_matching\t0\ngpus\t1\nverbose\t00001\nslow_flow_S\t2\nslow_flow_layers\t2\nslow_flow_niter_alter\t3\n"
                   "slow_flow_niter_outer\t3\nslow_flow_occlusion_reasoning\t1\nslow_flow_thres_outer\t0\nslow_flow_thres_inner\t0\nslow_flow_rho_0\t1\nslow_flow_omega_0\t0\n"
                   % (seqdir, tmp_path, JETS, tmp_path))
    r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg), "-overwrite"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "hfr_rate 2" in r.stdout and "lfr_rate 10" in r.stdout, r.stdout + r.stderr
    hi, lo = tmp_path / "out" / "high_fr", tmp_path / "out" / "low_fr"
    assert "jet_fps\t100" in (hi / "config.cfg").read_text() and "jet_fps\t20" in (lo / "config.cfg").read_text()
    for d, skip in ((hi, 2), (lo, 10)):
        assert len(json.load(open(str(d / "timings.json")))) == 2 * JETS
        u, v = read_flo(str(d / "f_0010.flo"))                                              # frames 10 -> 10 + skip
        inner = (slice(12, H - 12), slice(12, W - 12))
        assert np.abs(u[inner] - skip * fu[inner]).mean() < 0.1 * skip and np.abs(v[inner] - skip * fv[inner]).mean() < 0.1 * skip, (skip, np.abs(u[inner] - skip * fu[inner]).mean())
        for a in (1, 2):
            assert open(str(d / "tmp" / ("frame_10_%d.png" % a)), "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    # `-fr 1` alone redoes only the low frame rate
    r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg), "-overwrite", "-fr", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.count("Reading") == 1 + (JETS + 2) * steps
    # the last alternation's labels are the occlusion estimate that is written next to the flow
    import zlib
    png = open(str(hi / "tmp" / "frame_10_2.png"), "rb").read()
    i, idat = 8, b""
    while i < len(png):
        n, t = struct.unpack(">I4s", png[i:i + 8])
        if t == b"IDAT":
            idat += png[i + 8:i + 8 + n]
        i += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(H, W + 1)
    assert np.all(rows[:, 0] == 0)
    pgm = open(str(hi / "occlusion" / "frame_10.pgm"), "rb").read()
    assert np.array_equal(rows[:, 1:], np.frombuffer(pgm[-W * H:], dtype=np.uint8).reshape(H, W))
    assert set(np.unique(rows[:, 1:])) <= {0, 255}


def write_ppm16(path, img):        # img: (3,h,w) float 0..65535, big-endian samples as the PNM format has them
    h, w = img.shape[1:]
    data = np.clip(np.round(img), 0, 65535).astype(">u2").transpose(1, 2, 0).tobytes()
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n65535\n" % (w, h))
        f.write(data)


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [8, 16])
def test_driver_deep_matching_initialisation(host_build, tmp_path, bits):
    """deep_matching 1 (slow_flow.cpp:744-863, 960-1004): with match and edge files at the reference's locations (<output>tmp/matches_<a>_<b>.dat, edges_<n>.dat) the
    driver initialises every window with EpicFlow's interpolation and refines from there: the .flo equals the binding run from the same initial flow (epic through
    tests/host/epic_tool.cpp, divided by steps), and a large motion that the zero-initialised refinement cannot reach at one level is recovered."""
    import slowflow_amd as sfa
    from synth import texture_frame
    w, h, jets, S = 128, 96, 1, 2
    steps = S - 1
    nframes = 1 + (jets + 2) * steps
    DX, DY = 9.0, -6.0                                               # large motion: out of reach for a single-level variational refinement from zero
    frames = [np.clip(np.round(texture_frame(w, h, k, dx=DX, dy=DY)[:, :, :w]), 0, 255) for k in range(nframes)]
    if bits == 16:
        # 16-bit input: samples 0..65535 that are NOT multiples of 255, so the 8-bit copy EpicFlow's saliency works on
        # (img.convertTo(CV_8U, 1/255): slow_flow.cpp:472-474, :578) differs from both the frame and frame / 255
        frames = [np.clip(np.round(f * 255.0 + 100.0 * np.sin(0.37 * np.arange(w))[None, None, :] + 60.0), 0, 65535).astype(np.float32) for f in frames]
    for k, f in enumerate(frames):
        (write_ppm16 if bits == 16 else write_ppm)(str(tmp_path / ("f_%03d.ppm" % (10 - steps + k))), f)
    out = tmp_path / "out"
    (out / "tmp").mkdir(parents=True)
    rng = np.random.default_rng(0)
    for a, b, sx, sy in ((10, 11, DX, DY), (11, 10, -DX, -DY)):                           # forward and backward matches of the constant translation (+ noise)
        with open(str(out / "tmp" / ("matches_%d_%d.dat" % (a, b))), "w") as f:
            for _ in range(500):
                x, y = rng.uniform(12, w - 13), rng.uniform(8, h - 9)
                f.write("%.3f %.3f %.3f %.3f 3.1 1\n" % (x, y, x + sx + rng.normal(0, 0.2), y + sy + rng.normal(0, 0.2)))
    for n in (10, 11):
        (0.05 + 0.02 * rng.uniform(0, 1, (h, w))).astype(np.float32).tofile(str(out / "tmp" / ("edges_%d.dat" % n)))
    cfg = tmp_path / "run.cfg"
    cfg.write_text(("file\t%s/f_%%03i.ppm\noutput\t%s/out\nJets\t%d\nstart\t10\nmax_fps\t200\n16bit\t" + ("1" if bits == 16 else "0") + "\nraw\t0\nscale\t1.0\ndeep_matching\t1\ndm_scale\t1.0\nverbose\t00001\n"
                   "slow_flow_S\t%d\nslow_flow_layers\t1\nslow_flow_niter_alter\t1\nslow_flow_niter_outer\t5\nslow_flow_occlusion_reasoning\t0\n"
                   "slow_flow_thres_outer\t0\nslow_flow_thres_inner\t0\nslow_flow_rho_0\t1\nslow_flow_omega_0\t0\ngpus\t1\n") % (tmp_path, tmp_path, jets, S))
    r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg), "-overwrite"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    u, v = read_flo(str(out / "f_010.flo"))
    inner = (slice(10, h - 10), slice(14, w - 14))
    assert abs(np.median(u[inner]) - DX) < 0.3 and abs(np.median(v[inner]) - DY) < 0.3
    ub, vb = read_flo(str(out / "f_011_back.flo"))
    assert abs(np.median(ub[inner]) + DX) < 0.3 and abs(np.median(vb[inner]) + DY) < 0.3
    assert open(str(out / "tmp" / "frame_10_INIT.png"), "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    # without the initialisation the same schedule stays near zero: it IS the initial flow that carries the motion
    cfg0 = tmp_path / "run0.cfg"
    cfg0.write_text(cfg.read_text().replace("deep_matching\t1", "deep_matching\t0").replace("%s/out" % tmp_path, "%s/out0" % tmp_path))
    r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg0), "-overwrite"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0
    u0, _ = read_flo(str(tmp_path / "out0" / "f_010.flo"))
    assert abs(np.median(u0[inner]) - DX) > 3.0
    # the forward window through the binding from the same initial flow
    exe = _link_host_test(tmp_path, ["epic_tool.cpp"], "epic_tool")
    st = sfa.stride_of(w)
    rgb = np.zeros((3, h, st), np.float32)
    rgb[:, :, :w] = frames[1] if bits == 8 else np.clip(np.rint(frames[1] * np.float32(1.0 / 255)), 0, 255)      # un_seq: the 8-bit copy
    rgb.tofile(str(tmp_path / "epic_rgb.bin"))
    import shutil
    shutil.copy(str(out / "tmp" / "matches_10_11.dat"), str(tmp_path / "epic_matches.txt"))
    shutil.copy(str(out / "tmp" / "edges_10.dat"), str(tmp_path / "epic_edges.bin"))
    r = subprocess.run([exe, str(tmp_path), str(w), str(h), "LA", "0.045", "25", "5.0", "160", "1.1", "0.001", "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    ix = np.fromfile(str(tmp_path / "epic_fx.bin"), dtype=np.float32).reshape(h, st) * np.float32(1.0 / steps)
    iy = np.fromfile(str(tmp_path / "epic_fy.bin"), dtype=np.float32).reshape(h, st) * np.float32(1.0 / steps)
    ctx = sfa.Context(0)
    fr = []
    for f in frames:
        a = np.zeros((3, h, st), np.float32); a[:, :, :w] = f
        fr.append(a)
    avg, std = ctx.normalize(fr, w)
    p = sfa.default_params()
    p.S = S; p.layers = 1; p.niter_alter = 1; p.niter_outer = 5; p.occlusion_reasoning = 0; p.thres_outer = 0; p.thres_inner = 0; p.hbit = 1 if bits == 16 else 0; p.smoothing = 1
    p.rho[0] = 1; p.omega[0] = 0
    for k in range(3):
        p.norm_avg[k] = float("%g" % avg[k]); p.norm_std[k] = float("%g" % std[k])
    wx, wy = np.ascontiguousarray(ix), np.ascontiguousarray(iy)
    ctx.variational(p, wx, wy, fr[0:3], w)
    assert np.array_equal(u, wx[:, :w] * steps) and np.array_equal(v, wy[:, :w] * steps)
    ctx.close()


@pytest.mark.gpu
def test_driver_deep_matching_at_half_resolution(host_build, tmp_path):
    """dm_scale 0.5 (slow_flow.cpp:396-402, 571-586, 801-843): matches and edge maps belong to frames blurred with sigma = 1 / sqrt(2 dm_scale) and resized by dm_scale; the
    driver interpolates at that resolution from the same 8-bit copy (blur, resize, convertTo), resizes the field to the frames and multiplies it by the integer width
    ratio (:827-828).  The .flo equals the binding run from an initial flow built from the library's own operators in that order; the motion is recovered; a dm_scale
    whose integer ratio is 1 although the sizes differ (0.75) is refused by name (the reference would pass a field of the reduced size on); max_flow > 150 halves dm_scale
    (:397-402): the same files serve `dm_scale 1.0` + `max_flow 200`.  OpenCV is not vendored: the blur / resize arithmetic is the pyramid's (parity unpinned there too)."""
    import shutil
    import slowflow_amd as sfa
    from synth import texture_frame
    w, h, jets, S = 128, 96, 1, 2
    steps = S - 1
    ew, eh = w // 2, h // 2
    nframes = 1 + (jets + 2) * steps
    DX, DY = 9.0, -6.0
    frames = [np.clip(np.round(texture_frame(w, h, k, dx=DX, dy=DY)[:, :, :w]), 0, 255) for k in range(nframes)]
    for k, f in enumerate(frames):
        write_ppm(str(tmp_path / ("f_%03d.ppm" % (10 - steps + k))), f)
    out = tmp_path / "out"
    (out / "tmp").mkdir(parents=True)
    rng = np.random.default_rng(0)
    for a, b, sx, sy in ((10, 11, DX, DY), (11, 10, -DX, -DY)):                           # matches in the coordinates of the half-size frames
        with open(str(out / "tmp" / ("matches_%d_%d.dat" % (a, b))), "w") as f:
            for _ in range(500):
                x, y = rng.uniform(6, ew - 7), rng.uniform(4, eh - 5)
                f.write("%.3f %.3f %.3f %.3f 3.1 1\n" % (x, y, x + sx / 2 + rng.normal(0, 0.1), y + sy / 2 + rng.normal(0, 0.1)))
    for n in (10, 11):
        (0.05 + 0.02 * rng.uniform(0, 1, (eh, ew))).astype(np.float32).tofile(str(out / "tmp" / ("edges_%d.dat" % n)))
    cfg = tmp_path / "run.cfg"
    cfg.write_text(("file\t%s/f_%%03i.ppm\noutput\t%s/out\nJets\t%d\nstart\t10\nmax_fps\t200\n16bit\t0\nraw\t0\nscale\t1.0\ndeep_matching\t1\ndm_scale\t0.5\nverbose\t00001\n"
                    "slow_flow_S\t%d\nslow_flow_layers\t1\nslow_flow_niter_alter\t1\nslow_flow_niter_outer\t5\nslow_flow_occlusion_reasoning\t0\n"
                    "slow_flow_thres_outer\t0\nslow_flow_thres_inner\t0\nslow_flow_rho_0\t1\nslow_flow_omega_0\t0\ngpus\t1\n") % (tmp_path, tmp_path, jets, S))
    r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg), "-overwrite"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    u, v = read_flo(str(out / "f_010.flo"))
    inner = (slice(10, h - 10), slice(14, w - 14))
    assert abs(np.median(u[inner]) - DX) < 0.4 and abs(np.median(v[inner]) - DY) < 0.4
    # the same initial flow from the library's operators, then the binding
    ctx = sfa.Context(0)
    st, est = sfa.stride_of(w), sfa.stride_of(ew)
    sigma = np.float32(1 / np.sqrt(2 * 0.5))
    rgb = np.zeros((3, eh, est), np.float32)
    for ch in range(3):
        pl = np.zeros((h, st), np.float32); pl[:, :w] = frames[1][ch]
        small, dw = ctx.resize_linear_fx(ctx.gaussian_blur(pl, w, float(sigma)), w, 0.5, 0.5)
        assert dw == ew and small.shape == (eh, est)
        rgb[ch, :, :ew] = np.clip(np.rint(small[:, :ew]), 0, 255)                           # img.convertTo(CV_8U)
    exe = _link_host_test(tmp_path, ["epic_tool.cpp"], "epic_tool")
    rgb.tofile(str(tmp_path / "epic_rgb.bin"))
    shutil.copy(str(out / "tmp" / "matches_10_11.dat"), str(tmp_path / "epic_matches.txt"))
    shutil.copy(str(out / "tmp" / "edges_10.dat"), str(tmp_path / "epic_edges.bin"))
    r = subprocess.run([exe, str(tmp_path), str(ew), str(eh), "LA", "0.045", "25", "5.0", "160", "1.1", "0.001", "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    ex = np.fromfile(str(tmp_path / "epic_fx.bin"), dtype=np.float32).reshape(eh, est)
    ey = np.fromfile(str(tmp_path / "epic_fy.bin"), dtype=np.float32).reshape(eh, est)
    fac = np.float32(np.float32(w // ew) / np.float32(steps))                                # fx / steps: one float product (image_mul_scalar)
    ix = ctx.resize_linear(np.ascontiguousarray(ex), ew, w, h) * fac
    iy = ctx.resize_linear(np.ascontiguousarray(ey), ew, w, h) * fac
    fr = []
    for f in frames:
        a = np.zeros((3, h, st), np.float32); a[:, :, :w] = f
        fr.append(a)
    avg, std = ctx.normalize(fr, w)
    p = sfa.default_params()
    p.S = S; p.layers = 1; p.niter_alter = 1; p.niter_outer = 5; p.occlusion_reasoning = 0; p.thres_outer = 0; p.thres_inner = 0; p.hbit = 0; p.smoothing = 1
    p.rho[0] = 1; p.omega[0] = 0
    for k in range(3):
        p.norm_avg[k] = float("%g" % avg[k]); p.norm_std[k] = float("%g" % std[k])
    wx, wy = np.ascontiguousarray(ix), np.ascontiguousarray(iy)
    ctx.variational(p, wx, wy, fr[0:3], w)
    assert np.array_equal(u, wx[:, :w] * steps) and np.array_equal(v, wy[:, :w] * steps)
    ctx.close()
    # max_flow > 150 halves dm_scale: the same half-size files serve dm_scale 1.0
    cfg2 = tmp_path / "run2.cfg"
    cfg2.write_text(cfg.read_text().replace("dm_scale\t0.5", "dm_scale\t1.0\nmax_flow\t200"))
    r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg2), "-overwrite"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    u2, v2 = read_flo(str(out / "f_010.flo"))
    assert np.array_equal(u2, u) and np.array_equal(v2, v)
    # 1 < width ratio < 2: refused by name
    cfg3 = tmp_path / "run3.cfg"
    cfg3.write_text(cfg.read_text().replace("dm_scale\t0.5", "dm_scale\t0.75"))
    for n in (10, 11):
        np.zeros((72, 96), np.float32).tofile(str(out / "tmp" / ("edges_%d.dat" % n)))
    r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg3), "-overwrite"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "dm_scale" in r.stderr and "integer ratio" in r.stderr
    # a size whose rounded and truncated products differ (96 * 0.3 = 28.8: cv::resize makes 29 rows, slow_flow.cpp:584 allocates 28): the reference has no defined
    # result there -- refused by name
    cfg4 = tmp_path / "run4.cfg"
    cfg4.write_text(cfg.read_text().replace("dm_scale\t0.5", "dm_scale\t0.3"))
    r = subprocess.run([os.path.join(HOST, "slow_flow"), str(cfg4), "-overwrite"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "dm_scale" in r.stderr and "truncates" in r.stderr


@pytest.mark.gpu
def test_release_build_runs_the_path(tmp_path):
    """the release build of the library (-DSFA_RELEASE, tests/test_abi.py) on the GPU, each build in a process of its own (tools/release_report.py): the smoke parity
    (SOR bit-identical, a two-level refinement within 1e-4 of the oracle) through it, the default solver shapes at 1 and 16 systems of 1024x436 are the full build's,
    a sweep count that no chain shape divides (K = 7) takes the one fallback kernel each build has and is bit-identical to the oracle in both"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "release_report.py")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = json.loads(r.stdout[r.stdout.index("["):])
    full, rel = rows
    assert "error" not in full and "error" not in rel, rows
    assert full["debug_set_rc"] == 0 and rel["debug_set_rc"] != 0
    assert rel["solver_kernels_1_16_K7"][:2] == full["solver_kernels_1_16_K7"][:2] and rel["solver_kernels_1_16_K7"][1].startswith("k_sor_chain<2,6,3,1")
    assert rel["solver_kernels_1_16_K7"][2].startswith("k_sor_solve") and full["solver_kernels_1_16_K7"][2].startswith("k_sor_band")
    assert full["fallback_K7_bit_identical"] and rel["fallback_K7_bit_identical"]
    assert rel["bytes"] < 0.75 * full["bytes"] and (rel["gpu_kernels"] is None or rel["gpu_kernels"] < full["gpu_kernels"])
    print(json.dumps(rows))
