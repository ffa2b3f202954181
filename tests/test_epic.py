"""EpicFlow's sparse-to-dense interpolation (the initialisation of the path with deep_matching 1): slowflow_amd/host/epic.cpp against
the reference's own epic() compiled by oracle/Makefile (oracle/_ref/libslowflow_ref_epic.so; unmodified sources, LAPACK = the OpenBLAS inside the scipy wheel).
The Nadaraya-Watson route (distance transform, neighbourhood graph, graph search, kernel sums) must agree bit for bit; the locally-weighted affine route solves its
least-squares problems with an own routine instead of sgels and agrees to 1e-3 px; the saliency filter runs its filters on the GPU (-m gpu)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "slowflow_amd", "host")
REF_EPIC = os.path.join(ROOT, "oracle", "_ref", "libslowflow_ref_epic.so")
_f = C.POINTER(C.c_float)


class image_t(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("stride", C.c_int), ("data", _f)]


class color_image_t(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("stride", C.c_int), ("c1", _f), ("c2", _f), ("c3", _f)]


@pytest.fixture(scope="module")
def refepic():
    if not os.path.exists(REF_EPIC):
        pytest.skip("oracle/_ref/libslowflow_ref_epic.so not built (needs /root/reference and a LAPACK)")
    try:
        L = C.CDLL(REF_EPIC)
    except OSError as e:
        pytest.skip("reference epic library does not load here: %s" % e)
    L.rgb_to_lab.restype = C.POINTER(color_image_t)
    L.saliency.restype = C.POINTER(image_t)
    L.saliency.argtypes = [C.POINTER(color_image_t), C.c_float, C.c_float]
    return L


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    import slowflow_amd as sfa
    if not os.path.exists(sfa.LIB_PATH):
        sfa.build()
    r = subprocess.run(["make", "-C", HOST], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    exe = str(tmp_path_factory.mktemp("epic") / "epic_tool")
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-pthread", "-I", HOST, os.path.join(ROOT, "tests", "host", "epic_tool.cpp"), os.path.join(HOST, "libslowflow_host.a"),
                        "-L", os.path.join(ROOT, "slowflow_amd"), "-lslowflow_amd", "-lz", "-Wl,-rpath," + os.path.join(ROOT, "slowflow_amd"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def stride_of(w):
    return ((w + 3) // 4) * 4


def scene(w, h, seed, n_matches=400, outliers=20):
    """a textured image, an edge-cost map with two strong edges, and matches of a piecewise-affine motion plus a few outliers"""
    rng = np.random.default_rng(seed)
    st = stride_of(w)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    rgb = np.zeros((3, h, st), np.float32)
    for c in range(3):
        rgb[c, :, :w] = np.clip(127 + 60 * np.sin(0.21 * xx + c) * np.cos(0.17 * yy - c) + 40 * (xx > w * 0.55) + rng.uniform(-8, 8, (h, w)), 0, 255)
    edges = (0.02 + 0.01 * rng.uniform(0, 1, (h, w))).astype(np.float32)
    edges[:, int(w * 0.55)] = 0.9                                     # a vertical motion boundary
    edges[int(h * 0.4), : int(w * 0.55)] = 0.7
    x1 = rng.uniform(0, w - 1, n_matches); y1 = rng.uniform(0, h - 1, n_matches)
    right = x1 > w * 0.55
    u = np.where(right, 3.0 + 0.02 * (x1 - w / 2), -1.5 + 0.01 * y1)
    v = np.where(right, -2.0 + 0.015 * y1, 0.5 - 0.01 * (x1 - 10))
    bad = rng.choice(n_matches, outliers, replace=False)
    u[bad] += rng.uniform(15, 30, outliers); v[bad] -= rng.uniform(15, 30, outliers)
    m = np.stack([x1, y1, x1 + u, y1 + v], 1).astype(np.float32)
    m[:5, 0] = -3.0; m[5:8, 3] = h + 7.0                              # some outside the image: clamped (epic.cpp:15-28)
    return rgb, edges, m


def run_tool(tool, d, rgb, edges, m, w, h, method, sal, pref_nn, nn, gpu=False, coef=0.8, euc=0.001):
    rgb.tofile(os.path.join(d, "epic_rgb.bin"))
    edges.tofile(os.path.join(d, "epic_edges.bin"))
    with open(os.path.join(d, "epic_matches.txt"), "w") as f:
        for r in m:
            f.write("%.9g %.9g %.9g %.9g 1.0 17\n" % tuple(float(x) for x in r))          # DeepMatching writes a score and an index behind the coordinates
    r = subprocess.run([tool, d, str(w), str(h), method, repr(sal), str(pref_nn), "5.0", str(nn), repr(coef), repr(euc)] + (["gpu"] if gpu else []), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    st = stride_of(w)
    rd = lambda n, k=1: np.fromfile(os.path.join(d, n), dtype=np.float32).reshape((k, h, st) if k > 1 else (h, st))
    return rd("epic_fx.bin"), rd("epic_fy.bin"), rd("epic_lab.bin", 3)


def run_ref(L, lab, edges, m, w, h, method, sal, pref_nn, nn, coef=0.8, euc=0.001):
    st = stride_of(w)
    fx, fy = np.zeros((h, st), np.float32), np.zeros((h, st), np.float32)
    lab = np.ascontiguousarray(lab); e = np.ascontiguousarray(edges).copy(); mm = np.ascontiguousarray(m).copy()
    L.ref_epic(fx.ctypes.data_as(_f), fy.ctypes.data_as(_f), w, h, st, lab.ctypes.data_as(_f), mm.ctypes.data_as(_f), len(mm), e.ctypes.data_as(_f), method.encode(),
               C.c_float(sal), pref_nn, C.c_float(5.0), nn, C.c_float(coef), C.c_float(euc))
    return fx, fy


def ref_lab(L, rgb, w, h):
    st = stride_of(w)
    rgb = np.ascontiguousarray(rgb)
    im = color_image_t(w, h, st, rgb[0].ctypes.data_as(_f), rgb[1].ctypes.data_as(_f), rgb[2].ctypes.data_as(_f))
    res = L.rgb_to_lab(C.byref(im)).contents
    return np.stack([np.ctypeslib.as_array(p, shape=(h, st)).copy() for p in (res.c1, res.c2, res.c3)])


@pytest.mark.parametrize("w,h,seed", [(96, 64, 0), (131, 77, 1)])
def test_epic_against_the_compiled_reference(tool, refepic, tmp_path, w, h, seed):
    rgb, edges, m = scene(w, h, seed)
    # Nadaraya-Watson with the consistency filter, no saliency filter (that one needs the GPU): bit for bit
    fx, fy, lab = run_tool(tool, str(tmp_path), rgb, edges, m, w, h, "NW", 0.0, 25, 100)
    rl = ref_lab(refepic, rgb, w, h)
    assert np.array_equal(lab[:, :, :w], rl[:, :, :w])                                    # rgb_to_lab (image.c:694-726)
    rx, ry = run_ref(refepic, rl, edges, m, w, h, "NW", 0.0, 25, 100)
    assert np.array_equal(fx[:, :w], rx[:, :w]) and np.array_equal(fy[:, :w], ry[:, :w])
    assert len(np.unique(fx[:, :w])) > 50                                                # many seeds, many regions
    # no filter at all, fewer neighbours
    fx2, fy2, _ = run_tool(tool, str(tmp_path), rgb, edges, m, w, h, "NW", 0.0, 0, 30)
    rx2, ry2 = run_ref(refepic, rl, edges, m, w, h, "NW", 0.0, 0, 30)
    assert np.array_equal(fx2[:, :w], rx2[:, :w]) and np.array_equal(fy2[:, :w], ry2[:, :w])
    assert not np.array_equal(fx2, fx)                                                   # the outliers were filtered in the first run
    # locally-weighted affine (the default): own least squares vs LAPACK sgels
    fa, fb, _ = run_tool(tool, str(tmp_path), rgb, edges, m, w, h, "LA", 0.0, 25, 100)
    ra, rb = run_ref(refepic, rl, edges, m, w, h, "LA", 0.0, 25, 100)
    assert max(np.abs(fa[:, :w] - ra[:, :w]).max(), np.abs(fb[:, :w] - rb[:, :w]).max()) <= 1e-3
    # and it is an interpolation of the motion: right of the boundary (u, v) ~ (3 + .02 (x - w/2), -2 + .015 y), outliers removed
    yy, xx = np.mgrid[0:h, 0:w]
    right = xx > w * 0.55 + 3
    assert np.abs(fa[:, :w][right] - (3.0 + 0.02 * (xx[right] - w / 2))).mean() < 0.6          # (the region is small and edge-aware weights extrapolate)
    assert np.abs(fb[:, :w][right] - (-2.0 + 0.015 * yy[right])).mean() < 0.6
    left = xx < w * 0.55 - 3
    assert np.abs(fa[:, :w][left] - (-1.5 + 0.01 * yy[left])).mean() < 0.6                      # and the other side of the boundary keeps ITS motion


def test_local_affine_reproduces_an_affine_field(tool, tmp_path):
    """matches of an exactly affine motion: the locally-weighted affine interpolation returns that motion at every pixel (no reference needed)"""
    w, h = 80, 60
    rgb, edges, _ = scene(w, h, 3)
    rng = np.random.default_rng(3)
    x1 = rng.uniform(0, w - 6, 300).astype(np.float32); y1 = rng.uniform(1, h - 4, 300).astype(np.float32)   # targets stay inside the image: no clamping (epic.cpp:15-28)
    x1, y1 = np.floor(x1), np.floor(y1)                               # seeds are the integer parts (epic.cpp:31-42)
    u, v = 1.0 + 0.03 * x1 - 0.01 * y1, -0.5 + 0.02 * x1 + 0.015 * y1
    m = np.stack([x1, y1, x1 + u, y1 + v], 1).astype(np.float32)
    fx, fy, _ = run_tool(tool, str(tmp_path), rgb, edges, m, w, h, "LA", 0.0, 0, 50)
    yy, xx = np.mgrid[0:h, 0:w]
    assert np.abs(fx[:, :w] - (1.0 + 0.03 * xx - 0.01 * yy)).max() < 2e-3 and np.abs(fy[:, :w] - (-0.5 + 0.02 * xx + 0.015 * yy)).max() < 2e-3


@pytest.mark.gpu
def test_epic_with_saliency_filter_gpu(tool, refepic, tmp_path):
    """the default parameter set (saliency 0.045, 25-neighbour consistency, 100 neighbours): the saliency map -- Gaussian smoothing and derivative filters on the
    GPU through the path's own operators -- equals the reference's bit for bit, and with it the whole interpolation"""
    w, h = 131, 77
    rgb, edges, m = scene(w, h, 5, n_matches=600)
    rgb[:, :30, :40] = 128.0                                          # a textureless corner: matches from there are dropped by the saliency filter
    fx, fy, lab = run_tool(tool, str(tmp_path), rgb, edges, m, w, h, "NW", 0.045, 25, 100, gpu=True)
    st = stride_of(w)
    sal = np.fromfile(str(tmp_path / "epic_sal.bin"), dtype=np.float32).reshape(h, st)
    rl = ref_lab(refepic, rgb, w, h)
    im = color_image_t(w, h, st, rl[0].ctypes.data_as(_f), rl[1].ctypes.data_as(_f), rl[2].ctypes.data_as(_f))
    rs = refepic.saliency(C.byref(im), 0.8, 1.0).contents
    rsal = np.ctypeslib.as_array(rs.data, shape=(h, st)).copy()
    assert np.array_equal(sal[:, :w], rsal[:, :w])
    assert (sal[:25, :35] < 0.045).all() and (sal[:, :w] >= 0.045).mean() > 0.3    # the threshold does select
    rx, ry = run_ref(refepic, rl, edges, m, w, h, "NW", 0.045, 25, 100)
    assert np.array_equal(fx[:, :w], rx[:, :w]) and np.array_equal(fy[:, :w], ry[:, :w])
    fa, fb, _ = run_tool(tool, str(tmp_path), rgb, edges, m, w, h, "LA", 0.045, 25, 100, gpu=True)
    ra, rb = run_ref(refepic, rl, edges, m, w, h, "LA", 0.045, 25, 100)
    assert max(np.abs(fa[:, :w] - ra[:, :w]).max(), np.abs(fb[:, :w] - rb[:, :w]).max()) <= 1e-3
