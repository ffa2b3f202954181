"""Pins the CPU restatement (oracle/slowflow_oracle.c) against the reference's own compiled C
(oracle/_ref, built by oracle/Makefile from the sources under /root/reference).  Runs only where
that library exists; the committed fixtures in tests/golden cover the same ground elsewhere."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as orc
from synth import copy_sys, noise_color, noise_plane, smooth_noise_color, sor_system

SIZES = [(67, 45), (64, 48), (130, 98), (5, 4), (2, 2), (9, 33)]


def valid(a, w):
    return a[..., :w]


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("order", [1, 2])
def test_convolutions_bit_exact(oracle, reflib, w, h, order):
    if order == 2 and h < 4:
        pytest.skip("5-tap vertical fast path needs h>=4 (image.c:443)")
    rng = np.random.default_rng(w * 1000 + h + order)
    src = noise_plane(rng, w, h, -3, 3)
    for horiz in (True, False):
        a = oracle.convolve(src, w, order, horiz)
        b = reflib.convolve(src, w, order, horiz)
        assert np.array_equal(valid(a, w), valid(b, w)), (order, horiz)


@pytest.mark.parametrize("w,h", [(67, 45), (64, 48), (2, 2), (130, 98), (3, 7)])
@pytest.mark.parametrize("K,omega", [(1, 1.9), (2, 1.9), (30, 1.9), (7, 1.0)])
def test_sor_bit_exact(oracle, reflib, w, h, K, omega):
    rng = np.random.default_rng(w + 31 * h + K)
    sys0 = sor_system(rng, w, h)
    sys0["du"][:, :w] = rng.uniform(-.2, .2, (h, w))
    sys0["dv"][:, :w] = rng.uniform(-.2, .2, (h, w))
    a, b = copy_sys(sys0), copy_sys(sys0)
    oracle.sor(a["du"], a["dv"], a["a11"], a["a12"], a["a22"], a["b1"], a["b2"], a["sh"], a["sv"], w, K, omega)
    reflib.sor(b["du"], b["dv"], b["a11"], b["a12"], b["a22"], b["b1"], b["b2"], b["sh"], b["sv"], w, K, omega)
    for k in ("du", "dv", "a11", "a12", "a22"):
        assert np.array_equal(valid(a[k], w), valid(b[k], w)), k


def test_sor_generic_edge_weights_bit_exact(oracle, reflib):
    """sh[w-1] / sv[h-1] non-zero and non-zero initial guess: the drop-in sor_coupled accepts any planes"""
    w, h = 37, 21
    rng = np.random.default_rng(5)
    sys0 = sor_system(rng, w, h)
    sys0["sh"][:, :] = rng.uniform(0.1, 1, sys0["sh"].shape)
    sys0["sv"][:, :] = rng.uniform(0.1, 1, sys0["sv"].shape)
    sys0["du"][:, :] = rng.uniform(-1, 1, sys0["du"].shape)
    sys0["dv"][:, :] = rng.uniform(-1, 1, sys0["dv"].shape)
    a, b = copy_sys(sys0), copy_sys(sys0)
    oracle.sor(a["du"], a["dv"], a["a11"], a["a12"], a["a22"], a["b1"], a["b2"], a["sh"], a["sv"], w, 5, 1.9)
    reflib.sor(b["du"], b["dv"], b["a11"], b["a12"], b["a22"], b["b1"], b["b2"], b["sh"], b["sv"], w, 5, 1.9)
    for k in ("du", "dv"):
        assert np.array_equal(valid(a[k], w), valid(b[k], w)), k


def test_sor_readable_bit_exact_and_close_to_fast(oracle, reflib):
    w, h = 40, 30
    rng = np.random.default_rng(11)
    sys0 = sor_system(rng, w, h)
    a, b, c = copy_sys(sys0), copy_sys(sys0), copy_sys(sys0)
    oracle.sor(a["du"], a["dv"], a["a11"], a["a12"], a["a22"], a["b1"], a["b2"], a["sh"], a["sv"], w, 30, 1.9, readable=True)
    reflib.sor(b["du"], b["dv"], b["a11"], b["a12"], b["a22"], b["b1"], b["b2"], b["sh"], b["sv"], w, 30, 1.9, readable=True)
    assert np.array_equal(valid(a["du"], w), valid(b["du"], w))
    assert np.array_equal(valid(a["dv"], w), valid(b["dv"], w))
    oracle.sor(c["du"], c["dv"], c["a11"], c["a12"], c["a22"], c["b1"], c["b2"], c["sh"], c["sv"], w, 30, 1.9)
    assert np.max(np.abs(valid(a["du"], w) - valid(c["du"], w))) < 2e-5


@pytest.mark.parametrize("w,h", [(67, 45), (64, 48), (130, 98)])
@pytest.mark.parametrize("factor", [-2, -1, 1, 2, 3])
def test_image_warp_bit_exact(oracle, reflib, w, h, factor):
    rng = np.random.default_rng(w + h + factor + 100)
    src = smooth_noise_color(rng, w, h)
    wx, wy = noise_plane(rng, w, h, -4, 4), noise_plane(rng, w, h, -4, 4)
    # push some samples far outside to exercise clamping and the mask
    wx[0, :5] = 1000; wy[1, :5] = -1000; wx[2, 3] = -0.0
    a, ma = oracle.image_warp(src, wx, wy, w, factor)
    fwx = orc.plane(h, wx.shape[1]); fwy = orc.plane(h, wx.shape[1])
    fwx[...] = np.float32(factor) * wx
    fwy[...] = np.float32(factor) * wy
    b, mb = reflib.image_warp_prescaled(src, fwx, fwy, w)
    assert np.array_equal(valid(a, w), valid(b, w))
    assert np.array_equal(valid(ma, w), valid(mb, w))
    assert 0 < valid(ma, w).mean() < 1


@pytest.mark.parametrize("w,h", [(67, 45), (64, 48)])
def test_sub_laplacian_bit_exact(oracle, reflib, w, h):
    rng = np.random.default_rng(w * h)
    src, wh, wv = noise_plane(rng, w, h), noise_plane(rng, w, h, 0, 2), noise_plane(rng, w, h, 0, 2)
    d0 = noise_plane(rng, w, h)
    a = orc.plane(*d0.shape); a[...] = d0
    b = orc.plane(*d0.shape); b[...] = d0
    oracle.sub_laplacian(a, src, wh, wv, w)
    reflib.sub_laplacian(b, src, wh, wv, w)
    assert np.array_equal(valid(a, w), valid(b, w))


@pytest.mark.parametrize("w,h", [(67, 45), (64, 48)])
def test_dpsis_weight_bit_exact(oracle, reflib, w, h):
    rng = np.random.default_rng(w - h)
    im = smooth_noise_color(rng, w, h)
    a = oracle.dpsis_weight(im, w)
    b = reflib.dpsis_weight(im, w)
    assert np.array_equal(valid(a, w), valid(b, w))
    assert 0 < valid(a, w).min() and valid(a, w).max() <= 0.5


@pytest.mark.parametrize("w,h", [(67, 45), (64, 48)])
def test_derivative_stack_bit_exact(oracle, reflib, w, h):
    """MT stack (variational_mt.cpp:113-133): M=.5*(I2+I1), Iz=I1-I2 == 2-frame get_derivatives(im1=I2, im2=I1)"""
    rng = np.random.default_rng(w + 7 * h)
    I1, I2 = smooth_noise_color(rng, w, h), smooth_noise_color(rng, w, h)
    a = oracle.derivative_stack(I1, I2, w)
    b = reflib.get_derivatives(I2, I1, w)      # dt = im2 - im1 = I1 - I2
    names = ["Ix", "Iy", "Iz", "Ixx", "Ixy", "Iyy", "Ixz", "Iyz"]
    for i, n in enumerate(names):
        assert np.array_equal(valid(a[i], w), valid(b[i], w)), n


@pytest.mark.parametrize("pid", [0, 1, 2, 3, 4, 7])
def test_penalties_bit_exact(oracle, reflib, pid):
    rng = np.random.default_rng(pid)
    x = np.concatenate([rng.uniform(0, 1e-6, 64), rng.uniform(0, 1, 64), rng.uniform(0, 50, 64), [0, 0.25, 0.2499999, 1e-12]]).astype(np.float32)
    for eps, trunc in ((0.001, 0.5), (0.05, 0.5), (0.3, 2.0)):
        xs, s, v = reflib.penalty_derivative(pid, eps, trunc, x)
        pen = orc.Penalty(pid, eps, trunc)
        so, vo = oracle.psi_deriv(pen, xs)
        assert np.array_equal(s, so), (pid, eps, "scalar")
        assert np.array_equal(v, vo), (pid, eps, "vec")


def test_smoothness_method1_near_pinned(oracle, reflib):
    """2-frame compute_smoothness (variational_aux.c:86): same operator with half_alpha/sqrt(x+eps) instead of
    alpha*psi'(x); identical up to rounding order."""
    w, h = 67, 45
    rng = np.random.default_rng(3)
    uu, vv = noise_plane(rng, w, h, -2, 2), noise_plane(rng, w, h, -2, 2)
    dps = noise_plane(rng, w, h, 0.05, 0.5)
    alpha = 4.0
    pen = orc.Penalty(1, 0.001, 0.5)
    sh, sv = oracle.smoothness(1, uu, vv, dps, w, alpha, pen)
    rh, rv = reflib.compute_smoothness(uu, vv, dps, w, alpha / 2)
    for a, b in ((sh, rh), (sv, rv)):
        rel = np.abs(valid(a, w) - valid(b, w)) / np.maximum(np.abs(valid(b, w)), 1e-30)
        assert rel.max() < 4e-7
    assert np.all(sh[:, w - 1:] == 0) and np.all(sv[h - 1] == 0)


def test_data_term_near_pinned(oracle, reflib):
    """2-frame compute_data_and_match (variational_aux.c:226) == MT add_data_and_match with s=-1 (factor -1,
    factor+1 = 0) on the negated temporal derivatives, unit mask/weights, hd = 2*half_delta_over3/... up to
    rounding: MT psi' = 1/(2 sqrt(x+eps)) times hd, 2-frame = half_hd/sqrt(x+eps)."""
    w, h = 67, 45
    rng = np.random.default_rng(9)
    I1, I2 = smooth_noise_color(rng, w, h, 10), smooth_noise_color(rng, w, h, 10)
    D = oracle.derivative_stack(I1, I2, w)
    du, dv = noise_plane(rng, w, h, -.5, .5), noise_plane(rng, w, h, -.5, .5)
    mask = orc.plane(h, du.shape[1], 1.0)
    ones = orc.plane(h, du.shape[1], 1.0)
    delta_over3, gamma_over3 = 1.0 / 3.0, 6.0 / 3.0
    sysm = [orc.plane(h, du.shape[1]) for _ in range(5)]
    pen = orc.Penalty(1, 0.001, 0.5)
    # MT, s = 0: factor 0, factor+1 = 1: r = Iz - Ix*du - Iy*dv, tx = -Ix
    oracle.add_data(sysm, mask, du, dv, D, [ones, ones, ones], w, delta_over3, gamma_over3, 0.0, True, pen, pen)
    # 2-frame operator uses r = Iz' + Ix*du + Iy*dv with Iz' = im2-im1: feed the negated temporal stacks
    Dn = [orc.aligned_zeros(D[i].shape) for i in range(8)]
    for i in range(8):
        Dn[i][...] = D[i]
    for i in (2, 6, 7):
        Dn[i][...] = -D[i]
    ref = reflib.compute_data_and_match(mask, du, dv, Dn, w, delta_over3 / 2, gamma_over3 / 2)
    for a, b, n in zip(sysm, ref, ["a11", "a12", "a22", "b1", "b2"]):
        va, vb = valid(a, w), valid(b, w)
        scale = np.abs(vb).max()
        assert np.max(np.abs(va - vb)) <= 2e-6 * scale, n


def test_expf_restatement():
    """glibc's expf (called by the reference at variational_aux_mt.cpp:700) is not correctly rounded, so the device
    code restates its published algorithm (slowflow_amd/csrc/kernels.hip expf_glibc).  The same restatement in numpy
    must agree with this machine's libm bit for bit."""
    import ctypes
    import struct
    from decimal import Decimal, getcontext
    getcontext().prec = 60
    N = 32
    T = []
    for i in range(N):
        v = float(Decimal(2) ** (Decimal(i) / Decimal(N)))
        T.append((struct.unpack("<Q", struct.pack("<d", v))[0] - (i << 47)) & 0xFFFFFFFFFFFFFFFF)
    T = np.array(T, dtype=np.uint64)
    inv = float.fromhex("0x1.71547652b82fep+0") * N
    C = [float.fromhex("0x1.c6af84b912394p-5") / N / N / N, float.fromhex("0x1.ebfce50fac4f3p-3") / N / N, float.fromhex("0x1.62e42ff0c52d6p-1") / N]
    shift = float.fromhex("0x1.8p+52")
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-8, 0, 150000), rng.uniform(-0.01, 0, 25000), rng.uniform(-80, 5, 25000)]).astype(np.float32)
    z = inv * x.astype(np.float64)
    kd = z + shift
    ki = kd.view(np.uint64)
    kd = kd - shift
    r = z - kd
    t = T[(ki % N).astype(np.int64)] + (ki << np.uint64(47))
    s = t.view(np.float64)
    y = ((C[0] * r + C[1]) * (r * r) + (C[2] * r + 1)) * s
    mine = y.astype(np.float32)
    libm = ctypes.CDLL("libm.so.6")
    libm.expf.restype = ctypes.c_float
    libm.expf.argtypes = [ctypes.c_float]
    ref = np.array([libm.expf(float(v)) for v in x], dtype=np.float32)
    assert (ref != mine).mean() < 1e-5     # 0 here; an FMA build of libm may flip ~1e-9 of them
    # and the kernel's table is this table
    src = open(os.path.join(os.path.dirname(__file__), "..", "slowflow_amd", "csrc", "kernels.hip")).read()
    for v in T:
        assert ("0x%016xull" % int(v)) in src


# ------------------------------------------------------------------------------------------------------
# occlusion step (optimizeOcc): psi itself is pinned; the energies / cut are restated (parity unpinned, oracle header)
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pid,eps,trunc", [(0, .05, .5), (1, .001, .5), (2, .05, .5), (3, .001, .5), (3, .05, .02), (4, .05, .5)])
def test_psi_apply_pinned(oracle, reflib, pid, eps, trunc):
    rng = np.random.default_rng(pid)
    x = np.concatenate([rng.uniform(0, 1e-4, 2000), rng.uniform(0, 2, 2000), rng.uniform(0, 1e4, 2000), [0, 0, 0, 0]]).astype(np.float32)
    xx, v = reflib.penalty_apply(pid, eps, trunc, x)
    assert np.array_equal(v, oracle.psi_apply(orc.Penalty(pid, eps, trunc), xx))


def test_grid_cut_is_the_exact_minimum(oracle):
    """the oracle's Dinic cut against exhaustive search on 4x3 grids"""
    rng = np.random.default_rng(0)
    w, h = 4, 3
    st = orc.stride_of(w)
    for trial in range(40):
        d0, d1 = orc.plane(h, st), orc.plane(h, st)
        d0[:, :w] = rng.uniform(0, 2, (h, w)); d1[:, :w] = rng.uniform(0, 2, (h, w))
        alpha = float(rng.uniform(0, 1))
        occ, e = oracle.grid_cut(d0, d1, alpha, w)
        best = min(np.where(lab, d1[:, :w], d0[:, :w]).astype(np.float64).sum() + alpha * ((lab[:, 1:] != lab[:, :-1]).sum() + (lab[1:] != lab[:-1]).sum())
                   for lab in (np.array([(bits >> i) & 1 for i in range(w * h)]).reshape(h, w) for bits in range(1 << (w * h))))
        assert abs(best - e) < 1e-6
        assert abs(oracle.grid_cut_energy(occ, d0, d1, alpha, w) - e) < 1e-12


@pytest.mark.parametrize("w,h", [(67, 45), (64, 48), (130, 98)])
@pytest.mark.parametrize("kw", [dict(), dict(delta=0.5, niter_outer=3, niter_inner=2), dict(alpha=3.0, gamma=0.2, niter_solver=7, sor_omega=1.5)])
def test_two_frame_variational_end_to_end(oracle, reflib, w, h, kw):
    """the original two-frame refinement (variational.c:101): the restatement against the REAL compiled entry point, bit for bit
    on the whole output -- warp, derivative stack, smoothness, data term, laplacian, SOR, updates, all iterations"""
    rng = np.random.default_rng(w + h)
    im1 = smooth_noise_color(rng, w + 8, h + 8, 40)
    a, b = orc.aligned_zeros((3, h, orc.stride_of(w))), orc.aligned_zeros((3, h, orc.stride_of(w)))
    a[:, :, :w] = im1[:, 4:4 + h, 4:4 + w]
    b[:, :, :w] = im1[:, 3:3 + h, 2:2 + w]                       # translated by (2, 1)
    wx0, wy0 = noise_plane(rng, w, h, 1.5, 2.5), noise_plane(rng, w, h, 0.5, 1.5)
    p = orc.params_2f(**kw)
    wxr, wyr = orc.plane(*wx0.shape), orc.plane(*wx0.shape)
    wxr[...] = wx0; wyr[...] = wy0
    wxo, wyo = wxr.copy(), wyr.copy()
    wxo2, wyo2 = orc.plane(*wx0.shape), orc.plane(*wx0.shape)
    wxo2[...] = wx0; wyo2[...] = wy0
    reflib.variational_2frame(wxr, wyr, a, b, w, p)
    oracle.variational_2frame(wxo2, wyo2, a, b, w, p)
    assert np.array_equal(valid(wxr, w), valid(wxo2, w)) and np.array_equal(valid(wyr, w), valid(wyo2, w))
    assert np.abs(valid(wxr, w) - valid(wx0, w)).max() > 1e-3      # it did move


@pytest.mark.parametrize("w,h", [(67, 45), (64, 48), (130, 98)])
@pytest.mark.parametrize("sigma", [0.5, 0.8, 1.0, 2.3])
def test_gaussian_presmooth_bit_exact(oracle, reflib, w, h, sigma):
    """cfg sigma > 0 (variational_mt.cpp:590-597): gaussian_filter + the generic convolve_horiz / convolve_vert of image.c"""
    rng = np.random.default_rng(int(sigma * 10) + w)
    src = noise_plane(rng, w, h, 0, 255)
    assert np.array_equal(valid(oracle.gaussian_presmooth(src, w, sigma), w), valid(reflib.gaussian_presmooth(src, w, sigma), w))
