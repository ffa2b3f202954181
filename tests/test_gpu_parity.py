"""Parity of the HIP path (through the C-ABI) against the oracle on the same seeded inputs, against the committed
golden vectors, and -- at BASELINE.json's full sizes -- through properties.  fp32 everywhere; the bar is
bit-identical (IEEE ==) for every stage except where noted, and <= 1e-4 max-abs on (u,v) for the whole path
(north_star).  Needs a real MI355X."""
import os

import numpy as np
import pytest

import oracle as orc
import slowflow_amd as sfa
from synth import copy_sys, noise_color, noise_plane, smooth_noise_color, sor_system, texture_frame

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "ref_vectors.npz"))
TOL_UV = 1e-4          # north_star: (u,v) within 1e-4 max-abs of the CPU reference


@pytest.fixture(scope="module")
def ctx():
    c = sfa.Context(0)
    yield c
    c.close()


def valid(a, w):
    return a[..., :w]


def c_(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def mk_params(o, **kw):
    """oracle params and the layout-identical product params"""
    po = o.default_params()
    ps = sfa.default_params()
    for p in (po, ps):
        p.niter_alter = 1; p.niter_outer = 3; p.occlusion_reasoning = 0; p.thres_outer = 0; p.thres_inner = 0; p.hbit = 0
        for k, v in kw.items():
            if k in ("rho", "omega"):
                for i, x in enumerate(v):
                    getattr(p, k)[i] = x
            elif k in ("robust_color", "robust_grad", "robust_reg"):
                getattr(p, k).id, getattr(p, k).eps, getattr(p, k).trunc = v
            elif k in ("norm_avg", "norm_std"):
                for i, x in enumerate(v):
                    getattr(p, k)[i] = x
            else:
                setattr(p, k, v)
    return po, ps


# ------------------------------------------------------------------------------------------------------
# stages
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h", [(67, 45), (64, 48), (130, 98), (5, 4)])
def test_convolve(ctx, oracle, w, h):
    rng = np.random.default_rng(w * h)
    src = noise_plane(rng, w, h, -3, 3)
    for order in (1, 2):
        for horiz in (True, False):
            assert np.array_equal(valid(ctx.convolve(c_(src), w, order, horiz), w), valid(oracle.convolve(src, w, order, horiz), w))


def test_convolve_golden(ctx):
    for (w, h) in ((67, 45), (64, 48)):
        src = c_(G[f"conv_{w}x{h}_src"])
        for order in (1, 2):
            for horiz in (0, 1):
                assert np.array_equal(valid(ctx.convolve(src, w, order, horiz), w), G[f"conv_{w}x{h}_o{order}_h{horiz}"][:, :w])


@pytest.mark.parametrize("factor", [-2, -1, 0, 1, 2, 3])
def test_image_warp(ctx, oracle, factor):
    w, h = 130, 98
    rng = np.random.default_rng(factor + 10)
    src = smooth_noise_color(rng, w, h)
    wx, wy = noise_plane(rng, w, h, -4, 4), noise_plane(rng, w, h, -4, 4)
    wx[0, :5] = 1000; wy[1, :5] = -1000
    a, ma = oracle.image_warp(src, wx, wy, w, factor)
    b, mb = ctx.image_warp(c_(src), c_(wx), c_(wy), w, factor)
    assert np.array_equal(valid(a, w), valid(b, w))
    if factor != 0:
        assert np.array_equal(valid(ma, w), valid(mb, w))


def test_image_warp_with_flows_beyond_the_int_range(ctx, oracle):
    """round 6 (found by tools/fuzz_parity.py as a GPU memory fault): a flow value whose target column leaves the int range -- a refinement that diverges -- or is not a
    number.  The reference's x86 build converts such a coordinate to INT_MIN (`int x = floor(xx)`, variational_aux_mt.cpp:737: cvttsd2si's integer indefinite) and reads
    column / row 0; the GPU's conversion saturates to INT_MAX, `x + 1` overflowed, and the compiler's clamp let INT_MIN through as an index.  Now the kernels convert like
    the reference: every pixel, the poisoned ones included, equals the oracle's bit for bit (NaN == NaN), in the stage kernel and inside a level (k_warp_smooth, k_warp_jobs)"""
    w, h = 130, 98
    rng = np.random.default_rng(77)
    src = smooth_noise_color(rng, w, h)
    wx, wy = noise_plane(rng, w, h, -4, 4), noise_plane(rng, w, h, -4, 4)
    bad = [3e9, -3e9, 1e30, -1e30, np.inf, -np.inf, np.nan, 2147483520.0, -2147483648.0, 2147483648.0]
    for i, v in enumerate(bad):
        wx[3 + i, 10 + 3 * i] = v
        wy[40 + i, 20 + 5 * i] = v
        wx[70, 5 + i] = v; wy[70, 5 + i] = bad[(i + 3) % len(bad)]
    for factor in (-2, -1, 1, 2):
        a, ma = oracle.image_warp(src, wx, wy, w, factor)
        b, mb = ctx.image_warp(c_(src), c_(wx), c_(wy), w, factor)
        assert np.array_equal(valid(a, w), valid(b, w), equal_nan=True), factor
        assert np.array_equal(valid(ma, w), valid(mb, w)), factor
    # the same field as the initial flow of a level (S = 3: four warps per get_derivatives through k_warp_smooth; niter_inner = 2: through k_warp_jobs): no fault, and
    # the poison spreads over the same pixels as in the oracle
    frames, af, sf = normalized_frames(oracle, w, h, 5, seed=9)
    for inner in (1, 2):
        po, ps = mk_params(oracle, S=3, rho=[1, 0.5], omega=[0.5, 2], norm_avg=af, norm_std=sf, niter_outer=2, niter_inner=inner, niter_solver=5)
        o, g = run_both(ctx, oracle, po, ps, frames, w, h, init=(wx, wy))
        for k in range(2):
            fo, fg = valid(o[k], w), valid(g[k], w)
            assert np.array_equal(np.isfinite(fo), np.isfinite(fg)), (inner, k)
            fin = np.isfinite(fo)
            assert fin.sum() < fin.size                                                                      # (the sweeps carry the poison nearly, or really, everywhere)
            assert fin.sum() == 0 or np.abs(fo[fin] - fg[fin]).max() <= 1e-3 * max(1.0, np.abs(fo[fin]).max()), (inner, k)


def test_image_warp_golden(ctx):
    w = 67
    for factor in (-2, -1, 1, 2):
        d, m = ctx.image_warp(c_(G["warp_src"]), c_(G["warp_wx"]), c_(G["warp_wy"]), w, factor)
        assert np.array_equal(d[..., :w], G[f"warp_f{factor}_dst"][..., :w])
        assert np.array_equal(m[:, :w], G[f"warp_f{factor}_mask"][:, :w])


@pytest.mark.parametrize("w,h", [(67, 45), (64, 48), (35, 21)])
def test_derivative_stack(ctx, oracle, w, h):
    rng = np.random.default_rng(w + h)
    I1, I2 = smooth_noise_color(rng, w, h), smooth_noise_color(rng, w, h)
    assert np.array_equal(valid(ctx.derivative_stack(c_(I1), c_(I2), w), w), valid(oracle.derivative_stack(I1, I2, w), w))


def test_derivative_stack_golden(ctx):
    w2 = 35
    st = ctx.derivative_stack(c_(G["stack_I1"]), c_(G["stack_I2"]), w2)
    assert np.array_equal(st[..., :w2], G["stack_out"][..., :w2])


def test_dpsis_weight(ctx, oracle):
    """expf is glibc's algorithm restated on the device (glibc is not correctly rounded): expect bit equality; a
    different libm variant on the host (FMA ifunc) may flip ~1e-9 of the results by one ulp"""
    w, h = 130, 98
    rng = np.random.default_rng(1)
    im = smooth_noise_color(rng, w, h)
    for (avg, std, hbit) in (((0, 0, 0), (1, 1, 1), 0), ((127.3, 120.1, 99.9), (0.178, 0.21, 0.19), 0), ((3000, 2000, 1000), (40, 30, 50), 1)):
        a = valid(oracle.dpsis_weight(im, w, avg, std, hbit), w)
        b = valid(ctx.dpsis_weight(c_(im), w, avg, std, hbit), w)
        ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
        assert ulp.max() <= 1
        assert (ulp > 0).mean() < 1e-5
    assert np.array_equal(valid(ctx.dpsis_weight(c_(G["dpsis_im"]), 67), 67), G["dpsis_out"][:, :67])


@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("pid", [0, 1, 2, 3, 4])
def test_smoothness(ctx, oracle, method, pid):
    w, h = 67, 45
    rng = np.random.default_rng(method * 10 + pid)
    uu, vv = noise_plane(rng, w, h, -2, 2), noise_plane(rng, w, h, -2, 2)
    dps = noise_plane(rng, w, h, 0.05, 0.5)
    eps = 0.001 if pid in (1, 3) else 0.05
    a = oracle.smoothness(method, uu, vv, dps, w, 4.0, orc.Penalty(pid, eps, 0.5))
    b = ctx.smoothness(method, c_(uu), c_(vv), c_(dps), w, 4.0, sfa.Penalty(pid, eps, 0.5))
    for x, y in zip(a, b):
        assert np.array_equal(valid(x, w), valid(y, w))


def test_sub_laplacian(ctx, oracle):
    w, h = 67, 45
    rng = np.random.default_rng(4)
    src, wh, wv, d0 = noise_plane(rng, w, h), noise_plane(rng, w, h, 0, 2), noise_plane(rng, w, h, 0, 2), noise_plane(rng, w, h)
    a = orc.plane(*d0.shape); a[...] = d0
    oracle.sub_laplacian(a, src, wh, wv, w)
    b = ctx.sub_laplacian(c_(d0).copy(), c_(src), c_(wh), c_(wv), w)
    assert np.array_equal(valid(a, w), valid(b, w))
    d = ctx.sub_laplacian(c_(G["sublap_dst0"]).copy(), c_(G["sublap_src"]), c_(G["sublap_wh"]), c_(G["sublap_wv"]), 67)
    assert np.array_equal(d[:, :67], G["sublap_dst"][:, :67])


@pytest.mark.parametrize("ref_term", [False, True])
@pytest.mark.parametrize("dt_norm", [0, 1])
@pytest.mark.parametrize("pid", [0, 1, 2, 3, 4])
def test_data_terms(ctx, oracle, ref_term, dt_norm, pid):
    w, h = 67, 45
    rng = np.random.default_rng(pid + 7 * dt_norm + 13 * ref_term)
    I1, I2 = smooth_noise_color(rng, w, h, 10), smooth_noise_color(rng, w, h, 10)
    D = oracle.derivative_stack(I1, I2, w)
    du, dv = noise_plane(rng, w, h, -.5, .5), noise_plane(rng, w, h, -.5, .5)
    mask = noise_plane(rng, w, h, 0, 1)
    mask[:, :w] = (mask[:, :w] > 0.2) * 0.5
    chw = [noise_plane(rng, w, h, 0.5, 1.5) for _ in range(3)]
    eps = 0.001 if pid in (1, 3) else 0.05
    for s in ((-2.0, -1.0, 1.0, 2.0) if ref_term else (-2.0, -1.0, 0.0, 1.0)):
        for hd in (0.0, 1.0 / 3.0):
            sys_o = [noise_plane(rng, w, h) for _ in range(5)]
            sys_g = [c_(x).copy() for x in sys_o]
            rc_o = oracle.add_data(sys_o, mask, du, dv, D, chw, w, hd, 2.0, s, dt_norm, orc.Penalty(pid, eps, 0.5), orc.Penalty(pid, eps, 0.5), ref_term)
            rc_g = ctx.add_data(sys_g, c_(mask), c_(du), c_(dv), c_(D), [c_(x) for x in chw], w, hd, 2.0, s, dt_norm, sfa.Penalty(pid, eps, 0.5),
                                sfa.Penalty(pid, eps, 0.5), ref_term)
            assert rc_g == 0 and (rc_o in (0, None))
            for x, y, n in zip(sys_o, sys_g, ["a11", "a12", "a22", "b1", "b2"]):
                assert np.array_equal(valid(x, w), valid(y, w)), (n, s, hd)


def test_data_term_ref_rejects_reference_frame(ctx, oracle):
    w, h = 16, 8
    z = np.zeros((h, 16), np.float32)
    D = np.zeros((8, 3, h, 16), np.float32)
    rc = ctx.add_data([z.copy() for _ in range(5)], z, z, z, D, None, w, 1.0, 1.0, 0.0, 1, sfa.Penalty(1, .001, .5), sfa.Penalty(1, .001, .5), True)
    assert rc == -4        # SFA_ERR_REF_FRAME: the logic_error of variational_aux_mt.cpp:419


# ------------------------------------------------------------------------------------------------------
# SOR
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h", [(67, 45), (64, 48), (130, 98), (3, 7), (2, 2), (200, 3), (1, 9), (300, 70)])
@pytest.mark.parametrize("K,omega", [(1, 1.9), (2, 1.9), (30, 1.9), (5, 1.0)])
def test_sor_vs_oracle(ctx, oracle, w, h, K, omega):
    rng = np.random.default_rng(w * 7 + h * 3 + K)
    s0 = sor_system(rng, w, h)
    s0["du"][:, :w] = rng.uniform(-.2, .2, (h, w)); s0["dv"][:, :w] = rng.uniform(-.2, .2, (h, w))
    a = copy_sys(s0)
    oracle.sor(a["du"], a["dv"], a["a11"], a["a12"], a["a22"], a["b1"], a["b2"], a["sh"], a["sv"], w, K, omega)
    b = {k: c_(v).copy() for k, v in s0.items()}
    ctx.sor_coupled(b["du"], b["dv"], b["a11"], b["a12"], b["a22"], b["b1"], b["b2"], b["sh"], b["sv"], w, K, omega)
    for k in ("du", "dv"):
        assert np.array_equal(valid(a[k], w), valid(b[k], w)), k
    if w >= 2 and h >= 2:
        for k in ("a11", "a12", "a22"):     # in-place inverted blocks, solver.c:104-106
            assert np.array_equal(valid(a[k], w), valid(b[k], w)), k


def test_sor_generic_planes(ctx, oracle):
    """the drop-in accepts any planes: non-zero last column / row weights, non-zero start"""
    w, h = 37, 21
    rng = np.random.default_rng(5)
    s0 = sor_system(rng, w, h)
    for k in ("sh", "sv", "du", "dv"):
        s0[k][:, :] = rng.uniform(0.1, 1, s0[k].shape)
    a = copy_sys(s0)
    oracle.sor(a["du"], a["dv"], a["a11"], a["a12"], a["a22"], a["b1"], a["b2"], a["sh"], a["sv"], w, 7, 1.9)
    b = {k: c_(v).copy() for k, v in s0.items()}
    ctx.sor_coupled(b["du"], b["dv"], b["a11"], b["a12"], b["a22"], b["b1"], b["b2"], b["sh"], b["sv"], w, 7, 1.9)
    assert np.array_equal(valid(a["du"], w), valid(b["du"], w)) and np.array_equal(valid(a["dv"], w), valid(b["dv"], w))


@pytest.mark.parametrize("w,h", [(67, 45), (64, 48)])
def test_sor_golden(ctx, w, h):
    for K in (1, 2, 30):
        s = {k: c_(G[f"sor_{w}x{h}_in_{k}"]).copy() for k in ("du", "dv", "a11", "a12", "a22", "b1", "b2", "sh", "sv")}
        ctx.sor_coupled(s["du"], s["dv"], s["a11"], s["a12"], s["a22"], s["b1"], s["b2"], s["sh"], s["sv"], w, K, 1.9)
        assert np.array_equal(s["du"][:, :w], G[f"sor_{w}x{h}_K{K}_du"][:, :w])
        assert np.array_equal(s["dv"][:, :w], G[f"sor_{w}x{h}_K{K}_dv"][:, :w])


@pytest.mark.parametrize("nb,shape", [(1, "k_sor_chain<1,5,1,0,4,4,2,1,1"), (4, "k_sor_chain<1,5,1,0,4,4,2,1,1"), (8, "k_sor_chain<1,5,1,0,4,4,2,2,2"), (9, "k_sor_chain<1,5,1,0,4,4,2,2,2"), (10, "k_sor_chain<2,6,3,1,4,2,2,1,1"),
                                      (16, "k_sor_chain<2,6,3,1"), (64, "k_sor_chain<2,6,3,1"), (128, "k_sor_chain<2,6,3,1")])
def test_default_solver_shape_and_its_bits(ctx, oracle, nb, shape):
    """what the library launches by default at 1024x436 x 30 (8 bands per system) for 1 ... 64 systems per launch -- the chain kernel with the operand ring at every
    batch size: five stages of one sweep up to 72 bands per launch (with one-interval poll / publication lags up to 32 bands), seven stages of 2,2,2,2,2,2,3 sweeps
    on nine waves above (round 5; six stages of 3,3,3,2,2,2 in round 4) -- and that the first and the last system of the launch are the raster-order oracle's bits.
    128 systems = the launch the bench times at level 0 (2 048 workgroups, both words of the window mask): also the two systems around the word boundary"""
    w, h, K = 1024, 436, 30
    rng = np.random.default_rng(100 + nb)
    systems = [sor_system(rng, w, h) for _ in range(2)]
    sb = sfa.SorBatch(ctx, w, h, nb)
    for b in range(nb):
        sb.upload(b, *[c_(systems[b % 2][k]) for k in ("du", "dv", "a11", "a12", "a22", "b1", "b2", "sh", "sv")])
    ctx.profile_enable(True)
    sb.run(K, 1.9)
    kernel = ctx.profile_read_kernels()[3]
    ctx.profile_enable(False)
    assert kernel.startswith(shape), kernel
    for b in sorted({0, nb - 1} | ({63, 64} if nb > 64 else set())):
        a = copy_sys(systems[b % 2])
        oracle.sor(a["du"], a["dv"], a["a11"], a["a12"], a["a22"], a["b1"], a["b2"], a["sh"], a["sv"], w, K, 1.9)
        du, dv = sb.download(b)
        assert np.array_equal(valid(a["du"], w), valid(du, w)) and np.array_equal(valid(a["dv"], w), valid(dv, w)), b
    sb.close()


def test_sor_full_size_and_batch(ctx, oracle):
    """1024x436, K=30 (the metric's configuration): bit-identical to the raster-order oracle, for every element of a
    batch of different systems solved by one launch; repeated runs are deterministic."""
    w, h, K, nb = 1024, 436, 30, 3
    rng = np.random.default_rng(42)
    systems = [sor_system(rng, w, h) for _ in range(nb)]
    sb = sfa.SorBatch(ctx, w, h, nb)
    for b, s in enumerate(systems):
        sb.upload(b, *[c_(s[k]) for k in ("du", "dv", "a11", "a12", "a22", "b1", "b2", "sh", "sv")])
    sb.run(K, 1.9)
    outs = [sb.download(b) for b in range(nb)]
    for b, s in enumerate(systems):
        a = copy_sys(s)
        oracle.sor(a["du"], a["dv"], a["a11"], a["a12"], a["a22"], a["b1"], a["b2"], a["sh"], a["sv"], w, K, 1.9)
        assert np.array_equal(valid(a["du"], w), valid(outs[b][0], w))
        assert np.array_equal(valid(a["dv"], w), valid(outs[b][1], w))
    # the batch run inverted a11.. in place on the device: re-upload and run again -> same answer
    for b, s in enumerate(systems):
        sb.upload(b, *[c_(s[k]) for k in ("du", "dv", "a11", "a12", "a22", "b1", "b2", "sh", "sv")])
    sb.run(K, 1.9)
    for b in range(nb):
        du, dv = sb.download(b)
        assert np.array_equal(du, outs[b][0]) and np.array_equal(dv, outs[b][1])
    sb.close()


SOR_VARIANTS = {"task_f1": {"SFA_SOR_BAND": "0", "SFA_SOR_F": "1", "SFA_SOR_CH": "8"}, "task_f1_ch16": {"SFA_SOR_BAND": "0", "SFA_SOR_F": "1", "SFA_SOR_CH": "16"}, "task_f2": {"SFA_SOR_BAND": "0", "SFA_SOR_F": "2"},
                "task_f3": {"SFA_SOR_BAND": "0", "SFA_SOR_F": "3", "SFA_SOR_CH": "4"},
                "band_f1": {"SFA_SOR_BAND": "1"}, "band_f2": {"SFA_SOR_BAND": "2"}, "band_f3": {"SFA_SOR_BAND": "3"},
                "band_f5": {"SFA_SOR_BAND": "5"}, "band_f6": {"SFA_SOR_BAND": "6"}, "band_mixed_4x6_3x2": {"SFA_SOR_BAND": "43"},
                # sor_chain.hip (few windows per launch): groups of stages per workgroup + I/O wave; a shape whose sweeps per group do not divide K
                # falls back to the kernels above
                "chain_1x3": {"SFA_SOR_CHAIN": "1"}, "chain_2x3": {"SFA_SOR_CHAIN": "2"}, "chain_3x5": {"SFA_SOR_CHAIN": "3"}, "chain_2x5": {"SFA_SOR_CHAIN": "5"},
                "chain_1x5": {"SFA_SOR_CHAIN": "6"}, "chain_3x2": {"SFA_SOR_CHAIN": "8"}, "chain_5x6": {"SFA_SOR_CHAIN": "9"}, "chain_3x10": {"SFA_SOR_CHAIN": "10"},
                # six stages of mixed width (15 sweeps per group: K = 15, 30; other K fall back): the operand ring at its minimum depth
                "chain_3x3_2x3": {"SFA_SOR_CHAIN": "11"}, "chain_2x3_3x3": {"SFA_SOR_CHAIN": "12"},
                # round 5: seven stages of 3,2,2,2,2,2,2 sweeps on nine waves (at most 4 sweeps on a SIMD), the operand ring at its tight depth of 51 rows
                "chain_3x1_2x6": {"SFA_SOR_CHAIN": "13"},
                # 1 x 5 with one-interval poll / publication lags (the lone-solve default; 11 runs with them too)
                "chain_1x5_lags1": {"SFA_SOR_CHAIN": "16"},
                # round 5: the one-sweep shapes take their operand rows from a FILL wave (LDS-DMA) and the first stage reads the ring too; 1 x 3 and 1 x 6 with the lone solve's lags
                "chain_1x3_lags1": {"SFA_SOR_CHAIN": "17"}, "chain_1x6_lags1": {"SFA_SOR_CHAIN": "19"},
                # round 5: seven stages of 2,2,2,2,2,2,3 sweeps on nine waves, the last stage beside the I/O waves (the default from 73 bands on)
                "chain_2x6_3x1": {"SFA_SOR_CHAIN": "14"}}


@pytest.mark.parametrize("variant", sorted(SOR_VARIANTS))
@pytest.mark.parametrize("w,h,K", [(67, 45, 6), (130, 98, 12), (300, 70, 30), (64, 200, 6), (1024, 436, 30), (2, 2, 6), (700, 5, 30), (200, 150, 10),
                                   (150, 130, 7), (90, 140, 15), (129, 65, 16), (100, 100, 1), (257, 33, 31)])
def test_sor_kernel_variants(ctx, oracle, switches, variant, w, h, K):
    """every solver kernel (task pipeline with 1/2/3 fused sweeps per wave, band pipeline with 1/2/3/5/6: a shape that does not divide K falls back
    to the next that does) gives the
    raster-order result bit for bit, for each element of a batch of two different systems"""
    for k in ("SFA_SOR_BAND", "SFA_SOR_F", "SFA_SOR_CH", "SFA_SOR_CHAIN"):
        switches.unset(k)
    if not variant.startswith("chain"):
        switches.set("SFA_SOR_CHAIN", "0")
    for k, v in SOR_VARIANTS[variant].items():
        switches.set(k, v)
    rng = np.random.default_rng(w + 3 * h + K)
    systems = [sor_system(rng, w, h) for _ in range(2)]
    for s in systems:
        s["du"][:, :w] = rng.uniform(-.2, .2, (h, w)); s["dv"][:, :w] = rng.uniform(-.2, .2, (h, w))
    sb = sfa.SorBatch(ctx, w, h, 2)
    for b, s in enumerate(systems):
        sb.upload(b, *[c_(s[k]) for k in ("du", "dv", "a11", "a12", "a22", "b1", "b2", "sh", "sv")])
    sb.run(K, 1.9)
    for b, s in enumerate(systems):
        a = copy_sys(s)
        oracle.sor(a["du"], a["dv"], a["a11"], a["a12"], a["a22"], a["b1"], a["b2"], a["sh"], a["sv"], w, K, 1.9)
        du, dv = sb.download(b)
        assert np.array_equal(valid(a["du"], w), valid(du, w)) and np.array_equal(valid(a["dv"], w), valid(dv, w))
    sb.close()


def test_sor_fixed_point_property(ctx):
    """size-independent property: started from the converged solution one more sweep changes nothing beyond rounding,
    and the residual of the original system is small"""
    w, h = 1024, 436
    rng = np.random.default_rng(3)
    s = sor_system(rng, w, h)
    b = {k: c_(v).copy() for k, v in s.items()}
    ctx.sor_coupled(b["du"], b["dv"], b["a11"], b["a12"], b["a22"], b["b1"], b["b2"], b["sh"], b["sv"], w, 400, 1.0)
    du, dv = b["du"].copy(), b["dv"].copy()
    b2 = {k: c_(v).copy() for k, v in s.items()}
    b2["du"], b2["dv"] = du.copy(), dv.copy()
    ctx.sor_coupled(b2["du"], b2["dv"], b2["a11"], b2["a12"], b2["a22"], b2["b1"], b2["b2"], b2["sh"], b2["sv"], w, 1, 1.0)
    assert np.max(np.abs(b2["du"] - du)) < 5e-6 and np.max(np.abs(b2["dv"] - dv)) < 5e-6


# ------------------------------------------------------------------------------------------------------
# the level and the whole path
# ------------------------------------------------------------------------------------------------------
TOL_LEVEL = 2e-5       # rounding-level stage differences grow to 1e-6..2e-5 through the outer loop


def oracle_sensitivity(oracle, po, frames, w, h, level_only=True):
    """how much the ORACLE's own (u,v) moves when 20 pixels per frame change by one ulp: the conditioning of the
    configuration.  The GPU may differ from the oracle by a small multiple of this and no more."""
    stride = orc.stride_of(w)
    outs = []
    for pert in (False, True):
        fr = []
        rng = np.random.default_rng(1)
        for f in frames:
            g = orc.aligned_zeros(f.shape); g[...] = f
            if pert:
                for i in rng.integers(0, h * w, 20):
                    y, x = divmod(int(i), w)
                    g[0, y, x] = np.nextafter(g[0, y, x], np.float32(1e9))
            fr.append(g)
        wx, wy = orc.plane(h, stride), orc.plane(h, stride)
        if level_only:
            oracle.compute_one_level(po, wx, wy, fr, w)
        else:
            oracle.variational(po, wx, wy, fr, w)
        outs.append((wx, wy))
    return max(np.abs(outs[0][0][:, :w] - outs[1][0][:, :w]).max(), np.abs(outs[0][1][:, :w] - outs[1][1][:, :w]).max())


def run_both(ctx, oracle, po, ps, frames, w, h, level_only=True, chw=None, init=None):
    stride = orc.stride_of(w)
    wxo, wyo = orc.plane(h, stride), orc.plane(h, stride)
    if init is not None:
        wxo[...] = init[0]; wyo[...] = init[1]
    wxg, wyg = c_(wxo).copy(), c_(wyo).copy()
    if level_only:
        rc, cho, _ = oracle.compute_one_level(po, wxo, wyo, frames, w, chw)
        chg, _ = ctx.compute_one_level(ps, wxg, wyg, [c_(f) for f in frames], w, [c_(x) for x in chw] if chw else None)
    else:
        rc, cho = oracle.variational(po, wxo, wyo, frames, w, chw)
        chg, _ = ctx.variational(ps, wxg, wyg, [c_(f) for f in frames], w, [c_(x) for x in chw] if chw else None)
    assert rc == 0
    return (wxo, wyo, cho), (wxg, wyg, chg)


def normalized_frames(oracle, w, h, n, seed=None):
    if seed is None:
        frames = [texture_frame(w, h, k) for k in range(n)]
    else:
        rng = np.random.default_rng(seed)
        m = max(8, 2 * n)                                   # margin: frame k is the crop shifted by (2k, k)
        base = smooth_noise_color(rng, w + 2 * m, h + 2 * m, 40)
        frames = []
        for k in range(n):
            f = orc.aligned_zeros((3, h, orc.stride_of(w)))
            # integer-shifted crops: a known translation of (2,1) px per frame
            f[:, :, :w] = base[:, m - k:m - k + h, m - 2 * k:m - 2 * k + w]
            frames.append(f)
    _, _, af, sf = oracle.normalize(frames, w)
    return frames, af, sf


@pytest.mark.parametrize("w,h", [(67, 45), (128, 96)])
def test_level_symmetric_S2(ctx, oracle, w, h):
    frames, af, sf = normalized_frames(oracle, w, h, 3)
    po, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf)
    o, g = run_both(ctx, oracle, po, ps, frames, w, h)
    d = max(np.abs(valid(o[0], w) - valid(g[0], w)).max(), np.abs(valid(o[1], w) - valid(g[1], w)).max())
    assert d <= TOL_LEVEL, d
    assert abs(o[2][0] - g[2][0]) < 1e-6 and abs(o[2][1] - g[2][1]) < 1e-6


def test_level_cfg_default_S3(ctx, oracle):
    """cfgs/slow_flow.cfg terms: S=3, rho 1/1, omega 0/2, smoothing 1, normalised data term"""
    w, h = 96, 64
    frames, af, sf = normalized_frames(oracle, w, h, 5)
    po, ps = mk_params(oracle, S=3, rho=[1, 1], omega=[0, 2], norm_avg=af, norm_std=sf, niter_outer=4)
    o, g = run_both(ctx, oracle, po, ps, frames, w, h)
    d = max(np.abs(valid(o[0], w) - valid(g[0], w)).max(), np.abs(valid(o[1], w) - valid(g[1], w)).max())
    assert d <= TOL_LEVEL, d


@pytest.mark.parametrize("kw", [
    dict(one_direction=1), dict(smoothing=0), dict(smoothing=2), dict(dataterm_norm=0),
    dict(robust_color=(2, 0.05, 0.5), robust_grad=(2, 0.05, 0.5), robust_reg=(2, 0.05, 0.5)),
    dict(robust_color=(3, 0.001, 0.5), robust_grad=(3, 0.001, 0.5), robust_reg=(3, 0.001, 0.5)),
    dict(robust_color=(4, 0.05, 0.5), robust_grad=(4, 0.05, 0.5), robust_reg=(4, 0.05, 0.5)),
    dict(robust_color=(0, 0.05, 0.5), robust_grad=(0, 0.05, 0.5), robust_reg=(0, 0.05, 0.5)),
    dict(niter_inner=3), dict(delta=0.0), dict(occlusion_reasoning=1, niter_alter=1), dict(hbit=1), dict(niter_solver=7, sor_omega=1.5),
])
def test_level_variants(ctx, oracle, kw):
    w, h = 67, 45
    frames, af, sf = normalized_frames(oracle, w, h, 5, seed=7)
    po, ps = mk_params(oracle, S=3, rho=[1, 0.5], omega=[0.5, 2], norm_avg=af, norm_std=sf, niter_outer=2, **kw)
    o, g = run_both(ctx, oracle, po, ps, frames, w, h)
    d = max(np.abs(valid(o[0], w) - valid(g[0], w)).max(), np.abs(valid(o[1], w) - valid(g[1], w)).max())
    # ill-conditioned variants (e.g. all-quadratic penalties: the oracle itself moves by 6e-4 under 1-ulp input noise)
    # are held to a small multiple of the oracle's own sensitivity
    tol = max(TOL_LEVEL, 3 * oracle_sensitivity(oracle, po, frames, w, h))
    assert d <= tol, (kw, d, tol)


@pytest.mark.parametrize("kw", [
    dict(), dict(one_direction=1), dict(dataterm_norm=0), dict(delta=0.0), dict(occlusion_reasoning=1, niter_alter=1), dict(niter_inner=3),
    dict(robust_color=(3, 0.001, 0.5), robust_grad=(4, 0.05, 0.5)),
])
@pytest.mark.parametrize("w,h", [(67, 45), (200, 37), (64, 16), (129, 70)])
def test_fused_assembly_is_the_unfused_pipeline(ctx, oracle, switches, kw, w, h):
    """the image->system kernel (derivative filters in LDS, mask weights on the fly) against the materialised form
    (warp copies, 24-plane stacks, mask-weight pass, per-pixel assembly): the same bits, every term kind, with
    channel weights, across tile borders (sizes off the 64x8 tile grid)"""
    frames, af, sf = normalized_frames(oracle, w, h, 5, seed=3)
    rng = np.random.default_rng(1)
    chw = [noise_plane(rng, w, h, 0.5, 1.5) for _ in range(3)]
    _, ps = mk_params(oracle, S=3, rho=[1, 0.5], omega=[0.5, 2], norm_avg=af, norm_std=sf, niter_outer=2, **kw)
    out = []
    for unfused in ("1", "0"):
        switches.set("SFA_UNFUSED", unfused)
        wx, wy = np.zeros((h, sfa.stride_of(w)), np.float32), np.zeros((h, sfa.stride_of(w)), np.float32)
        ch, _ = ctx.compute_one_level(ps, wx, wy, [c_(f) for f in frames], w, [c_(x) for x in chw])
        out.append((wx, wy, ch))
    assert np.array_equal(valid(out[0][0], w), valid(out[1][0], w)) and np.array_equal(valid(out[0][1], w), valid(out[1][1], w))
    # the change norms are fp64 sums taken in another order by the two pipelines: equal to fp64 rounding, not bit for bit
    assert np.allclose(out[0][2], out[1][2], rtol=1e-6, atol=0)


def test_level_channel_weights_and_initial_flow(ctx, oracle):
    w, h = 67, 45
    frames, af, sf = normalized_frames(oracle, w, h, 3, seed=9)
    rng = np.random.default_rng(0)
    chw = [noise_plane(rng, w, h, 0.5, 1.5) for _ in range(3)]
    init = (noise_plane(rng, w, h, 1.5, 2.5), noise_plane(rng, w, h, 0.5, 1.5))
    po, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf)
    o, g = run_both(ctx, oracle, po, ps, frames, w, h, chw=chw, init=init)
    d = max(np.abs(valid(o[0], w) - valid(g[0], w)).max(), np.abs(valid(o[1], w) - valid(g[1], w)).max())
    assert d <= TOL_LEVEL, d


def test_thresholds_break_like_the_oracle(ctx, oracle):
    w, h = 67, 45
    frames, af, sf = normalized_frames(oracle, w, h, 3)
    po, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, niter_outer=30, thres_outer=2e-3, thres_inner=1e-9)
    o, g = run_both(ctx, oracle, po, ps, frames, w, h)
    d = max(np.abs(valid(o[0], w) - valid(g[0], w)).max(), np.abs(valid(o[1], w) - valid(g[1], w)).max())
    assert d <= TOL_LEVEL, d
    assert o[2][0] < 2e-3 and abs(o[2][0] - g[2][0]) < 1e-6      # both stopped at the same outer iteration


@pytest.mark.parametrize("w,h", [(67, 45), (200, 150)])
def test_break_decision_at_the_threshold_is_the_reference_arithmetic(ctx, oracle, switches, w, h):
    """VERDICT r4 "missing" 3: the reference decides the outer break on fp32 running sums in raster order (variational_mt.cpp:412-436), the GPU on fp64 tree sums -- which
    can fall on the other side of a threshold that the norm all but touches.  Round 5: a window whose fp64 norm lies within the band (1e-3 at these sizes) of the threshold is decided by the
    reference's own summation (k_exact_break).  The sharpest case there is: the threshold set to the oracle's own fp32 norm of outer iteration k (then `norm < thres` is
    false there and the run goes on) and to the next float above it (true: the run stops at k) -- both sides stop where the oracle stops and report the oracle's norm bit
    for bit; with the exact decision switched off (SFA_NO_EXACT_BREAK) the norms agree only to the fp64 / fp32 difference"""
    frames, af, sf = normalized_frames(oracle, w, h, 3)
    kw = dict(S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, niter_outer=8, thres_inner=1e-9)
    po, ps = mk_params(oracle, thres_outer=0, **kw)
    oracle.change_log(64)
    wxo, wyo = orc.plane(h, orc.stride_of(w)), orc.plane(h, orc.stride_of(w))
    rc, _, _ = oracle.compute_one_level(po, wxo, wyo, frames, w)
    rows = oracle.change_log_rows().copy()
    oracle.change_log(0)
    outer = [(float(max(np.float32(a), np.float32(b))), np.float32(a), np.float32(b)) for kind, it, a, b in rows if kind == 1]
    assert rc == 0 and len(outer) == 8
    k = 3
    assert all(outer[i][0] > outer[k][0] for i in range(k)) and outer[k + 1][0] < outer[k][0]          # the norms fall: the first iteration below m_k is k + 1
    m = np.float32(outer[k][0])
    for thres, stop in ((m, k + 1), (np.nextafter(m, np.float32(np.inf)), k)):
        po, ps = mk_params(oracle, thres_outer=float(thres), **kw)
        o, g = run_both(ctx, oracle, po, ps, frames, w, h)
        assert np.float32(o[2][0]) == outer[stop][1] and np.float32(o[2][1]) == outer[stop][2]        # the oracle stopped at `stop`
        if stop == k:       # ... and so did the GPU: the stop was decided inside the band, on the reference's own fp32 sums -- the norms are the oracle's bit for bit
            assert np.float32(g[2][0]) == np.float32(o[2][0]) and np.float32(g[2][1]) == np.float32(o[2][1]), (thres, g[2], o[2])
        else:               # iteration k was NOT a stop (decided inside the band), k + 1 is one far below the threshold: its norms are the fp64 sums'
            assert abs(g[2][0] - o[2][0]) <= 2e-5 * abs(o[2][0]) and abs(g[2][1] - o[2][1]) <= 2e-5 * abs(o[2][1]), (thres, g[2], o[2])
        d = max(np.abs(valid(o[0], w) - valid(g[0], w)).max(), np.abs(valid(o[1], w) - valid(g[1], w)).max())
        assert d <= TOL_LEVEL, d
    # the same decision inside a lockstep batch (two copies of the window around one that stops at once: the undecided windows share a mask word)
    po, ps = mk_params(oracle, thres_outer=float(np.nextafter(m, np.float32(np.inf))), **kw)
    o, g = run_both(ctx, oracle, po, ps, frames, w, h)
    job = sfa.Job(ctx, ps, w, h, 3)
    for b, f in enumerate((frames, [frames[1]] * 3, frames)):
        job.upload(b, [c_(x) for x in f])
    job.run()
    for b in (0, 2):
        gx, gy, chg = job.download(b)
        assert np.array_equal(gx, g[0]) and np.array_equal(gy, g[1]) and np.float32(chg[0]) == np.float32(o[2][0]) and np.float32(chg[1]) == np.float32(o[2][1]), b
    job.close()
    switches.set("SFA_NO_EXACT_BREAK", "1")
    po, ps = mk_params(oracle, thres_outer=float(m), **kw)
    o, g = run_both(ctx, oracle, po, ps, frames, w, h)
    assert abs(g[2][0] - o[2][0]) <= 2e-4 * abs(o[2][0]) or abs(g[2][0] - outer[k][1]) <= 2e-4 * abs(outer[k][1])      # either side of the threshold, to the sums' difference


def test_break_band_follows_the_image_size(ctx, oracle):
    """how far the reference's fp32 running sum of a change norm can lie from the fp64 sum grows with the number of additions -- ceil(w / 4) * h * 2^-24 of
    its value, 6.7e-3 at 1024 x 436 -- and need not average out (small block sums dropped one after the other, always downwards).  The band inside which the break is
    decided on the reference's own summation is that bound since round 6 (sfa_internal.h: break_band; a constant 1e-3 before).  At 1024 x 436 a threshold 3e-3 above the
    oracle's norm of the first outer iteration -- outside the old band, inside the new one -- is decided by k_exact_break: the GPU stops at k like the oracle AND reports the
    oracle's fp32 norms bit for bit (the fp64 sums would differ in their last digits); a threshold 3e-2 above is outside the band: same stop, norms to the sums' difference"""
    w, h = 1024, 436
    frames, af, sf = normalized_frames(oracle, w, h, 3, seed=21)
    kw = dict(S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, niter_outer=3, thres_inner=1e-9)
    po, ps = mk_params(oracle, thres_outer=0, **kw)
    oracle.change_log(64)
    wxo, wyo = orc.plane(h, orc.stride_of(w)), orc.plane(h, orc.stride_of(w))
    rc, _, _ = oracle.compute_one_level(po, wxo, wyo, frames, w)
    rows = oracle.change_log_rows().copy()
    oracle.change_log(0)
    outer = [(float(max(np.float32(a), np.float32(b))), np.float32(a), np.float32(b)) for kind, it, a, b in rows if kind == 1]
    assert rc == 0 and len(outer) == 3
    k = 0                                                                     # every threshold above the first iteration's norm is met there and nowhere before
    for rel, exact in ((3e-3, True), (3e-2, False)):
        thres = np.float32(outer[k][0] * (1.0 + rel))
        po, ps = mk_params(oracle, thres_outer=float(thres), **kw)
        o, g = run_both(ctx, oracle, po, ps, frames, w, h)
        assert np.float32(o[2][0]) == outer[k][1] and np.float32(o[2][1]) == outer[k][2]                     # the oracle stopped at k
        if exact:
            assert np.float32(g[2][0]) == outer[k][1] and np.float32(g[2][1]) == outer[k][2], (rel, g[2], outer[k])
        else:
            assert abs(g[2][0] - o[2][0]) <= 1e-4 * abs(o[2][0]) and abs(g[2][1] - o[2][1]) <= 1e-4 * abs(o[2][1]), (rel, g[2], o[2])
        d = max(np.abs(valid(o[0], w) - valid(g[0], w)).max(), np.abs(valid(o[1], w) - valid(g[1], w)).max())
        assert d <= TOL_LEVEL, (rel, d)


def test_inner_break_decision_at_the_threshold_is_the_reference_arithmetic(ctx, oracle):
    """the same for the inner break (variational_mt.cpp:371-407, taken on the host; `slow_flow_niter_inner` > 1): the threshold set to the oracle's own fp32 norm of the
    first inner iteration (no break there: all three run) and to the next float above it (break: one runs) -- two different flows, and the GPU follows the oracle
    into each"""
    w, h = 67, 45
    frames, af, sf = normalized_frames(oracle, w, h, 3)
    kw = dict(S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, niter_outer=1, niter_inner=3, thres_outer=0)
    po, ps = mk_params(oracle, thres_inner=1e-12, **kw)
    oracle.change_log(64)
    wxo, wyo = orc.plane(h, orc.stride_of(w)), orc.plane(h, orc.stride_of(w))
    rc, _, _ = oracle.compute_one_level(po, wxo, wyo, frames, w)
    rows = oracle.change_log_rows().copy()
    oracle.change_log(0)
    inner = [np.float32(max(np.float32(a), np.float32(b))) for kind, it, a, b in rows if kind == 0]
    assert rc == 0 and len(inner) == 3 and inner[1] > inner[0]      # (at the threshold m the second iteration then does not break either: one against three iterations)
    m = inner[0]
    flows = []
    for thres in (m, np.nextafter(m, np.float32(np.inf))):
        po, ps = mk_params(oracle, thres_inner=float(thres), **kw)
        o, g = run_both(ctx, oracle, po, ps, frames, w, h)
        d = max(np.abs(valid(o[0], w) - valid(g[0], w)).max(), np.abs(valid(o[1], w) - valid(g[1], w)).max())
        assert d <= TOL_LEVEL, (thres, d)
        flows.append(o)
    assert np.abs(valid(flows[0][0], w) - valid(flows[1][0], w)).max() > 100 * TOL_LEVEL           # the decision matters: one inner iteration more or less


def test_nan_norms_never_break(ctx, oracle):
    """std::max(a, b) = (a < b) ? b : a keeps a NaN first argument (variational_mt.cpp:407,436): a NaN change norm is never below a threshold, so neither
    side breaks, both report NaN norms, and the NaNs have spread over the same pixels"""
    w, h = 67, 45
    frames, af, sf = normalized_frames(oracle, w, h, 3)
    frames[1][1, h - 3, w - 4] = np.nan
    po, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, niter_outer=2, niter_solver=3, thres_outer=1e30, thres_inner=1e30)
    o, g = run_both(ctx, oracle, po, ps, frames, w, h)
    assert np.isnan(o[2][0]) and np.isnan(o[2][1]) and np.isnan(g[2][0]) and np.isnan(g[2][1])
    for k in range(2):
        a, b = valid(o[k], w), valid(g[k], w)
        assert np.array_equal(np.isnan(a), np.isnan(b))
        fin = ~np.isnan(a)
        assert fin.sum() == 0 or np.abs(a[fin] - b[fin]).max() <= TOL_LEVEL


def test_config1_translation_256(ctx, oracle):
    """BASELINE config 1: 256x256 constant translation (1.5,-0.75), 1 level, 30 SOR iterations"""
    w = h = 256
    frames, af, sf = normalized_frames(oracle, w, h, 3)
    po, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, niter_outer=5)
    o, g = run_both(ctx, oracle, po, ps, frames, w, h)
    d = max(np.abs(valid(o[0], w) - valid(g[0], w)).max(), np.abs(valid(o[1], w) - valid(g[1], w)).max())
    assert d <= TOL_UV, d
    assert abs(g[0][32:-32, 32:w - 32].mean() - 1.5) < 0.05 and abs(g[1][32:-32, 32:w - 32].mean() + 0.75) < 0.05


def test_pyramid_ops(ctx, oracle):
    w, h = 130, 98
    rng = np.random.default_rng(8)
    src = noise_plane(rng, w, h, 0, 255)
    a = oracle.gaussian_blur_cv(src, w, 0.745356)
    b = ctx.gaussian_blur(c_(src), w, 0.745356)
    assert np.array_equal(valid(a, w), valid(b, w))
    for (dw, dh) in ((117, 88), (65, 49), (200, 150)):
        a = oracle.resize_linear_cv(src, w, dw, dh)
        b = ctx.resize_linear(c_(src), w, dw, dh)
        assert np.array_equal(valid(a, dw), valid(b, dw))


@pytest.mark.parametrize("layers", [3, 5])
def test_variational_multilevel(ctx, oracle, layers):
    w, h = 130, 98
    frames, af, sf = normalized_frames(oracle, w, h, 3, seed=11)
    po, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, layers=layers, niter_outer=3)
    o, g = run_both(ctx, oracle, po, ps, frames, w, h, level_only=False)
    d = max(np.abs(valid(o[0], w) - valid(g[0], w)).max(), np.abs(valid(o[1], w) - valid(g[1], w)).max())
    assert d <= TOL_UV, d
    # the synthetic sequence moves by (2,1) px per frame
    assert abs(np.median(valid(g[0], w)) - 2.0) < 0.2 and abs(np.median(valid(g[1], w)) - 1.0) < 0.2


def test_variational_full_size_config2(ctx, oracle):
    """BASELINE config 2 shape: 1024x436, 5 levels, 5 outer x 30 SOR, S=2, against the oracle (<= 1e-4)"""
    w, h = 1024, 436
    frames, af, sf = normalized_frames(oracle, w, h, 3, seed=5)
    po, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, layers=5, niter_outer=5)
    o, g = run_both(ctx, oracle, po, ps, frames, w, h, level_only=False)
    d = max(np.abs(valid(o[0], w) - valid(g[0], w)).max(), np.abs(valid(o[1], w) - valid(g[1], w)).max())
    assert d <= TOL_UV, d


def test_job_batch_equals_single(ctx, oracle):
    """frame windows of a batch are independent: a lockstep batch gives each element what it gets alone"""
    w, h = 96, 64
    fa, af, sf = normalized_frames(oracle, w, h, 3, seed=1)
    fb, _, _ = normalized_frames(oracle, w, h, 3, seed=2)
    _, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, layers=2)
    singles = []
    for fr in (fa, fb):
        j = sfa.Job(ctx, ps, w, h, 1)
        j.upload(0, [c_(f) for f in fr])
        j.run()
        singles.append(j.download(0))
        j.close()
    j = sfa.Job(ctx, ps, w, h, 2)
    j.upload(0, [c_(f) for f in fa]); j.upload(1, [c_(f) for f in fb])
    j.run()
    for b in range(2):
        wx, wy, _ = j.download(b)
        assert np.array_equal(wx, singles[b][0]) and np.array_equal(wy, singles[b][1])
    j.run()   # re-running a resident job starts again from the uploaded initial flow
    wx, wy, _ = j.download(0)
    assert np.array_equal(wx, singles[0][0])
    j.close()


def test_normalize(ctx, oracle):
    """normalize() (variational_mt.cpp:17-85).  The reference sums I and the fp32 product I*I in fp64 in raster order; the GPU forms the same fp64 terms and adds them as
    a tree.  (1) For what the path is actually fed -- 8- and 16-bit pixel values, as every image file delivers them -- each term and each partial sum is an integer below
    2^53, fp64 addition is exact whatever its order, and statistics AND frames are the reference's to the last bit: asserted with ==.  (2) For arbitrary fp32 input the two
    summation orders round differently (~1e-13 relative on the statistics); the normalised pixel is a double expression stored to float on both sides, so a pixel whose
    exact value lies within 1e-13 of a rounding boundary can come out one ulp apart -- rare, and the only way to remove it would be a sequential sum on the GPU.  That
    case keeps the 1-ulp bound, counted in ulps."""
    w, h = 67, 45
    rng = np.random.default_rng(3)
    for top in (255, 65535):
        fo = []
        for k in range(5):
            f = orc.aligned_zeros((3, h, orc.stride_of(w)))
            f[:, :, :w] = rng.integers(0, top + 1, size=(3, h, w)).astype(np.float32)
            fo.append(f)
        fg = [c_(f).copy() for f in fo]
        avg_o, std_o, _, _ = oracle.normalize(fo, w)
        avg_g, std_g = ctx.normalize(fg, w)
        assert list(avg_o) == list(avg_g) and list(std_o) == list(std_g), top
        for a, b in zip(fo, fg):
            assert np.array_equal(valid(a, w), valid(b, w)), top
    fo = [texture_frame(w, h, k) for k in range(3)]
    fg = [c_(f).copy() for f in fo]
    avg_o, std_o, _, _ = oracle.normalize(fo, w)
    avg_g, std_g = ctx.normalize(fg, w)
    assert np.allclose(avg_o, avg_g, rtol=1e-12, atol=0) and np.allclose(std_o, std_g, rtol=1e-12, atol=0)
    for a, b in zip(fo, fg):
        ulp = np.abs(valid(a, w).view(np.int32).astype(np.int64) - valid(b, w).view(np.int32).astype(np.int64))
        assert ulp.max() <= 1 and (ulp > 0).mean() < 1e-3


# ------------------------------------------------------------------------------------------------------
# BASELINE configs 3 and 5 (parity cases, not bench lines): their shapes and penalties at full size
# ------------------------------------------------------------------------------------------------------
def test_config5_sor_2048(ctx, oracle):
    """2048x2048, K = 30: single solve (task kernel) and a lockstep batch (band kernel) against the raster-order oracle"""
    w = h = 2048
    rng = np.random.default_rng(55)
    s = sor_system(rng, w, h)
    a = copy_sys(s)
    oracle.sor(a["du"], a["dv"], a["a11"], a["a12"], a["a22"], a["b1"], a["b2"], a["sh"], a["sv"], w, 30, 1.9)
    for nb in (1, 16):
        sb = sfa.SorBatch(ctx, w, h, nb)
        for b in range(nb):
            sb.upload(b, *[c_(s[k]) for k in ("du", "dv", "a11", "a12", "a22", "b1", "b2", "sh", "sv")])
        sb.run(30, 1.9)
        for b in (0, nb - 1):
            du, dv = sb.download(b)
            assert np.array_equal(valid(a["du"], w), valid(du, w)) and np.array_equal(valid(a["dv"], w), valid(dv, w)), (nb, b)
        sb.close()


def test_config5_level_2048_lorentzian(ctx, oracle):
    """one 2048x2048 level under the Lorentzian penalty (config 5's operator set) against the oracle"""
    w = h = 2048
    frames, af, sf = normalized_frames(oracle, w, h, 3, seed=21)
    lor = (2, 0.05, 0.5)
    po, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, niter_outer=1, robust_color=lor, robust_grad=lor, robust_reg=lor)
    o, g = run_both(ctx, oracle, po, ps, frames, w, h)
    d = max(np.abs(valid(o[0], w) - valid(g[0], w)).max(), np.abs(valid(o[1], w) - valid(g[1], w)).max())
    assert d <= TOL_LEVEL, d


def test_config5_pyramid_6_levels_fused_equals_unfused(ctx, oracle, switches):
    """2048x2048, 6 levels, Lorentzian, whole coarse-to-fine run: the fused pipeline is the materialised one bit for bit
    (the oracle is too slow for the whole schedule at this size; its levels are pinned above and at smaller sizes)"""
    w = h = 2048
    frames, af, sf = normalized_frames(oracle, w, h, 3, seed=22)
    lor = (2, 0.05, 0.5)
    _, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, niter_outer=2, layers=6, robust_color=lor, robust_grad=lor, robust_reg=lor)
    out = []
    for unfused in ("1", "0"):
        switches.set("SFA_UNFUSED", unfused)
        wx, wy = np.zeros((h, sfa.stride_of(w)), np.float32), np.zeros((h, sfa.stride_of(w)), np.float32)
        ctx.variational(ps, wx, wy, [c_(f) for f in frames], w, None)
        out.append((wx, wy))
    assert np.array_equal(valid(out[0][0], w), valid(out[1][0], w)) and np.array_equal(valid(out[0][1], w), valid(out[1][1], w))
    assert abs(np.median(valid(out[1][0], w)) - 2.0) < 0.2 and abs(np.median(valid(out[1][1], w)) - 1.0) < 0.2   # the sequence moves by (2,1) px / frame


def test_config5_batch_32_fits_in_1_65_gb_per_window(ctx, oracle):
    """arena diet: one level is live at a time, so the work planes of all levels share one region and, for large frames, so do the
    solver workspaces.  2048x2048, 6 levels, Lorentzian: a lockstep batch of 32 windows costs <= 1.65 GB per window (round 2: ~4 GB, so config 5
    could not reach the batch sizes the throughput needs), and every window of the batch ends with the bits of the same window refined alone."""
    w = h = 2048
    frames, af, sf = normalized_frames(oracle, w, h, 3, seed=22)
    lor = (2, 0.05, 0.5)
    _, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, niter_outer=1, layers=6, robust_color=lor, robust_grad=lor, robust_reg=lor)
    fr = [c_(f) for f in frames]
    out = {}
    for nb in (1, 32):
        job = sfa.Job(ctx, ps, w, h, nb)
        for b in range(nb):
            job.upload(b, fr)
        job.run(); ctx.sync()
        if nb == 32:
            per_window = job.device_bytes() / nb
            assert per_window <= 1.65e9, per_window
        out[nb] = [job.download(b)[:2] for b in sorted({0, nb - 1})]
        job.close()
    for wx, wy in out[32]:
        assert np.array_equal(valid(wx, w), valid(out[1][0][0], w)) and np.array_equal(valid(wy, w), valid(out[1][0][1], w))
    assert abs(np.median(valid(out[1][0][0], w)) - 2.0) < 0.3


@pytest.mark.parametrize("w,h,layers", [(256, 200, 4), (130, 98, 3)])
def test_shared_solver_workspace_is_the_per_level_one(ctx, oracle, switches, w, h, layers):
    """large frames re-shape ONE solver workspace level by level instead of keeping one per level; forced on a small size (SFA_SHARE_SOR) it must not change a bit,
    (the memory figure of a bench window is test_bench_window_memory's)"""
    frames, af, sf = normalized_frames(oracle, w, h, 3, seed=31)
    _, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, niter_outer=2, layers=layers)
    out = []
    for share in (False, True):
        if share:
            switches.set("SFA_SHARE_SOR", "1")
        else:
            switches.unset("SFA_SHARE_SOR")
        wx, wy = np.zeros((h, sfa.stride_of(w)), np.float32), np.zeros((h, sfa.stride_of(w)), np.float32)
        for _ in range(2):                                   # twice: the second run re-shapes a workspace that already holds another level's data
            wx[...] = 0; wy[...] = 0
            ctx.variational(ps, wx, wy, [c_(f) for f in frames], w, None)
        out.append((wx.copy(), wy.copy()))
    assert np.array_equal(valid(out[0][0], w), valid(out[1][0], w)) and np.array_equal(valid(out[0][1], w), valid(out[1][1], w))


def test_bench_window_memory(ctx, oracle):
    """a 1024x436 window of the bench configuration (S = 2, 5 levels) holds <= 0.32 GB of device memory (round 2: 0.43): 0.12 GB of arena, the rest
    the five per-level solver workspaces (at this size they are not shared: re-zeroing them per level would cost 3-4 % of the step)"""
    import bench
    p = bench.bench_params()
    job = sfa.Job(ctx, p, bench.W, bench.H, 8)
    win = bench.synth_window(3)
    for b in range(8):
        job.upload(b, win)
    job.run(); ctx.sync()
    per_window = job.device_bytes() / 8
    job.close()
    assert per_window <= 0.32e9, per_window


def test_config3_shape_2560x1440_cfg_terms(ctx, oracle, switches):
    """config 3 stand-in: 2560x1440, cfgs/slow_flow.cfg terms (S=3, rho 1/1, omega 0/2, modified L1), 5 levels:
    fused == materialised bit for bit at full size, and the known (2,1) px/frame translation is recovered"""
    w, h = 2560, 1440
    frames, af, sf = normalized_frames(oracle, w, h, 5, seed=23)
    _, ps = mk_params(oracle, S=3, rho=[1, 1], omega=[0, 2], norm_avg=af, norm_std=sf, niter_outer=2, layers=5)
    out = []
    for unfused in ("1", "0"):
        switches.set("SFA_UNFUSED", unfused)
        wx, wy = np.zeros((h, sfa.stride_of(w)), np.float32), np.zeros((h, sfa.stride_of(w)), np.float32)
        ctx.variational(ps, wx, wy, [c_(f) for f in frames], w, None)
        out.append((wx, wy))
    assert np.array_equal(valid(out[0][0], w), valid(out[1][0], w)) and np.array_equal(valid(out[0][1], w), valid(out[1][1], w))
    assert abs(np.median(valid(out[1][0], w)) - 2.0) < 0.2 and abs(np.median(valid(out[1][1], w)) - 1.0) < 0.2


def test_config3_level_2560x1440_against_the_oracle(ctx, oracle):
    """config 3's frame size and cfg terms (S = 3: four image pairs, the to-reference terms omega = 0/2 included) through ONE full-size level, two
    outer iterations, against the oracle -- the comparison the full schedule is too slow for on the CPU (full-size configs 3 and 5 were
    property tests only)"""
    w, h = 2560, 1440
    frames, af, sf = normalized_frames(oracle, w, h, 5, seed=23)
    po, ps = mk_params(oracle, S=3, rho=[1, 1], omega=[0, 2], norm_avg=af, norm_std=sf, niter_outer=2)
    o, g = run_both(ctx, oracle, po, ps, frames, w, h)
    d = max(np.abs(valid(o[0], w) - valid(g[0], w)).max(), np.abs(valid(o[1], w) - valid(g[1], w)).max())
    assert d <= TOL_LEVEL, d
    assert np.abs(valid(g[0], w)).max() > 0.05                       # the level did move the flow


def test_config5_two_levels_2048_against_the_oracle(ctx, oracle):
    """config 5's frame size and operator set (Lorentzian) through the two finest levels of the pyramid (down-sampling, flow rescale and both levels'
    iterations), against the oracle at <= 1e-4"""
    w = h = 2048
    frames, af, sf = normalized_frames(oracle, w, h, 3, seed=21)
    lor = (2, 0.05, 0.5)
    po, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, niter_outer=1, layers=2, robust_color=lor, robust_grad=lor, robust_reg=lor)
    o, g = run_both(ctx, oracle, po, ps, frames, w, h, level_only=False)
    d = max(np.abs(valid(o[0], w) - valid(g[0], w)).max(), np.abs(valid(o[1], w) - valid(g[1], w)).max())
    assert d <= TOL_UV, d


# ------------------------------------------------------------------------------------------------------
# occlusion step between alternations (optimizeOcc, variational_aux_mt.cpp:758-887)
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pid", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("S,w,h", [(2, 67, 45), (3, 130, 70)])
def test_occlusion_costs(ctx, oracle, pid, S, w, h):
    """data costs from the image pairs (Iz, Ixz, Iyz formed in LDS) == the oracle's costs from materialised stacks"""
    ref = S - 1
    rng = np.random.default_rng(pid + 10 * S)
    eps = 0.001 if pid in (1, 3) else 0.05
    po, ps = mk_params(oracle, S=S, rho=[1, 0.5][:ref], omega=[0.5, 2][:ref], robust_color=(pid, eps, 0.5), robust_grad=(pid, eps, 0.3))
    imgs = [[smooth_noise_color(rng, w, h, 10) for _ in range(4)] for _ in range(2 * ref)]     # per slot: succ1, succ2, ref1, ref2
    masks = [noise_plane(rng, w, h, 0, 1) for _ in range(2 * ref)]
    for m in masks:
        m[:, :w] = (m[:, :w] > 0.2).astype(np.float32)
    masks[0][:3, :w] = 0; masks[-1][:3, :w] = 0                                                # a region where one label has no support at all
    succ = np.stack([oracle.derivative_stack(a, b, w) for a, b, _, _ in imgs])
    toref = np.stack([oracle.derivative_stack(c, d, w) for _, _, c, d in imgs])
    mo = orc.aligned_zeros((2 * ref,) + masks[0].shape); mo[...] = np.stack(masks)
    so = orc.aligned_zeros(succ.shape); so[...] = succ
    to = orc.aligned_zeros(toref.shape); to[...] = toref
    o0, o1 = oracle.occlusion_costs(mo, so, to, ref, list(po.rho)[:ref], list(po.omega)[:ref], po.delta / np.float32(3), po.gamma / np.float32(3),
                                    po.occlusion_penalty, po.robust_color, po.robust_grad, w)
    g0, g1 = ctx.occlusion_costs(ps, [c_(m) for m in masks], [c_(i[0]) for i in imgs], [c_(i[1]) for i in imgs], [c_(i[2]) for i in imgs],
                                 [c_(i[3]) for i in imgs], w)
    for a, b in ((o0, g0), (o1, g1)):
        if pid == 2:      # Lorentzian: log in fp64 on both sides, glibc vs ocml -> the float result may differ in the last bit
            ulp = np.abs(valid(a, w).view(np.int32).astype(np.int64) - valid(b, w).view(np.int32).astype(np.int64))
            assert ulp.max() <= 1 and (ulp > 0).mean() < 1e-3
        else:
            assert np.array_equal(valid(a, w), valid(b, w))


def _cut_case(rng, w, h, kind):
    st = sfa.stride_of(w)
    d0, d1 = np.zeros((h, st), np.float32), np.zeros((h, st), np.float32)
    if kind == "noise":
        d0[:, :w] = rng.uniform(0, 2, (h, w)); d1[:, :w] = rng.uniform(0, 2, (h, w))
    elif kind == "blobs":       # the typical shape: label 0 preferred everywhere except in a few compact regions
        d0[:, :w] = rng.uniform(0, 0.2, (h, w)); d1[:, :w] = 1.0 + rng.uniform(0, 0.2, (h, w))
        for _ in range(6):
            cx, cy, r = rng.integers(0, w), rng.integers(0, h), rng.integers(2, 9)
            yy, xx = np.mgrid[0:h, 0:w]
            d0[:, :w][(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] += rng.uniform(1.0, 4.0)
    else:                       # "stripes": long thin structures, long residual paths
        d0[:, :w] = 0.1; d1[:, :w] = 0.6
        d0[::7, :w] += 3.0; d1[3::7, :w] += 3.0
        d0[:, :w] += rng.uniform(0, 0.05, (h, w))
    return d0, d1


@pytest.mark.parametrize("kind", ["noise", "blobs", "stripes"])
@pytest.mark.parametrize("w,h,alpha", [(67, 45, 0.5), (130, 98, 0.5), (64, 64, 2.0), (5, 4, 0.3), (200, 33, 0.05), (33, 70, 0.0)])
def test_grid_cut_reaches_the_minimum_energy(ctx, oracle, kind, w, h, alpha):
    """the GPU push-relabel labelling has the energy of the oracle's exact (fp64 Dinic) minimum cut; labels agree
    except where the minimum is not unique"""
    rng = np.random.default_rng(w * h + int(alpha * 100))
    d0, d1 = _cut_case(rng, w, h, kind)
    a0 = orc.plane(*d0.shape); a0[...] = d0
    a1 = orc.plane(*d1.shape); a1[...] = d1
    occ_o, e_o = oracle.grid_cut(a0, a1, alpha, w)
    occ_g = ctx.grid_cut(c_(d0), c_(d1), alpha, w)
    assert set(np.unique(valid(occ_g, w))) <= {-1.0, 1.0}
    og = orc.plane(*d0.shape); og[...] = occ_g
    e_g = oracle.grid_cut_energy(og, a0, a1, alpha, w)
    assert abs(e_g - e_o) <= 1e-5 * max(1.0, abs(e_o)), (e_g, e_o)
    assert (valid(occ_o, w) != valid(occ_g, w)).mean() < 0.01
    assert np.array_equal(occ_g, ctx.grid_cut(c_(d0), c_(d1), alpha, w))       # deterministic


def test_grid_cut_full_size(ctx, oracle):
    """1024x436: energy against the oracle's exact cut"""
    w, h = 1024, 436
    rng = np.random.default_rng(5)
    d0, d1 = _cut_case(rng, w, h, "blobs")
    a0 = orc.plane(*d0.shape); a0[...] = d0
    a1 = orc.plane(*d1.shape); a1[...] = d1
    _, e_o = oracle.grid_cut(a0, a1, 0.5, w)
    occ_g = ctx.grid_cut(c_(d0), c_(d1), 0.5, w)
    og = orc.plane(*d0.shape); og[...] = occ_g
    assert abs(oracle.grid_cut_energy(og, a0, a1, 0.5, w) - e_o) <= 1e-5 * abs(e_o)


@pytest.mark.parametrize("kind,w,h", [("blobs", 1024, 436), ("stripes", 300, 200), ("noise", 130, 98), ("blobs", 67, 45)])
def test_grid_cut_sparse_phase_runs_the_same_rounds(ctx, switches, kind, w, h):
    """the one-workgroup-per-window kernel of the sparse phase executes the rounds the grid launches would: identical labels"""
    rng = np.random.default_rng(w + h)
    d0, d1 = _cut_case(rng, w, h, kind)
    switches.set("SFA_CUT_DISCHARGE", "0")                # the grid rounds (round 2's schedule, the cross-check of the tile discharge)
    switches.set("SFA_CUT_TAIL", "1")                     # a single window would not take it by itself
    occ_tail = ctx.grid_cut(c_(d0), c_(d1), 0.5, w)
    switches.unset("SFA_CUT_TAIL")
    switches.set("SFA_CUT_NO_TAIL", "1")
    occ_grid = ctx.grid_cut(c_(d0), c_(d1), 0.5, w)
    assert np.array_equal(occ_tail, occ_grid)


@pytest.mark.parametrize("kind,w,h", [("blobs", 1024, 436), ("stripes", 300, 200), ("noise", 130, 98), ("blobs", 67, 45), ("stripes", 64, 16), ("noise", 65, 17)])
def test_grid_cut_tile_discharge_finds_the_same_cut(ctx, oracle, switches, kind, w, h):
    """the tile-discharge schedule (rounds inside 64 x 16 tiles, four colours) and the grid rounds reach a maximum flow each; the labelling read off
    it (who still reaches the passive terminal) is the same set whichever maximum flow it is -- and the energy is the oracle's minimum"""
    rng = np.random.default_rng(w + 3 * h)
    d0, d1 = _cut_case(rng, w, h, kind)
    occ_tiles = ctx.grid_cut(c_(d0), c_(d1), 0.5, w)
    for inner, sup in ((3, 1), (200, 4)):
        switches.set("SFA_CUT_INNER", str(inner)); switches.set("SFA_CUT_SUPER", str(sup))
        assert np.array_equal(occ_tiles, ctx.grid_cut(c_(d0), c_(d1), 0.5, w))
    switches.set("SFA_CUT_DISCHARGE", "0")
    occ_grid = ctx.grid_cut(c_(d0), c_(d1), 0.5, w)
    a0 = orc.plane(*d0.shape); a0[...] = d0
    a1 = orc.plane(*d1.shape); a1[...] = d1
    og = orc.plane(*d0.shape)
    og[...] = occ_tiles; e_t = oracle.grid_cut_energy(og, a0, a1, 0.5, w)
    og[...] = occ_grid; e_g = oracle.grid_cut_energy(og, a0, a1, 0.5, w)
    assert abs(e_t - e_g) <= 1e-6 * max(1.0, abs(e_g)), (e_t, e_g)
    assert (valid(occ_tiles, w) != valid(occ_grid, w)).mean() < 1e-3


@pytest.mark.parametrize("S,rho,omega", [(2, [1], [0]), (3, [1, 1], [0, 2])])
def test_level_with_occlusion_reasoning(ctx, oracle, S, rho, omega):
    """alternations with the discrete occlusion step (cfgs/slow_flow.cfg: occlusion reasoning on).  A minimum cut need not be unique, so
    the comparison is made under ONE labelling: the GPU's labels of every alternation (sfa_job_keep_alternation_occlusions) are (a) checked
    to be minimum-energy labellings of the oracle's own costs at that alternation and (b) forced on the oracle (orc_force_labels), whose
    flow must then match the GPU's -- unconditionally."""
    w, h = 96, 64
    A = 3
    frames, af, sf = normalized_frames(oracle, w, h, 2 * S - 1, seed=4)
    po, ps = mk_params(oracle, S=S, rho=rho, omega=omega, norm_avg=af, norm_std=sf, niter_outer=2, niter_alter=A, occlusion_reasoning=1, layers=1)
    stride = orc.stride_of(w)
    job = sfa.Job(ctx, ps, w, h, 1)
    job.keep_alternation_occlusions(True)
    job.upload(0, [c_(f) for f in frames])
    job.run()
    wxg, wyg, _ = job.download(0)
    occ_g = job.download_occlusions(0)
    labels = orc.aligned_zeros((A, h, stride))
    for a in range(1, A):
        labels[a] = job.download_alternation_occlusions(0, a)
        assert set(np.unique(valid(labels[a], w))) <= {-1.0, 1.0}
    job.close()
    assert np.array_equal(occ_g, labels[A - 1])                                             # getOcclusions() = the last alternation's labels
    assert (valid(occ_g, w) > 0).any() and (valid(occ_g, w) < 0).any()                      # the cut did label something "future"
    # the oracle on its own: its labels may differ from the GPU's only where the minimum is not unique
    wxo, wyo = orc.plane(h, stride), orc.plane(h, stride)
    rc, _, occ_o = oracle.compute_one_level(po, wxo, wyo, frames, w, None, want_occ=True)
    assert rc == 0 and (valid(occ_o, w) != valid(occ_g, w)).mean() < 0.002
    # the oracle under the GPU's labels
    oracle.force_labels(labels)
    try:
        wxf, wyf = orc.plane(h, stride), orc.plane(h, stride)
        rc, _, occ_f = oracle.compute_one_level(po, wxf, wyf, frames, w, None, want_occ=True)
        gaps = [oracle.forced_gap(a) for a in range(1, A)]
    finally:
        oracle.force_labels(None)
    assert rc == 0 and np.array_equal(valid(occ_f, w), valid(occ_g, w))
    assert max(abs(g) for g in gaps) <= 1e-5, gaps                                          # each GPU labelling is a minimum of the oracle's energy
    d = max(np.abs(valid(wxf, w) - valid(wxg, w)).max(), np.abs(valid(wyf, w) - valid(wyg, w)).max())
    assert d <= max(TOL_LEVEL, 3 * oracle_sensitivity(oracle, po, frames, w, h)), d


def test_config3_full_schedule_2560x1440(ctx, oracle):
    """BASELINE config 3 (stand-in for the `sheeps` teaser, SURVEY.md 8d: 2560x1440 synthetic) under the FULL cfgs/slow_flow.cfg schedule:
    S=3, rho 1/1, omega 0/2, modified L1, 5 levels, 10 alternations x 10 outer x 30 sweeps, occlusion reasoning on, thresholds 1e-5.
    The oracle needs hours at this size, so the checks are properties: the known (2,1) px/frame translation is recovered, the labels are
    labels, a second run of the same job gives the same bits, and a lockstep batch of two windows gives each window's own result."""
    w, h = 2560, 1440
    frames, af, sf = normalized_frames(oracle, w, h, 5, seed=23)
    _, ps = mk_params(oracle, S=3, rho=[1, 1], omega=[0, 2], norm_avg=af, norm_std=sf, niter_alter=10, niter_outer=10, layers=5, occlusion_reasoning=1,
                      thres_outer=1e-5, thres_inner=1e-5, occlusion_penalty=0.1, occlusion_alpha=0.1)
    fr = [c_(f) for f in frames]
    job = sfa.Job(ctx, ps, w, h, 1)
    job.upload(0, fr[0:5])
    job.run()
    wx, wy, chg = job.download(0)
    occ = job.download_occlusions(0)
    job.run()
    wx2, wy2, _ = job.download(0)
    job.close()
    assert np.isfinite(wx).all() and np.isfinite(wy).all()
    assert abs(np.median(valid(wx, w)) - 2.0) < 0.1 and abs(np.median(valid(wy, w)) - 1.0) < 0.1
    inner = (slice(40, h - 40), slice(40, w - 40))
    assert np.abs(wx[inner] - 2.0).mean() < 0.1 and np.abs(wy[inner] - 1.0).mean() < 0.1
    assert set(np.unique(valid(occ, w))) <= {-1.0, 1.0}
    assert np.array_equal(wx, wx2) and np.array_equal(wy, wy2)
    assert 0 <= chg[0] < 1e-2 and 0 <= chg[1] < 1e-2                                        # the last outer iteration barely moves the flow
    job2 = sfa.Job(ctx, ps, w, h, 2)
    job2.upload(0, fr[::-1]); job2.upload(1, fr[0:5])                                       # the backward window rides along
    job2.run()
    bx, by, _ = job2.download(1)
    job2.close()
    assert np.array_equal(bx, wx) and np.array_equal(by, wy)


def test_batch_with_thresholds_keeps_every_window_exact(ctx, oracle):
    """cfg thresholds on: windows of a lockstep batch meet them at different outer iterations and ride along as passengers of the batched
    launches; each must end with exactly the result it gets alone (which test_thresholds_break_like_the_oracle ties to the oracle)"""
    w, h = 130, 98
    sets = [normalized_frames(oracle, w, h, 3, seed=s)[0] for s in (1, 2)]
    frames, af, sf = normalized_frames(oracle, w, h, 3, seed=3)
    still = [frames[1], frames[1], frames[1]]                                               # no motion: converges at once
    wins = [still, sets[0], sets[1], still, frames, sets[0], still, frames, sets[1]]        # 9 windows: the band kernel's batch path; window 0 stops first
    _, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, niter_outer=8, niter_inner=2, layers=2, thres_outer=2e-3, thres_inner=1e-3)
    alone = []
    for f in wins[:5]:
        j1 = sfa.Job(ctx, ps, w, h, 1)
        j1.upload(0, [c_(x) for x in f]); j1.run()
        alone.append(j1.download(0))
        j1.close()
    job = sfa.Job(ctx, ps, w, h, len(wins))
    for b, f in enumerate(wins):
        job.upload(b, [c_(x) for x in f])
    job.run()
    for b in range(len(wins)):
        gx, gy, chg = job.download(b)
        ref = alone[[0, 1, 2, 0, 4, 1, 0, 4, 2][b]]
        assert np.array_equal(gx, ref[0]) and np.array_equal(gy, ref[1]) and chg == ref[2], b
    job.close()
    assert np.abs(alone[0][0]).max() < 1e-3                                                 # the still window did stop early (and stayed put)


def test_full_batch_of_64_with_thresholds(ctx, oracle):
    """the largest lockstep batch (64 windows: every bit of the device-side mask in use) under break thresholds: windows that stop at different outer
    iterations, each with the result it gets alone"""
    w, h = 130, 98
    sets = [normalized_frames(oracle, w, h, 3, seed=s)[0] for s in (1, 2)]
    frames, af, sf = normalized_frames(oracle, w, h, 3, seed=3)
    still = [frames[1], frames[1], frames[1]]
    kinds = [still, sets[0], sets[1], frames]
    _, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, niter_outer=8, layers=2, thres_outer=2e-3, thres_inner=1e-3)
    alone = []
    for f in kinds:
        j1 = sfa.Job(ctx, ps, w, h, 1)
        j1.upload(0, [c_(x) for x in f]); j1.run()
        alone.append(j1.download(0))
        j1.close()
    job = sfa.Job(ctx, ps, w, h, 64)
    for b in range(64):
        job.upload(b, [c_(x) for x in kinds[(b * 7 + b // 4) % 4]])
    job.run()
    for b in (0, 1, 2, 3, 31, 32, 62, 63):
        gx, gy, chg = job.download(b)
        ref = alone[(b * 7 + b // 4) % 4]
        assert np.array_equal(gx, ref[0]) and np.array_equal(gy, ref[1]) and chg == ref[2], b
    job.close()


def test_full_batch_of_128_with_thresholds(ctx, oracle):
    """round 5: the set of windows that still iterate is two 64-bit words (sfa_internal.h: WMask).  The largest lockstep batch -- 128 windows, every bit of both words
    in use, windows on both sides of the word boundary stopping at different outer iterations (k_outer_threshold: one wave per word) and riding along as passengers --
    each with exactly the result it gets alone; and a batch of 100 (a partly filled second word)"""
    w, h = 67, 45
    sets = [normalized_frames(oracle, w, h, 3, seed=s)[0] for s in (1, 2)]
    frames, af, sf = normalized_frames(oracle, w, h, 3, seed=3)
    still = [frames[1], frames[1], frames[1]]
    kinds = [still, sets[0], sets[1], frames]
    _, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, niter_outer=8, niter_inner=2, layers=2, thres_outer=2e-3, thres_inner=1e-3)
    alone = []
    for f in kinds:
        j1 = sfa.Job(ctx, ps, w, h, 1)
        j1.upload(0, [c_(x) for x in f]); j1.run()
        alone.append(j1.download(0))
        j1.close()
    assert np.abs(alone[0][0]).max() < 1e-3                                                 # the still window stops early
    for nb in (128, 100):
        job = sfa.Job(ctx, ps, w, h, nb)
        pick = lambda b: (b * 7 + b // 4) % 4
        for b in range(nb):
            job.upload(b, [c_(x) for x in kinds[pick(b)]])
        job.run()
        for b in range(nb):
            gx, gy, chg = job.download(b)
            ref = alone[pick(b)]
            assert np.array_equal(gx, ref[0]) and np.array_equal(gy, ref[1]) and chg == ref[2], (nb, b)
        job.close()
    with pytest.raises(sfa.SlowflowError):
        sfa.Job(ctx, ps, w, h, 129)


def test_bench_job_128_windows_against_the_oracle(ctx, oracle):
    """THE TIMED GEOMETRY: what bench.py times is one lockstep group of 128 windows of 1024x436, 5 levels, 5 outer x 30 sweeps
    -- both words of the window mask, 2 048 solver workgroups at level 0 on the seven-stage shape, the path of more than four windows (k_update_outer_x<false> +
    k_reduce_partials).  Built from bench.py's own workload (synth_window, bench_params, the sequence statistics at six digits): (1) windows 0, 63, 64 and 127 of
    the 128-window job against oracle.variational on the same normalised frames, <= 1e-4 (north_star); (2) EVERY window of the job bit for bit what the same windows
    give as two jobs of 64 -- the geometry every full-size oracle comparison of rounds 1-5 ran at or below."""
    import bench
    w, h, nb = bench.W, bench.H, 128
    distinct = 12
    wins = [bench.synth_window(1000 + b) for b in range(distinct)]
    pick = lambda b: {0: 0, 63: 1, 64: 2, 127: 3}.get(b, 4 + b % (distinct - 4))
    avg, std = ctx.normalize([f for wdw in wins for f in wdw], w)
    ps = bench.bench_params()
    po = oracle.default_params()
    for p in (ps, po):
        p.S = bench.S; p.layers = bench.LAYERS; p.niter_alter = 1; p.niter_outer = bench.OUTER; p.niter_inner = bench.INNER; p.niter_solver = bench.SWEEPS
        p.thres_outer = 0; p.thres_inner = 0; p.occlusion_reasoning = 0; p.hbit = 0; p.rho[0] = 1; p.omega[0] = 0
        for k in range(3):
            p.norm_avg[k] = float("%g" % avg[k]); p.norm_std[k] = float("%g" % std[k])
    job = sfa.Job(ctx, ps, w, h, nb)
    for b in range(nb):
        job.upload(b, wins[pick(b)])
    job.run(); job.run()                                                  # the bench re-runs its resident job: the second pass is what it times
    got = [job.download(b) for b in range(nb)]
    job.close()
    for half in (0, 1):
        j64 = sfa.Job(ctx, ps, w, h, 64)
        for b in range(64):
            j64.upload(b, wins[pick(64 * half + b)])
        j64.run()
        for b in range(64):
            gx, gy, chg = j64.download(b)
            assert np.array_equal(gx, got[64 * half + b][0]) and np.array_equal(gy, got[64 * half + b][1]) and chg == got[64 * half + b][2], (half, b)
        j64.close()
    stride = orc.stride_of(w)
    for b in (0, 63, 64, 127):
        fr = []
        for f in wins[pick(b)]:
            a = orc.aligned_zeros(f.shape); a[...] = f
            fr.append(a)
        wxo, wyo = orc.plane(h, stride), orc.plane(h, stride)
        rc, _ = oracle.variational(po, wxo, wyo, fr, w)
        assert rc == 0
        d = max(np.abs(valid(wxo, w) - valid(got[b][0], w)).max(), np.abs(valid(wyo, w) - valid(got[b][1], w)).max())
        assert d <= TOL_UV, (b, d)
        assert abs(np.median(valid(got[b][0], w)) - 2.0) < 0.5              # and the synthetic sequence's motion (2 +- 1 px in x) was found


def test_solver_batch_beyond_one_mask_word(ctx, oracle):
    """100 systems in one solver launch (the second mask word partly filled): first, last and the two around the word boundary are the raster-order oracle's bits"""
    w, h, K, nb = 130, 98, 30, 100
    rng = np.random.default_rng(7)
    systems = [sor_system(rng, w, h) for _ in range(3)]
    sb = sfa.SorBatch(ctx, w, h, nb)
    for b in range(nb):
        sb.upload(b, *[c_(systems[b % 3][k]) for k in ("du", "dv", "a11", "a12", "a22", "b1", "b2", "sh", "sv")])
    sb.run(K, 1.9)
    for b in (0, 63, 64, 99):
        a = copy_sys(systems[b % 3])
        oracle.sor(a["du"], a["dv"], a["a11"], a["a12"], a["a22"], a["b1"], a["b2"], a["sh"], a["sv"], w, K, 1.9)
        du, dv = sb.download(b)
        assert np.array_equal(valid(a["du"], w), valid(du, w)) and np.array_equal(valid(a["dv"], w), valid(dv, w)), b
    sb.close()


def test_job_can_be_run_again_and_slots_reused(ctx, oracle):
    """a resident job is reused: run(); run() gives run() -- also with presmoothing (cfg sigma > 0), which replaces the uploaded frames once
    per upload -- and a slot that held channel weights forgets them when the next window comes without"""
    w, h = 130, 98
    frames, af, sf = normalized_frames(oracle, w, h, 3, seed=13)
    other, _, _ = normalized_frames(oracle, w, h, 3, seed=14)
    rng = np.random.default_rng(0)
    chw = [c_(noise_plane(rng, w, h, 0.5, 1.5)) for _ in range(3)]
    for sigma in (0.0, 0.8):
        _, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, layers=2, niter_outer=2, presmooth_sigma=sigma)
        fresh = sfa.Job(ctx, ps, w, h, 2)
        fresh.upload(0, [c_(f) for f in frames]); fresh.upload(1, [c_(f) for f in other])
        fresh.run()
        want = [fresh.download(b) for b in (0, 1)]
        fresh.run()
        for b in (0, 1):
            again = fresh.download(b)
            assert np.array_equal(again[0], want[b][0]) and np.array_equal(again[1], want[b][1]), (sigma, b)
        fresh.close()
        job = sfa.Job(ctx, ps, w, h, 2)
        job.upload(0, [c_(f) for f in other], chw=chw); job.upload(1, [c_(f) for f in frames], chw=chw)
        job.run()
        weighted = job.download(0)
        job.upload(0, [c_(f) for f in frames]); job.upload(1, [c_(f) for f in other])    # new windows into used slots, no weights
        job.run()
        for b in (0, 1):
            got = job.download(b)
            assert np.array_equal(got[0], want[b][0]) and np.array_equal(got[1], want[b][1]), (sigma, b)
        assert not np.array_equal(weighted[0], want[1][0])                                  # the weights did matter
        job.close()


@pytest.mark.parametrize("sw,sh,fx,fy", [(130, 98, 0.5, 0.5), (131, 97, 0.5, 0.5), (200, 150, 0.3, 0.3), (67, 45, 0.75, 0.6), (64, 48, 1.5, 1.25), (1024, 436, 0.4, 0.4)])
def test_resize_linear_fx(ctx, oracle, sw, sh, fx, fy):
    """the driver's input rescaling (slow_flow.cpp:552: cv::resize(Size(0,0), fx, fy, INTER_LINEAR)): source coordinate (dst + .5) / f - .5,
    which is NOT the explicit-dsize form when sw * fx is not an integer"""
    rng = np.random.default_rng(sw + sh)
    src = noise_plane(rng, sw, sh, 0, 255)
    a, dwa = oracle.resize_linear_fx(src, sw, fx, fy)
    b, dwb = ctx.resize_linear_fx(c_(src), sw, fx, fy)
    assert dwa == dwb and a.shape == b.shape and np.array_equal(valid(a, dwa), valid(b, dwb))
    if (sw * fx) % 1:
        assert not np.array_equal(valid(a, dwa), valid(oracle.resize_linear_cv(src, sw, dwa, a.shape[0]), dwa))


# ------------------------------------------------------------------------------------------------------
# the reference's original two-frame refinement (variational.c:101) -- the oracle is pinned end to end against the real one
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h", [(67, 45), (130, 98), (1024, 436)])
@pytest.mark.parametrize("kw", [dict(), dict(delta=0.5, niter_outer=3, niter_inner=2), dict(alpha=3.0, gamma=0.2, niter_solver=7, sor_omega=1.5)])
def test_two_frame_variational(ctx, oracle, w, h, kw):
    rng = np.random.default_rng(w + h)
    big = smooth_noise_color(rng, w + 8, h + 8, 40)
    a, b = orc.aligned_zeros((3, h, orc.stride_of(w))), orc.aligned_zeros((3, h, orc.stride_of(w)))
    a[:, :, :w] = big[:, 4:4 + h, 4:4 + w]
    b[:, :, :w] = big[:, 3:3 + h, 2:2 + w]                       # translated by (2, 1)
    wx0, wy0 = noise_plane(rng, w, h, 1.5, 2.5), noise_plane(rng, w, h, 0.5, 1.5)
    wxo, wyo = orc.plane(*wx0.shape), orc.plane(*wx0.shape)
    wxo[...] = wx0; wyo[...] = wy0
    oracle.variational_2frame(wxo, wyo, a, b, w, orc.params_2f(**kw))
    po = orc.params_2f(**kw)
    pg = sfa.Params2f(po.alpha, po.gamma, po.delta, po.sigma, po.niter_outer, po.niter_inner, po.niter_solver, po.sor_omega)
    wxg, wyg = c_(wx0).copy(), c_(wy0).copy()
    ctx.variational_2frame(wxg, wyg, c_(a), c_(b), w, pg)
    assert np.array_equal(valid(wxo, w), valid(wxg, w)) and np.array_equal(valid(wyo, w), valid(wyg, w))
    if orc.ref_available():                                      # and directly against the compiled reference where it travelled along
        wxr, wyr = orc.plane(*wx0.shape), orc.plane(*wx0.shape)
        wxr[...] = wx0; wyr[...] = wy0
        orc.RefLib().variational_2frame(wxr, wyr, a, b, w, po)
        assert np.array_equal(valid(wxr, w), valid(wxg, w)) and np.array_equal(valid(wyr, w), valid(wyg, w))


@pytest.mark.parametrize("case", ["default", "color_inner", "weights"])
def test_two_frame_golden_gpu(ctx, case):
    """GPU against the committed outputs of the compiled reference's own variational() (no oracle in between)"""
    cases = {"default": dict(), "color_inner": dict(delta=0.5, niter_outer=3, niter_inner=2), "weights": dict(alpha=3.0, gamma=0.2, niter_solver=7, sor_omega=1.5)}
    T = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_two_frame.npz"))
    w, h = (int(v) for v in T["size"])
    po = orc.params_2f(**cases[case])
    pg = sfa.Params2f(po.alpha, po.gamma, po.delta, po.sigma, po.niter_outer, po.niter_inner, po.niter_solver, po.sor_omega)
    wx, wy = c_(T["wx0"]).copy(), c_(T["wy0"]).copy()
    ctx.variational_2frame(wx, wy, c_(T["im1"]), c_(T["im2"]), w, pg)
    assert np.array_equal(wx[:, :w], T[f"{case}_wx"]) and np.array_equal(wy[:, :w], T[f"{case}_wy"])


@pytest.mark.parametrize("w,h,p_scale,layers", [(130, 98, 0.9, 4), (200, 150, 0.75, 3), (97, 61, 0.5, 3), (1024, 436, 0.9, 5)])
def test_fused_pyramid_step_is_blur_then_resize(ctx, oracle, switches, w, h, p_scale, layers):
    """k_pyr_down (blur rows, blur columns, bilinear sample from an LDS tile) gives the frames of the two-kernel pyramid bit for
    bit: whole runs agree exactly, for several scale factors (tile footprints) and sizes off the tile grid"""
    frames, af, sf = normalized_frames(oracle, w, h, 3, seed=31)
    _, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, niter_outer=2, layers=layers, p_scale=p_scale)
    out = []
    for unfused in ("1", "0"):
        if unfused == "1":
            switches.set("SFA_PYRAMID_UNFUSED", "1")
        else:
            switches.unset("SFA_PYRAMID_UNFUSED")
        wx, wy = np.zeros((h, sfa.stride_of(w)), np.float32), np.zeros((h, sfa.stride_of(w)), np.float32)
        ctx.variational(ps, wx, wy, [c_(f) for f in frames], w, None)
        out.append((wx, wy))
    assert np.array_equal(valid(out[0][0], w), valid(out[1][0], w)) and np.array_equal(valid(out[0][1], w), valid(out[1][1], w))


@pytest.mark.parametrize("w,h,S,smoothing", [(130, 98, 2, 1), (67, 45, 3, 1), (200, 150, 2, 0), (1024, 436, 2, 1)])
def test_fused_warp_smoothness_is_the_two_kernels(ctx, oracle, switches, w, h, S, smoothing):
    """k_warp_smooth (the warps of get_derivatives and compute_smoothness of the same flow field in one pass; S = 3: six warps, four of them behind the
    smoothness arithmetic) gives the bits of k_warp_jobs + k_smoothness_tiled -- whole runs agree exactly -- and so does the one-job-per-grid-z form of the warps"""
    frames, af, sf = normalized_frames(oracle, w, h, 2 * S - 1, seed=37)
    _, ps = mk_params(oracle, S=S, rho=[1] * (S - 1), omega=[0] * (S - 1), norm_avg=af, norm_std=sf, niter_outer=3, layers=2, smoothing=smoothing)
    out = []
    for env in ({}, {"SFA_NO_WARP_SMOOTH": "1"}, {"SFA_NO_WARP_SMOOTH": "1", "SFA_WARP_ALLJ": "0"}):
        for k in ("SFA_NO_WARP_SMOOTH", "SFA_WARP_ALLJ"):
            switches.unset(k)
        for k, v in env.items():
            switches.set(k, v)
        wx, wy = np.zeros((h, sfa.stride_of(w)), np.float32), np.zeros((h, sfa.stride_of(w)), np.float32)
        ctx.variational(ps, wx, wy, [c_(f) for f in frames], w, None)
        out.append((wx, wy))
    for o in out[1:]:
        assert np.array_equal(valid(out[0][0], w), valid(o[0], w)) and np.array_equal(valid(out[0][1], w), valid(o[1], w))
    assert np.abs(valid(out[0][0], w)).max() > 0


@pytest.mark.parametrize("sigma", [0.3, 0.5, 0.8, 1.7])
def test_variational_with_presmoothing(ctx, oracle, sigma):
    """cfg `sigma` > 0: level 0 is presmoothed with gaussian_filter + the generic / 3-tap / 5-tap convolutions of image.c (the oracle's
    presmoothing is pinned bit-exact against the compiled image.c); orders 1 (sigma .3), 2 (.5), 3 (.8), 6 (1.7)"""
    w, h = 130, 98
    frames, af, sf = normalized_frames(oracle, w, h, 3, seed=13)
    po, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, layers=3, niter_outer=3, presmooth_sigma=sigma)
    o, g = run_both(ctx, oracle, po, ps, frames, w, h, level_only=False)
    d = max(np.abs(valid(o[0], w) - valid(g[0], w)).max(), np.abs(valid(o[1], w) - valid(g[1], w)).max())
    assert d <= TOL_UV, d


@pytest.mark.parametrize("w,h", [(67, 45), (130, 98)])
@pytest.mark.parametrize("sigma", [0.3, 0.5, 0.8, 1.0, 2.3])
def test_gaussian_presmooth_stage(ctx, oracle, w, h, sigma):
    rng = np.random.default_rng(int(sigma * 10) + w)
    src = noise_plane(rng, w, h, 0, 255)
    assert np.array_equal(valid(oracle.gaussian_presmooth(src, w, sigma), w), valid(ctx.gaussian_presmooth(c_(src), w, sigma), w))


def test_smoothness_full_size_exact(ctx, oracle):
    """~1.8 million psi' evaluations of the default penalty through the fast fp64 form with its exact fallback: every bit as the oracle"""
    w, h = 1024, 436
    rng = np.random.default_rng(77)
    for scale in (2.0, 1e-3, 40.0):
        uu, vv = noise_plane(rng, w, h, -scale, scale), noise_plane(rng, w, h, -scale, scale)
        dps = noise_plane(rng, w, h, 0.05, 0.5)
        a = oracle.smoothness(1, uu, vv, dps, w, 4.0, orc.Penalty(1, 0.001, 0.5))
        b = ctx.smoothness(1, c_(uu), c_(vv), c_(dps), w, 4.0, sfa.Penalty(1, 0.001, 0.5))
        for x, y in zip(a, b):
            assert np.array_equal(valid(x, w), valid(y, w))


# ------------------------------------------------------------------------------------------------------
# LABELLED MODE slow_flow_sor_order red_black: a different algorithm (never the default); parity is stated against its own CPU twin,
# the deviation from the reference order is measured and reported, not hidden
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h", [(67, 45), (64, 48), (130, 98), (2, 2), (1024, 436)])
@pytest.mark.parametrize("K", [1, 30])
def test_red_black_solver_against_its_cpu_twin(ctx, oracle, w, h, K):
    """the labelled mode's kernels (LDS halo tiles, 5 sweeps per tile visit) against the CPU twin, bit for bit"""
    rng = np.random.default_rng(w + h + K)
    s0 = sor_system(rng, w, h)
    s0["du"][:, :w] = rng.uniform(-.2, .2, (h, w)); s0["dv"][:, :w] = rng.uniform(-.2, .2, (h, w))
    a = copy_sys(s0)
    oracle.sor(a["du"], a["dv"], a["a11"], a["a12"], a["a22"], a["b1"], a["b2"], a["sh"], a["sv"], w, K, 1.9, red_black=True)
    b = {k: c_(v).copy() for k, v in s0.items()}
    ctx.sor_coupled(b["du"], b["dv"], b["a11"], b["a12"], b["a22"], b["b1"], b["b2"], b["sh"], b["sv"], w, K, 1.9, red_black=True)
    for k in ("du", "dv", "a11", "a12", "a22"):
        assert np.array_equal(valid(a[k], w), valid(b[k], w)), k


@pytest.mark.parametrize("w,h,K", [(300, 70, 30), (131, 97, 7), (64, 16, 5), (1024, 436, 30), (65, 17, 11)])
def test_red_black_tile_kernel_is_the_pass_kernel(ctx, oracle, w, h, K):
    """one launch per colour pass (round 2's form, SFA_RB_TILE=0) and the tile kernel -- 3 or 5 sweeps per visit, image borders inside the halos, visits that
    do not divide K, odd numbers of visits (the result comes home from the scratch pair) -- give the same bits.  The environment switch is read once per
    process, so the three forms run in child processes."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import slowflow_amd as sfa
        from synth import sor_system
        w, h, K = %d, %d, %d
        rng = np.random.default_rng(w + h + K)
        s = sor_system(rng, w, h)
        s["du"][:, :w] = rng.uniform(-.2, .2, (h, w)); s["dv"][:, :w] = rng.uniform(-.2, .2, (h, w))
        b = {k: np.ascontiguousarray(v).copy() for k, v in s.items()}
        ctx = sfa.Context(0)
        ctx.sor_coupled(b["du"], b["dv"], b["a11"], b["a12"], b["a22"], b["b1"], b["b2"], b["sh"], b["sv"], w, K, 1.9, red_black=True)
        np.save(sys.argv[1], np.stack([b["du"][:, :w], b["dv"][:, :w]]))
    """) % (ROOT, os.path.join(ROOT, "tests"), w, h, K)
    import tempfile
    outs = []
    with tempfile.TemporaryDirectory() as d:
        for mode in ("0", "3", "5"):
            f = os.path.join(d, "rb_%s.npy" % mode)
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, SFA_DEBUG="1", SFA_RB_TILE=mode), capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr
            outs.append(np.load(f))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_red_black_mode_is_labelled_and_deviates(ctx, oracle):
    """the whole path with sor_order = 1: GPU == the oracle run in the same mode (<= 1e-4), and BOTH are measurably away from the reference order --
    the number that every results line of this mode carries"""
    w, h = 130, 98
    frames, af, sf = normalized_frames(oracle, w, h, 3, seed=5)
    po, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, niter_outer=3, layers=3)
    lex_o, lex_g = run_both(ctx, oracle, po, ps, frames, w, h, level_only=False)
    po.sor_order = 1; ps.sor_order = 1
    rb_o, rb_g = run_both(ctx, oracle, po, ps, frames, w, h, level_only=False)
    d_twin = max(np.abs(valid(rb_o[0], w) - valid(rb_g[0], w)).max(), np.abs(valid(rb_o[1], w) - valid(rb_g[1], w)).max())
    d_ref = max(np.abs(valid(lex_g[0], w) - valid(rb_g[0], w)).max(), np.abs(valid(lex_g[1], w) - valid(rb_g[1], w)).max())
    assert d_twin <= TOL_UV, d_twin
    assert d_ref > TOL_UV, d_ref                                   # it does NOT meet the reference's 1e-4: a different algorithm
    assert abs(np.median(valid(rb_g[0], w)) - 2.0) < 0.2           # still a flow estimate of the same motion
    # batches and thresholds go through the same mode
    ps.thres_outer = 1e-3
    job = sfa.Job(ctx, ps, w, h, 3)
    for b in range(3):
        job.upload(b, [c_(f) for f in frames])
    job.run()
    one = sfa.Job(ctx, ps, w, h, 1)
    one.upload(0, [c_(f) for f in frames]); one.run()
    ref = one.download(0)
    for b in range(3):
        got = job.download(b)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    job.close(); one.close()


def test_resident_sequence_is_normalize_plus_upload(ctx, oracle):
    """sfa_sequence (frames sent to the GPU once, normalised there, copied device-to-device into jobs) gives the statistics, the frames and the flow of
    the host-plane route: sfa_normalize + sfa_job_upload"""
    w, h, n = 130, 98, 5
    frames = [c_(texture_frame(w, h, k)) for k in range(n)]
    host = [f.copy() for f in frames]
    avg_h, std_h = ctx.normalize(host, w)
    seq = sfa.Sequence(ctx, w, h, n)
    for f in range(n):
        seq.upload(f, frames[f])
    avg_d, std_d = seq.normalize()
    assert avg_d == avg_h and std_d == std_h
    for f in range(n):
        assert np.array_equal(seq.download(f)[:, :, :w], host[f][:, :, :w])
    _, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=[float("%g" % a) for a in avg_h], norm_std=[float("%g" % s) for s in std_h], layers=2, niter_outer=2)
    a = sfa.Job(ctx, ps, w, h, 2)
    a.upload(0, host[0:3]); a.upload(1, host[4:1:-1])
    a.run()
    b = sfa.Job(ctx, ps, w, h, 2)
    b.upload_resident(0, seq, [0, 1, 2]); b.upload_resident(1, seq, [4, 3, 2])
    b.run()
    for e in (0, 1):
        x, y = a.download(e), b.download(e)
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
    # statistics over a sub-range (the reference's -jet k mode normalises over that jet's frames only)
    seq2 = sfa.Sequence(ctx, w, h, n)
    for f in range(n):
        seq2.upload(f, frames[f])
    sub = [f.copy() for f in frames[1:4]]
    got, want = seq2.normalize(1, 3), ctx.normalize(sub, w)
    assert list(got[0]) == list(want[0]) and list(got[1]) == list(want[1])
    assert np.array_equal(seq2.download(2)[:, :, :w], sub[1][:, :, :w]) and np.array_equal(seq2.download(0)[:, :, :w], frames[0][:, :, :w])   # frames outside the range untouched
    a.close(); b.close(); seq.close(); seq2.close()


def test_sequence_sharded_over_two_contexts_normalises_like_one(ctx, oracle):
    """the multi-GPU driver's ingest: every GPU holds a slice of the frames (here two contexts, frames 0..3 and 2..5: a halo of two), forms the
    per-frame sums of what it holds, the statistics come from the sums of ALL frames in frame order on the host, each slice is normalised with them -- the frames and the
    published statistics are those of one sequence normalised as a whole, bit for bit; a halo frame gets the same sums from both holders"""
    w, h, n = 130, 98, 6
    frames = [c_(texture_frame(w, h, k)) for k in range(n)]
    whole = sfa.Sequence(ctx, w, h, n)
    for f in range(n):
        whole.upload(f, frames[f])
    raw_sums = whole.frame_sums()
    avg_w, std_w = whole.normalize()
    other = sfa.Context(0)
    try:
        parts = [(ctx, 0, 4), (other, 2, 6)]
        seqs, sums = [], np.zeros((n, 6))
        have = np.zeros(n, bool)
        for c, lo, hi in parts:
            q = sfa.Sequence(c, w, h, hi - lo)
            for f in range(lo, hi):
                q.upload(f - lo, frames[f])
            mine = q.frame_sums()
            for f in range(lo, hi):
                if have[f]:
                    assert np.array_equal(sums[f], mine[f - lo])          # the halo: the same bits from both holders
                sums[f] = mine[f - lo]; have[f] = True
            seqs.append(q)
        assert np.array_equal(sums, raw_sums)
        avg, std = sfa.normalize_statistics(sums, w, h)
        assert avg == avg_w and std == std_w
        for (c, lo, hi), q in zip(parts, seqs):
            q.apply_normalization(avg, std)
            for f in range(lo, hi):
                assert np.array_equal(q.download(f - lo)[:, :, :w], whole.download(f)[:, :, :w])
    finally:
        other.close()


def test_debug_switch_hook_rejects_unknown_names(ctx):
    """sfa_debug_set knows the library's switches by name (sfa_internal.h: Switches); anything else is an error, not a silently ignored variable"""
    sfa.debug_set("SFA_SOR_CHAIN", 11); sfa.debug_set("SFA_SOR_CHAIN", None)
    with pytest.raises(sfa.SlowflowError):
        sfa.debug_set("SFA_NO_SUCH_SWITCH", 1)


@pytest.mark.gpu
def test_objects_may_be_finalised_in_any_order():
    """a context finalised before the jobs created on it (cyclic garbage, interpreter shutdown) must neither crash nor hang: close() on the context
    destroys its children first, and a child whose context is already gone does not call into the library"""
    import gc
    import subprocess
    import sys
    code = r'''
import sys, gc
sys.path.insert(0, %r)
import numpy as np, slowflow_amd as sfa
ctx = sfa.Context(0)
p = sfa.default_params(); p.layers = 1; p.niter_alter = 1; p.niter_outer = 1
job = sfa.Job(ctx, p, 64, 48, 2)
seq = sfa.Sequence(ctx, 64, 48, 3)
sb = sfa.SorBatch(ctx, 64, 48, 1)
ctx.close()                      # children first
assert not job.h_ and not seq.h_ and not sb.h_
job.close(); seq.close(); sb.close()
# the garbage collector's order: a cycle through the objects, context finalised first by hand
ctx2 = sfa.Context(0)
job2 = sfa.Job(ctx2, p, 64, 48, 1)
cyc = [ctx2, job2]; cyc.append(cyc)
sfa.lib().sfa_ctx_destroy(ctx2.h); ctx2.h = sfa.C.c_void_p()     # what a finaliser run out of order amounts to
del ctx2, job2, cyc
gc.collect()
# and objects simply left to the interpreter's shutdown, referenced from a cycle
ctx3 = sfa.Context(0)
jobs = [sfa.Job(ctx3, p, 64, 48, 1) for _ in range(2)]
def run_all():
    return [j for j in jobs], ctx3
run_all.self = run_all
print("alive")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "alive" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_boundary_rejects_bad_arguments_without_exiting(ctx):
    """the reference exit()s or throws on bad input (image.c:19-30, variational_aux_mt.cpp:419); the C-ABI returns a negative status with a
    message and the process lives on -- the context stays usable afterwards"""
    L = sfa.lib()
    C = sfa.C
    p = sfa.default_params()
    job = C.c_void_p()
    for (w, h, batch) in [(1, 48, 1), (64, 4, 1), (64, 48, 0), (64, 48, 129)]:
        assert L.sfa_job_create(ctx.h, C.byref(p), w, h, batch, C.byref(job)) == -1
        assert b"bad arguments" in L.sfa_last_error(ctx.h)
    assert L.sfa_job_create(None, C.byref(p), 64, 48, 1, C.byref(job)) == -1
    bad = sfa.default_params(); bad.S = 1                                        # a window needs at least two frames each side of ... S >= 2
    rc = L.sfa_job_create(ctx.h, C.byref(bad), 64, 48, 1, C.byref(job))
    assert rc < 0 and L.sfa_last_error(ctx.h)
    p.layers = 1; p.niter_alter = 1; p.niter_outer = 1
    j = sfa.Job(ctx, p, 64, 48, 2)
    stride = sfa.stride_of(64)
    fr = [np.zeros((3, 48, stride), np.float32) for _ in range(3)]
    with pytest.raises(sfa.SlowflowError):
        j.upload(0, fr[:2])                                                          # 2 frames where S = 2 needs 3
    with pytest.raises(sfa.SlowflowError):
        j.upload(2, fr)                                                              # window index out of range
    with pytest.raises(sfa.SlowflowError):
        j.download(-1)
    du = np.zeros((48, stride), np.float32)
    img = sfa.Image(64, 48, stride, sfa.fptr(du))
    nul = sfa.Image(64, 48, stride, None)                                            # a plane without data
    args = [C.byref(img)] * 9
    assert L.sfa_sor_coupled(ctx.h, None, *args[1:], 30, C.c_float(1.9)) < 0        # null image
    assert L.sfa_sor_coupled(ctx.h, C.byref(nul), *args[1:], 30, C.c_float(1.9)) < 0
    # still alive and working
    for f in fr:
        f[:, :, :64] = np.random.default_rng(0).uniform(0, 255, (3, 48, 64)).astype(np.float32)
    j.upload(0, fr); j.upload(1, fr)
    j.run()
    wx, wy, _ = j.download(1)
    assert np.isfinite(wx).all() and np.isfinite(wy).all()
    j.close()


def test_shared_reciprocal_division_is_ieee(ctx):
    """The cfg-default assembly kernel divides by a denominator two quotients share through one refined reciprocal (kernels.hip: recip_of / div_by), behind
    range guards; wherever the guards admit a pair the chain must BE the IEEE quotient.  Numerators: +0, every biased exponent around both guard edges
    (2^-87, 2^53) with random and extreme mantissas, squares of image-like values; denominators: 0.01 + sums of squares, the edges 2^-27 .. 2^33, exact
    powers of two and their neighbours.  Also: the guards do admit the values the data terms produce (otherwise the fast path would be dead code), and
    reject -0, negative, tiny and huge numerators."""
    rng = np.random.default_rng(7)
    def with_exp(e, n):                                   # n floats with biased exponent e, random mantissa
        return ((np.uint32(e) << np.uint32(23)) | rng.integers(0, 1 << 23, n, dtype=np.uint32)).view(np.float32)
    nums = [np.zeros(64, np.float32)]
    for e in list(range(30, 50)) + list(range(170, 190)) + list(range(100, 150, 7)):
        v = with_exp(e, 4096)
        v[:4] = np.array([e << 23, (e << 23) | 0x7fffff, (e << 23) | 1, (e << 23) | 0x400000], np.uint32).view(np.float32)
        nums.append(v)
    img = rng.normal(0, 1.5, 200000).astype(np.float32)
    nums.append(img * img)
    nums.append((rng.uniform(0, 500, 50000)).astype(np.float32))        # weights t
    nums = np.concatenate(nums)
    dens = [np.float32(0.01) + (rng.normal(0, 2, 100000).astype(np.float32)) ** 2 + (rng.normal(0, 2, 100000).astype(np.float32)) ** 2]
    for e in list(range(96, 104)) + list(range(118, 165)):
        v = with_exp(e, 512)
        v[:4] = np.array([e << 23, (e << 23) | 0x7fffff, (e << 23) | 1, (e << 23) | 0x400000], np.uint32).view(np.float32)
        dens.append(v)
    dens = np.concatenate(dens)
    n = 6_000_000
    a = nums[rng.integers(0, nums.size, n)]
    b = dens[rng.integers(0, dens.size, n)]
    # every edge numerator against every edge denominator as well
    ea = np.concatenate([with_exp(e, 4) for e in range(36, 46)] + [with_exp(e, 4) for e in range(176, 184)] + [np.zeros(1, np.float32)])
    eb = np.concatenate([with_exp(e, 4) for e in range(98, 102)] + [with_exp(e, 4) for e in range(156, 162)] + [np.float32([0.01, 0.0100001, 1.0, 3.0])])
    ga, gb = np.meshgrid(ea, eb)
    a = np.concatenate([a, ga.ravel().astype(np.float32)]); b = np.concatenate([b, gb.ravel().astype(np.float32)])
    qc, qe, ad = ctx.division_chain(a, b)
    adm = ad.astype(bool)
    assert np.array_equal(qe.view(np.uint32), (a.astype(np.float32) / b.astype(np.float32)).view(np.uint32)), "the GPU's own __fdiv_rn is the IEEE quotient"
    bad = adm & (qc.view(np.uint32) != qe.view(np.uint32))
    assert not bad.any(), f"{int(bad.sum())} admitted pairs differ, e.g. {a[bad][:3]} / {b[bad][:3]}"
    assert adm.mean() > 0.5                                # the guards admit the bulk of the sample
    typical = (a >= 1e-12) & (a <= 1e6) & (b >= 0.01) & (b <= 1e6)
    assert adm[typical].all(), "values of the size the data terms produce take the fast path"
    # rejected by construction
    rej_a = np.float32([-0.0, -1.0, 2.0 ** -100, 2.0 ** 60, np.inf])
    _, _, ad2 = ctx.division_chain(rej_a, np.full(rej_a.size, 1.0, np.float32))
    assert not ad2.any()
    _, _, ad3 = ctx.division_chain(np.float32([1.0, 1.0]), np.float32([2.0 ** 40, 2.0 ** -30]))
    assert not ad3.any()


# ------------------------------------------------------------------------------------------------------
# round 4: the reference's real schedule at the metric's size, the edges of the accepted parameter range, the timeout path
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["cfg_1e-5", "at_the_measured_change"])
def test_cfg_schedule_at_the_metric_size_with_thresholds_and_occlusions(ctx, oracle, mode):
    """1024x436, S = 3 (5 frames), rho 1/1, omega 0/2, break thresholds ON, occlusion reasoning ON -- cfgs/slow_flow.cfg's terms at BASELINE's size, reduced to
    what the CPU restatement finishes in about a minute: one level, 2 alternations x 4 outer iterations, started from a flow that is already close to the fixed
    point so that the outer loop MEETS its threshold (variational_mt.cpp:436).  The reference sums the change norms in fp32, one quad after the other
    (:376-402, 111 616 addends at this size); the GPU takes fp64 tree sums -- here is the size at which that difference is largest.
      cfg_1e-5:               thresholds 1e-5 as in the cfg
      at_the_measured_change: the outer threshold set 0.1 % above the change the GPU itself measures at the second outer iteration -- the comparison the two
                              summation orders are most likely to decide differently
    Same protocol as test_level_with_occlusion_reasoning (the GPU's labels forced on the oracle): the oracle must stop where the GPU stopped (its last change
    norms are the GPU's to 1e-3 relative; one outer iteration more or less moves them by tens of percent) and (u, v) must agree to 2e-5."""
    w, h, S, A = 1024, 436, 3, 2
    stride = orc.stride_of(w)
    import bench                                                  # the bench's own stand-in for config 2: band-limited texture under a smooth, non-constant flow of <= 3 px per frame
    frames = []
    for f in bench.synth_window(300, w=w, h=h, n=2 * S - 1):
        g_ = orc.aligned_zeros(f.shape); g_[...] = f
        frames.append(g_)
    _, _, af, sf = oracle.normalize(frames, w)
    base = dict(S=S, rho=[1, 1], omega=[0, 2], norm_avg=af, norm_std=sf, layers=1)
    # a flow near the fixed point: 3 x 10 iterations on the GPU without thresholds
    _, pw = mk_params(oracle, niter_outer=10, niter_alter=3, occlusion_reasoning=1, **base)
    job = sfa.Job(ctx, pw, w, h, 1)
    job.upload(0, [c_(f) for f in frames]); job.run()
    wx0, wy0, _ = job.download(0)
    job.close()
    thres = 1e-5
    if mode == "at_the_measured_change":
        _, pp = mk_params(oracle, niter_outer=2, niter_alter=1, occlusion_reasoning=1, **base)
        job = sfa.Job(ctx, pp, w, h, 1)
        job.upload(0, [c_(f) for f in frames], wx0, wy0); job.run()
        _, _, ch = job.download(0)
        job.close()
        thres = float(np.float32(max(ch) * 1.001))
    po, ps = mk_params(oracle, niter_outer=4, niter_alter=A, occlusion_reasoning=1, thres_outer=thres, thres_inner=1e-5, **base)
    job = sfa.Job(ctx, ps, w, h, 1)
    job.keep_alternation_occlusions(True)
    job.upload(0, [c_(f) for f in frames], wx0, wy0)
    job.run()
    wxg, wyg, chg = job.download(0)
    labels = orc.aligned_zeros((A, h, stride))
    for a in range(1, A):
        labels[a] = job.download_alternation_occlusions(0, a)
    occ_g = job.download_occlusions(0)
    job.close()
    oracle.force_labels(labels)
    try:
        wxo, wyo = orc.plane(h, stride), orc.plane(h, stride)
        wxo[...] = wx0; wyo[...] = wy0
        rc, cho, occ_o = oracle.compute_one_level(po, wxo, wyo, frames, w, None, want_occ=True)
        gaps = [oracle.forced_gap(a) for a in range(1, A)]
    finally:
        oracle.force_labels(None)
    print(f"[{mode}] threshold {thres:.6g}; last change norms oracle {cho} GPU {chg}; forced-label energy gaps {gaps}")
    assert rc == 0 and np.array_equal(valid(occ_o, w), valid(occ_g, w))
    assert max(abs(g) for g in gaps) <= 1e-5 * max(1.0, w * h * 1e-3), gaps                   # the GPU's labelling is a minimum of the oracle's energy
    assert abs(cho[0] - chg[0]) <= 1e-3 * cho[0] and abs(cho[1] - chg[1]) <= 1e-3 * cho[1], (mode, thres, cho, chg)   # stopped at the same iteration
    if mode == "at_the_measured_change":
        assert max(chg) < thres                                                                # ... and the loop did stop on the threshold
    d = max(np.abs(valid(wxo, w) - valid(wxg, w)).max(), np.abs(valid(wyo, w) - valid(wyg, w)).max())
    assert d <= TOL_LEVEL, (mode, d)


@pytest.mark.parametrize("S,rho,omega", [(4, [1, 0.5, 0.25], [0, 2, 1]), (5, [1, 1, 0.5, 0.5], [0.5, 0, 1, 2]),
                                         (6, [1, 1, 0.5, 0.5, 0.25], [0.5, 0, 1, 2, 1]), (9, [1, 1, 0.5, 0.5, 0.25, 0.25, 0.125, 0.125], [0.5, 0, 1, 2, 1, 0, 0.5, 1])])
def test_level_with_seven_and_nine_frames(ctx, oracle, S, rho, omega):
    """slow_flow_S = 4, 5, 6 and 9 (7 / 9 / 11 / 17 frames: SFA_MAX_REF = 8 is the widest window the boundary accepts since round 5, 4 before): to-reference terms up to
    eight frames from the reference frame, the staged pairs of an odd and an even number of terms, up to 32 terms per assembly"""
    w, h = 96, 64
    frames, af, sf = normalized_frames(oracle, w, h, 2 * S - 1, seed=5)
    po, ps = mk_params(oracle, S=S, rho=rho, omega=omega, norm_avg=af, norm_std=sf, niter_outer=2)
    o, g = run_both(ctx, oracle, po, ps, frames, w, h)
    d = max(np.abs(valid(o[0], w) - valid(g[0], w)).max(), np.abs(valid(o[1], w) - valid(g[1], w)).max())
    assert d <= max(TOL_LEVEL, 3 * oracle_sensitivity(oracle, po, frames, w, h)), (S, d)
    _, bad = mk_params(oracle, S=10, rho=[1], omega=[0], norm_avg=af, norm_std=sf)
    with pytest.raises(sfa.SlowflowError):                                                    # one more than SFA_MAX_REF + 1 is refused, not truncated
        ctx.compute_one_level(bad, np.zeros((h, sfa.stride_of(w)), np.float32), np.zeros((h, sfa.stride_of(w)), np.float32), ([c_(f) for f in frames] * 19)[:19], w)


def test_unknown_penalty_id_is_the_default_class(ctx, oracle):
    """select_robust_function (variational_aux_mt.cpp:909-925) maps every id it does not know to the modified L1 norm: ids 7 / 9 / -3 must give the bits of id 1
    on the GPU, and the oracle's flow within the level tolerance"""
    w, h = 67, 45
    frames, af, sf = normalized_frames(oracle, w, h, 5, seed=7)
    outs = []
    for ids in ((1, 1, 1), (7, 9, -3)):
        kw = dict(robust_color=(ids[0], 0.001, 0.5), robust_grad=(ids[1], 0.001, 0.5), robust_reg=(ids[2], 0.001, 0.5))
        po, ps = mk_params(oracle, S=3, rho=[1, 0.5], omega=[0.5, 2], norm_avg=af, norm_std=sf, niter_outer=2, **kw)
        outs.append(run_both(ctx, oracle, po, ps, frames, w, h))
    (o1, g1), (o7, g7) = outs
    assert np.array_equal(valid(g1[0], w), valid(g7[0], w)) and np.array_equal(valid(g1[1], w), valid(g7[1], w))
    assert np.array_equal(valid(o1[0], w), valid(o7[0], w))                                     # the oracle maps them the same way
    d = max(np.abs(valid(o7[0], w) - valid(g7[0], w)).max(), np.abs(valid(o7[1], w) - valid(g7[1], w)).max())
    assert d <= TOL_LEVEL, d


def test_poisoned_solve_returns_timeout_and_the_context_keeps_working(oracle):
    """Fault injection (sfa_ctx_set_wait_bound): with a bound of one poll the first wait of a band for the band above gives up, poisons the launch and every
    other wait follows; the kernel drains, the entry point returns SFA_ERR_TIMEOUT (-5) -- for the stand-alone solver, for a whole refinement and for a batch --
    and with the default bound restored the same context produces the reference's bits again."""
    c = sfa.Context(0)
    try:
        w, h = 300, 200                                          # 4 bands: bands 1.. wait for the band above
        s0 = sor_system(np.random.default_rng(3), w, h)
        ref = copy_sys(s0)
        oracle.sor(ref["du"], ref["dv"], ref["a11"], ref["a12"], ref["a22"], ref["b1"], ref["b2"], ref["sh"], ref["sv"], w, 30, 1.9)

        def solve():
            b = {k: np.ascontiguousarray(v).copy() for k, v in s0.items()}
            c.sor_coupled(b["du"], b["dv"], b["a11"], b["a12"], b["a22"], b["b1"], b["b2"], b["sh"], b["sv"], w, 30, 1.9)
            return b
        c.set_wait_bound(1)
        with pytest.raises(sfa.SlowflowError, match="-5"):
            solve()
        assert "wait gave up" in sfa.lib().sfa_last_error(c.h).decode()
        c.set_wait_bound(0)
        b = solve()
        assert np.array_equal(ref["du"][:, :w], b["du"][:, :w]) and np.array_equal(ref["dv"][:, :w], b["dv"][:, :w])
        # a whole refinement of a batch: the error comes back from the run, the job is usable afterwards
        fw, fh = 200, 150
        frames, af, sf = normalized_frames(oracle, fw, fh, 3, seed=2)
        _, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, niter_outer=2, layers=2)
        job = sfa.Job(c, ps, fw, fh, 3)
        for k in range(3):
            job.upload(k, [c_(f) for f in frames])
        job.run()
        good = job.download(1)
        c.set_wait_bound(1)
        with pytest.raises(sfa.SlowflowError, match="-5"):
            job.run(); c.sync()
        c.set_wait_bound(0)
        for k in range(3):
            job.upload(k, [c_(f) for f in frames])
        job.run()
        again = job.download(1)
        assert np.array_equal(good[0], again[0]) and np.array_equal(good[1], again[1])
        job.close()
    finally:
        c.close()


def test_verbose_change_lines(oracle, capfd, switches):
    """the reference prints "inner it i avg change a,b" / "outer it i avg change a,b" per iteration under verbosity(VER_CMD) (variational_mt.cpp:404-405, 431-432); the
    library prints the same lines for a context with sfa_ctx_set_verbose (the C++ class and the driver set it from the cfg's `verbose`): every line carries the oracle's
    value (the |old_du - du| norm on the inner lines, the last inner iteration included; fp32 raster sums there, fp64 tree sums here), the last outer line the change
    norms the call returns, and the flow is the flow of the quiet run bit for bit"""
    w, h = 67, 45
    frames, af, sf = normalized_frames(oracle, w, h, 3)
    results = {}
    for niter_inner in (2, 1):
        po, ps = mk_params(oracle, S=2, rho=[1], omega=[0], norm_avg=af, norm_std=sf, niter_outer=3, niter_inner=niter_inner)
        oracle.change_log(64)
        wxo, wyo = orc.plane(h, orc.stride_of(w)), orc.plane(h, orc.stride_of(w))
        rc, _, _ = oracle.compute_one_level(po, wxo, wyo, frames, w)
        ref = oracle.change_log_rows().copy()
        oracle.change_log(0)
        assert rc == 0 and len(ref) == 3 * niter_inner + 3
        c = sfa.Context(0)
        try:
            flows = []
            for verbose in (True, False):
                c.set_verbose(verbose)
                wx, wy = np.zeros((h, sfa.stride_of(w)), np.float32), np.zeros((h, sfa.stride_of(w)), np.float32)
                capfd.readouterr()
                ch, _ = c.compute_one_level(ps, wx, wy, [c_(f) for f in frames], w)
                out = capfd.readouterr().out.splitlines()
                flows.append((wx, wy))
                if not verbose:
                    assert not [l for l in out if "avg change" in l]
                    continue
                lines = [l for l in out if "avg change" in l]
                assert len(lines) == len(ref), out
                for l, (kind, it, ra, rb) in zip(lines, ref):
                    assert l.startswith(("\tinner it %d\tavg change " if kind == 0 else "outer it %d\tavg change ") % int(it)), (l, kind, it)
                    a, b = (float(x) for x in l.split("avg change ")[1].split(","))
                    assert abs(a - ra) <= 2e-4 * abs(ra) + 1e-9 and abs(b - rb) <= 2e-4 * abs(rb) + 1e-9, (l, ra, rb)
                a, b = (float(x) for x in lines[-1].split("avg change ")[1].split(","))
                assert abs(a - ch[0]) <= 1e-5 * abs(ch[0]) + 1e-12 and abs(b - ch[1]) <= 1e-5 * abs(ch[1]) + 1e-12
            assert np.array_equal(flows[0][0], flows[1][0]) and np.array_equal(flows[0][1], flows[1][1])
        finally:
            c.close()


@pytest.mark.parametrize("pen", [1, 2])
def test_assembly_instances_and_tile_orders_give_the_same_bits(ctx, oracle, switches, pen):
    """The fused assembly kernel's variants -- the folded instances (modified L1 / Lorentzian) against the run-time instance (SFA_ASSEMBLE_GENERIC), the
    shared-reciprocal divisions against __fdiv_rn only (SFA_EXACT_DIV), the XCD-contiguous tile order against the plain grid (SFA_ASM_XCD=0) -- on a batch whose
    launch is large enough for the XCD order (9 windows of 300x200: 675 tiles), S = 3 with to-reference terms, sizes off the tile grid: one set of bits."""
    w, h, nb = 300, 200, 9
    frames, af, sf = normalized_frames(oracle, w, h, 5, seed=21)
    kw = dict(robust_color=(pen, 0.05 if pen == 2 else 0.001, 0.5), robust_grad=(pen, 0.05 if pen == 2 else 0.001, 0.5))
    _, ps = mk_params(oracle, S=3, rho=[1, 0.5], omega=[0.5, 2], norm_avg=af, norm_std=sf, niter_outer=2, layers=2, **kw)
    outs = {}
    for name, env in (("default", {}), ("generic", {"SFA_ASSEMBLE_GENERIC": "1"}), ("exact_div", {"SFA_EXACT_DIV": "1"}), ("plain_grid", {"SFA_ASM_XCD": "0"})):
        for k in ("SFA_ASSEMBLE_GENERIC", "SFA_EXACT_DIV", "SFA_ASM_XCD"):
            switches.unset(k)
        for k, v in env.items():
            switches.set(k, v)
        job = sfa.Job(ctx, ps, w, h, nb)
        for b in range(nb):
            job.upload(b, [c_(np.roll(f, 3 * b, axis=2)) for f in frames])
        job.run()
        outs[name] = [job.download(b)[:2] for b in (0, 4, 8)]
        job.close()
    for name in ("generic", "exact_div", "plain_grid"):
        for (ax, ay), (bx, by) in zip(outs["default"], outs[name]):
            assert np.array_equal(ax, bx) and np.array_equal(ay, by), name
    assert not np.array_equal(outs["default"][0][0], outs["default"][1][0])                 # the windows do differ
