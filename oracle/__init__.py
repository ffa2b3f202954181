"""ctypes bindings of the TEST-ONLY checkers.

* ``Oracle``  -- oracle/libslowflow_oracle.so, our CPU restatement (oracle/slowflow_oracle.c).
* ``RefLib``  -- oracle/_ref/libslowflow_ref.so, the reference's own C sources compiled by
  oracle/Makefile (present only where it was built or shipped as a prebuilt file).

TEST INFRASTRUCTURE: only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline``
leg may import this package.  The product (slowflow_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.environ.get("SFA_ORACLE_SO") or os.path.join(_HERE, "libslowflow_oracle.so")   # SFA_ORACLE_SO: e.g. the -fsanitize build (make -C oracle asan)
REF_SO = os.path.join(_HERE, "_ref", "libslowflow_ref.so")
MAX_REF = 8

_f = C.POINTER(C.c_float)


def build(quiet=True):
    """(Re)build the checkers with oracle/Makefile.  Building the checker is not using it."""
    out = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


def stride_of(w):
    """image.c:25"""
    return ((w + 3) // 4) * 4


def aligned_zeros(shape, align=64):
    n = int(np.prod(shape))
    buf = np.zeros(n + align // 4, dtype=np.float32)
    off = (-buf.ctypes.data % align) // 4
    return buf[off:off + n].reshape(shape)


def plane(h, stride, fill=None):
    a = aligned_zeros((h, stride))
    if fill is not None:
        a[...] = fill
    return a


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_f)


class Penalty(C.Structure):
    _fields_ = [("id", C.c_int), ("eps", C.c_float), ("trunc", C.c_float)]


class Params(C.Structure):
    _fields_ = [
        ("S", C.c_int), ("one_direction", C.c_int), ("smoothing", C.c_int), ("dataterm_norm", C.c_int),
        ("niter_alter", C.c_int), ("niter_outer", C.c_int), ("niter_inner", C.c_int), ("niter_solver", C.c_int),
        ("thres_outer", C.c_float), ("thres_inner", C.c_float), ("sor_omega", C.c_float),
        ("alpha", C.c_float), ("gamma", C.c_float), ("delta", C.c_float),
        ("robust_color", Penalty), ("robust_grad", Penalty), ("robust_reg", Penalty),
        ("rho", C.c_float * MAX_REF), ("omega", C.c_float * MAX_REF),
        ("hbit", C.c_int), ("norm_avg", C.c_float * 3), ("norm_std", C.c_float * 3),
        ("occlusion_reasoning", C.c_int), ("layers", C.c_int), ("p_scale", C.c_float), ("presmooth_sigma", C.c_float),
        ("occlusion_penalty", C.c_float), ("occlusion_alpha", C.c_float), ("niter_graphc", C.c_int), ("sor_order", C.c_int),
    ]


class Oracle:
    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            build()
        self.lib = L = C.CDLL(path)
        L.orc_psi_deriv_scalar.restype = C.c_float
        L.orc_psi_deriv_scalar.argtypes = [C.POINTER(Penalty), C.c_float]
        L.orc_psi_deriv_vec.restype = C.c_float
        L.orc_psi_deriv_vec.argtypes = [C.POINTER(Penalty), C.c_float]
        L.orc_compute_one_level.restype = C.c_int
        L.orc_variational.restype = C.c_int
        L.orc_add_data_and_match_ref.restype = C.c_int
        L.orc_pyramid_sizes.restype = C.c_int

    def default_params(self):
        p = Params()
        self.lib.orc_params_default(C.byref(p))
        return p

    def psi_deriv(self, pen, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        s = np.array([self.lib.orc_psi_deriv_scalar(C.byref(pen), C.c_float(v)) for v in x.ravel()], dtype=np.float32)
        v = np.array([self.lib.orc_psi_deriv_vec(C.byref(pen), C.c_float(v)) for v in x.ravel()], dtype=np.float32)
        return s.reshape(x.shape), v.reshape(x.shape)

    def psi_apply(self, pen, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        self.lib.orc_psi_apply_vec.restype = C.c_float
        v = np.array([self.lib.orc_psi_apply_vec(C.byref(pen), C.c_float(t)) for t in x.ravel()], dtype=np.float32)
        return v.reshape(x.shape)

    def occlusion_costs(self, masks, succ, toref, ref, rho, omega, hd, hg, penalty, color, grad, w):
        """masks (2ref,h,stride), succ/toref (2ref,8,3,h,stride) -> d0, d1"""
        _, h, stride = masks.shape
        d0, d1 = plane(h, stride), plane(h, stride)
        r = (C.c_float * len(rho))(*rho); o = (C.c_float * len(omega))(*omega)
        self.lib.orc_occlusion_costs(fptr(d0), fptr(d1), fptr(masks), fptr(succ), fptr(toref), int(ref), r, o, C.c_float(hd), C.c_float(hg),
                                     C.c_float(penalty), C.byref(color), C.byref(grad), w, h, stride)
        return d0, d1

    def variational_2frame(self, wx, wy, im1, im2, w, p):
        h, stride = wx.shape
        self.lib.orc_variational_2frame(fptr(wx), fptr(wy), fptr(im1), fptr(im2), C.byref(p), w, h, stride)

    def gaussian_presmooth(self, src, w, sigma):
        h, stride = src.shape
        dst = plane(h, stride)
        self.lib.orc_gaussian_presmooth(fptr(dst), fptr(src), w, h, stride, C.c_float(sigma))
        return dst

    def grid_cut(self, d0, d1, alpha, w):
        h, stride = d0.shape
        occ = plane(h, stride)
        self.lib.orc_grid_cut.restype = C.c_double
        e = self.lib.orc_grid_cut(fptr(occ), fptr(d0), fptr(d1), C.c_float(alpha), w, h, stride)
        return occ, e

    def grid_cut_energy(self, occ, d0, d1, alpha, w):
        h, stride = d0.shape
        self.lib.orc_grid_cut_energy.restype = C.c_double
        return self.lib.orc_grid_cut_energy(fptr(occ), fptr(d0), fptr(d1), C.c_float(alpha), w, h, stride)

    def convolve(self, src, w, order, horiz):
        h, stride = src.shape
        dst = plane(h, stride)
        fn = self.lib.orc_convolve_horiz if horiz else self.lib.orc_convolve_vert
        fn(fptr(dst), fptr(src), w, h, stride, order)
        return dst

    def image_warp(self, src3, wx, wy, w, factor, want_mask=True):
        _, h, stride = src3.shape
        dst = aligned_zeros((3, h, stride))
        mask = plane(h, stride) if want_mask else None
        self.lib.orc_image_warp(fptr(dst), fptr(mask) if want_mask else None, fptr(src3), fptr(wx), fptr(wy), w, h, stride, factor)
        return dst, mask

    def derivative_stack(self, I1, I2, w):
        _, h, stride = I1.shape
        out = aligned_zeros((8, 3, h, stride))
        self.lib.orc_derivative_stack(fptr(out), fptr(I1), fptr(I2), w, h, stride)
        return out

    def dpsis_weight(self, im3, w, avg=(0, 0, 0), std=(1, 1, 1), hbit=0, coef=5.0):
        _, h, stride = im3.shape
        dst = plane(h, stride)
        a = (C.c_float * 3)(*avg)
        s = (C.c_float * 3)(*std)
        self.lib.orc_dpsis_weight(fptr(dst), fptr(im3), w, h, stride, C.c_float(coef), a, s, hbit)
        return dst

    def smoothness(self, method, uu, vv, dpsis, w, alpha, reg):
        h, stride = uu.shape
        sh, sv = plane(h, stride), plane(h, stride)
        self.lib.orc_smoothness(method, fptr(sh), fptr(sv), fptr(uu), fptr(vv), fptr(dpsis), w, h, stride, C.c_float(alpha), C.byref(reg))
        return sh, sv

    def sub_laplacian(self, dst, src, wh, wv, w):
        h, stride = src.shape
        self.lib.orc_sub_laplacian(fptr(dst), fptr(src), fptr(wh), fptr(wv), w, h, stride)
        return dst

    def add_data(self, sysm, mask, du, dv, D, chw, w, hd, hg, s, dt_norm, color, grad, ref_term=False):
        a11, a12, a22, b1, b2 = sysm
        h, stride = du.shape
        cw = (_f * 3)(fptr(chw[0]), fptr(chw[1]), fptr(chw[2]))
        fn = self.lib.orc_add_data_and_match_ref if ref_term else self.lib.orc_add_data_and_match
        return fn(fptr(a11), fptr(a12), fptr(a22), fptr(b1), fptr(b2), fptr(mask), fptr(du), fptr(dv), fptr(D), cw,
                  w, h, stride, C.c_float(hd), C.c_float(hg), C.c_float(s), int(dt_norm), C.byref(color), C.byref(grad))

    def sor(self, du, dv, a11, a12, a22, b1, b2, sh, sv, w, iterations, omega, readable=False, red_black=False):
        h, stride = du.shape
        fn = self.lib.orc_sor_red_black if red_black else (self.lib.orc_sor_coupled_readable if readable else self.lib.orc_sor_coupled)
        fn(fptr(du), fptr(dv), fptr(a11), fptr(a12), fptr(a22), fptr(b1), fptr(b2), fptr(sh), fptr(sv), w, h, stride, iterations, C.c_float(omega))

    def normalize(self, frames, w):
        """frames: list of (3,h,stride) arrays, normalised in place. returns (avg[3], std[3]) doubles and the
        published float round trip."""
        F = len(frames)
        _, h, stride = frames[0].shape
        arr = (_f * F)(*[fptr(f) for f in frames])
        avg = (C.c_double * 3)()
        std = (C.c_double * 3)()
        self.lib.orc_normalize(arr, F, w, h, stride, avg, std)
        af = (C.c_float * 3)()
        sf = (C.c_float * 3)()
        self.lib.orc_normalize_publish(avg, std, af, sf)
        return list(avg), list(std), list(af), list(sf)

    def change_log(self, cap=0):
        """start (cap > 0) or stop (cap = 0) recording the values of the reference's per-iteration "avg change" lines; returns the buffer, rows (kind, it, a, b)"""
        self._log = np.zeros((max(cap, 1), 4), np.float32)
        self.lib.orc_set_change_log.argtypes = [C.c_void_p, C.c_int]
        self.lib.orc_set_change_log(self._log.ctypes.data if cap > 0 else None, cap)
        return self._log

    def change_log_rows(self):
        return self._log[:self.lib.orc_change_log_count()]

    def compute_one_level(self, p, wx, wy, frames, w, chw=None, want_occ=False):
        h, stride = wx.shape
        F = len(frames)
        arr = (_f * F)(*[fptr(f) for f in frames])
        if chw is None:
            ones = plane(h, stride, 1.0)
            chw = [ones, ones, ones]
        cw = (_f * 3)(fptr(chw[0]), fptr(chw[1]), fptr(chw[2]))
        occ = plane(h, stride) if want_occ else None
        change = (C.c_float * 2)()
        rc = self.lib.orc_compute_one_level(C.byref(p), fptr(wx), fptr(wy), arr, cw, fptr(occ) if want_occ else None, w, h, stride, change)
        return rc, (change[0], change[1]), occ

    def variational(self, p, wx, wy, frames, w, chw=None):
        h, stride = wx.shape
        F = len(frames)
        arr = (_f * F)(*[fptr(f) for f in frames])
        cw = None
        if chw is not None:
            cw = (_f * 3)(fptr(chw[0]), fptr(chw[1]), fptr(chw[2]))
        change = (C.c_float * 2)()
        rc = self.lib.orc_variational(C.byref(p), fptr(wx), fptr(wy), arr, cw, w, h, stride, change)
        return rc, (change[0], change[1])

    def pyramid_sizes(self, w, h, layers, p_scale):
        ws = (C.c_int * 64)()
        hs = (C.c_int * 64)()
        L = self.lib.orc_pyramid_sizes(w, h, layers, C.c_float(p_scale), ws, hs)
        return [(ws[i], hs[i]) for i in range(L)]

    def gaussian_blur_cv(self, src, w, sigma):
        h, stride = src.shape
        dst = plane(h, stride)
        self.lib.orc_gaussian_blur_cv(fptr(dst), fptr(src), w, h, stride, C.c_float(sigma))
        return dst

    def resize_linear_fx(self, src, sw, fx, fy):
        """cv::resize(src, Size(0,0), fx, fy): dsize = cvRound(size * f) (round half to even), source coordinate (dst + .5) / f - .5"""
        sh, sstride = src.shape
        dw, dh = int(np.rint(sw * fx)), int(np.rint(sh * fy))
        dst = plane(dh, stride_of(dw))
        self.lib.orc_resize_linear_fx(fptr(dst), dw, dh, stride_of(dw), fptr(src), sw, sh, sstride, C.c_double(fx), C.c_double(fy))
        return dst, dw

    def force_labels(self, labels):
        """labels: (n_alter, h, stride) array kept alive by the caller, or None to switch the hook off (orc_force_labels)"""
        if labels is None:
            self.lib.orc_force_labels(None, 0)
        else:
            self.lib.orc_force_labels(fptr(labels), int(labels.shape[0]))

    def forced_gap(self, alter):
        self.lib.orc_forced_gap.restype = C.c_double
        return self.lib.orc_forced_gap(int(alter))

    def mask_weight(self, masks, occ, ref, data_norm, one_direction, w):
        """masks (2ref,h,stride) weighted in place (variational_mt.cpp:293-320)"""
        _, h, stride = masks.shape
        self.lib.orc_mask_weight(fptr(masks), fptr(occ), int(ref), C.c_float(data_norm), int(one_direction), w, h, stride)
        return masks

    def resize_linear_cv(self, src, sw, dw, dh):
        sh, sstride = src.shape
        dst = plane(dh, stride_of(dw))
        self.lib.orc_resize_linear_cv(fptr(dst), dw, dh, stride_of(dw), fptr(src), sw, sh, sstride)
        return dst


# ---------------------------------------------------------------------------------------------
# the compiled reference (image.h:17-43 structs)
# ---------------------------------------------------------------------------------------------
class image_t(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("stride", C.c_int), ("data", _f)]


class color_image_t(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("stride", C.c_int), ("c1", _f), ("c2", _f), ("c3", _f)]


class convolution_t(C.Structure):
    _fields_ = [("order", C.c_int), ("coeffs", _f), ("coeffs_accu", _f)]


def as_image(a, w):
    h, stride = a.shape
    assert a.ctypes.data % 16 == 0
    return image_t(w, h, stride, fptr(a))


def as_color(a, w):
    _, h, stride = a.shape
    assert a.ctypes.data % 16 == 0 and (h * stride) % 4 == 0
    base = a.ctypes.data
    pl = h * stride * 4
    return color_image_t(w, h, stride, C.cast(base, _f), C.cast(base + pl, _f), C.cast(base + 2 * pl, _f))


def ref_available():
    return os.path.exists(REF_SO)


class Params2f(C.Structure):
    """variational_params_t (variational.h:16-25) == orc_params_2f == sfa_params_2frame"""
    _fields_ = [("alpha", C.c_float), ("gamma", C.c_float), ("delta", C.c_float), ("sigma", C.c_float),
                ("niter_outer", C.c_int), ("niter_inner", C.c_int), ("niter_solver", C.c_int), ("sor_omega", C.c_float)]


def params_2f(alpha=1.0, gamma=0.71, delta=0.0, sigma=1.0, niter_outer=5, niter_inner=1, niter_solver=30, sor_omega=1.9):
    """defaults of variational_params_default (variational.c:86-98)"""
    return Params2f(alpha, gamma, delta, sigma, niter_outer, niter_inner, niter_solver, sor_omega)


class RefLib:
    """The reference's own compiled C (solver.c, image.c, variational_aux.c, penalty headers)."""

    def __init__(self, path=REF_SO):
        self.lib = L = C.CDLL(path)
        L.convolution_new.restype = C.POINTER(convolution_t)
        L.convolution_new.argtypes = [C.c_int, _f, C.c_int]
        L.compute_dpsis_weight.restype = C.POINTER(image_t)
        half5 = (C.c_float * 3)(0.0, np.float32(-8.0) / np.float32(12.0), np.float32(1.0) / np.float32(12.0))
        half3 = (C.c_float * 2)(0.0, -0.5)
        self.deriv = L.convolution_new(2, half5, 0)        # variational_mt.cpp:570-571
        self.deriv_flow = L.convolution_new(1, half3, 0)   # variational_mt.cpp:572-573

    def convolve(self, src, w, order, horiz):
        h, stride = src.shape
        s = src.copy() if False else src   # convolve_horiz writes replicate values into the source padding
        s2 = aligned_zeros(src.shape)
        s2[...] = s
        dst = plane(h, stride)
        conv = self.deriv if order == 2 else self.deriv_flow
        fn = self.lib.convolve_horiz if horiz else self.lib.convolve_vert
        d_i, s_i = as_image(dst, w), as_image(s2, w)
        fn(C.byref(d_i), C.byref(s_i), conv)
        return dst

    def gaussian_presmooth(self, src, w, sigma):
        """gaussian_filter (image.c:310) + convolution_new(even) + convolve_horiz, convolve_vert: what variational_mt.cpp:590-597 applies per channel"""
        h, stride = src.shape
        L = self.lib
        L.gaussian_filter.restype = _f
        L.gaussian_filter.argtypes = [C.c_float, C.POINTER(C.c_int)]
        n = C.c_int()
        filt = L.gaussian_filter(C.c_float(sigma), C.byref(n))
        conv = L.convolution_new(n.value, filt, 1)
        s2 = aligned_zeros(src.shape); s2[...] = src
        tmp, dst = plane(h, stride), plane(h, stride)
        L.convolve_horiz(C.byref(as_image(tmp, w)), C.byref(as_image(s2, w)), conv)
        L.convolve_vert(C.byref(as_image(dst, w)), C.byref(as_image(tmp, w)), conv)
        return dst

    def variational_2frame(self, wx, wy, im1, im2, w, p):
        """the reference's own two-frame `variational` (variational.c:101), in place on wx, wy"""
        a, b = as_image(wx, w), as_image(wy, w)
        i1, i2 = as_color(im1, w), as_color(im2, w)
        self.lib.variational(C.byref(a), C.byref(b), C.byref(i1), C.byref(i2), C.byref(p))

    def sor(self, du, dv, a11, a12, a22, b1, b2, sh, sv, w, iterations, omega, readable=False):
        imgs = [as_image(a, w) for a in (du, dv, a11, a12, a22, b1, b2, sh, sv)]
        fn = self.lib.sor_coupled_slow_but_readable if readable else self.lib.sor_coupled
        fn(*[C.byref(i) for i in imgs], C.c_int(iterations), C.c_float(omega))

    def image_warp_prescaled(self, src3, fwx, fwy, w):
        """variational_aux.c:18 (2-frame image_warp): xx = i + wx. Feeding factor*wx (one fp32 product, the
        same one variational_aux_mt.cpp:735 forms) reproduces the MT warp for any factor."""
        _, h, stride = src3.shape
        dst = aligned_zeros((3, h, stride))
        mask = plane(h, stride)
        d, s = as_color(dst, w), as_color(src3, w)
        m, x, y = as_image(mask, w), as_image(fwx, w), as_image(fwy, w)
        self.lib.image_warp(C.byref(d), C.byref(m), C.byref(s), C.byref(x), C.byref(y))
        return dst, mask

    def sub_laplacian(self, dst, src, wh, wv, w):
        d, s, a, b = as_image(dst, w), as_image(src, w), as_image(wh, w), as_image(wv, w)
        self.lib.sub_laplacian(C.byref(d), C.byref(s), C.byref(a), C.byref(b))
        return dst

    def dpsis_weight(self, im3, w, coef=5.0):
        """variational_aux.c:199 -- equals the MT 3-output version's first output for avg=0,std=1,8 bit"""
        s = as_color(im3, w)
        r = self.lib.compute_dpsis_weight(C.byref(s), C.c_float(coef), self.deriv)
        h, stride = r.contents.height, r.contents.stride
        out = np.ctypeslib.as_array(r.contents.data, shape=(h, stride)).copy()
        self.lib.image_delete(r)
        return out

    def get_derivatives(self, im1, im2, w):
        """variational_aux.c:56: mean = .5*(im2+im1), dt = im2-im1.  returns dx,dy,dt,dxx,dxy,dyy,dxt,dyt"""
        _, h, stride = im1.shape
        outs = [aligned_zeros((3, h, stride)) for _ in range(8)]
        a = aligned_zeros(im1.shape); a[...] = im1
        b = aligned_zeros(im2.shape); b[...] = im2
        cs = [as_color(o, w) for o in outs]
        ia, ib = as_color(a, w), as_color(b, w)
        self.lib.get_derivatives(C.byref(ia), C.byref(ib), self.deriv, *[C.byref(c) for c in cs])
        return outs

    def compute_smoothness(self, uu, vv, dpsis, w, half_alpha):
        h, stride = uu.shape
        sh, sv = plane(h, stride), plane(h, stride)
        u2 = aligned_zeros(uu.shape); u2[...] = uu
        v2 = aligned_zeros(vv.shape); v2[...] = vv
        args = [as_image(a, w) for a in (sh, sv, u2, v2, dpsis)]
        self.lib.compute_smoothness(*[C.byref(a) for a in args], self.deriv_flow, C.c_float(half_alpha))
        return sh, sv

    def compute_data_and_match(self, mask, du, dv, D, w, half_delta_over3, half_gamma_over3):
        """variational_aux.c:226.  D: list of 8 (3,h,stride) arrays Ix,Iy,Iz,Ixx,Ixy,Iyy,Ixz,Iyz"""
        h, stride = du.shape
        outs = [plane(h, stride) for _ in range(5)]
        imgs = [as_image(a, w) for a in outs] + [as_image(mask, w), as_image(du, w), as_image(dv, w)]
        cols = [as_color(d, w) for d in D]
        self.lib.compute_data_and_match(*[C.byref(i) for i in imgs], *[C.byref(c) for c in cols],
                                        C.c_float(half_delta_over3), C.c_float(half_gamma_over3))
        return outs

    def penalty_apply(self, pid, eps, trunc, x):
        x = np.ascontiguousarray(x, dtype=np.float32).ravel()
        n = (x.size // 4) * 4
        x = x[:n].copy()
        v = np.zeros(n, np.float32)
        self.lib.ref_penalty_apply(int(pid), C.c_float(eps), C.c_float(trunc), fptr(x), n, fptr(v))
        return x, v

    def penalty_derivative(self, pid, eps, trunc, x):
        x = np.ascontiguousarray(x, dtype=np.float32).ravel()
        n = (x.size // 4) * 4
        x = x[:n].copy()
        s = np.zeros(n, np.float32)
        v = np.zeros(n, np.float32)
        self.lib.ref_penalty_derivative(int(pid), C.c_float(eps), C.c_float(trunc), fptr(x), n, fptr(s), fptr(v))
        return x, s, v
