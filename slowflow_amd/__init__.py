"""slowflow_amd -- Python binding (ctypes) of libslowflow_amd.so, the MI355X-native implementation of
slowflow's variational optical-flow refinement path (include/slowflow_amd.h).

This package is plumbing around the C-ABI for tests, bench.py and Python callers.  It contains no
compute and no CPU fallback: if the HIP library is missing or no GPU is usable every call raises.
"""
import ctypes as C
import os
import subprocess
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SFA_LIB") or os.path.join(_HERE, "libslowflow_amd.so")     # SFA_LIB: an experimental build of the same C-ABI (tuning only)
MAX_REF = 8

_f = C.POINTER(C.c_float)


class SlowflowError(RuntimeError):
    pass


def build(verbose=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    out = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"], capture_output=True, text=True)
    if out.returncode != 0:
        raise SlowflowError("building libslowflow_amd.so failed:\n" + out.stdout + out.stderr)
    if verbose:
        print(out.stdout)
    return LIB_PATH


class Penalty(C.Structure):
    _fields_ = [("id", C.c_int), ("eps", C.c_float), ("trunc", C.c_float)]


class Params(C.Structure):
    """sfa_params: the cfg keys of the path (variational_mt.cpp:173-192, 533-568)."""
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


class Image(C.Structure):
    """sfa_image == image_t (epic_flow_extended/image.h:17-23)"""
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("stride", C.c_int), ("data", _f)]


class Params2f(C.Structure):
    """sfa_params_2frame == variational_params_t (variational.h:16-25)"""
    _fields_ = [("alpha", C.c_float), ("gamma", C.c_float), ("delta", C.c_float), ("sigma", C.c_float),
                ("niter_outer", C.c_int), ("niter_inner", C.c_int), ("niter_solver", C.c_int), ("sor_omega", C.c_float)]


EXPORTS = [
    "sfa_device_count", "sfa_ctx_create", "sfa_ctx_destroy", "sfa_last_error", "sfa_ctx_sync", "sfa_params_default",
    "sfa_variational", "sfa_variational_2frame", "sfa_params_2frame_default", "variational", "sfa_compute_one_level", "sfa_normalize", "sfa_sor_coupled", "sfa_sor_red_black", "sor_coupled",
    "sfa_image_warp", "sfa_derivative_stack", "sfa_convolve", "sfa_dpsis_weight", "sfa_smoothness", "sfa_sub_laplacian",
    "sfa_add_data_and_match", "sfa_occlusion_costs", "sfa_grid_cut", "sfa_gaussian_blur", "sfa_resize_linear", "sfa_resize_linear_fx", "sfa_gaussian_presmooth", "sfa_pyramid_sizes",
    "sfa_sequence_create", "sfa_sequence_destroy", "sfa_sequence_upload", "sfa_sequence_download", "sfa_sequence_normalize", "sfa_sequence_frame_sums", "sfa_normalize_statistics", "sfa_sequence_apply_normalization",
    "sfa_job_create", "sfa_job_destroy", "sfa_job_upload", "sfa_job_upload_resident", "sfa_job_reset_flow", "sfa_job_run", "sfa_job_download", "sfa_job_download_occlusions", "sfa_job_keep_alternation_occlusions", "sfa_job_download_alternation_occlusions", "sfa_job_mpix_iters", "sfa_job_device_bytes",
    "sfa_sor_batch_create", "sfa_sor_batch_destroy", "sfa_sor_batch_upload", "sfa_sor_batch_run", "sfa_sor_batch_download",
    "sfa_division_chain", "sfa_ctx_set_wait_bound", "sfa_debug_set", "sfa_ctx_set_verbose", "sfa_profile_enable", "sfa_profile_read", "sfa_profile_read_kernels", "sfa_timer_start", "sfa_timer_stop",
]

_lib = None


def lib():
    """The loaded C-ABI library.  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SlowflowError(f"{LIB_PATH} is missing: build it with slowflow_amd.build() / make -C slowflow_amd/csrc "
                                "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.sfa_last_error.restype = C.c_char_p
        L.sfa_last_error.argtypes = [C.c_void_p]
        L.sfa_job_mpix_iters.restype = C.c_double
        L.sfa_job_mpix_iters.argtypes = [C.c_void_p]
        L.sfa_job_device_bytes.restype = C.c_double
        L.sfa_job_device_bytes.argtypes = [C.c_void_p]
        for name in ("sfa_ctx_destroy", "sfa_job_destroy", "sfa_sor_batch_destroy", "sfa_sequence_destroy"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [C.c_void_p]
        _lib = L
    return _lib


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], "fp32 C-contiguous planes only"
    return a.ctypes.data_as(_f)


def stride_of(w):
    return ((w + 3) // 4) * 4


def pyramid_sizes(w, h, layers, p_scale):
    """level sizes of the coarse-to-fine pyramid (variational_mt.cpp:576-652); pure host logic, no GPU needed"""
    ws, hs = (C.c_int * 64)(), (C.c_int * 64)()
    n = lib().sfa_pyramid_sizes(int(w), int(h), int(layers), C.c_float(p_scale), ws, hs)
    return list(ws[:n]), list(hs[:n])


def default_params():
    p = Params()
    lib().sfa_params_default(C.byref(p))
    return p


def debug_set(name, value=None):
    """test / tooling hook (include/slowflow_amd.h: sfa_debug_set): one switch of the library's cross-check and what-if paths, e.g. debug_set("SFA_UNFUSED", 1);
    value None = back to the default.  The library itself reads these names from the environment only under SFA_DEBUG=1."""
    L = lib()
    L.sfa_debug_set.argtypes = [C.c_char_p, C.c_char_p]
    rc = L.sfa_debug_set(name.encode(), None if value is None else str(value).encode())
    if rc != 0:
        raise SlowflowError("sfa_debug_set(%s): %s" % (name, L.sfa_last_error(None).decode()))


class debug_switches:
    """with debug_switches(SFA_UNFUSED=1, SFA_SOR_CHAIN=5): ... -- sets the switches and restores the defaults on exit"""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            debug_set(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            debug_set(k, None)
        return False


def device_count():
    return lib().sfa_device_count()


class Context:
    """sfa_ctx: one GPU, one HIP stream."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        self._children = weakref.WeakSet()          # jobs, sequences, solver batches created on this context: they hold device memory and the stream
        rc = lib().sfa_ctx_create(int(device), C.byref(self.h))
        if rc != 0:
            raise SlowflowError(f"sfa_ctx_create({device}) -> {rc}: {lib().sfa_last_error(None).decode()}")

    def close(self):
        """destroys what was created on the context first: the library's objects keep a pointer to it, and the garbage collector (cycles,
        interpreter shutdown) may finalise a context before its jobs"""
        if self.h:
            for child in list(getattr(self, "_children", ())):
                child.close()
            lib().sfa_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise SlowflowError(f"{what} -> {rc}: {lib().sfa_last_error(self.h).decode()}")

    def sync(self):
        self._ck(lib().sfa_ctx_sync(self.h), "sfa_ctx_sync")

    # ---- stage entry points (host planes) -------------------------------------------------------
    def image_warp(self, src3, wx, wy, w, factor, want_mask=True):
        _, h, stride = src3.shape
        dst = np.zeros_like(src3)
        mask = np.zeros((h, stride), np.float32) if want_mask else None
        self._ck(lib().sfa_image_warp(self.h, fptr(dst), fptr(mask) if want_mask else None, fptr(src3), fptr(wx), fptr(wy), w, h, stride, int(factor)), "sfa_image_warp")
        return dst, mask

    def derivative_stack(self, I1, I2, w):
        _, h, stride = I1.shape
        out = np.zeros((8, 3, h, stride), np.float32)
        self._ck(lib().sfa_derivative_stack(self.h, fptr(out), fptr(I1), fptr(I2), w, h, stride), "sfa_derivative_stack")
        return out

    def convolve(self, src, w, order, horiz):
        h, stride = src.shape
        dst = np.zeros_like(src)
        self._ck(lib().sfa_convolve(self.h, fptr(dst), fptr(src), w, h, stride, int(order), int(bool(horiz))), "sfa_convolve")
        return dst

    def dpsis_weight(self, im3, w, avg=(0, 0, 0), std=(1, 1, 1), hbit=0, coef=5.0):
        _, h, stride = im3.shape
        dst = np.zeros((h, stride), np.float32)
        a, s = (C.c_float * 3)(*avg), (C.c_float * 3)(*std)
        self._ck(lib().sfa_dpsis_weight(self.h, fptr(dst), fptr(im3), w, h, stride, C.c_float(coef), a, s, int(hbit)), "sfa_dpsis_weight")
        return dst

    def smoothness(self, method, uu, vv, dpsis, w, alpha, reg):
        h, stride = uu.shape
        sh, sv = np.zeros_like(uu), np.zeros_like(uu)
        self._ck(lib().sfa_smoothness(self.h, int(method), fptr(sh), fptr(sv), fptr(uu), fptr(vv), fptr(dpsis), w, h, stride, C.c_float(alpha), C.byref(reg)), "sfa_smoothness")
        return sh, sv

    def sub_laplacian(self, dst, src, wh, wv, w):
        h, stride = src.shape
        self._ck(lib().sfa_sub_laplacian(self.h, fptr(dst), fptr(src), fptr(wh), fptr(wv), w, h, stride), "sfa_sub_laplacian")
        return dst

    def occlusion_costs(self, p, masks, succ1, succ2, ref1, ref2, w):
        """masks: list of 2ref planes; succ1/succ2/ref1/ref2: lists of 2ref colour images (3,h,stride) -> d0, d1"""
        h, stride = masks[0].shape
        d0, d1 = np.zeros((h, stride), np.float32), np.zeros((h, stride), np.float32)
        n = len(masks)
        arr = lambda xs: (_f * n)(*[fptr(x) for x in xs])
        self._ck(lib().sfa_occlusion_costs(self.h, C.byref(p), fptr(d0), fptr(d1), arr(masks), arr(succ1), arr(succ2), arr(ref1), arr(ref2), w, h, stride),
                 "sfa_occlusion_costs")
        return d0, d1

    def grid_cut(self, d0, d1, alpha, w):
        h, stride = d0.shape
        occ = np.zeros((h, stride), np.float32)
        self._ck(lib().sfa_grid_cut(self.h, fptr(occ), fptr(d0), fptr(d1), w, h, stride, C.c_float(alpha)), "sfa_grid_cut")
        return occ

    def add_data(self, sysm, mask, du, dv, D, chw, w, hd, hg, s, dt_norm, color, grad, ref_term=False):
        a11, a12, a22, b1, b2 = sysm
        h, stride = du.shape
        cw = (_f * 3)(fptr(chw[0]), fptr(chw[1]), fptr(chw[2])) if chw is not None else None
        rc = lib().sfa_add_data_and_match(self.h, fptr(a11), fptr(a12), fptr(a22), fptr(b1), fptr(b2), fptr(mask), fptr(du), fptr(dv), fptr(D), cw,
                                          w, h, stride, C.c_float(hd), C.c_float(hg), C.c_float(s), int(bool(ref_term)), int(dt_norm),
                                          C.byref(color), C.byref(grad))
        return rc

    def sor_coupled(self, du, dv, a11, a12, a22, b1, b2, sh, sv, w, iterations, omega, red_black=False):
        """drop-in for sor_coupled (solver.h:11) on host planes; red_black=True: the labelled two-colour mode (a different algorithm)"""
        h, stride = du.shape
        imgs = [Image(w, h, stride, fptr(a)) for a in (du, dv, a11, a12, a22, b1, b2, sh, sv)]
        fn, name = (lib().sfa_sor_red_black, "sfa_sor_red_black") if red_black else (lib().sfa_sor_coupled, "sfa_sor_coupled")
        self._ck(fn(self.h, *[C.byref(i) for i in imgs], int(iterations), C.c_float(omega)), name)

    def gaussian_blur(self, src, w, sigma):
        h, stride = src.shape
        dst = np.zeros_like(src)
        self._ck(lib().sfa_gaussian_blur(self.h, fptr(dst), fptr(src), w, h, stride, C.c_float(sigma)), "sfa_gaussian_blur")
        return dst

    def resize_linear(self, src, sw, dw, dh):
        sh, sstride = src.shape
        dst = np.zeros((dh, stride_of(dw)), np.float32)
        self._ck(lib().sfa_resize_linear(self.h, fptr(dst), dw, dh, stride_of(dw), fptr(src), sw, sh, sstride), "sfa_resize_linear")
        return dst

    def gaussian_presmooth(self, src, w, sigma):
        h, stride = src.shape
        dst = np.zeros_like(src)
        self._ck(lib().sfa_gaussian_presmooth(self.h, fptr(dst), fptr(src), w, h, stride, C.c_float(sigma)), "sfa_gaussian_presmooth")
        return dst

    def resize_linear_fx(self, src, sw, fx, fy):
        """cv::resize(src, Size(0,0), fx, fy): dsize = round(size * f), source coordinate (dst + .5) / f - .5"""
        sh, sstride = src.shape
        dw, dh = int(round(sw * fx)), int(round(sh * fy))
        dst = np.zeros((dh, stride_of(dw)), np.float32)
        self._ck(lib().sfa_resize_linear_fx(self.h, fptr(dst), dw, dh, stride_of(dw), fptr(src), sw, sh, sstride, C.c_double(fx), C.c_double(fy)),
                 "sfa_resize_linear_fx")
        return dst, dw

    def normalize(self, frames, w):
        F = len(frames)
        _, h, stride = frames[0].shape
        arr = (_f * F)(*[fptr(f) for f in frames])
        avg, std = (C.c_double * 3)(), (C.c_double * 3)()
        self._ck(lib().sfa_normalize(self.h, arr, F, w, h, stride, avg, std), "sfa_normalize")
        return list(avg), list(std)

    # ---- the path --------------------------------------------------------------------------------
    def _run(self, fn, name, p, wx, wy, frames, w, chw, want_occ):
        h, stride = wx.shape
        F = len(frames)
        arr = (_f * F)(*[fptr(f) for f in frames])
        cw = (_f * 3)(fptr(chw[0]), fptr(chw[1]), fptr(chw[2])) if chw is not None else None
        occ = np.zeros((h, stride), np.float32) if want_occ else None
        change = (C.c_float * 2)()
        rc = fn(self.h, C.byref(p), fptr(wx), fptr(wy), w, h, stride, arr, F, cw, fptr(occ) if want_occ else None, change)
        self._ck(rc, name)
        return (change[0], change[1]), occ

    def variational_2frame(self, wx, wy, im1, im2, w, p=None):
        """the reference's original two-frame `variational` (variational.c:101), in place on wx, wy"""
        h, stride = wx.shape
        self._ck(lib().sfa_variational_2frame(self.h, fptr(wx), fptr(wy), w, h, stride, fptr(im1), fptr(im2), C.byref(p) if p is not None else None),
                 "sfa_variational_2frame")

    def compute_one_level(self, p, wx, wy, frames, w, chw=None, want_occ=False):
        return self._run(lib().sfa_compute_one_level, "sfa_compute_one_level", p, wx, wy, frames, w, chw, want_occ)

    def variational(self, p, wx, wy, frames, w, chw=None, want_occ=False):
        return self._run(lib().sfa_variational, "sfa_variational", p, wx, wy, frames, w, chw, want_occ)

    # ---- profiling / timing ----------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._ck(lib().sfa_profile_enable(self.h, int(on)), "sfa_profile_enable")

    def profile_read(self):
        n, ms, by = C.c_int(), C.c_double(), C.c_double()
        self._ck(lib().sfa_profile_read(self.h, C.byref(n), C.byref(ms), C.byref(by)), "sfa_profile_read")
        return n.value, ms.value, by.value

    def profile_read_kernels(self):
        """(assembly launches, their ms, their pixels x terms, name of the solver kernel shape last launched); call BEFORE profile_enable(False)"""
        n, ms, px = C.c_int(), C.c_double(), C.c_double()
        name = C.create_string_buffer(160)
        self._ck(lib().sfa_profile_read_kernels(self.h, C.byref(n), C.byref(ms), C.byref(px), name, 160), "sfa_profile_read_kernels")
        return n.value, ms.value, px.value, name.value.decode()

    def set_verbose(self, on=True):
        """the reference's per-iteration "avg change" lines on stdout (variational_mt.cpp:404-405, 431-432)"""
        self._ck(lib().sfa_ctx_set_verbose(self.h, int(bool(on))), "sfa_ctx_set_verbose")

    def set_wait_bound(self, spins):
        """test hook: bound of the solver's in-kernel waits in polls (0 = the default of 2^22); see include/slowflow_amd.h"""
        self._ck(lib().sfa_ctx_set_wait_bound(self.h, C.c_uint(int(spins))), "sfa_ctx_set_wait_bound")

    def division_chain(self, a, b):
        """(q_chain, q_exact, admitted) of the shared-reciprocal division test hook (include/slowflow_amd.h: sfa_division_chain)"""
        import numpy as np
        a = np.ascontiguousarray(a, np.float32).ravel(); b = np.ascontiguousarray(b, np.float32).ravel()
        assert a.size == b.size
        qc, qe, ad = np.empty_like(a), np.empty_like(a), np.empty(a.size, np.uint8)
        L = lib()
        L.sfa_division_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        self._ck(L.sfa_division_chain(self.h, a.ctypes.data, b.ctypes.data, qc.ctypes.data, qe.ctypes.data, ad.ctypes.data, a.size), "sfa_division_chain")
        return qc, qe, ad

    def timer_start(self):
        self._ck(lib().sfa_timer_start(self.h), "sfa_timer_start")

    def timer_stop(self):
        ms = C.c_float()
        self._ck(lib().sfa_timer_stop(self.h, C.byref(ms)), "sfa_timer_stop")
        return ms.value


class Job:
    """sfa_job: `batch` frame windows of one size, resident in HBM, refined in lockstep."""

    def __init__(self, ctx, params, w, h, batch=1):
        self.ctx, self.w, self.h, self.batch = ctx, w, h, batch
        self.h_ = C.c_void_p()
        ctx._ck(lib().sfa_job_create(ctx.h, C.byref(params), w, h, batch, C.byref(self.h_)), "sfa_job_create")
        ctx._children.add(self)

    def upload(self, b, frames, wx=None, wy=None, chw=None):
        F = len(frames)
        _, h, stride = frames[0].shape
        arr = (_f * F)(*[fptr(f) for f in frames])
        cw = (_f * 3)(fptr(chw[0]), fptr(chw[1]), fptr(chw[2])) if chw is not None else None
        self.ctx._ck(lib().sfa_job_upload(self.h_, b, arr, F, fptr(wx) if wx is not None else None, fptr(wy) if wy is not None else None, stride, cw), "sfa_job_upload")

    def upload_resident(self, b, seq, frame_index, wx=None, wy=None, chw=None):
        idx = (C.c_int * len(frame_index))(*frame_index)
        cw = (_f * 3)(fptr(chw[0]), fptr(chw[1]), fptr(chw[2])) if chw is not None else None
        self.ctx._ck(lib().sfa_job_upload_resident(self.h_, b, seq.h_, idx, len(frame_index), fptr(wx) if wx is not None else None,
                                                   fptr(wy) if wy is not None else None, stride_of(self.w), cw), "sfa_job_upload_resident")

    def run(self):
        self.ctx._ck(lib().sfa_job_run(self.h_), "sfa_job_run")

    def download(self, b):
        stride = stride_of(self.w)
        wx, wy = np.zeros((self.h, stride), np.float32), np.zeros((self.h, stride), np.float32)
        change = (C.c_float * 2)()
        self.ctx._ck(lib().sfa_job_download(self.h_, b, fptr(wx), fptr(wy), stride, change), "sfa_job_download")
        return wx, wy, (change[0], change[1])

    def download_occlusions(self, b):
        stride = stride_of(self.w)
        occ = np.zeros((self.h, stride), np.float32)
        self.ctx._ck(lib().sfa_job_download_occlusions(self.h_, b, fptr(occ), stride), "sfa_job_download_occlusions")
        return occ

    def keep_alternation_occlusions(self, on=True):
        self.ctx._ck(lib().sfa_job_keep_alternation_occlusions(self.h_, int(on)), "sfa_job_keep_alternation_occlusions")

    def download_alternation_occlusions(self, b, alter):
        stride = stride_of(self.w)
        occ = np.zeros((self.h, stride), np.float32)
        self.ctx._ck(lib().sfa_job_download_alternation_occlusions(self.h_, b, int(alter), fptr(occ), stride), "sfa_job_download_alternation_occlusions")
        return occ

    def mpix_iters(self):
        return lib().sfa_job_mpix_iters(self.h_)

    def device_bytes(self):
        return lib().sfa_job_device_bytes(self.h_)

    def close(self):
        if self.h_:
            if self.ctx.h:                      # a context finalised first (cyclic garbage, interpreter shutdown) took its stream along: nothing to call into
                lib().sfa_job_destroy(self.h_)
            self.h_ = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def normalize_statistics(sums, w, h):
    """normalize()'s (avg, std) from per-frame sums in frame order (include/slowflow_amd.h: sfa_normalize_statistics; host arithmetic)"""
    s = np.ascontiguousarray(sums, np.float64)
    avg, std = (C.c_double * 3)(), (C.c_double * 3)()
    rc = lib().sfa_normalize_statistics(C.c_void_p(s.ctypes.data), int(s.shape[0]), int(w), int(h), avg, std)
    if rc != 0:
        raise SlowflowError("sfa_normalize_statistics: bad arguments")
    return list(avg), list(std)


class Sequence:
    """sfa_sequence: the frames of a sequence resident on the GPU, normalised there"""

    def __init__(self, ctx, w, h, n):
        self.ctx, self.w, self.h, self.n = ctx, w, h, n
        self.h_ = C.c_void_p()
        ctx._ck(lib().sfa_sequence_create(ctx.h, w, h, n, C.byref(self.h_)), "sfa_sequence_create")
        ctx._children.add(self)

    def upload(self, f, frame3):
        self.ctx._ck(lib().sfa_sequence_upload(self.h_, f, fptr(frame3), frame3.shape[2]), "sfa_sequence_upload")

    def download(self, f):
        a = np.zeros((3, self.h, stride_of(self.w)), np.float32)
        self.ctx._ck(lib().sfa_sequence_download(self.h_, f, fptr(a), a.shape[2]), "sfa_sequence_download")
        return a

    def normalize(self, f0=0, n=None):
        avg, std = (C.c_double * 3)(), (C.c_double * 3)()
        self.ctx._ck(lib().sfa_sequence_normalize(self.h_, f0, self.n - f0 if n is None else n, avg, std), "sfa_sequence_normalize")
        return list(avg), list(std)

    def frame_sums(self, f0=0, n=None):
        """(n, 6) fp64 sums of the raw frames (per channel: sum, sum of squares)"""
        n = self.n - f0 if n is None else n
        s = np.zeros((n, 6), np.float64)
        self.ctx._ck(lib().sfa_sequence_frame_sums(self.h_, f0, n, C.c_void_p(s.ctypes.data)), "sfa_sequence_frame_sums")
        return s

    def apply_normalization(self, avg, std, f0=0, n=None):
        a, s = (C.c_double * 3)(*avg), (C.c_double * 3)(*std)
        self.ctx._ck(lib().sfa_sequence_apply_normalization(self.h_, f0, self.n - f0 if n is None else n, a, s), "sfa_sequence_apply_normalization")

    def close(self):
        if self.h_:
            if self.ctx.h:                      # a context finalised first (cyclic garbage, interpreter shutdown) took its stream along: nothing to call into
                lib().sfa_sequence_destroy(self.h_)
            self.h_ = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SorBatch:
    """sfa_sor_batch: `batch` independent 2x2-block systems of one size, resident in HBM."""

    def __init__(self, ctx, w, h, batch=1):
        self.ctx, self.w, self.h, self.batch = ctx, w, h, batch
        self.h_ = C.c_void_p()
        ctx._ck(lib().sfa_sor_batch_create(ctx.h, w, h, batch, C.byref(self.h_)), "sfa_sor_batch_create")
        ctx._children.add(self)

    def upload(self, b, du, dv, a11, a12, a22, b1, b2, sh, sv):
        stride = du.shape[1]
        self.ctx._ck(lib().sfa_sor_batch_upload(self.h_, b, *[fptr(a) for a in (du, dv, a11, a12, a22, b1, b2, sh, sv)], stride), "sfa_sor_batch_upload")

    def run(self, iterations, omega):
        self.ctx._ck(lib().sfa_sor_batch_run(self.h_, int(iterations), C.c_float(omega)), "sfa_sor_batch_run")

    def download(self, b):
        stride = stride_of(self.w)
        du, dv = np.zeros((self.h, stride), np.float32), np.zeros((self.h, stride), np.float32)
        self.ctx._ck(lib().sfa_sor_batch_download(self.h_, b, fptr(du), fptr(dv), stride), "sfa_sor_batch_download")
        return du, dv

    def close(self):
        if self.h_:
            if self.ctx.h:                      # a context finalised first (cyclic garbage, interpreter shutdown) took its stream along: nothing to call into
                lib().sfa_sor_batch_destroy(self.h_)
            self.h_ = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
