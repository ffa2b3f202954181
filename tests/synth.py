"""Deterministic synthetic inputs shared by the tests, bench.py and smoke()."""
import numpy as np

from oracle import aligned_zeros, plane, stride_of


def texture_frame(w, h, k, dx=1.5, dy=-0.75):
    """Analytic sinusoid texture of BASELINE config 1, frame k sampled at (x - k*dx, y - k*dy);
    returns (3,h,stride) fp32 in 0..255."""
    stride = stride_of(w)
    out = aligned_zeros((3, h, stride))
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    xs, ys = x - k * dx, y - k * dy
    for c in range(3):
        v = (127.5 + 40 * np.sin(.11 * xs + .07 * ys + c) + 35 * np.sin(.05 * xs - .13 * ys + 2 * c)
             + 30 * np.cos(.23 * xs + .19 * ys) + 20 * np.sin(.41 * ys - .31 * xs + c))
        out[c, :, :w] = v.astype(np.float32)
    return out


def noise_plane(rng, w, h, lo=-1.0, hi=1.0):
    a = plane(h, stride_of(w))
    a[:, :w] = rng.uniform(lo, hi, size=(h, w)).astype(np.float32)
    return a


def noise_color(rng, w, h, lo=-1.0, hi=1.0):
    a = aligned_zeros((3, h, stride_of(w)))
    a[:, :, :w] = rng.uniform(lo, hi, size=(3, h, w)).astype(np.float32)
    return a


def smooth_noise_color(rng, w, h, scale=60.0):
    """band-limited texture: box-filtered uniform noise, 0..255-ish"""
    a = rng.uniform(0, 1, size=(3, h + 8, w + 8))
    k = np.ones(5) / 5
    for ax in (1, 2):
        a = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), ax, a)
    a = (a[:, 4:-4, 4:-4] - 0.5) * scale * 4 + 127.5
    out = aligned_zeros((3, h, stride_of(w)))
    out[:, :, :w] = a.astype(np.float32)
    return out


def sor_system(rng, w, h, spd=True):
    """A random SPD per-pixel 2x2 block system with positive edge weights, shaped like the ones the
    pipeline produces (sh last column 0, sv last row 0)."""
    s = stride_of(w)
    a11, a22 = noise_plane(rng, w, h, 0.5, 3.0), noise_plane(rng, w, h, 0.5, 3.0)
    a12 = noise_plane(rng, w, h, -0.4, 0.4)
    b1, b2 = noise_plane(rng, w, h, -1, 1), noise_plane(rng, w, h, -1, 1)
    sh, sv = noise_plane(rng, w, h, 0.05, 1.5), noise_plane(rng, w, h, 0.05, 1.5)
    sh[:, w - 1:] = 0
    sv[h - 1, :] = 0
    du, dv = plane(h, s), plane(h, s)
    return dict(du=du, dv=dv, a11=a11, a12=a12, a22=a22, b1=b1, b2=b2, sh=sh, sv=sv)


def copy_sys(d):
    out = {}
    for k, v in d.items():
        a = aligned_zeros(v.shape)
        a[...] = v
        out[k] = a
    return out
