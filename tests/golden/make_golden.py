"""Generates tests/golden/ref_vectors.npz: inputs and the outputs the REFERENCE ITSELF produced for them.

Run in the build container only (needs oracle/_ref/libslowflow_ref.so, i.e. /root/reference):
    python tests/golden/make_golden.py
The reference's compiled C (solver.c, image.c, variational_aux.c, penalty_functions headers; recipe in
oracle/Makefile) is called through ctypes on seeded inputs; only the resulting data is committed.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import oracle as orc  # noqa: E402
from synth import copy_sys, noise_plane, smooth_noise_color, sor_system  # noqa: E402


def main():
    ref = orc.RefLib()
    out = {}
    # --- convolutions -------------------------------------------------------------------------
    for (w, h) in ((67, 45), (64, 48)):
        rng = np.random.default_rng(1000 + w)
        src = noise_plane(rng, w, h, -3, 3)
        out[f"conv_{w}x{h}_src"] = src
        for order in (1, 2):
            for horiz in (0, 1):
                out[f"conv_{w}x{h}_o{order}_h{horiz}"] = ref.convolve(src, w, order, bool(horiz))
    # --- sor_coupled ---------------------------------------------------------------------------
    for (w, h) in ((67, 45), (64, 48)):
        rng = np.random.default_rng(2000 + w)
        s0 = sor_system(rng, w, h)
        for k, v in s0.items():
            out[f"sor_{w}x{h}_in_{k}"] = v
        for K in (1, 2, 30):
            s = copy_sys(s0)
            ref.sor(s["du"], s["dv"], s["a11"], s["a12"], s["a22"], s["b1"], s["b2"], s["sh"], s["sv"], w, K, 1.9)
            out[f"sor_{w}x{h}_K{K}_du"] = s["du"]
            out[f"sor_{w}x{h}_K{K}_dv"] = s["dv"]
            if K == 1:
                for k in ("a11", "a12", "a22"):
                    out[f"sor_{w}x{h}_inv_{k}"] = s[k]
        s = copy_sys(s0)
        ref.sor(s["du"], s["dv"], s["a11"], s["a12"], s["a22"], s["b1"], s["b2"], s["sh"], s["sv"], w, 30, 1.9, readable=True)
        out[f"sor_{w}x{h}_readable30_du"] = s["du"]
        out[f"sor_{w}x{h}_readable30_dv"] = s["dv"]
    # --- image_warp (2-frame routine fed factor*flow, see oracle.RefLib.image_warp_prescaled) -------
    w, h = 67, 45
    rng = np.random.default_rng(3000)
    src = smooth_noise_color(rng, w, h)
    wx, wy = noise_plane(rng, w, h, -4, 4), noise_plane(rng, w, h, -4, 4)
    wx[0, :5] = 1000
    wy[1, :5] = -1000
    out["warp_src"], out["warp_wx"], out["warp_wy"] = src, wx, wy
    for factor in (-2, -1, 1, 2):
        fwx = orc.plane(*wx.shape); fwy = orc.plane(*wx.shape)
        fwx[...] = np.float32(factor) * wx
        fwy[...] = np.float32(factor) * wy
        d, m = ref.image_warp_prescaled(src, fwx, fwy, w)
        out[f"warp_f{factor}_dst"], out[f"warp_f{factor}_mask"] = d, m
    # --- sub_laplacian, dpsis weight, derivative stack ------------------------------------------------
    rng = np.random.default_rng(4000)
    sl_src, sl_wh, sl_wv, sl_d0 = noise_plane(rng, w, h), noise_plane(rng, w, h, 0, 2), noise_plane(rng, w, h, 0, 2), noise_plane(rng, w, h)
    d = orc.plane(*sl_d0.shape); d[...] = sl_d0
    ref.sub_laplacian(d, sl_src, sl_wh, sl_wv, w)
    out.update(sublap_src=sl_src, sublap_wh=sl_wh, sublap_wv=sl_wv, sublap_dst0=sl_d0, sublap_dst=d)
    im = smooth_noise_color(rng, w, h)
    out["dpsis_im"], out["dpsis_out"] = im, ref.dpsis_weight(im, w)
    w2, h2 = 35, 21
    I1, I2 = smooth_noise_color(rng, w2, h2), smooth_noise_color(rng, w2, h2)
    out["stack_I1"], out["stack_I2"] = I1, I2
    out["stack_out"] = np.stack(ref.get_derivatives(I2, I1, w2))
    # --- penalties ---------------------------------------------------------------------------------------
    x = np.concatenate([rng.uniform(0, 1e-6, 32), rng.uniform(0, 1, 32), rng.uniform(0, 50, 32), [0, 0.25, 0.2499999, 1e-12]]).astype(np.float32)
    out["pen_x"] = x
    for pid in (0, 1, 2, 3, 4):
        for eps in (0.001, 0.05):
            _, s, v = ref.penalty_derivative(pid, eps, 0.5, x)
            out[f"pen_{pid}_{eps}_scalar"], out[f"pen_{pid}_{eps}_vec"] = s, v
    out["meta_sizes"] = np.array([67, 45, 64, 48, 35, 21], dtype=np.int32)
    path = os.path.join(HERE, "ref_vectors.npz")
    np.savez_compressed(path, **{k: np.ascontiguousarray(v) for k, v in out.items()})
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
