#!/usr/bin/env python3
"""Golden vectors of the reference's ORIGINAL two-frame refinement: inputs and the (wx, wy) that the compiled reference's own
`variational()` (epic_flow_extended/variational.c:101, built by oracle/Makefile into oracle/_ref) returns for them.
Run in the build container (needs /root/reference):  python tests/golden/make_golden_2frame.py -> tests/golden/ref_two_frame.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
import oracle as orc  # noqa: E402
from synth import noise_plane, smooth_noise_color  # noqa: E402

CASES = {"default": dict(), "color_inner": dict(delta=0.5, niter_outer=3, niter_inner=2), "weights": dict(alpha=3.0, gamma=0.2, niter_solver=7, sor_omega=1.5)}


def main():
    ref = orc.RefLib()
    w, h = 67, 45
    rng = np.random.default_rng(7)
    big = smooth_noise_color(rng, w + 8, h + 8, 40)
    a, b = orc.aligned_zeros((3, h, orc.stride_of(w))), orc.aligned_zeros((3, h, orc.stride_of(w)))
    a[:, :, :w] = big[:, 4:4 + h, 4:4 + w]
    b[:, :, :w] = big[:, 3:3 + h, 2:2 + w]
    wx0, wy0 = noise_plane(rng, w, h, 1.5, 2.5), noise_plane(rng, w, h, 0.5, 1.5)
    out = {"im1": a, "im2": b, "wx0": wx0, "wy0": wy0, "size": np.array([w, h], np.int32)}
    for name, kw in CASES.items():
        wx, wy = orc.plane(*wx0.shape), orc.plane(*wx0.shape)
        wx[...] = wx0; wy[...] = wy0
        ref.variational_2frame(wx, wy, a, b, w, orc.params_2f(**kw))
        out[f"{name}_wx"], out[f"{name}_wy"] = wx[:, :w].copy(), wy[:, :w].copy()
    path = os.path.join(HERE, "ref_two_frame.npz")
    np.savez_compressed(path, **{k: np.ascontiguousarray(v) for k, v in out.items()})
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
