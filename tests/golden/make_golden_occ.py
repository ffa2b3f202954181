#!/usr/bin/env python3
"""Golden vectors for psi (PenaltyFunction::apply, the v4sf overload optimizeOcc evaluates, variational_aux_mt.cpp:817-827),
produced by the reference's own penalty classes compiled into oracle/_ref (recipe: oracle/Makefile).  Run in the build
container (needs /root/reference):  python tests/golden/make_golden_occ.py  ->  tests/golden/ref_psi_apply.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle as orc  # noqa: E402


def main():
    ref = orc.RefLib()
    rng = np.random.default_rng(2026)
    x = np.concatenate([rng.uniform(0, 1e-6, 64), rng.uniform(0, 1, 64), rng.uniform(0, 50, 64), rng.uniform(0, 1e4, 60), [0, 0.25, 0.2499999, 1e-12]]).astype(np.float32)
    out = {"x": x}
    for pid in (0, 1, 2, 3, 4):
        for eps, trunc in ((0.001, 0.5), (0.05, 0.5), (0.05, 0.02)):
            _, v = ref.penalty_apply(pid, eps, trunc, x)
            out[f"apply_{pid}_{eps}_{trunc}"] = v
    path = os.path.join(HERE, "ref_psi_apply.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
