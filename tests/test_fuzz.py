"""Randomised parity on the GPU box (round 6): a short, seeded slice of tools/fuzz_parity.py and tools/fuzz_driver.py in every `-m gpu` run -- the long runs are recorded in
profiles/r06_fuzz_parity.txt (about 500 000 cases; the first one found a memory fault in the warp kernels' index conversion, fixed since)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_randomised_parity_slice():
    """25 s of seeded random cases -- solver batches (IEEE ==), levels and pyramids (the tolerance of test_gpu_parity.py), lockstep batches of different windows under
    thresholds (bit for bit what each gives alone), the operator entry points down to one pixel, exact cuts, the two-frame refinement and the red-black mode (bit for bit),
    occlusion runs -- without a failure; each line of the child's output names its case"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "25", "900000"], capture_output=True, text=True, timeout=600)
    tail = "\n".join(l for l in r.stdout.splitlines() if " ok  " not in l)[-3000:]
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert "Memory access fault" not in r.stdout + r.stderr
    n = sum(1 for l in r.stdout.splitlines() if " ok  " in l)
    assert n >= 100, (n, tail)
    kinds = {l.split()[4] for l in r.stdout.splitlines() if " ok  " in l and len(l.split()) > 4}
    assert {"sor", "level", "batch", "stage", "cut"} <= kinds, kinds


@pytest.mark.gpu
def test_randomised_driver_layouts_slice():
    """20 s of random sequences through the slow_flow driver under random layouts (virtual GPUs, workers, windows per job) against the plain layout: every .flo byte for byte"""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "slowflow_amd", "host")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_driver.py"), "20", "7000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert sum(1 for l in r.stdout.splitlines() if " ok  " in l) >= 5, r.stdout[-2000:]
