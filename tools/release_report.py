#!/usr/bin/env python3
"""The two builds of the library side by side: file size, kernels in the code object, exported symbols, and -- on a GPU -- the time of the first
sfa_ctx_create of a process (code-object load + context) and a small parity run (smoke()'s) through each.

usage: tools/release_report.py            (builds both; the GPU part runs when a device is present)
       tools/release_report.py child LIB  (internal: one library in a fresh process)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = os.path.join(ROOT, "slowflow_amd", "libslowflow_amd.so")
REL = os.path.join(ROOT, "slowflow_amd", "csrc", "build_release", "libslowflow_amd.so")


def child(lib):
    os.environ["SFA_LIB"] = lib
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import slowflow_amd as sfa
    L = sfa.lib()
    out = {"lib": os.path.relpath(lib, ROOT), "debug_set_rc": L.sfa_debug_set(b"SFA_UNFUSED", b"1")}
    L.sfa_debug_set(b"SFA_UNFUSED", None)
    if sfa.device_count() > 0:
        t0 = time.perf_counter()
        ctx = sfa.Context(0)
        ctx.sync()
        out["first_ctx_create_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        t0 = time.perf_counter()
        c2 = sfa.Context(0); c2.sync()
        out["second_ctx_create_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        c2.close(); ctx.close()
        import __graft_entry__ as g
        t0 = time.perf_counter()
        g.smoke()                                   # SOR bit-identical + a two-level refinement within 1e-4 of the oracle, through THIS library
        out["smoke_s"] = round(time.perf_counter() - t0, 2)
        # the bench's solver shapes are in both builds: 16 systems at 1024 x 436 pick the seven-stage shape, one system the one-sweep shape
        import numpy as np
        from synth import sor_system
        ctx = sfa.Context(0)
        names = []
        for nb in (1, 16):
            sb = sfa.SorBatch(ctx, 1024, 436, nb)
            s = sor_system(np.random.default_rng(1), 1024, 436)
            for b in range(nb):
                sb.upload(b, *[np.ascontiguousarray(s[k]) for k in ("du", "dv", "a11", "a12", "a22", "b1", "b2", "sh", "sv")])
            ctx.profile_enable(True); sb.run(30, 1.9); names.append(ctx.profile_read_kernels()[3].split(" ")[0]); ctx.profile_enable(False)
            sb.close()
        # a sweep count no chain shape divides: the fallback kernel of each build (full: band kernel for batches; release: the task kernel)
        sb = sfa.SorBatch(ctx, 130, 98, 8)
        s = sor_system(np.random.default_rng(2), 130, 98)
        for b in range(8):
            sb.upload(b, *[np.ascontiguousarray(s[k]) for k in ("du", "dv", "a11", "a12", "a22", "b1", "b2", "sh", "sv")])
        ctx.profile_enable(True); sb.run(7, 1.9); names.append(ctx.profile_read_kernels()[3].split(" ")[0]); ctx.profile_enable(False)
        du, dv = sb.download(3)
        import oracle as orc
        from synth import copy_sys
        a = copy_sys(s)
        orc.Oracle().sor(a["du"], a["dv"], a["a11"], a["a12"], a["a22"], a["b1"], a["b2"], a["sh"], a["sv"], 130, 7, 1.9)
        out["fallback_K7_bit_identical"] = bool(np.array_equal(a["du"][:, :130], du[:, :130]) and np.array_equal(a["dv"][:, :130], dv[:, :130]))
        sb.close(); ctx.close()
        out["solver_kernels_1_16_K7"] = names
    print("REPORT " + json.dumps(out))


def kernels_in(lib):
    """kernel descriptors of the embedded gfx950 code objects: the `<kernel>.kd` names in their symbol tables (the .hip_fatbin section holds one code object per
    translation unit)"""
    import re
    blob = open(lib, "rb").read()
    return len(set(re.findall(rb"(_Z[\w]+)\.kd\x00", blob)))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        return child(sys.argv[2])
    csrc = os.path.join(ROOT, "slowflow_amd", "csrc")
    t0 = time.perf_counter()
    subprocess.run(["make", "-C", csrc, "-j4"], check=True, capture_output=True)
    subprocess.run(["make", "-C", csrc, "-j4", "release"], check=True, capture_output=True)
    rows = []
    for lib in (FULL, REL):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", lib], capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("REPORT ")]
        row = json.loads(line[0][7:]) if line else {"lib": lib, "error": (r.stdout + r.stderr)[-400:]}
        row["bytes"] = os.path.getsize(lib)
        nm = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
        row["exported_functions"] = sum(1 for l in nm.splitlines() if " T " in l)
        row["gpu_kernels"] = kernels_in(lib)
        rows.append(row)
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
