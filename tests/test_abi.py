"""CPU-side checks of the drop-in boundary: the HIP library builds for gfx950, loads, exports every symbol that
include/slowflow_amd.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

import slowflow_amd as sfa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    if not os.path.exists(sfa.LIB_PATH):
        sfa.build()
    return sfa.LIB_PATH


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "slowflow_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(sfa_[a-z0-9_]+|sor_coupled|variational)\s*\(", src)
    return sorted(set(names))


def test_header_symbols_exported(libpath):
    L = C.CDLL(libpath)
    decl = declared_symbols()
    assert len(decl) >= 30
    missing = [n for n in decl if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(set(sfa.EXPORTS)) == decl


def test_struct_layouts_match_reference_images():
    # image_t: int width,height,stride; float* data (epic_flow_extended/image.h:17-23)
    assert C.sizeof(sfa.Image) == 24 and sfa.Image.data.offset == 16
    assert C.sizeof(sfa.Params) == C.sizeof(C.c_int) * 8 + 4 * 6 + 12 * 3 + 2 * 4 * sfa.MAX_REF + 4 + 12 * 2 + 4 * 4 + 4 * 3 + 4      # rho, omega: SFA_MAX_REF floats each (8 since round 5); + sor_order (additive)
    assert sfa.MAX_REF == 8


def test_params_default_matches_driver_defaults(libpath):
    p = sfa.default_params()       # slow_flow.cpp:64-128
    assert (p.S, p.smoothing, p.dataterm_norm, p.niter_alter, p.niter_outer, p.niter_inner, p.niter_solver) == (2, 1, 1, 10, 10, 1, 30)
    assert abs(p.sor_omega - 1.9) < 1e-6 and p.alpha == 4.0 and p.gamma == 6.0 and p.delta == 1.0
    assert p.robust_color.id == 1 and abs(p.robust_color.eps - 0.001) < 1e-9
    assert list(p.omega)[:2] == [0.0, 2.0] and p.layers == 1 and abs(p.p_scale - 0.9) < 1e-6
    assert abs(p.occlusion_penalty - 0.1) < 1e-7 and abs(p.occlusion_alpha - 0.1) < 1e-7 and p.niter_graphc == 10       # slow_flow.cpp:117-118


def test_pyramid_sizes_host_logic(libpath):
    L = sfa.lib()
    ws, hs = (C.c_int * 64)(), (C.c_int * 64)()
    n = L.sfa_pyramid_sizes(1024, 436, 5, C.c_float(0.9), ws, hs)
    assert n == 5
    assert [ws[i] for i in range(5)] == [1024, 921, 828, 745, 670]
    assert [hs[i] for i in range(5)] == [436, 392, 352, 316, 284]
    n = L.sfa_pyramid_sizes(2048, 2048, 6, C.c_float(0.9), ws, hs)
    assert [ws[i] for i in range(n)] == [2048, 1843, 1658, 1492, 1342, 1207]
    n = L.sfa_pyramid_sizes(256, 256, 4, C.c_float(0.9), ws, hs)
    assert [ws[i] for i in range(n)] == [256, 230, 207, 186]
    # the size break: levels stop when floor(w*p) <= order+1 = 4 (variational_mt.cpp:647)
    n = L.sfa_pyramid_sizes(9, 9, 10, C.c_float(0.9), ws, hs)
    assert 1 <= n < 10


def test_pyramid_sizes_match_oracle(libpath, oracle):
    L = sfa.lib()
    ws, hs = (C.c_int * 64)(), (C.c_int * 64)()
    for (w, h, layers, p) in [(1024, 436, 5, 0.9), (67, 45, 8, 0.9), (130, 98, 6, 0.8), (33, 200, 30, 0.95)]:
        n = L.sfa_pyramid_sizes(w, h, layers, C.c_float(p), ws, hs)
        assert [(ws[i], hs[i]) for i in range(n)] == oracle.pyramid_sizes(w, h, layers, p)


def test_no_cpu_fallback(libpath):
    """Without a GPU the product must fail loudly; with one this test is vacuous (covered by the gpu tests)."""
    if sfa.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(sfa.SlowflowError) as e:
        sfa.Context(0)
    assert "no HIP device" in str(e.value) or "-3" in str(e.value)


def test_product_does_not_touch_the_oracle():
    """the product path may not import, link or call anything under oracle/"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "slowflow_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "slowflow_oracle" not in txt and "import oracle" not in txt and "orc_" not in txt, fn


def test_counted_publish_covers_the_edge_store():
    """the band kernel publishes a band's progress behind a hand-counted `s_waitcnt vmcnt(N)`; the count is only right while at least
    N vector-memory instructions are issued between the edge store and the wait.  tools/check_publish_vmcnt.py verifies that on the ISA of every
    instantiated shape and role; a toolchain bump or a change of the refill code that breaks it fails here, not as a rare stale read on the GPU."""
    import shutil
    import subprocess
    import sys
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_publish_vmcnt.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "counted publishes cover their edge store" in r.stdout, r.stdout + r.stderr


def test_no_wait_inside_the_dma_issue_of_the_assembly_kernel():
    """Round 4: twice a register of the DMA issue loop's address arithmetic had a load pending (the prologue's, then the masks') and the compiler's wait-count
    pass put an `s_waitcnt vmcnt(0)` in front of the first piece -- a whole memory round trip in every staging round, 4-30 % of the kernel.  Checked on the ISA
    of every instance (tools/isa_dma_waits.py compiles kernels.hip with the product's flags)."""
    import shutil
    import subprocess
    import sys
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_dma_waits.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "no wait inside any DMA issue phase" in r.stdout, r.stdout + r.stderr


def test_the_assembly_kernel_keeps_three_blocks_per_cu():
    """Round 4: the data-term kernel's 15 % came from a third block per CU (79 registers, 48 KB of LDS).  tools/check_asm_occupancy.py reads the compiler's own
    resource report for every instance of kernels.hip built with the product's flags."""
    import shutil
    import subprocess
    import sys
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_occupancy.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "three blocks per CU for every folded instance" in r.stdout, r.stdout + r.stderr


def test_release_build_is_the_same_abi_without_the_switches(libpath):
    """`make -C slowflow_amd/csrc release` (-DSFA_RELEASE) compiles out every cross-check / what-if path, the SFA_DEBUG environment gate and the
    non-default solver shapes.  The result exports exactly the functions of the full build (= include/slowflow_amd.h), refuses sfa_debug_set by name, holds no
    switch name and no getenv, and is smaller; tools/release_report.py prints both builds side by side (its GPU half: tests/test_host.py)."""
    import shutil
    import subprocess
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "slowflow_amd", "csrc"), "-j4", "release"], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rel = os.path.join(ROOT, "slowflow_amd", "csrc", "build_release", "libslowflow_amd.so")

    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
        return sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert exported(rel) == exported(libpath)
    L = C.CDLL(rel)
    assert not [n for n in declared_symbols() if not hasattr(L, n)]
    L.sfa_last_error.restype = C.c_char_p
    assert L.sfa_debug_set(b"SFA_UNFUSED", b"1") != 0 and b"release build" in L.sfa_last_error(None)
    blob = open(rel, "rb").read()
    full = open(libpath, "rb").read()
    assert b"SFA_SOR_CHAIN" in full and b"SFA_DEBUG" in full                       # (the full build does carry the table and the gate)
    assert b"SFA_SOR_CHAIN" not in blob and b"SFA_UNFUSED" not in blob and b"SFA_DEBUG\0" not in blob
    und = subprocess.run(["nm", "-D", "--undefined-only", rel], capture_output=True, text=True).stdout
    assert "getenv" not in und
    assert len(blob) < 0.75 * len(full)
