#!/usr/bin/env python3
"""Build-time check of the band kernel's counted publish (sor.hip publishes a band's progress word behind a hand-counted
`s_waitcnt vmcnt(N)` and relies on at least N+1 vector-memory instructions having been issued, in program order, after the edge store it
must cover -- vmcnt retires in order, so with <= N operations outstanding the store, older than all of them, has completed).

The script compiles sor.hip to ISA and, for every instantiated k_sor_band shape, walks each `s_waitcnt vmcnt(N)` that is followed by the
progress-word store (a global_store_dword ... sc1 within the next instructions, the only sc1 dword stores of the kernel) BACKWARDS to the
preceding edge store (global_store_dwordx2 ... sc1) and counts the VMEM instructions in between along the straight-line layout of the macro
chunk.  It fails when any count is < N.  Run: python tools/check_publish_vmcnt.py   (tests/test_abi.py runs it when hipcc is present)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "slowflow_amd", "csrc", "sor.hip")
VMEM = re.compile(r"^\s*(buffer_|global_|flat_|scratch_)(load|store|atomic)")


def isa():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "sor.s")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.dirname(SRC), "-S",
                            "--cuda-device-only", "-o", out, SRC], capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit("hipcc failed:\n" + r.stderr)
        return open(out).read().split("\n")


def kernels(lines):
    cur, body = None, []
    for l in lines:
        m = re.match(r"^(_ZN3sfa1[06]k_sor_band\S*):", l)
        if m:
            cur, body = m.group(1), []
        elif cur and ".end_amdhsa_kernel" in l:
            yield cur, body
            cur = None
        elif cur:
            body.append(l)


def check(name, body):
    """returns list of (N, lower bound on the VMEM instructions between the macro-chunk loop's top and the hand-written wait) for every counted
    publish of this kernel.  Dynamic order: edge store (end of the macro-chunk body) -> branch to the loop top -> chunk 0 of the next macro chunk
    (its operand refills) -> the wait.  Only the refills (buffer_load_dwordx4: never part of a poll loop) are counted, walking back from the wait
    to the label of the enclosing depth-1 loop -- a lower bound on what has been issued since the store."""
    found = []
    ins, hand, header = [], set(), set()   # layout order; indices written by hand (inline asm); indices that are depth-1 loop headers
    in_app = False
    for l in body:
        t = l.strip()
        if t.startswith((";APP", ";;#ASMSTART")):
            in_app = True
        elif t.startswith((";NO_APP", ";;#ASMEND")):
            in_app = False
        elif re.match(r"^\.LBB\d+_\d+:", l):
            if "Loop Header: Depth=1" in l:
                header.add(len(ins))
            ins.append(l)
        elif l.startswith("\t") and not t.startswith((".", ";")):
            if in_app:
                hand.add(len(ins))
            ins.append(l)
    for i, l in enumerate(ins):
        m = re.match(r"\s*s_waitcnt vmcnt\((\d+)\)\s*$", l)
        if not m or int(m.group(1)) == 0 or i not in hand:
            continue                       # the compiler's own data-dependency waits are not the protocol's
        n = int(m.group(1))
        cnt = 0
        j = i - 1
        while j >= 0 and j not in header:
            if re.match(r"\s*buffer_load_dwordx4\s", ins[j]):
                cnt += 1
            j -= 1
        found.append((n, cnt if j >= 0 else None))
    return found


def main():
    bad = 0
    total = 0
    for name, body in kernels(isa()):
        res = check(name, body)
        shape = re.findall(r"Li(\d+)E", name)
        tag = ("k_sor_band_mixed<%s>" if "mixed" in name else "k_sor_band<%s>") % ",".join(shape) if shape else name
        for n, cnt in res:
            total += 1
            ok = cnt is not None and cnt >= n   # <= n outstanding and >= n issued after the store: the store is not among them
            print("%-28s s_waitcnt vmcnt(%d): >= %s operand loads issued since the edge store -> %s" % (tag, n, cnt if cnt is not None else "n/a", "ok" if ok else "VIOLATED"))
            bad += not ok
    if total == 0:
        raise SystemExit("no counted publish found: the kernel changed shape, update this checker")
    if bad:
        raise SystemExit("%d counted publish(es) no longer cover their edge store" % bad)
    print("all %d counted publishes cover their edge store" % total)


if __name__ == "__main__":
    main()
