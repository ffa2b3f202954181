#!/usr/bin/env python3
"""Build-time check of the solver kernels' counted publishes.

Three kernels publish progress words behind hand-counted `s_waitcnt vmcnt(N)` (vmcnt retires in issue order -- loads, stores and atomics
together -- so with <= N operations outstanding everything older than the N youngest has completed):
  * k_sor_band / k_sor_band_mixed (sor.hip): the band's edge store, see below;
  * k_sor_solve<F,CH> (sor.hip): a chunk's iterate stores, covered by the operand loads of the next chunk (2 * PUB - 1 a quarter into the chunk for the
    long-chunk shapes, SFA_PUBLISH_VMCNT behind the chunk for the others);
  * k_sor_chain (sor_chain.hip): the OUT wave issues exactly T buffer atomics per barrier interval and waits for vmcnt(PUBD * T - 1).
`flat_*` memory instructions return out of order and would void every one of these waits: none may appear in any k_sor_* kernel.

Original description (band kernel): build-time check of the band kernel's counted publish (sor.hip publishes a band's progress word behind a hand-counted
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
SRC_CHAIN = os.path.join(ROOT, "slowflow_amd", "csrc", "sor_chain.hip")
VMEM = re.compile(r"^\s*(buffer_|global_|flat_|scratch_)(load|store|atomic)")


def isa(src=SRC):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "sor.s")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.dirname(src), "-S",
                            "--cuda-device-only", "-o", out, src], capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit("hipcc failed:\n" + r.stderr)
        return open(out).read().split("\n")


def kernels(lines, pat=r"^(_ZN3sfa1[06]k_sor_band\S*):"):
    cur, body = None, []
    for l in lines:
        m = re.match(pat, l)
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


def instructions(body):
    """layout-ordered instructions and labels; which of them were written by hand (inline asm); which labels head a depth-1 loop"""
    ins, hand, header = [], set(), set()
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
    return ins, hand, header


def check_task(name, body):
    """k_sor_solve: every hand-written counted wait must have at least N operand loads (global_load_dwordx4) between the top of the chunk loop and
    itself -- the iterate stores it covers were issued in the previous trip of that loop."""
    ins, hand, header = instructions(body)
    found = []
    for i, l in enumerate(ins):
        m = re.match(r"\s*s_waitcnt vmcnt\((\d+)\)\s*$", l)
        if not m or int(m.group(1)) == 0 or i not in hand:
            continue
        n, cnt, j = int(m.group(1)), 0, i - 1
        while j >= 0 and j not in header:
            if re.match(r"\s*global_load_dwordx4\s", ins[j]):
                cnt += 1
            j -= 1
        found.append((n, cnt if j >= 0 else None))
    return found


def check_chain(name, body):
    """k_sor_chain: in the OUT wave's loop (the depth-1 loop whose hand-written waits are followed by a buffer_atomic_umax) consecutive counted
    waits must be separated by exactly T = (N + 1) / PUBD vector-memory instructions, around the loop's back edge too."""
    pubd = int(re.findall(r"Li(\d+)E", name)[-1])
    ins, hand, header = instructions(body)
    found = []
    for h in sorted(header):
        # extent of the loop: up to the last branch back to this label
        lab = re.match(r"^(\.LBB\d+_\d+):", ins[h]).group(1)
        ends = [i for i in range(h, len(ins)) if re.match(r"\s*s_cbranch\S*\s+" + re.escape(lab) + r"\b|\s*s_branch\s+" + re.escape(lab) + r"\b", ins[i])]
        if not ends:
            continue
        e = ends[-1]
        waits = [i for i in range(h, e) if i in hand and re.match(r"\s*s_waitcnt vmcnt\((\d+)\)\s*$", ins[i]) and int(re.match(r"\s*s_waitcnt vmcnt\((\d+)\)", ins[i]).group(1)) > 0]
        waits = [i for i in waits if any("buffer_atomic_umax" in ins[k] for k in range(i, min(i + 12, e)))]
        if not waits:
            continue
        for k, i in enumerate(waits):
            n = int(re.match(r"\s*s_waitcnt vmcnt\((\d+)\)", ins[i]).group(1))
            prev = waits[k - 1]
            span = list(range(prev + 1, i)) if k > 0 else list(range(prev + 1, e)) + list(range(h, i))
            cnt = sum(1 for j in span if VMEM.match(ins[j]))
            found.append((n, cnt, (n + 1) // pubd, (n + 1) % pubd == 0))
    return found


def main():
    bad = 0
    total = 0
    lines = isa()
    lines_chain = isa(SRC_CHAIN)
    # no flat_* memory instruction in any solver kernel
    for src_lines in (lines, lines_chain):
        for name, body in kernels(src_lines, r"^(_ZN3sfa\d+k_sor_\S*):"):
            nflat = sum(1 for l in body if re.match(r"^\s*flat_(load|store|atomic)", l))
            if nflat:
                print("%s: %d flat_* memory instructions -> VIOLATED" % (name, nflat))
                bad += 1
    for name, body in kernels(lines, r"^(_ZN3sfa11k_sor_solve\S*):"):
        shape = re.findall(r"Li(\d+)E", name)
        for n, cnt in check_task(name, body):
            total += 1
            ok = cnt is not None and cnt >= n
            print("%-28s s_waitcnt vmcnt(%d): >= %s operand loads issued since the iterate stores -> %s" % ("k_sor_solve<%s>" % ",".join(shape), n, cnt if cnt is not None else "n/a", "ok" if ok else "VIOLATED"))
            bad += not ok
    nchain = 0
    for name, body in kernels(lines_chain, r"^(_ZN3sfa11k_sor_chain\S*):"):
        shape = re.findall(r"Li(\d+)E", name)
        for n, cnt, T, divisible in check_chain(name, body):
            total += 1; nchain += 1
            ok = divisible and cnt == T
            print("%-40s OUT wave s_waitcnt vmcnt(%d): %d memory instructions per interval, T = %d -> %s" % ("k_sor_chain<%s>" % ",".join(shape), n, cnt, T, "ok" if ok else "VIOLATED"))
            bad += not ok
    if nchain == 0:
        raise SystemExit("no counted publish found in k_sor_chain: the kernel changed shape, update this checker")
    for name, body in kernels(lines):
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
