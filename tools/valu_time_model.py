#!/usr/bin/env python3
"""VALU-time model of the two VALU-bound kernels, from their ISA and the measured pass costs.

What it does: compiles sor_chain.hip / kernels.hip to gfx950 assembly (device code only, the product's flags), finds
  * the compute loops of a k_sor_chain shape -- one per chain_compute instance: the unrolled body of PD x CH hyperplane steps of a stage --, the IN / OUT / FILL
    waves' per-interval loops, and
  * the static VALU mix of a k_assemble_images instance,
classifies every VALU instruction (packed fp32, DPP, transcendental, the rest) and prices it with the pass costs tools/ubench/valu_rates.hip measured on MI355X
with two waves per SIMD: v_pk_*_f32 4.4 cycles of the SIMD per wave64 instruction, a DPP move 4.4 (two passes each), a plain fp32 / integer
VALU instruction 2.3; v_rcp / v_sqrt / v_rsq ... are priced at two plain instructions (MI355X_MICROARCH.md: transcendentals issue at half rate).

Output (profiles/<tag>_valu_model.json): per solver stage the VALU cycles of ONE hyperplane step, per I/O wave the VALU cycles of one barrier interval, the
wave -> SIMD placement of the shape (waves i, i + 4, i + 8 of a workgroup share a SIMD; ChainShape::stage_of_wave), and per assembly instance the cycles per VALU
instruction of its mix.  bench.py turns these into `valu_time_floor_frac` with the LIVE launch durations and the launch geometry:

    solver:    floor = workgroups / CUs x (VALU cycles the busiest SIMD of a workgroup needs over the workgroup's life) / clock;   frac = floor / launch duration
    assembly:  floor = waves x dynamic VALU instructions per wave (SQ counters) x cycles per instruction of the static mix / (SIMDs x clock)

usage: tools/valu_time_model.py [tag]        (container: needs hipcc, no GPU)"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "slowflow_amd", "csrc")
COST = {"packed": 4.4, "dpp": 4.4, "trans": 4.6, "plain": 2.3}
COST_SOURCE = ("tools/ubench/valu_rates.hip on MI355X, cycles of the SIMD per wave64 instruction at two waves per SIMD (DESIGN.md 5.1): v_pk_mul/add_f32 8.78 / 2, "
               "v_mov_b32_dpp 8.73 / 2, v_mul/add_f32 4.63 / 2; transcendentals = two plain instructions (MI355X_MICROARCH.md)")


def compile_s(src, extra=()):
    out = os.path.join(tempfile.mkdtemp(prefix="sfa_isa_"), os.path.basename(src).replace(".hip", ".s"))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only",
           "-I" + os.path.join(ROOT, "include"), *extra, src, "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def kernel_instructions(lines, mangled_substr):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and mangled_substr in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
    labels, ins = {}, []
    for i in range(start, end):
        l = lines[i].split(";")[0].rstrip()
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        t = l.strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        ins.append(t)
    return ins, labels


def valu_class(t):
    op = t.split()[0]
    if not op.startswith("v_"):
        return None
    if op.startswith("v_pk_"):
        return "packed"
    if "dpp" in op or "row_" in t or "wave_shr" in t or "wave_shl" in t:
        return "dpp"
    if op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")):
        return "trans"
    return "plain"


def mix_of(body):
    c = collections.Counter()
    for t in body:
        k = valu_class(t)
        if k:
            c[k] += 1
        else:
            op = t.split()[0]
            c["lds" if op.startswith("ds_") else "vmem" if op.startswith(("buffer_", "global_", "flat_", "scratch_")) else "wait" if op.startswith("s_waitcnt")
              else "barrier" if op.startswith("s_barrier") else "salu"] += 1
    return c


def valu_cycles(c):
    return sum(COST[k] * c.get(k, 0) for k in COST)


def loops(ins, labels, minlen=30):
    """backward branches = loops: (first, last) instruction indices, outermost duplicates removed"""
    out = []
    for idx, t in enumerate(ins):
        op = t.split()[0]
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= idx and idx - labels[tgt] >= minlen:
                out.append((labels[tgt], idx))
    return out


def solver_shape(lines, FA, NA, FB, NB_, CH, PD):
    """the stages of k_sor_chain<FA, NA, FB, NB_, CH, PD, ...>: (role, F) -> VALU cycles per step"""
    sub = "k_sor_chainILi%dELi%dELi%dELi%dELi%dELi%dE" % (FA, NA, FB, NB_, CH, PD)
    ins, labels = kernel_instructions(lines, sub)
    steps = PD * CH
    stages, io = [], {}
    byhead = {}
    for a, b in loops(ins, labels):                                   # several backward branches to one label: the loop is the longest of them
        byhead[a] = max(b, byhead.get(a, b))
    all_loops = sorted(byhead.items())
    for a, b in all_loops:
        if any(a2 <= a and b <= b2 and (a2, b2) != (a, b) for a2, b2 in all_loops):
            continue                                                  # nested (the bounded waits inside the IN wave's interval loop)
        body = ins[a:b + 1]
        c = mix_of(body)
        if c.get("packed", 0) >= 13 * steps:                       # a compute loop: 13 packed operations per sweep and step (sor_point2)
            F = round(c["packed"] / (13.0 * steps))
            role = "first (operand loads + ring fill)" if c.get("vmem", 0) else "ring-fed"
            stages.append({"F": F, "role": role, "steps_per_body": steps, "instructions_per_step": round(len(body) / steps, 2),
                           "per_step": {k: round(v / steps, 3) for k, v in sorted(c.items())}, "valu_cycles_per_step": round(valu_cycles(c) / steps, 2)})
        elif c.get("barrier", 0) >= 1:
            who = "OUT" if any("buffer_atomic" in t for t in body) else "FILL" if any(" lds" in t and t.startswith(("global_load", "buffer_load")) for t in body) else "IN"
            io[who] = {"barriers_per_body": c["barrier"], "instructions_per_interval": round(len(body) / c["barrier"], 1),
                       "per_interval": {k: round(v / c["barrier"], 2) for k, v in sorted(c.items())}, "valu_cycles_per_interval": round(valu_cycles(c) / c["barrier"], 2)}
    return {"mangled": sub, "static_instructions": len(ins), "stages": stages, "io_waves": io}


def placement(FA, NA, FB, NB_):
    """wave -> role of ChainShape<FA, NA, FB, NB_> (sor_chain.hip: stage_of_wave; wave 0 = IN, NW + 1 = OUT, NW + 2 = FILL for the one-sweep shapes) and the SIMD
    of every wave (wave mod 4)"""
    NW = NA + NB_
    infill = FA == 1 and NB_ == 0 and NW < 8
    nwaves = NW + 2 + (1 if infill else 0)
    perm9 = NW == 7 and NA == 1
    perm9l = NW == 7 and NA == 6 and NB_ == 1

    def stage_of_wave(wave):
        if perm9:
            return 0 if wave == 4 else (wave if wave < 4 else wave - 1)
        if perm9l:
            return 6 if wave == 4 else (wave - 1 if wave < 4 else wave - 2)
        return wave - 1
    simds = [[] for _ in range(4)]
    for wv in range(nwaves):
        if wv == 0:
            r = "IN"
        elif wv == NW + 1:
            r = "OUT"
        elif wv == NW + 2:
            r = "FILL"
        else:
            st = stage_of_wave(wv)
            r = "stage %d (F=%d)" % (st, FA if st < NA else FB)
        simds[wv % 4].append(r)
    return simds


def assemble_instance(lines, mangled_substr):
    ins, labels = kernel_instructions(lines, mangled_substr)
    c = mix_of(ins)
    nv = sum(c.get(k, 0) for k in COST)
    # the term loop (the largest loop) carries nearly all dynamic instructions: its mix is the one the dynamic count is priced with
    ls = loops(ins, labels, 200)
    big = max(ls, key=lambda ab: ab[1] - ab[0]) if ls else (0, len(ins) - 1)
    cl = mix_of(ins[big[0]:big[1] + 1])
    nvl = sum(cl.get(k, 0) for k in COST)
    return {"mangled": mangled_substr, "static_valu_instructions": nv, "static_mix": {k: c.get(k, 0) for k in COST},
            "term_loop_valu_instructions": nvl, "term_loop_mix": {k: cl.get(k, 0) for k in COST},
            "cycles_per_valu_instruction": round(valu_cycles(c) / nv, 4), "cycles_per_valu_instruction_term_loop": round(valu_cycles(cl) / max(nvl, 1), 4)}


def build_model():
    return _build()


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    model = _build()
    try:
        model["head"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    out = os.path.join(ROOT, "profiles", tag + "_valu_model.json")
    with open(out, "w") as f:
        json.dump(model, f, indent=1)
    for name, m in model["solver"].items():
        print(name, [(s["F"], s["role"][:5], s["valu_cycles_per_step"], s["instructions_per_step"]) for s in m["stages"]], {k: v["valu_cycles_per_interval"] for k, v in m["io_waves"].items()})
        print("   SIMDs:", m["simd_placement"])
    for name, m in model["assemble"].items():
        print(name, m["static_valu_instructions"], m["cycles_per_valu_instruction"], m["cycles_per_valu_instruction_term_loop"])
    print("wrote", out)


def _build():
    sor_s = open(compile_s(os.path.join(CSRC, "sor_chain.hip"))).read().split("\n")
    ker_s = open(compile_s(os.path.join(CSRC, "kernels.hip"), ["-fno-slp-vectorize"])).read().split("\n")
    model = {"costs_cycles_per_wave64_instruction": COST, "cost_source": COST_SOURCE, "clock_ghz": 2.4, "solver": {}, "assemble": {}}
    # the shapes the library launches by default (sor.hip chain_choice): seven stages of 2,2,2,2,2,2,3 from 73 bands on, five one-sweep stages below
    for name, (FA, NA, FB, NB_, CH, PD) in {"k_sor_chain<2,6,3,1,4,2": (2, 6, 3, 1, 4, 2), "k_sor_chain<1,5,1,0,4,4": (1, 5, 1, 0, 4, 4), "k_sor_chain<3,3,2,3,4,2": (3, 3, 2, 3, 4, 2)}.items():
        m = solver_shape(sor_s, FA, NA, FB, NB_, CH, PD)
        m["shape"] = {"FA": FA, "NA": NA, "FB": FB, "NB": NB_, "CH": CH, "PD": PD, "KG": FA * NA + FB * NB_, "NW": NA + NB_}
        m["simd_placement"] = placement(FA, NA, FB, NB_)
        model["solver"][name] = m
    # k_assemble_images<TY, threads, blocks per CU (launch bound), ZUV, FAST, XT>: FAST 1 = the cfg's defaults folded in (the bench), 2 = Lorentzian (config 5), 0 = generic
    for fast in (1, 2, 0):
        for zuv in (1, 0):
            sub = "k_assemble_imagesILi8ELi512ELi6ELb%dELi%dELb1E" % (zuv, fast)
            model["assemble"]["k_assemble_images<8,512,6,%s,%d,true>" % ("true" if zuv else "false", fast)] = assemble_instance(ker_s, sub)
    return model


if __name__ == "__main__":
    main()
