"""The few-windows solver kernel (slowflow_amd/csrc/sor_chain.hip) splits sor_coupled over workgroups that hand data over through memory
behind progress words.  Its index arithmetic and wait thresholds are transcribed into a CPU model (tools/sim_sor_chain.py) that runs
them under adversarial visibility and workgroup order; the model must reproduce the raster-order oracle bit for bit, and it must FAIL
when a threshold is weakened by one interval (the thresholds are tight, the model is sensitive)."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sim(mode, slack, pubd=2):
    os.environ["SIM_MODE"], os.environ["SIM_SLACK"], os.environ["SIM_PUBD"] = mode, str(slack), str(pubd)
    spec = importlib.util.spec_from_file_location(f"sim_sor_chain_{mode}_{slack}_{pubd}", os.path.join(ROOT, "tools", "sim_sor_chain.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    os.environ.pop("SIM_MODE"); os.environ.pop("SIM_SLACK"); os.environ.pop("SIM_PUBD")
    return m


CASES = [(40, 70, 12, (2, 3, 2, 0), 1), (50, 130, 9, (1, 3, 1, 0), 2), (45, 100, 14, (4, 1, 3, 1), 1), (33, 150, 12, (3, 2, 3, 0), 1), (37, 200, 10, (2, 5, 2, 0), 1),
         (41, 140, 30, (3, 3, 2, 3), 1), (36, 90, 15, (2, 3, 3, 3), 2),          # round 4: the six-stage shapes of mixed width (3,3,3,2,2,2 is the default from 97 bands on)
         (43, 150, 30, (3, 1, 2, 6), 1),                                          # round 5: seven stages of 3,2,2,2,2,2,2 sweeps on nine waves
         (43, 150, 30, (2, 6, 3, 1), 1), (37, 90, 15, (2, 6, 3, 1), 2)]           # ... and of 2,2,2,2,2,2,3: the default from 73 bands on


@pytest.mark.parametrize("mode", ["raw", "war"])
@pytest.mark.parametrize("w,h,K,shape,nb", CASES)
def test_chain_protocol_model_reproduces_the_oracle(mode, w, h, K, shape, nb):
    m = _sim(mode, 0)
    bad, _ = m.run(w, h, K, m.Shape(*shape), nb)
    assert bad == 0


def test_chain_protocol_thresholds_are_tight():
    m = _sim("raw", 1)
    bad, _ = m.run(40, 70, 12, m.Shape(2, 3, 2, 0), 1)
    assert bad > 0


@pytest.mark.parametrize("mode", ["raw", "war"])
@pytest.mark.parametrize("w,h,K,shape,nb", [(41, 140, 30, (3, 3, 2, 3), 1), (50, 130, 10, (1, 5, 1, 0), 2), (40, 70, 12, (2, 3, 2, 0), 1), (43, 150, 30, (2, 6, 3, 1), 1)])
def test_chain_protocol_model_with_one_interval_of_publication_delay(mode, w, h, K, shape, nb):
    """round 4: the default shapes for one to four windows (1 x 5) and from 13 windows on (3,3,3,2,2,2) are launched with PUBD = 1 -- the progress word covers
    the stores of the previous interval (a counted vmcnt wait over T instead of 2 T memory instructions); the consumers' thresholds are in published counts and do
    not change"""
    m = _sim(mode, 0, pubd=1)
    bad, _ = m.run(w, h, K, m.Shape(*shape), nb)
    assert bad == 0


def test_chain_protocol_thresholds_are_tight_with_one_interval_too():
    m = _sim("raw", 1, pubd=1)
    bad, _ = m.run(40, 70, 12, m.Shape(2, 3, 2, 0), 1)
    assert bad > 0


ALL_SHAPES = [(1, 3, 1, 0), (2, 3, 2, 0), (3, 5, 3, 0), (2, 5, 2, 0), (1, 5, 1, 0), (3, 2, 3, 0), (3, 3, 2, 3), (2, 3, 3, 3), (3, 1, 2, 6), (1, 6, 1, 0), (2, 6, 3, 1)]      # kChainShapes with an operand ring


@pytest.mark.parametrize("shape", ALL_SHAPES)
def test_operand_ring_depth_and_read_ahead(shape):
    """the write-after-read argument behind the operand ring's depth lived in a comment.  The model walks every (stage, sweep, step) of the barrier
    lockstep: at the kernel's depth and read-ahead no row is read before it is written or after its slot is rewritten; the derived minimum is tight (one row less: a
    write-after-read clash at one step of read-ahead); a step more of read-ahead than the kernel takes reads a row too early"""
    m = _sim("raw", 0)
    S = m.Shape(*shape)
    opr, oprmin = m.ring_rows(S)
    pf = m.ring_prefetch(S)
    if m.ring_infill(S):
        # round 5: the one-sweep shapes' rows come from a FILL wave by LDS-DMA, four at a time, in flight from the start of interval c to the end of interval c + 1; the
        # first stage reads the ring too.  No hazard at the kernel's depth (a multiple of four rows: a group never wraps), a write-after-read clash two groups below it
        assert opr % 4 == 0 and m.ring_hazards(S, opr, pf, infill=True) == []
        assert {b[0] for b in m.ring_hazards(S, opr - 8, pf, infill=True)} == {"war"}
        return
    assert opr > 0 and m.ring_hazards(S, opr, pf) == []
    assert m.ring_hazards(S, oprmin, 1) == [] and {b[0] for b in m.ring_hazards(S, oprmin - 1, 1)} == {"war"}
    if pf[1] == "chunk":                             # one step more: the row of step 4 c + 5 at the top of chunk c is not written yet for the second stage
        assert m.ring_hazards(S, opr, 1) == [] and {b[0] for b in m.ring_hazards(S, opr, 2)} & {"raw", "raw-in"}
    else:
        assert {b[0] for b in m.ring_hazards(S, opr, (pf[0] + 1, pf[1]))} & {"raw", "raw-self", "raw-in"}          # the first stage cannot read further ahead
        assert {b[0] for b in m.ring_hazards(S, opr - 1, pf)} == {"war"} or opr > oprmin - (pf[1] - 1)               # a shape at its tight depth has no row to spare
    if shape in ((3, 1, 2, 6), (2, 6, 3, 1)):
        assert opr == 51 and {b[0] for b in m.ring_hazards(S, opr, (2, 2))} == {"war"}                            # ... and needs the later stages' third step of read-ahead
