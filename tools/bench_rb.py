"""the labelled red-black mode (slow_flow_sor_order = 1): one window's whole path and the solver's share, for the form SFA_RB_TILE selects
(0 = one launch per colour pass, 3 / 5 = LDS tiles with that many sweeps per visit).  The switch is read once per process: run it once per form."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import slowflow_amd as sfa, bench
ctx = sfa.Context(0)
lex_ms, lex_sor, rb = bench.one_window_latency(ctx, reps=5)
print("SFA_RB_TILE=%s  reference order: %.2f ms / window, %.4f ms / solve;  red-black: %.2f ms / window, %.4f ms / solve, deviation %.5f px" % (
    os.environ.get("SFA_RB_TILE", "(default 5)"), lex_ms, lex_sor, rb["latency_one_window_ms"], rb["sor_ms_per_solve_avg_over_levels"],
    rb["max_abs_flow_deviation_from_reference_order_px"]), flush=True)
ctx.close()
