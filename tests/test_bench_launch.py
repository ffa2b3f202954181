"""bench.py --gpus N must really run N ranks: the launch line, the child-process launch itself and the
exchange sequence (barrier, max over ranks, gather of per-window timings), rehearsed on CPU over gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launch_line_is_the_drivers_form():
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # exactly one JSON line: rank 0's
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks():
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--selftest-launch"])
    assert out["n_gpus"] == 2 and out["steps"] == 3
    assert out["seconds_per_window"] == {"n": 8, "nonzero": 8}      # both ranks' windows arrived
    assert out["ms_per_step"] >= 19.0                                  # the MAX over ranks (rank 1 sleeps twice as long)


def test_gpus_1_stays_in_process():
    out = _run(["--gpus", "1", "--steps", "2", "--selftest-launch"])
    assert out["n_gpus"] == 1


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)


import pytest


def _fracs(o, path=""):
    if isinstance(o, dict):
        for k, v in o.items():
            yield from _fracs(v, path + "/" + k)
    elif "frac" in path.rsplit("/", 1)[-1] and isinstance(o, (int, float)):      # `frac` and every *_frac* key
        yield path, o


def test_valu_time_model_is_the_shipped_kernels():
    """`roofline.valu_time_floor_frac` is derived from the ISA of the shipped kernels (tools/valu_time_model.py -> profiles/r06_valu_model.json).  The committed
    model must be what the current sources compile to -- a kernel edit without a new model fails here --, and with it the default 128-window launch of the last
    round's record (1.58 ms mean launch) lies between a third and all of its VALU-time floor"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    import valu_time_model
    fresh = valu_time_model.build_model()
    with open(os.path.join(ROOT, "profiles", "r06_valu_model.json")) as f:
        committed = json.load(f)
    for name, m in fresh["solver"].items():
        assert [(s["F"], s["role"], s["valu_cycles_per_step"]) for s in m["stages"]] == [(s["F"], s["role"], s["valu_cycles_per_step"]) for s in committed["solver"][name]["stages"]], name
        assert {k: v["valu_cycles_per_interval"] for k, v in m["io_waves"].items()} == {k: v["valu_cycles_per_interval"] for k, v in committed["solver"][name]["io_waves"].items()}, name
        assert m["simd_placement"] == committed["solver"][name]["simd_placement"]
    for name, m in fresh["assemble"].items():
        assert m["cycles_per_valu_instruction_term_loop"] == committed["assemble"][name]["cycles_per_valu_instruction_term_loop"], name
    busiest, mean, per_simd, _ = bench.solver_valu_floor("k_sor_chain<2,6,3,1,4,2,2,1,1> (6 stages of 2 + 1 of 3 sweeps, 2 groups per band)", 128)
    assert 0.33 < busiest / 1.58e-3 < 1.0 and mean <= busiest and max(per_simd) == per_simd[1]          # stage 0 (loads + ring fill) and its partner: the busiest SIMD


@pytest.mark.gpu
def test_two_ranks_for_real_on_one_gpu():
    """the whole N = 2 path of the real bench -- child launch, two ranks with two contexts each, the timed region, max over ranks, the gather of the
    per-window timings, rank 0's line -- with both ranks on the one GPU of this box and the exchange over gloo (SFA_BENCH_BACKEND=gloo: a rehearsal
    switch; the line says so and is never a result).  Two processes on the card: within the box's limit."""
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "8", "--no-cpu-baseline"], {"SFA_BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["steps"] == 1
    assert out["config"]["frame_windows_per_gpu"] == 8
    assert out["seconds_per_window"]["n"] == 16                         # both ranks' windows arrived
    assert out["value"] > 0 and "gloo" in json.dumps(out)
    # the strong-scaling sections (BASELINE configs 4 and 5: a FIXED set of windows partitioned over the ranks): every window was refined by exactly one rank,
    # rank 0's line carries the rank count, every rank's seconds and their maximum
    for key, total in (("config4_strong", 128), ("config5_strong", 32)):
        st = out[key]
        assert "error" not in st, st
        assert st["n_gpus"] == 2 and st["windows_total"] == total and st["windows_all_ranks"] == total and st["windows_this_rank"] == total // 2
        assert len(st["seconds_per_rank"]) == 2 and all(v > 0 for v in st["seconds_per_rank"])
        assert abs(st["seconds"] - max(st["seconds_per_rank"])) < 1e-3 and st["scaling"] == "strong"
    # no key named frac above 1 anywhere in the line (a fraction is a fraction)
    assert all(0 <= v <= 1 for _, v in _fracs(out)), list(_fracs(out))


@pytest.mark.gpu
def test_one_rank_over_rccl():
    """the RCCL leg as far as one card allows.  bench.py started the way the round driver starts N ranks (`python -m torch.distributed.run --nnodes=1
    --nproc-per-node 1 ... bench.py --gpus 1`) initialises backend "nccl" (= RCCL) with its device id and walks the N-rank sequence -- barrier, timed region, barrier,
    MAX over ranks, the gather of the per-window timings and the strong sections' SUM / MAX all-reduces -- on DEVICE tensors (xdev = "cuda"): the branch no earlier
    round had executed.  One process on the card."""
    import bench
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SFA_BENCH_BACKEND")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["TORCH_DISTRIBUTED_DEBUG"] = "DETAIL"                           # every collective is logged and checked
    cmd = bench.launch_command(1, ["--gpus", "1", "--steps", "1", "--warmup", "1", "--batch", "8", "--no-cpu-baseline"], bench.free_port())
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["seconds_per_window"]["n"] == 8
    assert out["config"]["parallelism"] == "frame-window data parallel x1"          # backend nccl: no rehearsal label
    assert out["distributed"] == {"backend": "nccl", "world_size": 1, "exchange_tensors_on": "cuda", "collectives": "barrier, all_reduce(MAX), all_reduce(SUM)"}
    assert all(0 <= v <= 1 for _, v in _fracs(out)), list(_fracs(out))
    assert out["roofline"]["valu_time_floor_frac"] is not None or out["config"]["windows_per_launch"] < 10
    for key, total in (("config4_strong", 128), ("config5_strong", 32)):
        st = out[key]
        assert "error" not in st, st
        assert st["windows_all_ranks"] == total and len(st["seconds_per_rank"]) == 1 and st["seconds"] > 0


@pytest.mark.gpu
def test_timing_exchange_on_device_tensors_over_rccl(tmp_path):
    """shard.gather_timings / max_over_ranks / sum_over_ranks on cuda tensors through a one-rank RCCL communicator: the values survive the collectives"""
    script = tmp_path / "w.py"
    script.write_text('''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from slowflow_amd import shard
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
full = shard.gather_timings(dist, {i: 0.25 * (i + 1) for i in range(7)}, 7, device="cuda")
mx = shard.max_over_ranks(dist, 3.5, device="cuda")
sm = shard.sum_over_ranks(dist, 2.0, device="cuda")
dist.barrier()
print("RESULT", ",".join("%%.2f" %% v for v in full), mx, sm, dist.get_backend())
dist.destroy_process_group()
''' % ROOT)
    import bench
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(bench.free_port()), str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert [float(v) for v in line[1].split(",")] == [0.25 * (i + 1) for i in range(7)]
    assert float(line[2]) == 3.5 and float(line[3]) == 2.0 and line[4] == "nccl"
