"""
bench.py's launch contract: `python bench.py --gpus N` from a plain shell starts the N rank processes itself (the driver's
command, BENCH_rNN.json "cmd"), the launching process never touches the GPU (it does not import torch), rank 0's JSON line is
relayed and a failing rank makes the command fail.  BASELINE config 4 (8 ranks, RCCL) cannot run on a one-GPU box: the -m gpu
test rehearses the same code path with every rank on cuda:0 over gloo (US_BENCH_REHEARSE=1), which shows that the ranks start,
rendezvous, step in lock-step and report -- its numbers mean nothing.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_launcher_does_not_import_torch():
    code = "import sys; sys.argv=['bench.py']; import bench; assert bench.torch is None; assert 'torch' not in sys.modules; print('ok')"
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-2000:]


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the failure path of a box without a GPU")
def test_plain_multi_gpu_command_spawns_ranks_and_propagates_failure():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0                                        # no GPU here: every rank refuses, the launcher reports it
    assert out.stderr.count("needs the MI355X") == 2, out.stderr[-2000:]   # ... and two ranks were started and rendezvoused
    assert out.stdout.strip() == ""


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the failure path of a box without a GPU")
def test_a_side_process_that_dies_costs_its_own_entry_only():
    """bench.side_process: the dp_rank_local side run (an RCCL group + collectives captured into graphs) has a process of its own since
    torch's process-group watchdog once terminated bench.py in it; a child that ends badly is tried once more and then leaves an error
    text in its entry -- here both attempts end with 'needs the MI355X'"""
    import bench
    args = bench.parse_args(["--steps", "2", "--warmup", "1"])
    out = bench.side_process("dp_rank_local", args, timeout_s=120)
    assert set(out) == {"error"} and "dp_rank_local" in out["error"] and "code 1" in out["error"], out


@pytest.mark.gpu
def test_the_side_process_returns_the_rank_local_table():
    import bench
    args = bench.parse_args(["--steps", "3", "--warmup", "1"])
    out = bench.side_process("dp_rank_local", args, timeout_s=280)
    assert "error" not in out and out["local_fast"]["poses_fixed_graph_segments"] >= 2 and out["local_fast"]["poses_fixed_eager_ms"] > 0, out


@pytest.mark.gpu
@pytest.mark.parametrize("n,extra", [(2, []), (2, ["--sharded-adam"]), (4, []), (2, ["--grad-comm", "bf16"])])
def test_plain_command_rehearsal_on_one_gpu(n, extra):
    """n ranks on cuda:0 over gloo through bench.py's own launcher.  (At most 6 processes may use the card of a test box at once:
    4 ranks + this process is the largest rehearsal that fits; the 8-rank protocol itself runs on the CPU in test_dist_gloo.py.)"""
    env = dict(os.environ, US_BENCH_REHEARSE="1", US_BENCH_WATCHDOG="300")         # (a rank still running after 5 minutes says where it waits)
    out = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "3", "--warmup", "1", "--rays", "512", "--probe-steps", "1",
                          "--no-tracking", "--no-cpu-baseline", "--prewarm-s", "0"] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["steps"] == 3 and rec["value"] > 0 and rec["scaling"] == "weak"
    assert rec["config"]["rays_per_gpu"] == 512 and "roofline" in rec and rec["roofline"]["frac"] > 0
    assert abs(rec["value"] - n * 512 / (rec["ms_per_step"] * 1e-3)) < 1e-6 * rec["value"]
    assert rec["final_loss"] == rec["final_loss"]                     # not NaN


def test_request_roofline_of_the_encoder_from_the_committed_curve():
    """bench.py's `roofline_by_kernel.hashgrid_fwd_joint.l2_request` (r6): the random-gather curve is read from profiles/r06_ta_bench.txt as
    tools/ta_bench.hip printed it; lane loads are counted per point, level and grid as gather_corners issues them (4, + 4 on a hashed level
    when the cell's x is odd); the ceiling is the sum of loads / rate(level slab).  Checked on the CPU with a hand-made descriptor."""
    import types
    import bench
    curve = bench.gather_rate_curve()
    assert len(curve) >= 10 and curve == sorted(curve)
    sizes, rates = [c[0] for c in curve], [c[1] for c in curve]
    assert rates[0] > 900 and 250 < bench._rate_at(curve, 4 * 2 ** 20) < 280 and rates[-1] < 60          # 16 KiB, 4 MiB (the L2), 128 MiB
    assert all(a >= b for a, b in zip(rates, rates[1:]))              # bigger tables never gather faster
    assert bench._rate_at(curve, 1) == rates[0] and bench._rate_at(curve, 1e12) == rates[-1]
    mid = bench._rate_at(curve, (sizes[3] * sizes[4]) ** 0.5)
    assert min(rates[3], rates[4]) <= mid <= max(rates[3], rates[4])
    bench.torch = torch
    try:
        # two levels of one grid: a dense 4^3 level (64 entries) and a hashed level of 32 entries at resolution 8
        d = types.SimpleNamespace(n_levels=2, offset=[0, 64, 96], resolution=[4, 8], scale=[3.0, 7.0])
        x = torch.tensor([[0.10, 0.5, 0.5], [0.30, 0.5, 0.5], [0.90, 0.2, 0.1]])     # hashed level: cells x = floor(7 x + 0.5) = 1, 2, 6 -> one odd
        r = bench.encoder_request_roofline((d,), x, kernel_ms=1e-3)
    finally:
        bench.torch = None
    assert r["lane_loads_per_launch"] == 4 * 3 + (4 * 3 + 4 * 1)
    assert abs(r["achieved"] - r["lane_loads_per_launch"] / 1e-6 / 1e9) < 1e-6 and r["frac"] == pytest.approx(r["ceiling_ms"] / 1e-3)
