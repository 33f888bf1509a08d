"""-m gpu: bench.py end to end at reduced size -- the single-GPU contract line, and the N > 1 launch path (two ranks sharing the one
GPU of the test box over gloo) that the driver only ever runs on an 8-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--cams", "20000", "--edges", "400000", "--steps", "2", "--warmup", "1", "--cpu-baseline", "0", "--small-graphs", "0", "--sweep-reps", "3"]


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_contract_line():
    # (with the cpu_baseline leg: the oracle solves the same graph, so the line carries a device-vs-CPU comparison that is asserted below)
    small = [a for a in SMALL]
    small[small.index("--cpu-baseline") + 1] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + small + ["--cpu-single-cams", "0"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = _json_line(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0 and out["dtype"] == "f64" and "workload" in out["config"]
    ro = out["roofline"]
    assert ro["bound"] == "hbm" and "k_matvec" in ro["kernel"] and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12
    assert abs(ro["achieved"] - ro["algorithmic_bytes_per_launch"] / (ro["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * ro["achieved"]
    other = out["roofline_other"]
    assert {"k_lin", "k_cost_trial", "k_cost_full", "k_cost_s_only", "sigma_consensus_K6"} <= set(other)
    assert other["k_cost_full"]["kernel_ms"] >= other["k_cost_trial"]["kernel_ms"] * 0.9   # the stores cost something
    # the CPU baseline solved the benchmark graph itself: it doubles as the parity check of the measured configuration (default PCG schedule)
    cmp_ = out["cpu_baseline"]["device_vs_cpu"]
    assert cmp_["mean_angular_difference_rad"] <= 1e-6 and cmp_["lm_iterations_device"] == cmp_["lm_iterations_cpu"], cmp_
    # both schedules of the PCG tolerance are reported, `value` is the default one, and they agree
    assert out["pcg_schedule"]["pcg_forcing"] == 1 and out["exact_schedule"]["pcg_forcing"] == 0
    assert out["exact_schedule"]["default_schedule_vs_this"]["mean_angular_difference_rad"] <= 1e-6
    assert out["exact_schedule"]["cg_iterations"] >= out["cg_iterations_per_solve"]
    # BASELINE's second metric, on both schedules: the same count on the benchmark graph; and the line names the start `value` used
    its = out["iters_to_1e-6_by_schedule"]
    assert its["default"] == its["exact"] == out["iters_to_1e-6"] and its["lm_iterations_default"] == its["lm_iterations_exact"], its
    assert "init" in out["config"] and "spanning" in out["config"]["init"]


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`: camera partition, sharded problem creation (with its
    agreement all-reduce), all-gathers inside PCG, max-over-ranks timing, rank 0's JSON line."""
    env = dict(os.environ, GSFM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", GSFM_BENCH_PEER="1")   # (+ the peer-store variant, which nccl launches try by default)
    port = 29900 + os.getpid() % 90
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["parallelism"] == "camera-slice x2" and out["config"]["collectives"] in ("gloo", "peer-store+gloo")
    assert out["collectives_per_pcg_iteration"] == 1.0 and out["collectives_per_solve"] > out["cg_iterations_per_solve"]   # exactly one collective per PCG iteration
    ps = out["peer_store_exchange"]
    assert ps["status"] == "ok" and ps["reproduces_the_collective_run"] and ps["backend"] == "peer-store+gloo" and ps["pcg_chunks_replayed_as_hipgraphs"] > 0, ps
    # the weak-scaling point rides on the same line (round-5 review, 1c): the graph grown with the rank count, same options, converged
    ws = out["weak_scaling"]
    assert ws["status"] == "ok" and ws["cams"] == 2 * out["config"]["cams"] and ws["edges"] == 2 * out["config"]["edges"] and ws["value"] > 0, ws
    assert ws["termination"] == out["termination"] and ws["mean_angular_error_vs_ground_truth_deg"] < 1.0, ws
    one = _json_line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + ["--sigma-pass", "0"], cwd=ROOT, capture_output=True, text=True, timeout=900).stdout)
    assert out["lm_iterations"] == one["lm_iterations"] and out["residual_sweeps_per_solve"] == one["residual_sweeps_per_solve"]
    assert abs(out["final_cost"] - one["final_cost"]) <= 1e-6 * one["final_cost"]


def test_bench_single_rank_over_native_rccl_prints_exactly_one_stdout_line():
    """GSFM_FORCE_SHARD=1 under a one-process torchrun: the N > 1 code path of bench.py on real RCCL (process group over nccl, partition,
    the C++ communicator, collectives inside the solve).  RCCL prints a version banner on the C-level stdout when its first communicator is
    created; the bench line must still be the only thing on stdout."""
    env = dict(os.environ, GSFM_FORCE_SHARD="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29990 + os.getpid() % 9
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL + ["--sigma-pass", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:5]
    out = json.loads(lines[0])
    assert out["config"]["collectives"] == "rccl-native" and out["value"] > 0 and out["n_gpus"] == 1
    # the native communicator's callbacks only enqueue on the solver's stream: the PCG chunks are captured WITH their all-gathers and
    # replayed as hipGraphs by default, one collective per PCG iteration
    assert out["collectives_per_pcg_iteration"] == 1.0 and out["pcg_chunks_replayed_as_hipgraphs"] > 0
    # ... measured AFTER a plain-launch pass (no hipGraph in a sharded solve), which is what the line falls back to should the captured variant hang
    assert out["hipgraph_with_collectives"]["status"] == "ok" and out["hipgraph_with_collectives"]["same_final_cost"]
    assert out["plain_launches"]["pcg_chunks_replayed_as_hipgraphs"] == 0 and out["plain_launches"]["value"] > 0


def test_bench_falls_back_to_the_plain_launch_line_when_the_captured_variant_does_not_return():
    """The watchdog of the hipGraph-with-collectives attempt, forced to fire (limit 1 ms): rank 0 still prints ONE line -- the plain-launch
    measurement -- and the process ends with status 0 instead of 124 and no line (round-3 review)."""
    env = dict(os.environ, GSFM_FORCE_SHARD="1", HSA_ENABLE_IPC_MODE_LEGACY="0", GSFM_BENCH_CAPTURE_WATCHDOG_S="0.001")
    port = 29980 + os.getpid() % 9
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL + ["--sigma-pass", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert "watchdog" in out["hipgraph_with_collectives"]["status"] and out["value"] > 0 and out["pcg_chunks_replayed_as_hipgraphs"] == 0


def test_bench_prints_the_plain_launch_line_even_if_a_later_variant_kills_the_process():
    """A GPU fault in one of the never-measured multi-rank variants ends the process through abort(); the armed C-level handler writes the
    line measured with plain launches, marked with a `crashed_variant` field carrying the signal number (test hook: the process aborts itself right after arming)."""
    env = dict(os.environ, GSFM_FORCE_SHARD="1", HSA_ENABLE_IPC_MODE_LEGACY="0", GSFM_BENCH_TEST_CRASH="1")
    port = 29970 + os.getpid() % 9
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL + ["--sigma-pass", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = _json_line(r.stdout)
    # the line itself says that a later variant died and by which signal (SIGABRT = 6): nobody can read it as a clean run (round-4 advisor)
    assert out["crashed_variant"]["signal"] == 6 and out["value"] > 0 and out["pcg_chunks_replayed_as_hipgraphs"] == 0
