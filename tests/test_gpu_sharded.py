"""-m gpu: the camera-slice sharded HIP path against the unsharded one, on a single-GPU box.
  * 2 ranks sharing the GPU, collectives over gloo (host-staged): the real multi-rank data path;
  * 1 rank with GSFM_FORCE_SHARD=1 over nccl (= RCCL): the device-pointer callbacks used at 8 GPUs.
Per-camera sums are complete on their owner, so the sharded trajectory must match the unsharded one
to rounding in the cost all-reduce only."""
import os
import subprocess
import sys

import numpy as np
import pytest

from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reference():
    from globalsfmpy_amd.solver import RotationProblem
    g = synth.make_graph(1203, 40000, seed=23, outlier_frac=0.3)
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
    p.set_loss(MAGSACWeightBasedLoss(0.02))
    rot, s = p.solve(g["init_aa"])
    return rot, s, p.trace()


def _launch(nproc, backend, out, extra_env=None, mode="torch"):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    port = 29600 + (os.getpid() % 300) + (0 if backend == "gloo" else 1)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "sharded_worker.py"), backend, out, mode]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return np.load(out)


def _compare(res, ref):
    rot, s, trace = ref
    assert int(res["iters"]) == s["num_iterations"]
    assert int(res["term"]) == s["termination"]
    assert abs(float(res["cost"]) - s["final_cost"]) <= 1e-6 * s["final_cost"]
    assert synth.angular_distance(synth.align_rotations(res["rot"], rot), rot).mean() <= 1e-6


@pytest.mark.parametrize("world", [2, 3, 8])
def test_ranks_share_one_gpu_over_gloo(tmp_path, world):
    """1203 cameras: slices of 602 / 401 / 151 cameras, the last rank's slice short (padding rows) in every case."""
    res = _launch(world, "gloo", str(tmp_path / ("gloo%d.npz" % world)))
    _compare(res, _reference())
    assert int(res["n_ag"]) > int(res["cg"])      # one all-gather per PCG iteration + per linearisation
    assert int(res["n_ar"]) >= 2                  # cost all-reduces


def test_forced_single_rank_shard_over_rccl(tmp_path):
    res = _launch(1, "nccl", str(tmp_path / "nccl1.npz"), {"GSFM_FORCE_SHARD": "1"})
    _compare(res, _reference())
    assert int(res["n_ag"]) > 0


def test_forced_single_rank_shard_over_native_rccl(tmp_path):
    """The C++ RCCL communicator (libgsfm_rccl.so): ncclCommInitRank + all-gather / all-reduce on the solver's stream."""
    res = _launch(1, "nccl", str(tmp_path / "native1.npz"), {"GSFM_FORCE_SHARD": "1"}, mode="native")
    assert str(res["backend"]) == "rccl-native"
    _compare(res, _reference())
