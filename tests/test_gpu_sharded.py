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


def _reference(case="default"):
    from globalsfmpy_amd.solver import RotationProblem
    g = synth.make_graph(1203, 40000, seed=23, outlier_frac=0.3)
    if case == "isolated":
        keep = (g["edge_i"] < 1100) & (g["edge_j"] < 1100)
        for k in ("edge_i", "edge_j", "rel_aa", "cov6", "inlier_weight"):
            g[k] = g[k][keep]
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
    p.set_loss(MAGSACWeightBasedLoss(0.02))
    rot, s = p.solve(g["init_aa"])
    return rot, s, p.trace()


def _launch(nproc, backend, out, extra_env=None, mode="torch", case="default"):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    port = 29600 + (os.getpid() % 300) + (0 if backend == "gloo" else 1)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "sharded_worker.py"), backend, out, mode, case]
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
    """1203 cameras cut into slices of equal directed-entry counts (unequal camera counts: every slice but the widest is padded)."""
    res = _launch(world, "gloo", str(tmp_path / ("gloo%d.npz" % world)))
    _compare(res, _reference())
    assert int(res["n_ag"]) > int(res["cg"])      # one all-gather per PCG iteration + per linearisation
    assert int(res["n_ar"]) >= 2                  # cost all-reduces


def test_a_rank_that_holds_only_isolated_cameras(tmp_path):
    """8 ranks, the last 103 cameras without any edge and a partition that gives exactly those to the last rank: it holds no edge at all, still
    has to take part in every collective (problem creation with zero local edges), and the isolated cameras keep their input rotation."""
    res = _launch(8, "gloo", str(tmp_path / "iso.npz"), case="isolated")
    ref = _reference("isolated")
    _compare(res, ref)
    g0 = synth.make_graph(1203, 40000, seed=23, outlier_frac=0.3)
    assert np.array_equal(res["rot"][1100:], g0["init_aa"][1100:])


@pytest.mark.parametrize("world", [2, 3])
def test_sigma_consensus_on_a_sharded_problem(tmp_path, world):
    """EstimateRotationsWithSigmaConsensus (estimator.cpp:314-457) across ranks: every rank weights ALL the edges it holds from the same
    rotations, the mean weight change is summed over each edge's cost owner.  Same outer iterations, same weights, same rotations."""
    from globalsfmpy_amd.loss_functions import TrivialLoss
    from globalsfmpy_amd.solver import RotationProblem
    g = synth.make_graph(1203, 40000, seed=23, outlier_frac=0.3)
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS)
    p.set_loss(TrivialLoss())
    rot, s = p.solve_sigma_consensus(g["init_aa"], 4, 0.05, dense_cholesky_max_cams=0)
    res = _launch(world, "gloo", str(tmp_path / ("sigma%d.npz" % world)), case="sigma")
    assert int(res["outer"]) == s["outer_iterations"] and int(res["iters"]) == s["num_iterations"]
    assert abs(float(res["wchange"]) - s["last_weight_change"]) <= 1e-9 * max(1e-12, s["last_weight_change"])
    assert abs(float(res["cost"]) - s["final_cost"]) <= 1e-9 * s["final_cost"]
    assert synth.angular_distance(synth.align_rotations(res["rot"], rot), rot).mean() <= 1e-6


def test_host_callback_loss_on_a_sharded_problem(tmp_path):
    """A Python loss without a native descriptor (the reference's trampoline case, bind_src/GlobalSfMpy.cpp:33-65) on two ranks: each rank
    evaluates it for every edge it holds, from bitwise identical s."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from sharded_worker import python_only_loss
    from globalsfmpy_amd.solver import RotationProblem
    g = synth.make_graph(1203, 40000, seed=23, outlier_frac=0.3)
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
    p.set_loss_callback(python_only_loss)
    rot, s = p.solve(g["init_aa"], max_num_iterations=6, dense_cholesky_max_cams=0)
    res = _launch(2, "gloo", str(tmp_path / "callback.npz"), case="callback")
    assert int(res["iters"]) == s["num_iterations"] and int(res["term"]) == s["termination"]
    assert abs(float(res["cost"]) - s["final_cost"]) <= 1e-9 * s["final_cost"]
    assert synth.angular_distance(synth.align_rotations(res["rot"], rot), rot).mean() <= 1e-6


def test_disconnected_graph_keeps_its_tighter_pcg_tolerance_when_sharded(tmp_path):
    """A rank sees only its own edges; gsfm_rot_problem_create merges every rank's local components (one all-gather of a label per camera)
    and finds the global graph disconnected without being told: same PCG iteration count as the single-GPU solve, which counts the
    components itself.  (Round 2 needed the partitioner to set GSFM_SHARD_DISCONNECTED; sharding.py no longer does.)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from sharded_worker import two_component_graph
    from globalsfmpy_amd.solver import RotationProblem
    g = two_component_graph()
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
    p.set_loss(MAGSACWeightBasedLoss(0.02))
    # (dense_cholesky_max_cams=0: round 5 factorises the small component of an UNSHARDED disconnected problem exactly, solver_components.hpp;
    # here the one-PCG-over-everything path is what the sharded solve is held against)
    rot, s = p.solve(g["init_aa"], pcg_single_reduction=0, dense_cholesky_max_cams=0)
    _, s12 = p.solve(g["init_aa"], pcg_single_reduction=0, cg_relative_tolerance=1e-12, dense_cholesky_max_cams=0)   # (still solved at 1e-14: the floor applies)
    assert s12["num_cg_iterations"] == s["num_cg_iterations"]
    res = _launch(2, "gloo", str(tmp_path / "disc.npz"), case="disconnected")
    assert int(res["iters"]) == s["num_iterations"] and int(res["cg"]) == s["num_cg_iterations"]
    assert synth.angular_distance(synth.align_rotations(res["rot"], rot), rot).mean() <= 1e-6


@pytest.mark.parametrize("world,case", [(2, "default"), (3, "default"), (2, "callback"), (2, "sigma")])
def test_column_sorted_layout_on_a_sharded_problem(tmp_path, world, case):
    """K2c / K3c on every rank (GSFM_K3_COLSORT=1 forces the layout on this small graph; at C5 size each rank of an 8-GPU run chooses it by
    itself: 512 rows x degree 200 entries per block for 100k cameras, whatever the number of rows a rank owns): same solve as one GPU on
    the row-major kernels, for a native loss, for a host-callback loss (k_col_s: s of every held edge) and for sigma consensus."""
    from globalsfmpy_amd.solver import RotationProblem
    res = _launch(world, "gloo", str(tmp_path / ("cs_%s%d.npz" % (case, world))), {"GSFM_K3_COLSORT": "1"}, case=case)
    if case == "default":
        _compare(res, _reference())
        return
    g = synth.make_graph(1203, 40000, seed=23, outlier_frac=0.3)
    if case == "callback":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from sharded_worker import python_only_loss
        p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
        p.set_loss_callback(python_only_loss)
        rot, s = p.solve(g["init_aa"], max_num_iterations=6, dense_cholesky_max_cams=0)
    else:
        from globalsfmpy_amd.loss_functions import TrivialLoss
        p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS)
        p.set_loss(TrivialLoss())
        rot, s = p.solve_sigma_consensus(g["init_aa"], 4, 0.05, dense_cholesky_max_cams=0)
        assert int(res["outer"]) == s["outer_iterations"] and abs(float(res["wchange"]) - s["last_weight_change"]) <= 1e-9 * max(1.0, s["last_weight_change"])
    assert int(res["iters"]) == s["num_iterations"]
    assert abs(float(res["cost"]) - s["final_cost"]) <= 1e-8 * s["final_cost"]
    assert synth.angular_distance(synth.align_rotations(res["rot"], rot), rot).mean() <= 1e-6


def test_forced_single_rank_shard_over_rccl(tmp_path):
    res = _launch(1, "nccl", str(tmp_path / "nccl1.npz"), {"GSFM_FORCE_SHARD": "1"})
    _compare(res, _reference())
    assert int(res["n_ag"]) > 0


def test_forced_single_rank_shard_over_native_rccl(tmp_path):
    """The C++ RCCL communicator (libgsfm_rccl.so): ncclCommInitRank + all-gather / all-reduce on the solver's stream."""
    res = _launch(1, "nccl", str(tmp_path / "native1.npz"), {"GSFM_FORCE_SHARD": "1"}, mode="native")
    assert str(res["backend"]) == "rccl-native"
    _compare(res, _reference())


def test_sharded_pcg_chunk_replays_as_a_hipgraph_with_the_native_communicator(tmp_path):
    """pcg_hip_graph = 2: the chunk of PCG iterations INCLUDING its all-gathers is captured once and replayed (RCCL collectives are
    stream-capturable; the native communicator enqueues them on the solver's stream and does nothing else).  Same trajectory."""
    res = _launch(1, "nccl", str(tmp_path / "native_graph.npz"), {"GSFM_FORCE_SHARD": "1", "GSFM_TEST_PCG_GRAPH": "2"}, mode="native")
    assert str(res["backend"]) == "rccl-native"
    _compare(res, _reference())
    assert int(res["graph_launches"]) > 0


def test_host_staged_collectives_fall_back_to_plain_launches_when_capture_is_requested(tmp_path):
    res = _launch(2, "gloo", str(tmp_path / "gloo_graph.npz"), {"GSFM_TEST_PCG_GRAPH": "2"})
    _compare(res, _reference())
    assert int(res["graph_launches"]) == 0


@pytest.mark.parametrize("world,seed", [(3, 102), (5, 103), (8, 104), (6, 106), (7, 222)])
def test_random_graph_random_partition_equals_the_single_gpu_solve(tmp_path, world, seed):
    """Random size, error type (the non-Laplacian 9-residual functor included), loss, PCG variant and a random partition -- arbitrary cut points,
    so slices of very different widths and ranks that own no camera at all: same LM and PCG iteration counts, same cost, same rotations."""
    res = _launch(world, "gloo", str(tmp_path / ("rand%d.npz" % seed)), case="random:%d" % seed)
    info = "n=%d e=%d et=%d %s slice sizes %s" % (int(res["n"]), int(res["e"]), int(res["et"]), str(res["loss"]), res["widths"].tolist())
    # (per-camera sums are bitwise those of one GPU; the PCG dot products add the same numbers in the partition's camera order, so an
    # ill-conditioned step may need a few PCG iterations more or less: tests/manual/fuzz_sharded.sh, 7 of 25 random cases.  Round 4: the loose
    # solves of the forcing schedule stop on an ESTIMATE of the energy error extrapolated from the last four iterations, which moves by an
    # iteration or two with the summation order -- per solve, and a solve is now 5-15 iterations instead of 30-150: 5 % + 4 instead of 2 % + 2)
    assert int(res["iters"]) == int(res["ref_iters"]) and abs(int(res["cg"]) - int(res["ref_cg"])) <= 0.05 * int(res["ref_cg"]) + 4, info
    assert abs(float(res["cost"]) - float(res["ref_cost"])) <= 1e-8 * float(res["ref_cost"]), info
    assert synth.angular_distance(res["rot"], res["ref_rot"]).max() < (1e-4 if "MAGSAC" in str(res["loss"]) else 1e-6), info


@pytest.mark.parametrize("world", [2, 3])
def test_two_level_preconditioner_on_a_sharded_coherent_graph(tmp_path, world):
    """Every rank judges its own share coherent, the vote at creation is unanimous, the coarse matrix is summed over the ranks once per LM step
    and everything else of the preconditioner runs replicated: a fraction of the block-Jacobi iterations, the single-GPU answer."""
    res = _launch(world, "gloo", str(tmp_path / ("coarse%d.npz" % world)), case="coarse")
    assert int(res["cg"]) * 3 <= int(res["plain_cg"]), (int(res["cg"]), int(res["plain_cg"]))
    assert int(res["iters"]) == int(res["ref_iters"]) and abs(int(res["cg"]) - int(res["ref_cg"])) <= 0.05 * int(res["ref_cg"]) + 2
    assert abs(float(res["cost"]) - float(res["ref_cost"])) <= 1e-9 * float(res["ref_cost"])
    assert synth.angular_distance(res["rot"], res["ref_rot"]).max() < 1e-8


@pytest.mark.parametrize("world", [2, 3, 8])
def test_peer_store_exchange_equals_the_host_staged_collectives_bit_for_bit(tmp_path, world):
    """csrc/gsfm_peer.hip: every rank writes its slice into the other ranks' IPC-mapped mailboxes and a flag says it has arrived -- the
    all-gather of every PCG iteration and of every linearisation, and the scalar all-reduces, without a collective-library call.  N processes
    sharing the one GPU of the test box map each other's mailboxes exactly as N GPUs of a node would (hipIpcMemHandle); the solve must
    reproduce the gloo path's rotations bit for bit (an all-gather moves bits; the scalar all-reduce adds the ranks' shares in rank order on every rank)."""
    a = _launch(world, "gloo", str(tmp_path / ("peer%d.npz" % world)), mode="peer")
    b = _launch(world, "gloo", str(tmp_path / ("gloo%d.npz" % world)))
    assert str(a["backend"]) == "peer-store+gloo" and not bool(a["peer_error"])
    served, fell_back = (int(v) for v in a["peer_calls"])
    # (host-side count: a captured chunk's exchanges are counted once, at capture; nothing of the solve may need the fallback)
    assert served > 0 and int(a["pcg_collectives"]) > 0 and fell_back <= 2, (served, fell_back)
    assert int(a["cg"]) == int(b["cg"]) and int(a["iters"]) == int(b["iters"])
    if world == 2:   # an all-gather moves bits and a two-term sum has one order: the whole solve is bit-identical
        assert np.array_equal(a["rot"], b["rot"]) and float(a["cost"]) == float(b["cost"]) and np.array_equal(a["trace"], b["trace"])
    else:            # the cost all-reduce: gloo adds the ranks' shares in its ring's order, the mailbox in rank order -- last-bit differences in the
        #              costs, which the trust-region radius law feeds back into the steps
        assert synth.angular_distance(a["rot"], b["rot"]).max() <= 1e-12 and abs(float(a["cost"]) - float(b["cost"])) <= 1e-13 * float(b["cost"])
        assert np.allclose(a["trace"], b["trace"], rtol=1e-9, atol=0.0)
    _compare(a, _reference())


def test_peer_store_exchange_inside_the_captured_pcg_chunks(tmp_path):
    """The two kernels of a peer-store collective take no per-call arguments (the sequence number lives on the device), so the chunks of PCG
    iterations replay as hipGraphs WITH their exchanges: same bits as plain launches."""
    a = _launch(2, "gloo", str(tmp_path / "peer_graph.npz"), mode="peer", extra_env={"GSFM_TEST_PCG_GRAPH": "1"})
    b = _launch(2, "gloo", str(tmp_path / "peer_plain.npz"), mode="peer", extra_env={"GSFM_TEST_PCG_GRAPH": "0"})
    assert int(a["graph_launches"]) > 0 and int(b["graph_launches"]) == 0 and not bool(a["peer_error"])
    assert np.array_equal(a["rot"], b["rot"]) and int(a["cg"]) == int(b["cg"])


def test_sharded_equals_unsharded_at_the_default_tolerance_on_a_large_ill_conditioned_graph(tmp_path):
    """Round-3 advisor's open item: 40 000 cameras / 1.05 M edges (2.1 M directed entries: the unsharded solve takes the textbook recurrence, the
    sharded one the single-reduction one), spatially coherent and MAGSAC-weighted, DEFAULT options on both sides, to convergence."""
    res = _launch(2, "gloo", str(tmp_path / "bigcoh.npz"), case="bigcoherent")
    assert int(res["directed"]) > 2000000
    print("sharded %d LM / %d PCG (restarts %d, capped %d); unsharded %d LM / %d PCG (restarts %d, capped %d)" % (
        res["iters"], res["cg"], res["restarts"], res["capped"], res["ref_iters"], res["ref_cg"], res["ref_restarts"], res["ref_capped"]))
    assert int(res["iters"]) == int(res["ref_iters"]) and int(res["capped"]) == 0 and int(res["ref_capped"]) == 0
    assert abs(float(res["cost"]) - float(res["ref_cost"])) <= 1e-9 * float(res["ref_cost"])
    d = synth.angular_distance(synth.align_rotations(res["rot"], res["ref_rot"]), res["ref_rot"])
    print("rotations: mean %.2e max %.2e rad" % (d.mean(), d.max()))
    assert d.mean() <= 1e-6


def test_a_forcing_restart_under_magsac_is_taken_by_every_rank_together(tmp_path):
    """Round-5 advisor (medium): the staircase band of the restart decision was computed from a rank's OWN edge count, so ranks with unequal
    shares could disagree on it -- one restarting, the others entering the next all-gather.  Three ranks with random cut points on a MAGSAC
    problem whose default solve is redone (fuzz seed 9 trial 93): every rank reports the same restart and iteration counts, the run ends, and it
    equals the unsharded default solve."""
    res = _launch(3, "gloo", str(tmp_path / "restart.npz"), case="magsacrestart")
    print("%s, %d edges; per rank: cost edges %s, restarts %s, LM iterations %s; unsharded: %d restarts, %d iterations" % (
        res["loss"], res["n_edges"], res["rank_edges"].tolist(), res["restarts"].tolist(), res["rank_iters"].tolist(), res["ref_restarts"], res["ref_iters"]))
    assert str(res["loss"]) == "MAGSACWeightBasedLoss"
    assert len(set(res["rank_edges"].tolist())) == 3 and int(res["rank_edges"].sum()) == int(res["n_edges"])   # unequal shares, every edge counted once
    assert len(set(res["restarts"].tolist())) == 1 and len(set(res["rank_iters"].tolist())) == 1
    assert int(res["restarts"][0]) == int(res["ref_restarts"]) == 1
    assert int(res["iters"]) == int(res["ref_iters"])
    assert abs(float(res["cost"]) - float(res["ref_cost"])) <= 1e-9 * float(res["ref_cost"])
    assert synth.angular_distance(synth.align_rotations(res["rot"], res["ref_rot"]), res["ref_rot"]).mean() <= 1e-6


def test_peer_store_time_out_fails_the_solve_and_the_next_one_runs_on_the_fallback(tmp_path):
    res = _launch(2, "gloo", str(tmp_path / "peererr.npz"), mode="peer", case="peererror")
    failed, flagged, same, few_peer_calls, fallback_used = [int(v) for v in res["flags"]]
    assert failed == 1, "the solve in which a wait timed out must fail on every rank"
    assert flagged == 1 and fallback_used == 1 and few_peer_calls == 1
    assert same == 1 and int(res["iters"]) == int(res["ref_iters"])


@pytest.mark.parametrize("world", [2, 4])
def test_packed_components_solve_without_a_collective_in_the_pcg_loop(tmp_path, world):
    """SURVEY 8(e) / round-4 review item 3: six scenes, whole components per rank -- num_pcg_collectives == 0 on every rank, the ranks' own PCG
    solves end after different iteration counts, the answer equals the single-GPU solve per component."""
    res = _launch(world, "gloo", str(tmp_path / ("packed%d.npz" % world)), case="packed")
    print("packed, %d ranks: %d LM iterations (one GPU: %d), PCG iterations per rank %d..%d, %d collectives per solve, none in the PCG loop" % (
        world, res["iters"], res["ref_iters"], res["cg_min"], res["cg_max"], res["collectives"]))
    print("   component steps (most on any rank): %d (one GPU: %d)" % (res["dense"], res["ref_dense"]))
    assert int(res["pcg_collectives_max"]) == 0 and int(res["capped"]) == 0
    assert int(res["dense"]) == int(res["iters"])   # (the ranks that hold scenes of at most 512 cameras factorise them in every LM iteration, like the single GPU)
    assert int(res["iters"]) == int(res["ref_iters"])
    assert abs(float(res["cost"]) - float(res["ref_cost"])) <= 1e-9 * float(res["ref_cost"])
    offs = res["offs"]
    for c in range(len(offs) - 1):
        sl = slice(int(offs[c]), int(offs[c + 1]))
        assert synth.angular_distance(synth.align_rotations(res["rot"][sl], res["ref_rot"][sl]), res["ref_rot"][sl]).mean() <= 1e-6, c
