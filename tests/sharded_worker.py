"""Helper launched by tests/test_gpu_sharded.py under torch.distributed.run: solves one seeded graph
with the camera-slice sharded HIP path and writes rank 0's result to an .npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from globalsfmpy_amd import _abi, sharding, synth  # noqa: E402


def python_only_loss(s, out):
    """Cauchy(0.7) written out by hand (no native descriptor)."""
    b = 0.49
    out[0] = b * np.log1p(s / b)
    out[1] = 1.0 / (1.0 + s / b)
    out[2] = -(1.0 / b) * out[1] * out[1]


def two_component_graph():
    """Two scenes batched as one disconnected problem (BASELINE C4 in small)."""
    a = synth.make_graph(700, 20000, seed=41, outlier_frac=0.2)
    b = synth.make_graph(500, 9000, seed=42, outlier_frac=0.1)
    g = {"n_cams": 1200}
    for k in ("rel_aa", "cov6", "inlier_weight", "init_aa", "gt_aa"):
        g[k] = np.concatenate([a[k], b[k]])
    g["edge_i"] = np.concatenate([a["edge_i"], b["edge_i"] + 700]).astype(np.uint32)
    g["edge_j"] = np.concatenate([a["edge_j"], b["edge_j"] + 700]).astype(np.uint32)
    return g


def coarse_case(out, prefer_native):
    """A spatially coherent 10k-camera graph with shuffled ids: the partitioner's locality order makes every rank's share coherent, all ranks vote
    for the two-level preconditioner, its coarse matrix is all-reduced once per LM step."""
    import torch.distributed as dist
    from globalsfmpy_amd import loss_functions as LF
    from globalsfmpy_amd.solver import RotationProblem
    g = synth.make_graph(10000, 150000, 3, outlier_frac=0.1, local_window=300)
    loss = LF.MAGSACWeightBasedLoss(0.02)
    prob, part = sharding.make_sharded_problem(g, _abi.ANGLE_AXIS_COVARIANCE, loss=loss, prefer_native=prefer_native)
    # (pcg_forcing=0 throughout: the ranks' aggregates follow the partition's order, i.e. this is preconditioner against preconditioner, on the exact step)
    rot, summ = prob.solve(part.scatter(g["init_aa"]), pcg_forcing=0)
    n_ar = prob._comm.n_all_reduce
    os.environ["GSFM_PCG_COARSE"] = "0"
    plain, _ = sharding.make_sharded_problem(g, _abi.ANGLE_AXIS_COVARIANCE, loss=loss, prefer_native=prefer_native, part=part)
    _, s_plain = plain.solve(part.scatter(g["init_aa"]), pcg_forcing=0)
    os.environ.pop("GSFM_PCG_COARSE")
    if dist.get_rank() == 0:
        ref = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"]); ref.set_loss(loss)
        r1, s1 = ref.solve(g["init_aa"], pcg_forcing=0)
        np.savez(out, rot=part.gather(rot), ref_rot=r1, cost=summ["final_cost"], ref_cost=s1["final_cost"], iters=summ["num_iterations"], ref_iters=s1["num_iterations"],
                 cg=summ["num_cg_iterations"], ref_cg=s1["num_cg_iterations"], plain_cg=s_plain["num_cg_iterations"], n_ar=n_ar)
    dist.barrier()
    dist.destroy_process_group()


def random_case(seed, out, prefer_native, big=False):
    """Random graph, error type, loss and a random partition (arbitrary cut points: slices of very different widths, ranks that own no camera at
    all); rank 0 also solves the same problem unsharded and stores both answers."""
    import torch.distributed as dist
    from globalsfmpy_amd import loss_functions as LF
    from globalsfmpy_amd.solver import RotationProblem
    rng = np.random.default_rng(seed)          # the same stream on every rank
    world, rank = dist.get_world_size(), dist.get_rank()
    n = int(rng.integers(max(8, world), 1500)); window = int(rng.choice([0, 0, 50]))
    if big:   # large enough and coherent enough for the two-level preconditioner to be voted in
        n = int(rng.integers(5000, 9000)); window = int(rng.choice([60, 100, 160]))
    e_max = min(n * (n - 1) // 2, 30 * n) if window == 0 else n - 1 + sum(max(0, n - d) for d in range(2, window // 2 + 1)) // 2
    e = int(rng.integers(n - 1, e_max + 1))
    g = synth.make_graph(n, e, int(rng.integers(1 << 30)), outlier_frac=float(rng.uniform(0, 0.3)), local_window=window)
    et = [_abi.ANGLE_AXIS, _abi.ANGLE_AXIS_COVARIANCE, _abi.QUATERNION_COSINE, _abi.ROTATION_MAT_FNORM, _abi.ANGLE_AXIS_COV_INLIERS][int(rng.integers(5))]
    loss = [LF.HuberLoss(0.1), LF.SoftLOneLoss(0.1), LF.MAGSACWeightBasedLoss(0.02), LF.GemanMcClureLoss(0.1, 1.0)][int(rng.integers(4))]
    order = rng.permutation(n) if rng.random() < 0.5 else np.arange(n)
    cuts = np.sort(rng.integers(0, n + 1, world - 1)) if rng.random() < 0.5 else (np.arange(1, world) * n) // world
    cuts = [0] + [int(c) for c in cuts] + [n]
    width = max(1, max(cuts[r + 1] - cuts[r] for r in range(world)))
    new_id = np.empty(n, dtype=np.int64)
    for r in range(world):
        members = order[cuts[r]:cuts[r + 1]]
        new_id[members] = r * width + np.arange(members.size)
    part = sharding.Partition(n, world, width, new_id, [0] * world)
    prob, part = sharding.make_sharded_problem(g, et, loss=loss, prefer_native=prefer_native, part=part)
    opts = dict(max_num_iterations=12, pcg_single_reduction=int(rng.integers(0, 2)))
    rot, summ = prob.solve(part.scatter(g["init_aa"]), **opts)
    res = prob.residuals(rot) if False else None
    if rank == 0:
        ref = RotationProblem(n, g["edge_i"], g["edge_j"], g["rel_aa"], et, cov6=g["cov6"], inlier_weight=g["inlier_weight"]); ref.set_loss(loss)
        r1, s1 = ref.solve(g["init_aa"], dense_cholesky_max_cams=0, **opts)
        np.savez(out, rot=part.gather(rot), ref_rot=r1, cost=summ["final_cost"], ref_cost=s1["final_cost"], iters=summ["num_iterations"], ref_iters=s1["num_iterations"],
                 cg=summ["num_cg_iterations"], ref_cg=s1["num_cg_iterations"], n=n, e=e, et=et, widths=np.diff(cuts), loss=type(loss).__name__)
    dist.barrier()
    dist.destroy_process_group()


def big_coherent_case(out, prefer_native):
    """Round-3 advisor: sharded against unsharded at the DEFAULT tolerance (1e-12) on an ill-conditioned problem of more than 2 M directed
    entries -- spatially coherent, MAGSAC-weighted: the sharded solve runs the single-reduction recurrence (run_pcg2), the unsharded one of this
    size the textbook one; both to convergence, default options."""
    import torch.distributed as dist
    from globalsfmpy_amd import loss_functions as LF
    from globalsfmpy_amd.solver import RotationProblem
    g = synth.make_graph(40000, 1050000, 29, outlier_frac=0.15, local_window=240)
    loss = LF.MAGSACWeightBasedLoss(0.02)
    prob, part = sharding.make_sharded_problem(g, _abi.ANGLE_AXIS_COVARIANCE, loss=loss, prefer_native=prefer_native)
    rot, summ = prob.solve(part.scatter(g["init_aa"]))
    if dist.get_rank() == 0:
        ref = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"]); ref.set_loss(loss)
        r1, s1 = ref.solve(g["init_aa"])
        np.savez(out, rot=part.gather(rot), ref_rot=r1, cost=summ["final_cost"], ref_cost=s1["final_cost"], iters=summ["num_iterations"], ref_iters=s1["num_iterations"],
                 cg=summ["num_cg_iterations"], ref_cg=s1["num_cg_iterations"], directed=2 * len(g["edge_i"]), capped=summ["num_pcg_capped_steps"], ref_capped=s1["num_pcg_capped_steps"],
                 restarts=summ["num_forcing_restarts"], ref_restarts=s1["num_forcing_restarts"])
    dist.barrier()
    dist.destroy_process_group()


def magsac_restart_case(out, prefer_native):
    """Round-5 advisor: a MAGSAC solve on which the forcing schedule is abandoned and the run redone (seed 9 trial 93 of tests/manual/fuzz_forcing.py:
    1 388 cameras / 12 999 edges, 14 LM iterations), on ranks whose shares of the edges differ widely (random cut points).  Every rank must take the
    restart in the same LM iteration -- the staircase band of that decision is taken from the GLOBAL edge count (create-time agreement), not from a
    rank's own -- otherwise one rank re-enters the solve while the others wait in a collective and the run hangs."""
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests", "manual"))
    import fuzz_forcing
    from globalsfmpy_amd.solver import RotationProblem
    (t, g, et, loss, init, coherent), = list(fuzz_forcing.cases(94, 9, [93]))
    world, rank = dist.get_world_size(), dist.get_rank()
    n = g["n_cams"]
    rng = np.random.default_rng(5)
    cuts = [0] + sorted(int(c) for c in rng.integers(n // 10, n, world - 1)) + [n]
    width = max(1, max(cuts[r + 1] - cuts[r] for r in range(world)))
    new_id = np.empty(n, dtype=np.int64)
    for r in range(world):
        new_id[cuts[r]:cuts[r + 1]] = r * width + np.arange(cuts[r + 1] - cuts[r])
    part = sharding.Partition(n, world, width, new_id, [0] * world)
    prob, part = sharding.make_sharded_problem(g, et, loss=loss, prefer_native=prefer_native, part=part)
    rot, summ = prob.solve(part.scatter(init))
    mine = (int(summ["num_forcing_restarts"]), int(summ["num_iterations"]), int(summ["num_edges_used"]))
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    if rank == 0:
        ref = RotationProblem(n, g["edge_i"], g["edge_j"], g["rel_aa"], et, cov6=g["cov6"], inlier_weight=g["inlier_weight"]); ref.set_loss(loss)
        r1, s1 = ref.solve(init, dense_cholesky_max_cams=0, dense_cholesky_auto_cams=0)
        np.savez(out, rot=part.gather(rot), ref_rot=r1, cost=summ["final_cost"], ref_cost=s1["final_cost"], iters=summ["num_iterations"], ref_iters=s1["num_iterations"],
                 restarts=np.array([e[0] for e in everyone]), rank_iters=np.array([e[1] for e in everyone]), rank_edges=np.array([e[2] for e in everyone]),
                 ref_restarts=s1["num_forcing_restarts"], loss=type(loss).__name__, n_edges=len(g["edge_i"]))
    dist.barrier()
    dist.destroy_process_group()


def packed_case(out, prefer_native):
    """Six scenes as one disconnected problem, whole components per rank (sharding.pack_components): the library finds that no rank holds an
    edge leaving its slice (PACKED) and every rank solves its own block with its own PCG -- no collective inside the PCG loop."""
    import torch.distributed as dist
    from globalsfmpy_amd import loss_functions as LF
    from globalsfmpy_amd.solver import RotationProblem
    sizes = [700, 300, 900, 250, 620, 410]
    scenes = [synth.make_graph(n, 12 * n, seed=900 + k, outlier_frac=0.1) for k, n in enumerate(sizes)]
    offs = np.cumsum([0] + sizes)
    g = {"n_cams": int(offs[-1])}
    for k in ("rel_aa", "cov6", "inlier_weight", "init_aa", "gt_aa"):
        g[k] = np.concatenate([s_[k] for s_ in scenes])
    g["edge_i"] = np.concatenate([s_["edge_i"] + o for o, s_ in zip(offs, scenes)]).astype(np.uint32)
    g["edge_j"] = np.concatenate([s_["edge_j"] + o for o, s_ in zip(offs, scenes)]).astype(np.uint32)
    loss = LF.HuberLoss(0.1)
    prob, part = sharding.make_sharded_problem(g, _abi.ANGLE_AXIS_COVTRACE, loss=loss, prefer_native=prefer_native)
    rot, summ = prob.solve(part.scatter(g["init_aa"]))
    import torch
    cg = torch.tensor([summ["num_pcg_collectives"], summ["num_cg_iterations"], summ["num_dense_solves"]], dtype=torch.int64)
    lo = cg.clone(); dist.all_reduce(cg, op=dist.ReduceOp.MAX); dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    if dist.get_rank() == 0:
        ref = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVTRACE, cov6=g["cov6"]); ref.set_loss(loss)
        r1, s1 = ref.solve(g["init_aa"])
        np.savez(out, rot=part.gather(rot), ref_rot=r1, cost=summ["final_cost"], ref_cost=s1["final_cost"], iters=summ["num_iterations"], ref_iters=s1["num_iterations"],
                 pcg_collectives_max=int(cg[0]), cg_max=int(cg[1]), cg_min=int(lo[1]), collectives=summ["num_collectives"], offs=offs, capped=summ["num_pcg_capped_steps"],
                 dense=int(cg[2]), ref_dense=s1["num_dense_solves"])
    dist.barrier()
    dist.destroy_process_group()


def peer_error_case(out):
    """A peer-store wait that times out (injected on rank 0) fails THAT solve on every rank -- rank 0 at its next collective call, the others after
    their own 5 s bound, because rank 0 has stopped storing -- and every later call of the communicator goes to its fallback collectives:
    the next solve succeeds and equals the first one (csrc/gsfm_peer.hip; round-4 advisor / review item 6c)."""
    import torch.distributed as dist
    from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
    from globalsfmpy_amd.solver import SolverError
    g = synth.make_graph(1203, 40000, seed=23, outlier_frac=0.3)
    prob, part = sharding.make_sharded_problem(g, _abi.ANGLE_AXIS_COVARIANCE, loss=MAGSACWeightBasedLoss(0.02), prefer_native=False, exchange="peer")
    comm = prob._comm
    assert comm.backend.startswith("peer-store"), comm.backend
    init = part.scatter(g["init_aa"])
    rot0, s0 = prob.solve(init)
    peer0, fb0 = comm.calls()
    if dist.get_rank() == 0:
        comm._lib.gsfm_peer_inject_error(comm._ctx)
    failed = False
    try:
        prob.solve(init)
    except SolverError as e:
        failed = "peer" in str(e) or "callback failed" in str(e)
    dist.barrier()
    rot2, s2 = prob.solve(init)           # every call through the fallback now
    peer2, fb2 = comm.calls()
    ok = np.array_equal(rot0, rot2) or float(np.abs(rot0 - rot2).max()) <= 1e-9
    flags = np.array([int(failed), int(comm.error()), int(ok), int(peer2 - peer0 <= 2 * s0["num_iterations"] + 64), int(fb2 > fb0)])
    import torch
    t = torch.tensor(flags, dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if dist.get_rank() == 0:
        np.savez(out, flags=t.numpy(), iters=s2["num_iterations"], ref_iters=s0["num_iterations"])
    prob.close()
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


def main():
    backend, out = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)  # every rank shares the single GPU of the test box
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
    case = sys.argv[4] if len(sys.argv) > 4 else "default"
    if case == "coarse":
        return coarse_case(out, len(sys.argv) > 3 and sys.argv[3] == "native")
    if case == "bigcoherent":
        return big_coherent_case(out, len(sys.argv) > 3 and sys.argv[3] == "native")
    if case == "packed":
        return packed_case(out, len(sys.argv) > 3 and sys.argv[3] == "native")
    if case == "peererror":
        return peer_error_case(out)
    if case == "magsacrestart":
        return magsac_restart_case(out, len(sys.argv) > 3 and sys.argv[3] == "native")
    if case.startswith("random"):
        return random_case(int(case.split(":")[1]), out, len(sys.argv) > 3 and sys.argv[3] == "native", big=case.startswith("randomcoarse"))
    g = synth.make_graph(1203, 40000, seed=23, outlier_frac=0.3)
    if case == "isolated":   # the last cameras carry no edge at all: with 8 ranks the last slice holds nothing but isolated cameras
        keep = (g["edge_i"] < 1100) & (g["edge_j"] < 1100)
        for k in ("edge_i", "edge_j", "rel_aa", "cov6", "inlier_weight"):
            g[k] = g[k][keep]
    if case == "disconnected":
        g = two_component_graph()
    prefer_native = len(sys.argv) > 3 and sys.argv[3] == "native"
    exchange = "peer" if len(sys.argv) > 3 and sys.argv[3].startswith("peer") else None   # "peer": peer-store mailboxes (csrc/gsfm_peer.hip) over the chosen collectives
    if exchange and sys.argv[3] == "peer-native":
        prefer_native = True
    world = dist.get_world_size()
    if case == "isolated":   # hand-made partition: ranks 0..world-2 share the 1100 connected cameras, the last rank owns only isolated ones
        cuts = [int(v) for v in np.linspace(0, 1100, world)] + [g["n_cams"]]
        width = max(cuts[r + 1] - cuts[r] for r in range(world))
        new_id = np.concatenate([r * width + np.arange(cuts[r + 1] - cuts[r]) for r in range(world)])
        part = sharding.Partition(g["n_cams"], world, width, new_id, [0] * world)
    else:
        part = sharding.partition_cameras(g["n_cams"], g["edge_i"], g["edge_j"], world)
    if case == "sigma":   # EstimateRotationsWithSigmaConsensus on a sharded problem
        from globalsfmpy_amd.loss_functions import TrivialLoss
        prob, part = sharding.make_sharded_problem(g, _abi.ANGLE_AXIS, loss=TrivialLoss(), prefer_native=prefer_native, part=part)
        comm = prob._comm
        init = part.scatter(g["init_aa"])
        rot, summ = prob.solve_sigma_consensus(init, 4, 0.05)
    elif case == "callback":   # a loss the device cannot describe: evaluated on the host per held edge, on every rank
        prob, part = sharding.make_sharded_problem(g, _abi.ANGLE_AXIS_COVARIANCE, loss=None, prefer_native=prefer_native, part=part)
        prob.set_loss_callback(python_only_loss)
        comm = prob._comm
        init = part.scatter(g["init_aa"])
        rot, summ = prob.solve(init, max_num_iterations=6)
    else:
        prob, part = sharding.make_sharded_problem(g, _abi.ANGLE_AXIS_COVARIANCE, loss=MAGSACWeightBasedLoss(0.02), prefer_native=prefer_native, part=part, exchange=exchange)
        comm = prob._comm
        init = part.scatter(g["init_aa"])
        opts = {"pcg_hip_graph": int(os.environ["GSFM_TEST_PCG_GRAPH"])} if "GSFM_TEST_PCG_GRAPH" in os.environ else {}
        rot, summ = prob.solve(init, **opts)
    sweep_ms = prob.time_sweep(init, reps=3) if case != "callback" else 0.0
    if dist.get_rank() == 0:
        np.savez(out, rot=part.gather(rot), cost=summ["final_cost"], iters=summ["num_iterations"], cg=summ["num_cg_iterations"],
                 term=summ["termination"], backend=comm.backend, n_ag=comm.n_all_gather, n_ar=comm.n_all_reduce, sweep_ms=sweep_ms,
                 trace=prob.trace(), outer=summ["outer_iterations"], wchange=summ["last_weight_change"],
                 graph_launches=summ["num_graph_launches"], peer_calls=(comm.calls() if hasattr(comm, "calls") else (0, 0)),
                 peer_error=(comm.error() if hasattr(comm, "error") else False), pcg_collectives=summ["num_pcg_collectives"])
    if hasattr(comm, "calls"):
        prob.close()
        comm.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
