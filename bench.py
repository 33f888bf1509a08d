#!/usr/bin/env python3
"""bench.py -- the hot path's headline metric on MI355X.

Metric (BASELINE.json): edge-residuals/sec (+ IRLS iterations to 1e-6) of the robust
rotation-averaging solve on the synthetic 100k-camera / 10M-edge pose graph with 30 % outlier
edges, covariance-whitened residuals (ANGLE_AXIS_COVARIANCE) and the MAGSAC sigma-consensus loss
(the reference pipeline's defaults, scripts/sfm_pipeline.py:136-141).

A "step" is ONE full solve (gsfm_rot_solve) from the same initial guess with the graph already
resident in HBM.  value = E * (full-edge residual sweeps per solve) / (time per solve), aggregated
over all ranks.  N > 1 shards the SAME graph by camera slices (strong scaling, one process per GPU,
RCCL all-gather of the per-camera slices), launched as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cams", type=int, default=100000)
    ap.add_argument("--edges", type=int, default=10000000)
    ap.add_argument("--outliers", type=float, default=0.3)
    ap.add_argument("--seed", type=int, default=2023)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--sweep-reps", type=int, default=20)
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-sample-cams", type=int, default=0, help="0 = the benchmark graph itself (same cameras, edges and seed)")
    ap.add_argument("--cpu-sample-edges", type=int, default=0)
    ap.add_argument("--cpu-single-cams", type=int, default=10000, help="1-thread CPU sample (0 = skip)")
    ap.add_argument("--cpu-single-edges", type=int, default=1000000)
    ap.add_argument("--verbose", type=int, default=0)
    ap.add_argument("--sigma-pass", type=int, default=1, help="time the sigma-consensus weight pass on an ANGLE_AXIS problem of the same graph")
    ap.add_argument("--small-graphs", type=int, default=1, help="also time the latency-regime configurations (C1 Madrid, C2), each beside the CPU oracle")
    ap.add_argument("--coherent", type=int, default=0, help="also time the spatially coherent 100k / 2M graph with and without the two-level preconditioner (not a BASELINE config)")
    ap.add_argument("--tree-init", type=int, default=1, help="also solve from the maximum-spanning-tree initialisation of SURVEY 8(d)")
    ap.add_argument("--weak-leg", type=int, default=1, help="N > 1, --scaling strong: also time the weak-scaling point (cams x N, edges x N) and report it as `weak_scaling` on the same line")
    return ap.parse_args()


def kernel_source_sha16():
    """Identity of the device code a committed PMC measurement belongs to: bench.py reports `traffic` only if it still matches."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "globalsfmpy_amd", "csrc")
    for f in ("kernels.hpp", "colsort_kernels.hpp", "loss_dev.hpp", "so3_dev.hpp", "devmath.hpp"):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(args, loss_ctor, error_type, device_rot=None, device_summary=None):
    """The CPU oracle (restatement of the reference's Ceres path) timed on this host's cores: by default one full solve of THE
    BENCHMARK GRAPH itself (about 17 s on 16 cores; --cpu-sample-cams/-edges select a smaller graph of the same generator instead),
    which doubles as a full-size parity check.  SURVEY 8d asks for one thread and for all host cores: the all-cores figure is
    `value`, the 1-thread one (a smaller sample) rides along."""
    from globalsfmpy_amd import synth
    from oracle import pyoracle

    def timed(n_cams, n_edges, threads, seed):
        g = synth.make_graph(n_cams, n_edges, seed, outlier_frac=args.outliers)
        p = pyoracle.OracleProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], error_type, cov6=g["cov6"])
        p.set_loss(loss_ctor())
        prev = pyoracle.lib().orc_set_num_threads(threads)
        try:
            t0 = time.perf_counter()
            r, s = p.solve(g["init_aa"])
            dt = time.perf_counter() - t0
        finally:
            pyoracle.lib().orc_set_num_threads(prev)
        s["rotations"] = r
        return n_edges * s["num_residual_sweeps"] / dt, s, dt

    cores = pyoracle.usable_cores()   # affinity mask capped by the cgroup CPU quota
    same = args.cpu_sample_cams <= 0 or args.cpu_sample_edges <= 0
    sc, se = (args.cams, args.edges) if same else (args.cpu_sample_cams, args.cpu_sample_edges)
    rate, s, dt = timed(sc, se, cores, args.seed if same else args.seed + 1)
    out = {
        "value": rate,
        "unit": "edge-residuals/s",
        "cores": cores,
        "kind": "port",
        "host": "%d hardware threads visible, %d usable under the cgroup CPU quota" % (os.cpu_count() or 0, cores),
        "sample": "oracle/ (C++ restatement of the reference's Ceres LM path, OpenMP over edges, PCG 1e-14): "
                  "1 full solve of %s (%d cameras / %d edges, %g outliers), %d sweeps, %d LM iterations, %.1f s"
                  % ("THE BENCHMARK GRAPH itself, same seed" if same else "a smaller graph from the same generator (same mean degree)", sc, se, args.outliers,
                     s["num_residual_sweeps"], s["num_iterations"], dt),
        "final_cost": s["final_cost"],
    }
    if same and device_rot is not None:   # the same problem was just solved on the device: the oracle doubles as the full-size parity check
        d = synth.angular_distance(synth.align_rotations(device_rot, s["rotations"]), s["rotations"])
        out["device_vs_cpu"] = {"mean_angular_difference_rad": float(d.mean()), "max_angular_difference_rad": float(d.max()),
                                "final_cost_relative_difference": abs(device_summary["final_cost"] - s["final_cost"]) / s["final_cost"],
                                "lm_iterations_device": device_summary["num_iterations"], "lm_iterations_cpu": s["num_iterations"]}
    ceres_bin = os.path.join(ROOT, "tools", "bench_ceres", "build", "bench_ceres")
    if os.path.exists(ceres_bin):   # optional target: only on a box that has Ceres + Eigen; never in this image
        import subprocess
        import tempfile
        try:
            with tempfile.TemporaryDirectory() as td:
                gb = os.path.join(td, "graph.bin")
                subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "bench_ceres", "dump_graph.py"), str(args.cpu_single_cams or 10000),
                                       str(args.cpu_single_edges or 1000000), gb, str(args.seed + 1), str(args.outliers)], stdout=subprocess.DEVNULL)
                line = subprocess.run([ceres_bin, gb, str(cores)], capture_output=True, text=True, timeout=1800).stdout.strip().splitlines()[-1]
                out["ceres"] = json.loads(line)
                out["ceres"]["sample"] = "this build's own functors on the Ceres API (tools/bench_ceres), SPARSE_NORMAL_CHOLESKY, same generator"
        except Exception as e:  # noqa: BLE001
            out["ceres"] = {"error": repr(e)}
    if args.cpu_single_cams > 0:
        r1, s1, dt1 = timed(args.cpu_single_cams, args.cpu_single_edges, 1, args.seed + 1)
        out["single_thread"] = {"value": r1, "unit": "edge-residuals/s", "cores": 1,
                                "sample": "same oracle, OpenMP pinned to 1 thread: %d cameras / %d edges, %d sweeps, %.1f s"
                                          % (args.cpu_single_cams, args.cpu_single_edges, s1["num_residual_sweeps"], dt1)}
    return out


def small_graph_timings(args):
    """The BASELINE configurations that live in the launch-latency regime, one full solve each (best of 3, host wall time of
    gsfm_rot_solve): C1 = the real Madrid_Metropolis view graph (394 views / 23 784 edges, tests/golden/madrid_graph.npz; synthetic
    covariances as in the tests, spanning-tree initialisation), C2 = synthetic 10k cameras / 200k edges."""
    from globalsfmpy_amd import _abi, synth
    from globalsfmpy_amd import loss_functions as LF
    from globalsfmpy_amd.solver import RotationProblem
    from oracle import pyoracle
    out = {}
    cores = pyoracle.usable_cores()

    def best(p, x0, oracle_args=None, comps=None):   # comps: camera ranges of the connected components (each has its own gauge)
        ts, s = [], None
        rd, _ = p.solve(x0)
        for _ in range(3):
            t = time.perf_counter()
            rd, s = p.solve(x0)
            ts.append(time.perf_counter() - t)
        r = {"ms": 1e3 * min(ts), "lm_iterations": s["num_iterations"], "cg_iterations": s["num_cg_iterations"], "dense_cholesky_steps": s["num_dense_solves"],
             "inexact_steps": s["num_inexact_steps"], "forcing_restarts": s["num_forcing_restarts"], "pcg_capped_steps": s["num_pcg_capped_steps"]}
        if oracle_args is not None and args.cpu_baseline:   # the same solve by the CPU oracle on all usable cores, beside it
            n, ei, ej, rel, et, c6, loss = oracle_args
            o = pyoracle.OracleProblem(n, ei, ej, rel, et, cov6=c6)
            o.set_loss(loss)
            t = time.perf_counter()
            ro, so = o.solve(x0)
            r["cpu_oracle_ms"] = 1e3 * (time.perf_counter() - t)
            r["cpu_oracle_cores"] = cores
            r["cpu_oracle_lm_iterations"] = so["num_iterations"]
            if comps is None:
                r["device_vs_cpu_mean_rad"] = float(synth.angular_distance(synth.align_rotations(rd, ro), ro).mean())
            else:
                d = np.concatenate([synth.angular_distance(synth.align_rotations(rd[a:b], ro[a:b]), ro[a:b]) for a, b in comps])
                r["device_vs_cpu_mean_rad"] = float(d.mean())
                r["device_vs_cpu_max_rad"] = float(d.max())
        return r

    g = synth.make_graph(10000, 200000, 11, outlier_frac=0.1)
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS)
    p.set_loss(LF.GemanMcClureLoss(0.1, 1.0))
    out["C2_10k_cams_200k_edges_geman_mcclure"] = best(p, g["init_aa"], (g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS, None, LF.GemanMcClureLoss(0.1, 1.0)))
    p.close()
    p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
    p.set_loss(LF.MAGSACWeightBasedLoss(0.02))
    out["C2_10k_cams_200k_edges_cov_magsac"] = best(p, g["init_aa"], (g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, g["cov6"], LF.MAGSACWeightBasedLoss(0.02)))
    p.close()
    # a spatially coherent graph (what real large view graphs look like, unlike the uniformly random C5): neighbours within +-500 of a hidden
    # ordering, ids shuffled.  Block-Jacobi PCG against the two-level preconditioner the library chooses for such graphs (same answer).
    gc = synth.make_graph(100000, 2000000, 7, outlier_frac=0.1, local_window=1000) if args.coherent else None
    for label, env in ((("two_level_auto", None), ("block_jacobi_only", "0")) if args.coherent else ()):
        if env is None:
            os.environ.pop("GSFM_PCG_COARSE", None)
        else:
            os.environ["GSFM_PCG_COARSE"] = env
        p = RotationProblem(gc["n_cams"], gc["edge_i"], gc["edge_j"], gc["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=gc["cov6"])
        p.set_loss(LF.MAGSACWeightBasedLoss(0.02))
        out["coherent_100k_cams_2M_edges_cov_magsac_" + label] = best(p, gc["init_aa"])
        p.close()
    os.environ.pop("GSFM_PCG_COARSE", None)
    path = os.path.join(ROOT, "tests", "golden", "madrid_graph.npz")
    if os.path.exists(path):
        try:
            sys.path.insert(0, os.path.join(ROOT, "globalsfmpy_amd"))
            import GlobalSfMpy as sfm
            m = np.load(path)
            ids = np.sort(m["view_ids"])
            idx = {int(v): k for k, v in enumerate(ids)}
            vg = sfm.ViewGraph()
            for a_, b_, r_ in zip(m["edge_a"], m["edge_b"], m["rel_aa"]):
                info = sfm.TwoViewInfo()
                info.rotation_2 = r_
                info.num_verified_matches = 1
                vg.AddEdge(int(a_), int(b_), info)
            init = sfm.MapViewIdVector3d()
            sfm.OrientationsFromMaximumSpanningTree(vg, init)
            x0 = np.array([init[int(v)] for v in ids])
            ei = np.array([idx[int(v)] for v in m["edge_a"]], dtype=np.uint32)
            ej = np.array([idx[int(v)] for v in m["edge_b"]], dtype=np.uint32)
            rng = np.random.default_rng(7)
            A = rng.standard_normal((len(ei), 3, 3))
            S = (A @ np.transpose(A, (0, 2, 1)) + 0.5 * np.eye(3)) * 3e-8
            c6 = np.stack([S[:, 0, 0], S[:, 1, 1], S[:, 2, 2], S[:, 0, 1], S[:, 0, 2], S[:, 1, 2]], axis=1)
            p = RotationProblem(len(ids), ei, ej, m["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=c6)
            p.set_loss(LF.MAGSACWeightBasedLoss(0.02))
            out["C1_madrid_394_views_23784_edges_cov_magsac"] = best(p, x0, (len(ids), ei, ej, m["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, c6, LF.MAGSACWeightBasedLoss(0.02)))
            p.close()
            p = RotationProblem(len(ids), ei, ej, m["rel_aa"], _abi.ANGLE_AXIS)
            p.set_loss(LF.SoftLOneLoss(0.1))
            out["C1_madrid_softl1_EstimateRotations_default"] = best(p, x0, (len(ids), ei, ej, m["rel_aa"], _abi.ANGLE_AXIS, None, LF.SoftLOneLoss(0.1)))
            p.close()
            # C4: the 14 1DSfM scenes as ONE disconnected problem (the construction of tests/test_gpu_fullsize.py: the real Madrid graph + thirteen
            # synthetic graphs with the scenes' camera counts, Trafalgar's 5 288 included; Huber(0.1), trace-weighted covariances)
            sizes = [577, 227, 450, 553, 332, 328, 2152, 1084, 572, 789, 836, 437, 5288]
            scenes = [synth.make_graph(n, 12 * n, seed=400 + k, outlier_frac=0.1) for k, n in enumerate(sizes)]
            scenes.insert(2, {"n_cams": len(ids), "edge_i": ei, "edge_j": ej, "rel_aa": m["rel_aa"], "cov6": c6, "init_aa": x0})
            offs = np.cumsum([0] + [g4["n_cams"] for g4 in scenes])
            ei4 = np.concatenate([g4["edge_i"] + o for o, g4 in zip(offs, scenes)]).astype(np.uint32)
            ej4 = np.concatenate([g4["edge_j"] + o for o, g4 in zip(offs, scenes)]).astype(np.uint32)
            rel4 = np.concatenate([g4["rel_aa"] for g4 in scenes]); cov4 = np.concatenate([g4["cov6"] for g4 in scenes]); init4 = np.concatenate([g4["init_aa"] for g4 in scenes])
            p = RotationProblem(int(offs[-1]), ei4, ej4, rel4, _abi.ANGLE_AXIS_COVTRACE, cov6=cov4)
            p.set_loss(LF.HuberLoss(0.1))
            out["C4_14_scenes_%d_cams_%d_edges_one_disconnected_problem" % (int(offs[-1]), len(ei4))] = best(p, init4, (int(offs[-1]), ei4, ej4, rel4, _abi.ANGLE_AXIS_COVTRACE, cov4, LF.HuberLoss(0.1)),
                                                                                                          comps=[(int(a), int(b)) for a, b in zip(offs[:-1], offs[1:])])
            p.close()
        except Exception as e:  # noqa: BLE001  (the extra keys never take the headline down with them)
            out["C1_madrid_error"] = repr(e)
    return out


def main():
    t_process = time.perf_counter()
    args = parse()
    # Native libraries print to the C-level stdout (RCCL writes a five-line version banner there when its first communicator is created).
    # The contract is ONE JSON line on stdout: everything else of this process goes to stderr, the line is written to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the solver has no CPU fallback")
    local_rank %= torch.cuda.device_count()   # (dev smoke test: several gloo ranks may share one GPU)
    torch.cuda.set_device(local_rank)
    dist = None
    backend = os.environ.get("GSFM_BENCH_BACKEND", "nccl")  # "nccl" = RCCL over xGMI; "gloo" only for smoke tests
    # GSFM_FORCE_SHARD=1 under a one-process torchrun: the N > 1 code path (process group, partition, native RCCL communicator, collective
    # callbacks inside the solve) with a single rank -- the closest a 1-GPU box gets to the 8-GPU launch
    force_shard = world == 1 and os.environ.get("GSFM_FORCE_SHARD") and "RANK" in os.environ
    if world > 1 or force_shard:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    watchdog = None
    if world > 1:
        # a collective that never returns (a communicator that did not come up on some rank) must not hold the launcher until ITS limit:
        # every rank gives set-up, warm-up and the timed solves this long, then says so and leaves
        import threading
        limit = float(os.environ.get("GSFM_BENCH_WATCHDOG_S", "900"))

        def _give_up():
            sys.stderr.write("bench.py rank %d: no progress for %.0f s (set-up / collectives hung?): giving up\n" % (rank, limit))
            sys.stderr.flush()
            os._exit(124)
        watchdog = threading.Timer(limit, _give_up)
        watchdog.daemon = True
        watchdog.start()

    from globalsfmpy_amd import _abi, synth, sharding
    from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
    from globalsfmpy_amd.solver import RotationProblem

    error_type = _abi.ANGLE_AXIS_COVARIANCE
    loss_ctor = lambda: MAGSACWeightBasedLoss(0.02)  # noqa: E731

    n_cams, n_edges = args.cams, args.edges
    if args.scaling == "weak":
        n_cams, n_edges = args.cams * world, args.edges * world
    def shared_graph(n_cams, n_edges):
        """The synthetic graph, generated once per node (the contract is one node): rank 0 generates it and the others map its arrays from shared memory."""
        g = None
        if dist is not None and world > 1:
            # one node (the contract): rank 0 generates the graph once and the others map its arrays from shared memory instead of
            # generating 8 copies (6 s and 3 GB each); any failure falls back to generating locally -- the generator is deterministic
            shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
            path = None if shm is None else os.path.join(shm, "gsfm_bench_%s_%d_%d_%d" % (os.environ.get("MASTER_PORT", "0"), n_cams, n_edges, args.seed))
            ok = torch.zeros(1, dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
            if rank == 0 and path is not None:
                try:
                    g = synth.make_graph(n_cams, n_edges, args.seed, outlier_frac=args.outliers)
                    os.makedirs(path, exist_ok=True)
                    for k, v in g.items():
                        np.save(os.path.join(path, k + ".npy"), np.asarray(v))
                    ok += 1
                except OSError:
                    pass
            dist.broadcast(ok, src=0)
            if int(ok.item()) == 1 and rank != 0:
                try:
                    g = {f[:-4]: np.load(os.path.join(path, f), mmap_mode="r") for f in os.listdir(path) if f.endswith(".npy")}
                    g = {k: (v if v.ndim else v.item()) for k, v in g.items()}
                except (OSError, ValueError):
                    g = None
            dist.barrier()
            if rank == 0 and path is not None and int(ok.item()) == 1:
                import shutil
                shutil.rmtree(path, ignore_errors=True)   # (the others hold their mappings; the pages live until they drop them)
        if g is None:
            g = synth.make_graph(n_cams, n_edges, args.seed, outlier_frac=args.outliers)

        return g

    t_gen = time.perf_counter()
    g = shared_graph(n_cams, n_edges)
    t_gen = time.perf_counter() - t_gen

    t_create = time.perf_counter()
    part = None
    if world > 1 or force_shard:
        prob, part = sharding.make_sharded_problem(g, error_type, loss=loss_ctor())
        comm = prob._comm
        init, gt = part.scatter(g["init_aa"]), part.scatter(g["gt_aa"])
    else:
        prob = RotationProblem(n_cams, g["edge_i"], g["edge_j"], g["rel_aa"], error_type, cov6=g["cov6"])
        prob.set_loss(loss_ctor())
        init, gt = g["init_aa"], g["gt_aa"]
    torch.cuda.synchronize()
    t_create = time.perf_counter() - t_create

    prob_main, init_main = prob, init

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_solves(n_warm, n_steps, problem=None, **opts):
        """W untimed + exactly K timed solves bracketed by barrier + synchronize; returns (seconds, max over ranks; sweeps; last summary; last rotations)."""
        prob = problem if problem is not None else prob_main
        summ_, rot_ = None, None
        resident = opts.pop("resident", True)
        init = opts.pop("init", init_main)
        if resident:
            # Inputs resident in HBM when the timed region starts (the contract): the start rotations are a device buffer, refreshed from a
            # second device buffer before every solve (a 2.4 MB device-to-device copy inside the timed region), the result stays on the device
            # until the clock has stopped.  resident=False: host arrays in and out through gsfm_rot_solve (the PCIe-inclusive figure).
            init_d = torch.tensor(np.ascontiguousarray(init), dtype=torch.float64, device="cuda")
            work_d = torch.empty_like(init_d)
            def one(**kw):
                work_d.copy_(init_d)
                torch.cuda.current_stream().synchronize()   # (the problem runs on its own stream: the buffer must be complete before it reads it)
                return prob.solve_resident(work_d, **kw)
        for _ in range(n_warm):
            if resident: summ_ = one(verbose=args.verbose if rank == 0 else 0, **opts)
            else: _, summ_ = prob.solve(init, verbose=args.verbose if rank == 0 else 0, **opts)
        barrier()
        t0 = time.perf_counter()
        sweeps_ = 0
        for _ in range(n_steps):
            if resident: summ_ = one(**opts)
            else: rot_, summ_ = prob.solve(init, **opts)
            sweeps_ += summ_["num_residual_sweeps"]
        barrier()
        el = time.perf_counter() - t0
        if resident: rot_ = work_d.cpu().numpy()
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, sweeps_, summ_, rot_

    # N > 1 with the native RCCL communicator: the PCG chunks would be captured into hipGraphs TOGETHER with their all-gathers (the library's
    # default).  That path has never run on more than one GPU, so the contract line is first measured with plain launches (pcg_hip_graph = 0)
    # and the captured variant is tried afterwards, under its own watchdog, which prints the plain-launch line if the capture hangs.
    sharded_capture = part is not None and getattr(comm, "backend", "") == "rccl-native" and os.environ.get("GSFM_PCG_GRAPH_COLLECTIVES", "1") != "0"
    base_opts = dict(pcg_hip_graph=0) if sharded_capture else {}
    elapsed, sweeps, summ, rot = timed_solves(args.warmup, args.steps, **base_opts)
    el_h, sw_h, _, rot_h = timed_solves(1, args.steps, resident=False, **base_opts)   # (every rank: the sharded solve is a collective)
    host_leg = {"what": "gsfm_rot_solve: host arrays in and out (the reference's calling convention), PCIe-inclusive -- reported, never `value`",
                "ms_per_step": 1e3 * el_h / args.steps, "value": n_edges * sw_h / el_h, "same_rotations_as_value": bool(np.array_equal(rot, rot_h))}
    if watchdog is not None:
        watchdog.cancel()

    # the other schedule of the PCG tolerance: every LM step at cg_relative_tolerance (rounds 1-3; pcg_forcing = 0).  `value` is the default schedule.
    exact = None
    if world == 1:
        el_x, sw_x, sx, rot_x = timed_solves(1, max(1, min(args.steps, 5)), pcg_forcing=0)
        dev = synth.angular_distance(synth.align_rotations(rot, rot_x), rot_x)
        nx = max(1, min(args.steps, 5))
        exact = {"pcg_forcing": 0, "value": n_edges * sw_x / el_x, "ms_per_solve": 1e3 * el_x / nx, "lm_iterations": sx["num_iterations"], "iters_to_1e-6": sx["iters_to_1e6"], "cg_iterations": sx["num_cg_iterations"],
                 "final_cost": sx["final_cost"], "default_schedule_vs_this": {"mean_angular_difference_rad": float(dev.mean()), "max_angular_difference_rad": float(dev.max()),
                                                                                "final_cost_relative_difference": abs(summ["final_cost"] - sx["final_cost"]) / sx["final_cost"]}}

    # SURVEY 8(d)'s initialisation: rotations composed along a maximum spanning tree of the view graph (what the reference pipeline feeds
    # the solver, OrientationsFromMaximumSpanningTree), instead of ground truth + 2 degrees of noise.  Same graph, same kernels; a harder start.
    tree = None
    if world == 1 and args.tree_init:
        t_tree = time.perf_counter()
        init_tree, _ = synth.spanning_tree_init(g, args.seed)
        t_tree = time.perf_counter() - t_tree
        e0 = synth.angular_distance(synth.align_rotations(init_tree, g["gt_aa"]), g["gt_aa"])
        prob.solve(init_tree, pcg_forcing=0)
        t1 = time.perf_counter()
        rot_tx, stx = prob.solve(init_tree, pcg_forcing=0)
        dt_tree_x = time.perf_counter() - t1
        prob.solve(init_tree)
        t1 = time.perf_counter()
        rot_t, st = prob.solve(init_tree)
        dt_tree = time.perf_counter() - t1
        # the schedule without its exclusions by loss and conditioning (pcg_forcing = 3; include/gsfm_rot.h): what the default gives up on this start
        prob.solve(init_tree, pcg_forcing=3)
        t1 = time.perf_counter()
        rot_t3, st3 = prob.solve(init_tree, pcg_forcing=3)
        dt_tree3 = time.perf_counter() - t1
        dd3 = synth.angular_distance(synth.align_rotations(rot_t3, rot_tx), rot_tx)
        ddx = synth.angular_distance(synth.align_rotations(rot_t, rot_tx), rot_tx)
        e1 = synth.angular_distance(synth.align_rotations(rot_t, g["gt_aa"]), g["gt_aa"])
        dd = synth.angular_distance(synth.align_rotations(rot_t, rot), rot)
        tree = {"init": "maximum spanning tree by match count (inlier pairs 150-1500 matches, outlier pairs 16-150), composed from camera 0",
                "init_mean_error_deg": float(np.rad2deg(e0.mean())), "ms_per_solve": 1e3 * dt_tree, "value": n_edges * st["num_residual_sweeps"] / dt_tree,
                "iters_to_1e-6": st["iters_to_1e6"], "lm_iterations": st["num_iterations"], "cg_iterations": st["num_cg_iterations"],
                "residual_sweeps": st["num_residual_sweeps"], "termination": st["termination_name"], "final_cost": st["final_cost"],
                "mean_angular_error_vs_ground_truth_deg": float(np.rad2deg(e1.mean())),
                "mean_difference_to_the_noise_init_solution_rad": float(dd.mean()), "host_tree_build_s": t_tree,
                "inexact_steps": st["num_inexact_steps"], "continued_solves": st["num_forcing_refinements"], "forcing_restarts": st["num_forcing_restarts"],
                "pcg_capped_steps": st["num_pcg_capped_steps"],
                "pcg_forcing_3": {"what": "the forcing schedule without its conditioning gate (the default gives it up under MAGSAC once a loose solve needs more than 64 iterations: this start's first takes 308)",
                                  "ms_per_solve": 1e3 * dt_tree3, "value": n_edges * st3["num_residual_sweeps"] / dt_tree3, "lm_iterations": st3["num_iterations"], "cg_iterations": st3["num_cg_iterations"],
                                  "inexact_steps": st3["num_inexact_steps"], "forcing_restarts": st3["num_forcing_restarts"],
                                  "vs_exact_schedule_mean_rad": float(dd3.mean()), "vs_exact_schedule_max_rad": float(dd3.max())},
                "exact_schedule": {"pcg_forcing": 0, "ms_per_solve": 1e3 * dt_tree_x, "lm_iterations": stx["num_iterations"], "cg_iterations": stx["num_cg_iterations"],
                                   "default_schedule_vs_this_mean_rad": float(ddx.mean()), "default_schedule_vs_this_max_rad": float(ddx.max())}}

    # ---- kernels, timed live with HIP events on the solver's stream (this rank's share of the problem) ----
    e_local = summ["num_edges_used"]
    kt = prob.time_kernels(init, reps=10)   # k_cost (trial cost), k_lin, k_matvec; contains collectives when sharded: every rank must call it
    variants = prob.time_sweep_variants(init, reps=args.sweep_reps) if world == 1 else None
    alg_b, lay_b = prob.sweep_bytes()
    mv_layout_bytes, mv_form = prob.matvec_bytes()
    lin_layout_bytes = prob.linearize_bytes()

    sigma_pass = None
    if world == 1 and args.sigma_pass:
        # K6, the sigma-consensus weight pass (EstimateRotationsWithSigmaConsensus): needs an ANGLE_AXIS problem of the same graph
        from globalsfmpy_amd.loss_functions import TrivialLoss
        p6 = RotationProblem(n_cams, g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS)
        p6.set_loss(TrivialLoss())
        p6.set_edge_weights(np.ones(n_edges))
        v6 = p6.time_sweep_variants(init, reps=args.sweep_reps)
        sigma_pass = {k: v6[k] for k in ("k1_sigma_fused", "k1_sigma_plain", "k2_sigma_fused", "k2_sigma_plain")}
        p6.close()

    small = None
    if world == 1 and args.small_graphs:
        small = small_graph_timings(args)

    pmc, pmc_note = None, None
    try:  # committed PMC measurement (bench.py cannot run rocprofv3 on itself): used only if it was taken on THESE kernel sources and this workload
        path = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")
        if os.path.exists(path):
            pm = json.load(open(path))
            if world == 1 and pm["workload"] == {"cams": n_cams, "edges": n_edges} and pm.get("kernel_source_sha16") == kernel_source_sha16():
                pmc = (pm, "profiles/r06_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH x2 gfx950 correction; kernel sources %s)" % pm["kernel_source_sha16"])
            else:
                pmc_note = "profiles/r06_pmc_traffic.json is for other kernel sources or another workload: not reported"
    except Exception:
        pass

    def pmc_bytes(kernel):
        if pmc is None or kernel not in pmc[0]:
            return None
        b = pmc[0][kernel]["fetch_bytes"] + pmc[0][kernel]["write_bytes"]
        if kernel == "k_matvec" and "k_matvec_finish" in pmc[0]:    # kernel_ms covers the mat-vec AND its finishing kernel: so does the traffic
            b += pmc[0]["k_matvec_finish"]["fetch_bytes"] + pmc[0]["k_matvec_finish"]["write_bytes"]
        return b

    if rank == 0:
        rot_cmp, gt_cmp = (part.gather(rot), g["gt_aa"]) if part is not None else (rot, gt)
        err = synth.angular_distance(synth.align_rotations(rot_cmp, gt_cmp), gt_cmp)
        value = n_edges * sweeps / elapsed
        # algorithmic bytes per launch.  Directed entries = 2 per edge on one GPU.
        lap = os.environ.get("GSFM_LAPLACIAN", "1") != "0"     # angle-axis / quaternion-cosine problems: 6 doubles per directed entry instead of 9
        blk = 48.0 if lap else 72.0
        nd = 2.0 * e_local if part is None else float(part.entries_per_rank[rank])   # (sharded: rank 0's rows)
        mv_bytes = mv_layout_bytes      # this rank's mat-vec as laid out (gsfm_rot_matvec_bytes): row-major 52 B per directed entry, column-sorted 54 B per position + ...
        mv_kernel = {0: "k_matvec<false> (K3: general 9-value blocks, row-major)", 1: "k_matvec<LAP> (K3: Laplacian form, row-major, 52 B per directed entry)",
                     2: ("k_mv_col + k_mv_col_finish (K3c: Laplacian form, column-sorted row blocks, 52 B per position (48 block + 4-byte record: GSFM_K3C_K16=0); csrc/colsort_kernels.hpp)" if os.environ.get("GSFM_K3C_K16") == "0" else
                         "k_mv_col + k_mv_col_finish (K3c: Laplacian form, column-sorted row blocks, 50 B per position (48 block + 2-byte delta-coded record, round 5; 52 where the layout is too sparse for it); csrc/colsort_kernels.hpp)")}[mv_form]
        lin_bytes = lin_layout_bytes    # the linearisation as laid out (row-major: col 4 + q_rel 32 + whitening + block per directed entry; column-sorted: 8 + 32 + whitening + 48 per position)
        sweep_bytes = e_local * lay_b + 32.0 * n_cams          # as laid out: idx 8 + q_rel 32 + Lt 48 per edge, the quaternions once

        def roof(bytes_per_launch, ms, **extra):
            ach = bytes_per_launch / (ms * 1e-3) / 1e9
            d = {"achieved": ach, "unit": "GB/s", "peak": HBM_PEAK_GBPS, "frac": ach / HBM_PEAK_GBPS, "kernel_ms": ms,
                 "algorithmic_bytes_per_launch": bytes_per_launch}
            d.update(extra)
            return d

        mv_share = summ["num_cg_iterations"] * kt["k_matvec"] / max(1e-9, summ["t_linearize_ms"] + summ["t_sweep_ms"] + summ["t_cg_ms"])
        out = {
            "metric": "edge_residuals_per_sec", "value": value, "unit": "edge-residuals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "synthetic SO(3) pose graph %d cams / %d edges, %g outlier edges, ANGLE_AXIS_COVARIANCE + "
                                   "MAGSACWeightBasedLoss(0.02), one full LM solve per step (BASELINE.json configs[4] graph)"
                                   % (n_cams, n_edges, args.outliers),
                       # the start `value` is measured from (round-5 review): NOT SURVEY 8(d)'s chain / spanning-tree composition -- that start is
                       # `value_tree_init` / `spanning_tree_init` below, same graph, same options
                       "init": "ground truth + 2 degrees of isotropic noise per camera (synth.make_graph: init_aa); the maximum-spanning-tree start of SURVEY 8(d) is reported beside it as value_tree_init",
                       "cams": n_cams, "edges": n_edges, "outlier_frac": args.outliers, "seed": args.seed,
                       "parallelism": "camera-slice x%d" % world,
                       "collectives": (comm.backend if part is not None else "none")},
            "iters_to_1e-6": summ["iters_to_1e6"], "lm_iterations": summ["num_iterations"],
            "residual_sweeps_per_solve": summ["num_residual_sweeps"], "cg_iterations_per_solve": summ["num_cg_iterations"],
            "termination": summ["termination_name"], "final_cost": summ["final_cost"],
            "collectives_per_solve": summ["num_collectives"],
            # one all-gather (the A.p slices) per LAUNCHED PCG iteration and nothing else inside the PCG loop (launched = chunks of cg_check_interval)
            "collectives_per_pcg_iteration": (summ["num_pcg_collectives"] / summ["num_pcg_launched"]) if (summ["num_pcg_launched"] and (world > 1 or force_shard)) else 0.0,
            "pcg_chunks_replayed_as_hipgraphs": summ["num_graph_launches"],
            "mean_angular_error_vs_ground_truth_deg": float(np.rad2deg(err.mean())),
            "gpu_ms_per_solve": {"linearize": summ["t_linearize_ms"], "sweep": summ["t_sweep_ms"], "pcg": summ["t_cg_ms"]},
            "setup_s": {"generate": t_gen, "create_problem": t_create},
            # the time-dominant kernel: one launch per PCG iteration
            "roofline": dict(roof(mv_bytes, kt["k_matvec"]), bound="hbm",
                             kernel="%s: the normal-equation mat-vec, one per PCG iteration, %.0f %% of the solve's GPU time%s"
                                    % (mv_kernel, 100.0 * mv_share, "" if world == 1 else "; rank 0's rows, kernel_ms includes the all-gather of A.p that follows every launch"),
                             frac_on_survey_8d_bytes=((80.0 * e_local + 2 * 24.0 * n_cams) / (kt["k_matvec"] * 1e-3) / 1e9 / HBM_PEAK_GBPS) if part is None else None,
                             traffic=pmc_bytes("k_matvec"), traffic_unit="bytes per launch (mat-vec + its finishing kernel, as kernel_ms)", traffic_source=(pmc[1] if pmc else pmc_note),
                             directed_entries_per_launch=int(nd), launches_per_solve=summ["num_cg_iterations"]),
        }
        # What `value` times (the contract: inputs resident in HBM when the timed region starts): the graph, the measurements and the covariances live
        # on the device since create; the camera rotations -- 2.4 MB in, 2.4 MB out -- enter and leave as a device buffer (gsfm_rot_solve_resident).
        # The same K steps through gsfm_rot_solve, host arrays in and out (the reference's calling convention, PCIe-inclusive), ride along.
        out["state_transfer"] = {"value_times": "gsfm_rot_solve_resident: start rotations and result are device buffers (inputs resident in HBM); a 2.4 MB device-to-device refresh of the start per step is inside the timed region",
                                 "host_buffers": host_leg}
        kept = summ["num_inexact_steps"] > 0 and summ["num_forcing_restarts"] == 0
        out["pcg_schedule"] = {"pcg_forcing": 1, "pcg_forcing_tolerance_rad": 1e-8, "inexact_steps_per_solve": summ["num_inexact_steps"], "continued_solves_per_solve": summ["num_forcing_refinements"],
                               "forcing_restarts_per_solve": summ["num_forcing_restarts"], "pcg_capped_steps_per_solve": summ["num_pcg_capped_steps"],
                               "worst_accepted_cg_residual": summ["worst_accepted_cg_residual"],
                               # which schedule `value` actually ran (round-4 review): the library's default options; the gate decides per run
                               "value_ran": ("forcing schedule kept: every accepted step contracted below 0.3 x its predecessor, no restart" if kept else
                                             "exact schedule (the forcing schedule was abandoned and the solve redone exactly: both attempts are inside `value`)"
                                             if summ["num_forcing_restarts"] else "exact schedule (no inexact step was taken)"),
                               "what": "every LM step is solved loosely first (estimated deviation from the exact step <= 1e-8 rad rms, no camera's block-Jacobi estimate above 1e-7); "
                                       "decisions are taken from it only when a factor two away from their thresholds, otherwise PCG continues towards cg_relative_tolerance 1e-12; "
                                       "the schedule is kept only while every accepted step is below 0.3 x its predecessor (and, under MAGSAC, no loose solve needs more than 64 iterations) -- otherwise the run is redone with exact steps "
                                       "(include/gsfm_rot.h: pcg_forcing; tests/manual/fuzz_forcing.py: 0 mismatches in 4 840 default-option trials, profiles/r05_fuzz_forcing.txt)"}
        if exact is not None:
            out["exact_schedule"] = exact
            # BASELINE's second metric for BOTH schedules (round-5 review): under MAGSAC a run that ends on rejected candidates hovering at the
            # function tolerance can end some rejections earlier or later on another schedule (include/gsfm_rot.h: pcg_forcing, "count-only")
            out["iters_to_1e-6_by_schedule"] = {"default": summ["iters_to_1e6"], "exact": exact["iters_to_1e-6"], "lm_iterations_default": summ["num_iterations"], "lm_iterations_exact": exact["lm_iterations"]}
        if tree is not None:
            out["spanning_tree_init"] = tree
            # SURVEY 8(d)'s own initialisation as a first-class number next to `value` (round-4 review): same graph, same library defaults
            out["value_tree_init"] = tree["value"]
            out["ms_per_step_tree_init"] = tree["ms_per_solve"]
        out["kernels_us"] = {k: 1e3 * v for k, v in kt.items()}
        if world == 1:
            def k1_entry(ms, out_bytes, what, pmc_key):
                d = dict(roof(sweep_bytes + out_bytes * e_local, ms), kernel=what, traffic=pmc_bytes(pmc_key), bytes_per_edge=lay_b + out_bytes,
                         sweep_rate_edges_per_s=e_local / (ms * 1e-3))
                # the same launch priced on SURVEY 8(d)'s own accounting (88 B per edge in covariance mode + 24 B per camera), whatever it stores
                d["frac_on_survey_8d_bytes"] = (e_local * alg_b + 24.0 * n_cams) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
                return d
            other = {
                "k_lin": dict(roof(lin_bytes, kt["k_lin"]), kernel=("K2c k_lin_col + finish (column-sorted layout)" if mv_form == 2 else "K2 k_lin_fast (row-major)") +
                                                                   ": residual, row Jacobian, robust weight rho', gradient, diagonal blocks, edge blocks; once per accepted LM step "
                                                                   "(fast path: losses with rho'' <= 0, see kernels.hpp lin_entry_eval)",
                              traffic=pmc_bytes("k_lin"), frac_on_survey_8d_bytes=(168.0 * e_local + 72.0 * n_cams) / (kt["k_lin"] * 1e-3) / 1e9 / HBM_PEAK_GBPS),
                "k_cost_trial": k1_entry(variants["trial_cost"], 0.0, "K1 k_cost<MODE 0>: residual + rho VALUE, block-reduced; the solver's trial-cost sweep (no rho', no per-edge store)", "k_cost"),
                "k_cost_reweight": k1_entry(variants["rho1_only"], 8.0, "K1 k_cost<MODE 2>, the reweight sweep as SURVEY 8(d) defines it (own lean instantiation since round 4): residual, loss, rho' stored per edge "
                                                                        "(problem edge order, coalesced non-temporal stores)", "k_cost_reweight"),
                "k_cost_full": k1_entry(variants["full_reweight"], 32.0, "K1 k_cost<MODE 1>: residual, s, (rho, rho', rho'') stored per edge in the problem's edge order "
                                                                         "(what gsfm_rot_residuals runs; the caller's order is restored at the C-ABI boundary)", "k_cost_full"),
                "k_cost_s_only": k1_entry(variants["s_only"], 8.0, "K1 s-only mode: s stored per edge (pass 1 of host-callback losses)", "k_cost_s_only"),
            }
            if sigma_pass is not None:
                # ANGLE_AXIS problem with scalar weights: K1 streams idx 8 + q_rel 32 + weight 8 per edge, K2 col 4 + q_rel 32 + weight 8 in, block 48 out per
                # directed entry; the fused forms add one 8-byte weight store per edge / entry.  There is no separate weight pass any more.
                e6, nd6 = float(e_local), 2.0 * e_local
                other["sigma_consensus_K6"] = {
                    "kernel": "sigma consensus on an ANGLE_AXIS problem of the same graph: the weights are computed inside the inner solve's first cost sweep (K1) and first "
                              "linearisation (K2), stored to each kernel's own weight plane; `plain` = the same kernels without the weight computation",
                    "k1_fused": roof(e6 * (48.0 + 8.0) + 32.0 * n_cams, sigma_pass["k1_sigma_fused"]), "k1_plain": roof(e6 * 48.0 + 32.0 * n_cams, sigma_pass["k1_sigma_plain"]),
                    "k2_fused": roof(nd6 * (44.0 + 48.0 + 8.0) + 72.0 * n_cams, sigma_pass["k2_sigma_fused"]), "k2_plain": roof(nd6 * (44.0 + 48.0) + 72.0 * n_cams, sigma_pass["k2_sigma_plain"]),
                    "added_ms_per_outer_iteration": (sigma_pass["k1_sigma_fused"] - sigma_pass["k1_sigma_plain"]) + (sigma_pass["k2_sigma_fused"] - sigma_pass["k2_sigma_plain"])}
            out["roofline_other"] = other
        if small is not None:
            out["small_graph_ms"] = small
        if args.cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, loss_ctor, error_type, rot_cmp, summ)
    else:
        out = None
    # From here on the line is armed: the variants below have never run across GPUs, and a GPU fault in one of them would end the process
    # from inside the runtime (abort()) -- a C-level handler then writes the plain-launch line measured above, with a `crashed_variant` field carrying the signal number, and leaves.
    crash_lib = None
    if part is not None:
        try:
            import ctypes
            crash_lib = ctypes.CDLL(os.path.join(ROOT, "globalsfmpy_amd", "libgsfm_benchguard.so"))   # (bench-only code: csrc/bench_guard.c)
            # the line says that -- and by which signal -- a later variant died: the handler patches the two digits of "signal": 00
            marker = '"signal": 00'
            line0 = (json.dumps(dict(out, crashed_variant={"what": "a variant tried AFTER the plain-launch measurement (captured collectives / peer stores) ended the process; "
                                                                  "`value` is the plain-launch measurement", "signal": 0})).replace('"signal": 0}', marker + "}") + "\n").encode() if rank == 0 else b""
            off = line0.find(marker.encode()) + len(marker) - 2 if rank == 0 else 0
            crash_lib.gsfm_crash_line_arm(ctypes.c_int(real_stdout if rank == 0 else -1), line0, ctypes.c_size_t(len(line0)), ctypes.c_size_t(max(off, 0)))
        except OSError:
            crash_lib = None
        if os.environ.get("GSFM_BENCH_TEST_CRASH") == "1":   # (test hook: what a GPU fault does)
            os.abort()
    if sharded_capture:
        # The library's default for the native communicator: PCG chunks replayed as hipGraphs WITH their all-gathers.  Tried only now that the
        # plain-launch line exists; if it does not come back in time, rank 0 prints that line and every rank leaves with status 0.
        import threading
        limit = float(os.environ.get("GSFM_BENCH_CAPTURE_WATCHDOG_S", "300"))

        def _fall_back():
            sys.stderr.write("bench.py rank %d: the hipGraph-with-collectives variant did not finish within %.0f s: reporting the plain-launch measurement\n" % (rank, limit))
            if rank == 0:
                out["hipgraph_with_collectives"] = {"status": "no result within %.0f s (watchdog): `value` is the plain-launch measurement" % limit}
                os.write(real_stdout, (json.dumps(out) + "\n").encode())
            os._exit(0)
        wd = threading.Timer(limit, _fall_back)
        wd.daemon = True
        wd.start()
        el_c, sw_c, s_c, _ = timed_solves(args.warmup, args.steps)
        wd.cancel()
        if rank == 0:
            out["plain_launches"] = {"value": out["value"], "ms_per_step": out["ms_per_step"], "pcg_chunks_replayed_as_hipgraphs": out["pcg_chunks_replayed_as_hipgraphs"]}
            out["hipgraph_with_collectives"] = {"status": "ok", "value": n_edges * sw_c / el_c, "ms_per_step": 1e3 * el_c / args.steps, "pcg_chunks_replayed_as_hipgraphs": s_c["num_graph_launches"],
                                                "same_final_cost": s_c["final_cost"] == out["final_cost"]}
            out["value"], out["ms_per_step"] = n_edges * sw_c / el_c, 1e3 * el_c / args.steps     # the library's default configuration
            out["pcg_chunks_replayed_as_hipgraphs"] = s_c["num_graph_launches"]
    want_peer = os.environ.get("GSFM_BENCH_PEER", "1" if backend == "nccl" else "0") != "0"
    if part is not None and world > 1 and want_peer:
        # Third variant: the per-iteration all-gather as PEER STORES into IPC-mapped mailboxes (csrc/gsfm_peer.hip) instead of a collective-library
        # call -- never run across GPUs before this launch either, so: own watchdog, accepted only if it reproduces the final cost, and `value`
        # takes it only if it is faster.
        import threading
        limit = float(os.environ.get("GSFM_BENCH_CAPTURE_WATCHDOG_S", "300"))

        def _fall_back_peer():
            sys.stderr.write("bench.py rank %d: the peer-store variant did not finish within %.0f s: reporting what was measured before\n" % (rank, limit))
            if rank == 0:
                out["peer_store_exchange"] = {"status": "no result within %.0f s (watchdog)" % limit}
                os.write(real_stdout, (json.dumps(out) + "\n").encode())
            os._exit(0)
        wd = threading.Timer(limit, _fall_back_peer)
        wd.daemon = True
        wd.start()
        info = {"status": "unavailable"}
        try:
            prob_p, _ = sharding.make_sharded_problem(g, error_type, loss=loss_ctor(), part=part, exchange="peer")
            if prob_p._comm.backend.startswith("peer-store"):
                el_p, sw_p, s_p, _ = timed_solves(args.warmup, args.steps, problem=prob_p)
                bad = prob_p._comm.error()
                info = {"status": "ok" if not bad else "a wait for a peer's flag ran into its bound: result discarded", "backend": prob_p._comm.backend,
                        "value": n_edges * sw_p / el_p, "ms_per_step": 1e3 * el_p / args.steps, "pcg_chunks_replayed_as_hipgraphs": s_p["num_graph_launches"],
                        "final_cost": s_p["final_cost"], "lm_iterations": s_p["num_iterations"], "cg_iterations": s_p["num_cg_iterations"]}
                if rank == 0:
                    same = abs(s_p["final_cost"] - out["final_cost"]) <= 1e-12 * abs(out["final_cost"]) and s_p["num_iterations"] == out["lm_iterations"]
                    info["reproduces_the_collective_run"] = bool(same)
                    if same and not bad and info["value"] > out["value"]:
                        out["value"], out["ms_per_step"] = info["value"], info["ms_per_step"]
                        out["config"]["collectives"] = prob_p._comm.backend
                        out["pcg_chunks_replayed_as_hipgraphs"] = info["pcg_chunks_replayed_as_hipgraphs"]
            prob_p.close()
            prob_p._comm.close()
        except Exception as e:  # noqa: BLE001  (the extra variant never takes the line down with it)
            info = {"status": "failed: %r" % (e,)}
        wd.cancel()
        if rank == 0:
            out["peer_store_exchange"] = info
    # (bounded: skipped when the strong-scaling legs have already taken GSFM_BENCH_WEAK_AFTER_S (default 240 s) of wall clock, own watchdog of
    # GSFM_BENCH_WEAK_WATCHDOG_S (default 420 s) -- whoever launched this with a clock of its own gets the strong line in any case)
    weak_ok = (time.perf_counter() - t_process) <= float(os.environ.get("GSFM_BENCH_WEAK_AFTER_S", "240"))
    if part is not None and world > 1 and args.scaling == "strong" and args.weak_leg and not weak_ok and rank == 0:
        out["weak_scaling"] = {"status": "skipped: the strong-scaling legs took %.0f s (GSFM_BENCH_WEAK_AFTER_S)" % (time.perf_counter() - t_process)}
    if part is not None and world > 1 and args.scaling == "strong" and args.weak_leg and weak_ok:
        # The WEAK-scaling point next to the strong-scaling line (round-5 review, item 1c): the graph grown with the rank count so that a rank's
        # share is the one-GPU workload (cams x world, edges x world: 800k cameras / 80 M edges at 8 ranks), same generator, same options, same
        # timing rules -- so that the first run on a real node yields both curves.  Own watchdog: whatever happens here, the line measured so far is printed.
        import threading
        limit = float(os.environ.get("GSFM_BENCH_WEAK_WATCHDOG_S", "420"))

        def _fall_back_weak():
            sys.stderr.write("bench.py rank %d: the weak-scaling leg did not finish within %.0f s: reporting what was measured before\n" % (rank, limit))
            if rank == 0:
                out["weak_scaling"] = {"status": "no result within %.0f s (watchdog)" % limit}
                os.write(real_stdout, (json.dumps(out) + "\n").encode())
            os._exit(0)
        wd = threading.Timer(limit, _fall_back_weak)
        wd.daemon = True
        wd.start()
        info = {"status": "unavailable"}
        try:
            t_w = time.perf_counter()
            prob.close()   # (the strong-scaling problem: its planes make room for the larger graph's)
            del g
            wc, we = args.cams * world, args.edges * world
            gw = shared_graph(wc, we)
            prob_w, part_w = sharding.make_sharded_problem(gw, error_type, loss=loss_ctor())
            init_w = part_w.scatter(gw["init_aa"])
            t_w = time.perf_counter() - t_w
            ksteps = max(1, min(args.steps, 3))
            el_w, sw_w, s_w, rot_w = timed_solves(1, ksteps, problem=prob_w, init=init_w, **base_opts)
            info = {"status": "ok", "scaling": "weak", "cams": wc, "edges": we, "value": we * sw_w / el_w, "unit": "edge-residuals/s", "ms_per_step": 1e3 * el_w / ksteps, "steps": ksteps, "warmup": 1,
                    "lm_iterations": s_w["num_iterations"], "cg_iterations": s_w["num_cg_iterations"], "residual_sweeps_per_solve": s_w["num_residual_sweeps"],
                    "final_cost": s_w["final_cost"], "termination": s_w["termination_name"], "collectives": prob_w._comm.backend,
                    "matvec_layout_form_rank0": prob_w.matvec_bytes()[1], "setup_s": t_w,
                    "what": "the same generator and options on a graph grown with the rank count (a rank's share = the one-GPU workload); value = all ranks' edges x sweeps / max-over-ranks time; "
                            "efficiency against the one-GPU line is for the reader to compute (layout form 2 = column-sorted K2c / K3c on rank 0, 1 = row-major)"}
            if rank == 0:
                ew = synth.angular_distance(synth.align_rotations(part_w.gather(rot_w), gw["gt_aa"]), gw["gt_aa"])
                info["mean_angular_error_vs_ground_truth_deg"] = float(np.rad2deg(ew.mean()))
            prob_w.close()
        except Exception as e:  # noqa: BLE001  (the extra leg never takes the line down with it)
            info = {"status": "failed: %r" % (e,)}
        wd.cancel()
        if rank == 0:
            out["weak_scaling"] = info
    if crash_lib is not None:
        crash_lib.gsfm_crash_line_disarm()
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
