"""The N > 1 path on CPU: world_size-2 gloo processes.

The HIP kernels cannot run here, so the sharded ALGORITHM is exercised with the CPU oracle standing in
for the per-rank kernels (tests may use the oracle): each rank builds the edge subset the C-ABI asks for
(edges touching its camera slice), evaluates the rows it owns, and the slices are exchanged with the same
collectives the product uses (all-gather of per-camera slices, all-reduce of the cost).  The result must
equal the unsharded linearisation.  Partition invariants are checked exhaustively as well."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from globalsfmpy_amd import _abi, sharding, synth
    from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
    from oracle import pyoracle
    g = synth.make_graph(203, 3000, seed=17, outlier_frac=0.2)
    n = g["n_cams"]
    perm = sharding.balance_permutation(n, g["edge_i"], g["edge_j"], world)
    ei = perm[g["edge_i"].astype(np.int64)].astype(np.uint32)
    ej = perm[g["edge_j"].astype(np.int64)].astype(np.uint32)
    init = np.empty_like(g["init_aa"]); init[perm] = g["init_aa"]
    P = sharding.slice_width(n, world)
    lo, hi = rank * P, min((rank + 1) * P, n)
    m = sharding.local_edge_mask(n, ei, ej, rank, world)
    loss = MAGSACWeightBasedLoss(0.02)
    local = pyoracle.OracleProblem(n, ei[m], ej[m], g["rel_aa"][m], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"][m])
    local.set_loss(loss)
    lin = local.linearize(init)
    # owned rows are complete on this rank: pack [g(3) | D(9)] per camera into the slice of a padded buffer
    buf = torch.zeros(world * P, 12, dtype=torch.float64)
    buf[lo:hi, :3] = torch.from_numpy(lin["gradient"][lo:hi])
    buf[lo:hi, 3:] = torch.from_numpy(lin["diag_blocks"][lo:hi].reshape(-1, 9))
    flat = buf.view(-1)
    dist.all_gather_into_tensor(flat, flat[rank * P * 12:(rank + 1) * P * 12].clone())
    # cost: every edge is counted by exactly one rank
    owner = sharding.cost_owner(n, ei[m], ej[m], world)
    rho = local.residuals(init)["rho"][:, 0]
    cost = torch.tensor([0.5 * rho[owner == rank].sum()], dtype=torch.float64)
    dist.all_reduce(cost)
    if rank == 0:
        full = pyoracle.OracleProblem(n, ei, ej, g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
        full.set_loss(loss)
        ref = full.linearize(init)
        got = buf.numpy()[:n]
        ret["grad_err"] = float(np.abs(got[:, :3] - ref["gradient"]).max() / np.abs(ref["gradient"]).max())
        ret["blk_err"] = float(np.abs(got[:, 3:].reshape(-1, 3, 3) - ref["diag_blocks"]).max() / np.abs(ref["diag_blocks"]).max())
        ret["cost_err"] = float(abs(cost.item() - ref["cost"]) / ref["cost"])
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_linearisation_equals_unsharded_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret["grad_err"] < 1e-12
    assert ret["blk_err"] < 1e-12
    assert ret["cost_err"] < 1e-13


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_partition_invariants(world):
    from globalsfmpy_amd import sharding, synth
    g = synth.make_graph(1001, 20000, seed=5)
    n = g["n_cams"]
    perm = sharding.balance_permutation(n, g["edge_i"], g["edge_j"], world)
    assert np.array_equal(np.sort(perm), np.arange(n))
    ei = perm[g["edge_i"].astype(np.int64)]
    ej = perm[g["edge_j"].astype(np.int64)]
    P = sharding.slice_width(n, world)
    assert P * world >= n
    masks = [sharding.local_edge_mask(n, ei, ej, r, world) for r in range(world)]
    held = np.sum(masks, axis=0)
    assert held.min() >= 1 and held.max() <= 2                      # an edge lives on the owners of its two cameras
    owner = sharding.cost_owner(n, ei, ej, world)
    for r in range(world):
        assert masks[r][owner == r].all()                           # the cost owner always holds the edge
    # directed entries: each (camera, edge) incidence is evaluated exactly once, on the camera's owner
    incid = np.zeros(world, dtype=np.int64)
    for r in range(world):
        lo, hi = r * P, min((r + 1) * P, n)
        incid[r] = ((ei >= lo) & (ei < hi)).sum() + ((ej >= lo) & (ej < hi)).sum()
    assert incid.sum() == 2 * ei.size
    assert incid.max() <= 1.05 * incid.mean() + 64                  # balanced rows
    counts = np.bincount(owner, minlength=world)
    assert counts.max() <= 1.1 * counts.mean() + 64                 # balanced cost sweeps
