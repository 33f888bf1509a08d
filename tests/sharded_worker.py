"""Helper launched by tests/test_gpu_sharded.py under torch.distributed.run: solves one seeded graph
with the camera-slice sharded HIP path and writes rank 0's result to an .npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    backend, out = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)  # every rank shares the single GPU of the test box
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    from globalsfmpy_amd import _abi, sharding, synth
    from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
    g = synth.make_graph(1203, 40000, seed=23, outlier_frac=0.3)
    comm = sharding.make_comm(g["n_cams"], prefer_native=(len(sys.argv) > 3 and sys.argv[3] == "native"))
    prob, perm = sharding.make_sharded_problem(g, _abi.ANGLE_AXIS_COVARIANCE, comm, loss=MAGSACWeightBasedLoss(0.02))
    init = np.empty_like(g["init_aa"]); init[perm] = g["init_aa"]
    rot, summ = prob.solve(init)
    sweep_ms = prob.time_sweep(init, reps=3)
    if dist.get_rank() == 0:
        np.savez(out, rot=rot[perm], cost=summ["final_cost"], iters=summ["num_iterations"], cg=summ["num_cg_iterations"],
                 term=summ["termination"], backend=comm.backend, n_ag=comm.n_all_gather, n_ar=comm.n_all_reduce, sweep_ms=sweep_ms,
                 trace=prob.trace())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
