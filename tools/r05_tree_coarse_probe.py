"""Round-4 review item 4, the experiment before the kernel family: what would a coarse space whose aggregates follow the STRONG edges buy from the
spanning-tree start?  The benchmark graph with its cameras renumbered in depth-first order of the very spanning tree the initialisation is composed
along (so that contiguous chunks of the numbering are connected subtrees: the best aggregates a device-side aggregation over the strong edges of
the first LM steps could find), solved from that start with exact steps on the row-major layout -- block-Jacobi alone against the existing
two-level preconditioner (kernels.hpp, k_coarse_*) with n aggregates.  One process per setting (the switches are read once):
  GSFM_K3_COLSORT=0 GSFM_REORDER=0 GSFM_PCG_COARSE=<0|64|128> python tools/r05_tree_coarse_probe.py [cams edges]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import depth_first_order, minimum_spanning_tree
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
n, e = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 10000000)
g = synth.make_graph(n, e, 2023, outlier_frac=0.3)
init, m = synth.spanning_tree_init(g, 2023)
ei, ej = g["edge_i"].astype(np.int64), g["edge_j"].astype(np.int64)
tree = minimum_spanning_tree(sp.coo_matrix((-m.astype(np.float64), (ei, ej)), shape=(n, n)).tocsr())
order, _ = depth_first_order(tree + tree.T, 0, directed=False)
new_id = np.empty(n, dtype=np.int64); new_id[order] = np.arange(n)
a, b = new_id[ei], new_id[ej]
sw = a > b
rel = g["rel_aa"].copy(); rel[sw] = -rel[sw]          # the pair the other way round: the inverse measurement
ei2, ej2 = np.where(sw, b, a).astype(np.uint32), np.where(sw, a, b).astype(np.uint32)
init2 = np.empty_like(init); init2[new_id] = init
p = RotationProblem(n, ei2, ej2, rel, _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
p.set_loss(LF.MAGSACWeightBasedLoss(0.02))
p.solve(init2, pcg_forcing=0)
t = time.perf_counter(); r, s = p.solve(init2, pcg_forcing=0); dt = time.perf_counter() - t
tr = p.trace()
print("GSFM_PCG_COARSE=%s: %.1f ms, %d LM iterations, %d PCG iterations (per step: %s), mat-vec form %d" % (
    os.environ.get("GSFM_PCG_COARSE", "auto"), 1e3 * dt, s["num_iterations"], s["num_cg_iterations"], " ".join("%d" % v for v in tr[1:, 7]), p.matvec_bytes()[1]), flush=True)
