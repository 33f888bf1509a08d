"""C4 by component (round 5): what each of the 14 scenes needs on its own -- LM iterations, PCG iterations per step (PCG only, 1e-12), exact-step
time -- against the batch solved as one disconnected problem.  Same construction as tests/test_gpu_fullsize.py::test_c4 and bench.py."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd import loss_functions as LF
from globalsfmpy_amd.solver import RotationProblem
from test_gpu_fullsize import _madrid_component
sizes = [577, 227, 450, 553, 332, 328, 2152, 1084, 572, 789, 836, 437, 5288]
scenes = [synth.make_graph(n, 12 * n, seed=400 + k, outlier_frac=0.1) for k, n in enumerate(sizes)]
scenes.insert(2, _madrid_component(os.path.join(ROOT, "tests", "golden")))
loss = LF.HuberLoss(0.1)
def solve(g_list, **kw):
    offs = np.cumsum([0] + [g["n_cams"] for g in g_list])
    ei = np.concatenate([g["edge_i"] + o for o, g in zip(offs, g_list)]).astype(np.uint32)
    ej = np.concatenate([g["edge_j"] + o for o, g in zip(offs, g_list)]).astype(np.uint32)
    rel = np.concatenate([g["rel_aa"] for g in g_list]); cov = np.concatenate([g["cov6"] for g in g_list]); init = np.concatenate([g["init_aa"] for g in g_list])
    p = RotationProblem(int(offs[-1]), ei, ej, rel, _abi.ANGLE_AXIS_COVTRACE, cov6=cov)
    p.set_loss(loss)
    p.solve(init, **kw)
    t = time.perf_counter(); r, s = p.solve(init, **kw); dt = time.perf_counter() - t
    tr = p.trace()
    p.close()
    return 1e3 * dt, s, tr
for c, g in enumerate(scenes):
    ms, s, tr = solve([g], dense_cholesky_max_cams=0, dense_cholesky_auto_cams=0, pcg_forcing=0)
    ms2, s2, _ = solve([g])
    print("component %2d: %4d cameras %6d edges: PCG only %6.2f ms, %2d LM, %5d PCG (per step: %s); default options %6.2f ms, %2d LM, %d dense, %d PCG"
          % (c, g["n_cams"], len(g["edge_i"]), ms, s["num_iterations"], s["num_cg_iterations"], " ".join("%d" % v for v in tr[1:, 7]), ms2, s2["num_iterations"], s2["num_dense_solves"], s2["num_cg_iterations"]), flush=True)
ms, s, tr = solve(scenes)
print("all 14 as one problem: %.2f ms, %d LM, %d PCG (per step: %s), %d dense" % (ms, s["num_iterations"], s["num_cg_iterations"], " ".join("%d" % v for v in tr[1:, 7]), s["num_dense_solves"]))
ms, s, tr = solve([g for k, g in enumerate(scenes) if k != 2])
print("the 13 synthetic ones as one problem: %.2f ms, %d LM, %d PCG (per step: %s)" % (ms, s["num_iterations"], s["num_cg_iterations"], " ".join("%d" % v for v in tr[1:, 7])))
