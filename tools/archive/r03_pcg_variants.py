#!/usr/bin/env python3
"""C5 solves under environment / option variants given as arguments KEY=VALUE[,KEY=VALUE] (each in its own process; OPT_<name>=<int> sets a solver
option instead of an environment variable): ms per solve, PCG ms."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] != "--one":
    for spec in sys.argv[1:]:
        env = dict(os.environ)
        for kv in spec.split(","):
            if kv and kv != "default":
                k, v = kv.split("="); env[k] = v
        r = subprocess.run([sys.executable, __file__, "--one"], env=env, capture_output=True, text=True)
        print(spec, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
    sys.exit(0)
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
from globalsfmpy_amd.solver import RotationProblem
g = synth.make_graph(100000, 10000000, 2023, outlier_frac=0.3)
p = RotationProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
p.set_loss(MAGSACWeightBasedLoss(0.02))
opts = {k[4:]: int(v) for k, v in os.environ.items() if k.startswith("OPT_")}
best = None
for rep in range(4):
    t = time.perf_counter(); r, s = p.solve(g["init_aa"], **opts); dt = time.perf_counter() - t
    if best is None or dt < best[0]: best = (dt, s)
dt, s = best
print("ms %.2f lm %d cg %d pcg ms %.2f cost %.12e" % (1e3 * dt, s["num_iterations"], s["num_cg_iterations"], s["t_cg_ms"], s["final_cost"]))
