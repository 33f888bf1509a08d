"""Round-5 review item 3, first question: where does the column-sorted layout (K2c / K3c, row blocks of 512 rows) stop paying?  The rule of rounds
3-5 (problem_create.hpp) switches it off once a block holds fewer entries than half the cameras (512 rows x mean degree < cameras / 2) -- C5 sits
right at it (102 k entries per block for 100 k cameras).  Same generator, mean degree 200, growing camera counts: the default choice against the
forced layout (GSFM_K3_COLSORT=1), kernels and one whole solve each.  usage: r06_density_probe.py cams[,cams...] [degree]"""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 3 and sys.argv[3] == "child":
    import numpy as np
    from globalsfmpy_amd import _abi, synth
    from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
    from globalsfmpy_amd.solver import RotationProblem
    n, deg = int(sys.argv[1]), int(sys.argv[2])
    e = n * deg // 2
    g = synth.make_graph(n, e, 77, outlier_frac=0.3)
    t = time.perf_counter()
    p = RotationProblem(n, g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"]); p.set_loss(MAGSACWeightBasedLoss(0.02))
    tc = time.perf_counter() - t
    kt = p.time_kernels(g["init_aa"], reps=5)
    p.solve(g["init_aa"])
    t = time.perf_counter(); rot, s = p.solve(g["init_aa"]); dt = time.perf_counter() - t
    err = synth.angular_distance(synth.align_rotations(rot, g["gt_aa"]), g["gt_aa"])
    print(json.dumps({"cams": n, "edges": e, "form": int(p.matvec_bytes()[1]), "create_s": tc, "k_cost_us": 1e3 * kt["k_cost"], "k_lin_us": 1e3 * kt["k_lin"], "k_matvec_us": 1e3 * kt["k_matvec"],
                      "solve_ms": 1e3 * dt, "lm": s["num_iterations"], "pcg": s["num_cg_iterations"], "cost": s["final_cost"], "err_deg": float(np.rad2deg(err.mean())),
                      "matvec_ns_per_1000_directed_entries": 1e6 * kt["k_matvec"] / (2 * e)}))
    sys.exit(0)
cams = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "200000,400000,800000").split(",")]
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for n in cams:
    for mode in ("default", "forced"):
        env = dict(os.environ)
        env.pop("GSFM_K3_COLSORT", None)
        if mode == "forced": env["GSFM_K3_COLSORT"] = "1"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), str(deg), "child"], env=env, capture_output=True, text=True, timeout=1500)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print("%-8s %s" % (mode, line[-1] if line else "FAILED: " + r.stderr[-400:]), flush=True)
