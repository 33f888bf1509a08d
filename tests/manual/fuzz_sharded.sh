#!/bin/bash
# Random sharded cases (tests/sharded_worker.py "random:<seed>") over gloo on one GPU: usage fuzz_sharded.sh <first seed> <last seed>
# each seed runs with 2..8 ranks (seed mod 7 + 2); prints one verdict line per seed.  FUZZ_CASE=randomcoarse: 5-9k-camera coherent graphs
# (the two-level preconditioner is voted in by the ranks).
cd "$(dirname "$0")/../.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
for seed in $(seq $1 $2); do
  world=$(( seed % 7 + 2 ))
  out=/tmp/fz_sharded_$seed.npz
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $world --master-addr 127.0.0.1 --master-port $((29800 + seed % 150)) \
      tests/sharded_worker.py gloo $out torch ${FUZZ_CASE:-random}:$seed > /tmp/fz_sharded_$seed.log 2>&1
  rc=$?
  python - "$out" "$seed" "$world" "$rc" <<'PY'
import sys, numpy as np
out, seed, world, rc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sys.path.insert(0, ".")
from globalsfmpy_amd import synth
if rc != 0:
    print("seed %d world %d: worker failed rc=%d (see /tmp/fz_sharded_%d.log)" % (seed, world, rc, seed)); sys.exit(0)
r = np.load(out)
d = synth.angular_distance(r["rot"], r["ref_rot"]).max()
# Per-camera sums are complete on their owner and bitwise equal to the single-GPU ones; the PCG scalars are dot products over the cameras in
# the PARTITION's order (relabelled, padded), i.e. the same numbers added in another order: rounding-level differences, which an
# ill-conditioned step (thousands of PCG iterations) turns into a PCG count off by a few and a staircase loss into 1e-6 rad.
same = int(r["iters"]) == int(r["ref_iters"]) and int(r["cg"]) == int(r["ref_cg"]) and d < 1e-9
close = int(r["iters"]) == int(r["ref_iters"]) and abs(int(r["cg"]) - int(r["ref_cg"])) <= 0.02 * int(r["ref_cg"]) + 2 and \
    abs(float(r["cost"]) - float(r["ref_cost"])) <= 1e-8 * float(r["ref_cost"]) and d < (1e-4 if "MAGSAC" in str(r["loss"]) else 1e-6)
# (round 6) the same LM iterations, cost and rotations with a PCG COUNT further apart than 2 % + 2: the default (forcing) schedule stops loose solves on an
# estimate that moves by an iteration or two with the summation order, and a restart bills its abandoned attempt -- identical under the round-5 library
# (profiles/r06_fuzz.txt); the answer is the same, so not a mismatch
counts = int(r["iters"]) == int(r["ref_iters"]) and abs(float(r["cost"]) - float(r["ref_cost"])) <= 1e-8 * float(r["ref_cost"]) and d < (1e-4 if "MAGSAC" in str(r["loss"]) else 1e-6)
print("seed %d world %d n=%d e=%d et=%d %s slices %s: %s (LM %d/%d, PCG %d/%d, max dR %.1e)" % (seed, world, int(r["n"]), int(r["e"]), int(r["et"]), str(r["loss"]), r["widths"].tolist(),
      "same" if same else "rounding-level" if close else "pcg-count-only" if counts else "MISMATCH", int(r["iters"]), int(r["ref_iters"]), int(r["cg"]), int(r["ref_cg"]), d))
PY
done
