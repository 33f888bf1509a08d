# tests/manual/fuzz_differential.py (device against oracle on adversarial small graphs) with the column-sorted layout forced on, K3c's 2-byte
# record as the builder chooses it and FORCED (escape-heavy layouts: small graphs are sparse in the cameras), six further seeds
cd "$(dirname "$0")/.."
export GSFM_K3_COLSORT=1
for k16 in auto 1; do
  for seed in 22 23 24 25 26 27; do
    if [ $k16 = 1 ]; then export GSFM_K3C_K16=1; else unset GSFM_K3C_K16; fi
    echo "## GSFM_K3_COLSORT=1 GSFM_K3C_K16=$k16 seed $seed"
    timeout 600 python - <<P 2>&1 | grep -v amdgpu | tail -3
import sys
sys.path.insert(0, "tests/manual"); sys.path.insert(0, ".")
import fuzz_differential
rc = fuzz_differential.run(trials=120, seed=$seed, quick=True)
print("mismatches:", rc)
P
  done
done
