for f in 5e-6 5e-5 1e-4 3e-4; do
  echo "=== kappa $f"
  GSFM_FORCING_KAPPA=$f python tools/archive/r04_forcing_probe.py c5 tree 2>&1 | grep "eps 1e-08\|forcing off"
  GSFM_FORCING_KAPPA=$f python tools/archive/r04_colsort_case.py 2>&1 | grep "^{}"
  GSFM_FORCING_KAPPA=$f timeout 600 python tests/manual/fuzz_forcing.py 40 9 2>&1 | grep -v amdgpu | grep "MISMATCH\|forcing fuzz" | tail -6
done
