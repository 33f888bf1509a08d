for f in 0 2e-6 5e-6 1e-5 2e-5; do
  echo "=== kappa $f"
  GSFM_FORCING_KAPPA=$f python tools/archive/r04_forcing_probe.py c5 tree 2>&1 | grep "eps 1e-08\|forcing off"
  GSFM_FORCING_KAPPA=$f python tools/archive/r04_colsort_case.py 2>&1 | grep "^{}"
done
