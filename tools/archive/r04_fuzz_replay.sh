export FUZZ_ONLY=30,38,53,54
for cfg in "GSFM_FORCING_KAPPA=5e-6" "GSFM_FORCING_KAPPA=0" "GSFM_FORCING_KAPPA=0 GSFM_FORCING_STRICT_AFTER_REJECTION=1"; do
  echo "=== $cfg"
  env $cfg python - <<'PY' 2>&1 | grep "^trial" | cut -c1-175
import sys; sys.path.insert(0, "tests/manual")
import fuzz_forcing
fuzz_forcing.run(60, 3, with_oracle=False)
PY
done
