# round 6: the randomised suites with the three-component measurement planes FORCED onto small problems (GSFM_QREL3=1: by itself the compact form starts at a million
# edges), so that the decode meets the differential fuzz's exact-zero / exact-pi / near-identity / repeated-pair rotations and the forcing fuzz's trajectories
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06/fuzz
export GSFM_QREL3=1
for seed in 201 202 203; do timeout 1200 python tests/manual/fuzz_differential.py 600 $seed 2>&1 | tail -3 > gpurun_out/r06/fuzz/q3_differential_$seed.txt; done
for seed in 41 42; do timeout 1500 python tests/manual/fuzz_forcing.py 100 $seed 2>&1 | tail -2 > gpurun_out/r06/fuzz/q3_forcing_$seed.txt; done
timeout 1200 python tests/manual/fuzz_forcing.py 60 43 dense 2>&1 | tail -2 > gpurun_out/r06/fuzz/q3_forcing_43_dense.txt
timeout 1500 python tests/manual/fuzz_components.py 30 14 2>&1 | tail -2 > gpurun_out/r06/fuzz/q3_components_14.txt
tail -n 2 gpurun_out/r06/fuzz/q3_*.txt | grep -v amdgpu | cut -c1-400
