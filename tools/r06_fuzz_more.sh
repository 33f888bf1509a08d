# round 6: more seeds of the forcing fuzz (default options) on the final tree
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06/fuzz
for seed in 51 52 53 54 55 56 57 58 59 60; do timeout 900 python tests/manual/fuzz_forcing.py 100 $seed 2>&1 | grep -v amdgpu > gpurun_out/r06/fuzz/more_forcing_$seed.txt; tail -1 gpurun_out/r06/fuzz/more_forcing_$seed.txt | cut -c1-300; grep -c "MISMATCH\|beyond-PCG\|ill-posed" gpurun_out/r06/fuzz/more_forcing_$seed.txt; done
for seed in 61 62 63 64; do timeout 900 python tests/manual/fuzz_forcing.py 60 $seed dense 2>&1 | grep -v amdgpu > gpurun_out/r06/fuzz/more_forcing_${seed}_dense.txt; tail -1 gpurun_out/r06/fuzz/more_forcing_${seed}_dense.txt | cut -c1-300; done
