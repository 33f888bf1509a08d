# round 6: the randomised suites on the final tree (three-component measurement planes, finish kernels, component freeze rule, layout rule), one box call
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06/fuzz
for seed in 101 102 103; do timeout 1200 python tests/manual/fuzz_differential.py 600 $seed 2>&1 | tail -4 > gpurun_out/r06/fuzz/differential_$seed.txt; done
for seed in 31 32; do timeout 1500 python tests/manual/fuzz_forcing.py 100 $seed 2>&1 | tail -3 > gpurun_out/r06/fuzz/forcing_$seed.txt; done
timeout 1200 python tests/manual/fuzz_forcing.py 60 33 dense 2>&1 | tail -3 > gpurun_out/r06/fuzz/forcing_33_dense.txt
for seed in 11 12 13; do timeout 1500 python tests/manual/fuzz_components.py 30 $seed 2>&1 | tail -3 > gpurun_out/r06/fuzz/components_$seed.txt; done
timeout 900 python tests/manual/fuzz_sigma.py 60 5 2>&1 | tail -3 > gpurun_out/r06/fuzz/sigma_5.txt
timeout 900 python tests/manual/fuzz_host_layer.py 2>&1 | tail -3 > gpurun_out/r06/fuzz/host_layer.txt
timeout 1500 bash tests/manual/fuzz_sharded.sh 12 200 2>&1 | tail -6 > gpurun_out/r06/fuzz/sharded.txt
tail -n 3 gpurun_out/r06/fuzz/*.txt | cut -c1-400
