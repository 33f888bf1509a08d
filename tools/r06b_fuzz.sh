# last session of round 6: the randomised suites once more on the final tree (new seeds), one box call
cd "$(dirname "$0")/.."
O=gpurun_out/r06b_fuzz; rm -rf $O; mkdir -p $O
for seed in 105 106; do timeout 700 python tests/manual/fuzz_differential.py 600 $seed 2>&1 | tail -4 > $O/differential_$seed.txt; done
timeout 900 python tests/manual/fuzz_forcing.py 100 34 2>&1 | tail -3 > $O/forcing_34.txt
timeout 700 python tests/manual/fuzz_forcing.py 60 35 dense 2>&1 | tail -3 > $O/forcing_35_dense.txt
timeout 600 python tests/manual/fuzz_components.py 30 16 2>&1 | tail -3 > $O/components_16.txt
timeout 600 python tests/manual/fuzz_sigma.py 60 6 2>&1 | tail -3 > $O/sigma_6.txt
timeout 600 python tests/manual/fuzz_host_layer.py 2>&1 | tail -3 > $O/host_layer.txt
timeout 900 bash tests/manual/fuzz_sharded.sh 6 200 2>&1 | tail -6 > $O/sharded.txt
tail -n 3 $O/*.txt | cut -c1-300
