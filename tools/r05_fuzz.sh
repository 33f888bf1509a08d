mkdir -p gpurun_out/r05
for spec in "$@"; do
  set -- $(echo $spec | tr ':' ' ')
  timeout 1500 python tests/manual/fuzz_forcing.py $1 $2 $3 2>&1 | grep -v amdgpu > gpurun_out/r05/fuzz_forcing_seed$2${3:+_$3}.txt
  tail -1 gpurun_out/r05/fuzz_forcing_seed$2${3:+_$3}.txt
done
