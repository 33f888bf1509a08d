# A/B: K2c with 512-lane workgroups (one position per lane: 104-124 VGPRs, 4 waves per SIMD) against the product's 256-lane form (two positions per lane: 216 VGPRs, 2 waves per SIMD)
# (build the second library first: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DGSFM_COLLIN_THREADS=512 -o globalsfmpy_amd/libgsfm_rot_t512.so.alt globalsfmpy_amd/csrc/gsfm_rot.hip -ldl)
set -x
cd "$(dirname "$0")/.."
L=globalsfmpy_amd/libgsfm_rot.so
cp $L /tmp/lib_default.so
for m in default t512 default t512; do
  if [ $m = t512 ]; then cp globalsfmpy_amd/libgsfm_rot_t512.so.alt $L; else cp /tmp/lib_default.so $L; fi
  python bench.py --steps 8 --warmup 2 --cpu-baseline 0 --small-graphs 0 --tree-init 0 > gpurun_out/k2c_$m.json 2> gpurun_out/k2c_$m.err
  python - <<P
import json
d=json.loads(open('gpurun_out/k2c_$m.json').read().strip().splitlines()[-1])
print('K2C=$m', d['ms_per_step'], d['cg_iterations_per_solve'], d['final_cost'], d['kernels_us'], d['roofline_other']['k_lin']['frac'])
P
done
cp /tmp/lib_default.so $L
