set -x
python -m pytest tests/test_gpu_round5.py -m gpu -x -q -k two_byte > gpurun_out/k16_test.log 2>&1; tail -15 gpurun_out/k16_test.log
for m in 0 1 0 1; do
  GSFM_K3C_K16=$m GSFM_CREATE_TIMING=1 python bench.py --steps 8 --warmup 2 --cpu-baseline 0 --small-graphs 0 --tree-init 0 > gpurun_out/k16_bench_$m.json 2> gpurun_out/k16_bench_$m.err
  grep "2-byte" gpurun_out/k16_bench_$m.err | head -2
  python - <<P
import json
d=json.loads(open('gpurun_out/k16_bench_$m.json').read().strip().splitlines()[-1])
print('K16=$m', d['ms_per_step'], d['value'], d['cg_iterations_per_solve'], d['final_cost'], d['roofline']['kernel_ms'], d['roofline']['algorithmic_bytes_per_launch'], d['roofline']['frac'], d['roofline'].get('frac_on_survey_8d_bytes'), d['kernels_us'])
P
done
