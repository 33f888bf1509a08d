# round 6, review item 4: the measurement planes as 24 bytes per position (three quaternion components, kernels.hpp GSFM_QREL3) against the
# 32-byte quaternion (a second library built with -DGSFM_QREL3=0: tools/_ab/libgsfm_rot_q4.so); same box, alternating; C5
cd "$(dirname "$0")/.."
for m in q3 q4 q3 q4; do
  if [ $m = q4 ]; then export GSFM_ROT_LIB=$PWD/tools/_ab/libgsfm_rot_q4.so; else unset GSFM_ROT_LIB; fi
  python bench.py --steps 8 --warmup 2 --cpu-baseline 0 --small-graphs 0 --tree-init 0 > gpurun_out/r06/q_$m.json 2> gpurun_out/r06/q_$m.err
  python - <<P
import json
d=json.loads(open('gpurun_out/r06/q_$m.json').read().strip().splitlines()[-1])
o=d['roofline_other']
print('QREL=$m', 'ms_per_step %.3f' % d['ms_per_step'], d['cg_iterations_per_solve'], repr(d['final_cost']), {k: round(v,1) for k,v in d['kernels_us'].items()},
      'trial %.1f us  reweight %.1f us (8d frac %.3f)  full %.1f  s_only %.1f  | k_lin 8d frac %.3f own %.3f | sigma k1 %.1f/%.1f k2 %.1f/%.1f' % (
      1e3*o['k_cost_trial']['kernel_ms'], 1e3*o['k_cost_reweight']['kernel_ms'], o['k_cost_reweight']['frac_on_survey_8d_bytes'], 1e3*o['k_cost_full']['kernel_ms'], 1e3*o['k_cost_s_only']['kernel_ms'],
      o['k_lin']['frac_on_survey_8d_bytes'], o['k_lin']['frac'], 1e3*o['sigma_consensus_K6']['k1_fused']['kernel_ms'], 1e3*o['sigma_consensus_K6']['k1_plain']['kernel_ms'], 1e3*o['sigma_consensus_K6']['k2_fused']['kernel_ms'], 1e3*o['sigma_consensus_K6']['k2_plain']['kernel_ms']))
P
done
