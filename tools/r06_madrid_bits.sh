# Which cluster of its bimodal outcome does Madrid / MAGSAC land in under a given build?  (62 or 63 LM iterations, 2.0e-4 rad apart; the answer moves with last bits.)
cd "$(dirname "$0")/.."
for lib in default tools/_ab/libgsfm_rot_q4.so tools/_ab/libgsfm_rot_r05.so; do
  if [ $lib = default ]; then unset GSFM_ROT_LIB; else export GSFM_ROT_LIB=$PWD/$lib; fi
  python bench.py --cams 20000 --edges 400000 --steps 1 --warmup 1 --tree-init 0 --sigma-pass 0 --cpu-baseline 1 --cpu-single-cams 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
m=d['small_graph_ms']['C1_madrid_394_views_23784_edges_cov_magsac']
print('$lib', 'Madrid MAGSAC: %d LM iterations (oracle %d), %.2e rad from the oracle, %.2f ms' % (m['lm_iterations'], m['cpu_oracle_lm_iterations'], m['device_vs_cpu_mean_rad'], m['ms']))"
done
