cd /root/repo
for w in default 490 784 980 1176; do
  if [ $w = default ]; then unset GSFM_COL_WGS; else export GSFM_COL_WGS=$w; fi
  python bench.py --steps 5 --warmup 2 --cpu-baseline 0 --small-graphs 0 --tree-init 0 --sigma-pass 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('WGS=$w', round(d['ms_per_step'],3), d['cg_iterations_per_solve'], d['final_cost'], {k:round(v,1) for k,v in d['kernels_us'].items()})
"
done
