# round-3 artefacts (one box call): kernel trace of the benchmark command, PMC traffic of the hot kernels, the plain bench line
set -x
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r03_final
rm -rf "$OUT"; mkdir -p "$OUT"
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o r03 -- python "$ROOT/bench.py" --steps 5 --warmup 2 --cpu-baseline 0 --small-graphs 0 --tree-init 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err")
DB=$(find "$OUT/trace" -name '*.db' | head -1)
python profiles/summarize_rocpd.py "$DB" "round 3 (final tree): python bench.py --steps 5 --warmup 2 --cpu-baseline 0 --small-graphs 0 --tree-init 0, C5, 1x MI355X" > "$OUT/r03_kernel_stats.txt"
find "$OUT/trace" -name '*.db' -size +40M -delete
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc_$c" -o pmc -- python "$ROOT/tools/r03_k3_probe.py" > "$OUT/pmc_$c.log" 2>&1)
done
python profiles/make_pmc_traffic.py "$(find $OUT/pmc_FETCH_SIZE -name '*.db' | head -1)" "$(find $OUT/pmc_WRITE_SIZE -name '*.db' | head -1)" "$OUT/r03_pmc_traffic.json" > "$OUT/pmc_traffic.log" 2>&1
python profiles/summarize_pmc.py "$(find $OUT/pmc_FETCH_SIZE -name '*.db' | head -1)" "$(find $OUT/pmc_WRITE_SIZE -name '*.db' | head -1)" > "$OUT/r03_pmc_hbm_traffic.txt" 2>&1
find "$OUT" -name '*.db' -size +30M -delete
K3_PMC_OUT="$OUT/k3c_pmc" K3_PMC_GROUPS="tcp_req tcp_stall tcc_hit tcc_ea sq_a" bash tools/archive/r03_k3_pmc.sh > "$OUT/k3c_pmc.log" 2>&1
cp "$OUT/r03_pmc_traffic.json" profiles/r03_pmc_traffic.json   # so that the bench run below reports it (same kernel sources)
timeout 900 python bench.py > "$OUT/r03_bench.json" 2> "$OUT/r03_bench.err"
head -5 "$OUT/r03_kernel_stats.txt"; head -c 600 "$OUT/r03_bench.json"
