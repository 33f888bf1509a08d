# round-6 artefacts (one box call): kernel trace of the benchmark command, PMC traffic of the hot kernels, SQ counters of the sweep
# specialisations that sit below 0.60 of the HBM roofline (what binds them), the plain bench line
set -x
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r06_final
rm -rf "$OUT"; mkdir -p "$OUT"
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o r06 -- python "$ROOT/bench.py" --steps 5 --warmup 2 --cpu-baseline 0 --small-graphs 0 --tree-init 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err")
DB=$(find "$OUT/trace" -name '*.db' | head -1)
python profiles/summarize_rocpd.py "$DB" "round 6 (final tree): python bench.py --steps 5 --warmup 2 --cpu-baseline 0 --small-graphs 0 --tree-init 0, C5, 1x MI355X" > "$OUT/r06_kernel_stats.txt"
find "$OUT/trace" -name '*.db' -size +40M -delete
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc_$c" -o pmc -- python "$ROOT/tools/r04_pmc_probe.py" > "$OUT/pmc_$c.log" 2>&1)
done
python profiles/make_pmc_traffic.py "$(find $OUT/pmc_FETCH_SIZE -name '*.db' | head -1)" "$(find $OUT/pmc_WRITE_SIZE -name '*.db' | head -1)" "$OUT/r06_pmc_traffic.json" > "$OUT/pmc_traffic.log" 2>&1
python profiles/summarize_pmc.py "$(find $OUT/pmc_FETCH_SIZE -name '*.db' | head -1)" "$(find $OUT/pmc_WRITE_SIZE -name '*.db' | head -1)" > "$OUT/r06_pmc_hbm_traffic.txt" 2>&1
find "$OUT" -name '*.db' -size +30M -delete
mkdir -p "$OUT/sq"
for grp in "sq_a:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "sq_b:SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  name=${grp%%:*}; ctr=${grp#*:}
  d=$OUT/sq/$name; mkdir -p "$d"
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc $ctr -d "$d" -o pmc -- python "$ROOT/tools/r04_pmc_probe.py" > "$d/run.log" 2>&1)
  f=$(find "$d" -name '*.db' | head -1)
  if [ -n "$f" ]; then python profiles/summarize_pmc_db.py "$f" k_cost k_lin k_mv > "$OUT/sq/$name.txt" 2>&1; find "$d" -name '*.db' -size +30M -delete; fi
done
cp "$OUT/r06_pmc_traffic.json" profiles/r06_pmc_traffic.json   # so that the bench run below reports it (same kernel sources)
timeout 900 python bench.py > "$OUT/r06_bench.json" 2> "$OUT/r06_bench.err"
head -5 "$OUT/r06_kernel_stats.txt"; head -c 600 "$OUT/r06_bench.json"
