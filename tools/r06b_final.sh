# last session of round 6: the artefacts that changed with the exact step's new factorisation -- the bench line, the kernel traces of the
# latency-bound configurations, smoke, and the randomised suites that exercise the dense / component steps
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r06b_final; rm -rf $OUT; mkdir -p $OUT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06b_bench.json 2> $OUT/r06b_bench.err
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
bash tools/r06b_latency_traces.sh > $OUT/latency.log 2>&1
mkdir -p $OUT/latency_after; cp gpurun_out/r06b_latency/*_gaps.txt $OUT/latency_after/
bash tools/r06b_c4_timeline.sh > $OUT/c4_timeline.log 2>&1
cp gpurun_out/r06b_c4_timeline/spans.txt $OUT/latency_after/c4_iteration_spans.txt
timeout 700 python tests/manual/fuzz_differential.py 600 104 2>&1 | tail -4 > $OUT/fuzz_differential_104.txt
for seed in 14 15; do timeout 600 python tests/manual/fuzz_components.py 30 $seed 2>&1 | tail -3 > $OUT/fuzz_components_$seed.txt; done
tail -n 4 $OUT/fuzz_*.txt $OUT/smoke.txt | cut -c1-300; head -c 400 $OUT/r06b_bench.json
