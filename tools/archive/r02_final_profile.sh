# round-2 final artefacts: kernel trace of the benchmark command + the plain bench line (one box call)
set -x
cd /root/repo
export TMPDIR=/tmp
rm -rf gpurun_out/prof_r02f
(cd /tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r02f -o r02f -- python /root/repo/bench.py --steps 5 --warmup 2 --cpu-baseline 0 --small-graphs 0 > /root/repo/gpurun_out/r02f_bench_under_rocprof.json 2> /root/repo/gpurun_out/r02f_rocprof.err)
DB=$(find gpurun_out/prof_r02f -name '*.db' | head -1)
python profiles/summarize_rocpd.py "$DB" "round 2 (f, final tree): python bench.py --steps 5 --warmup 2 --cpu-baseline 0 --small-graphs 0, C5, 1x MI355X" > gpurun_out/r02_f_kernel_stats.txt
find gpurun_out/prof_r02f -name '*.db' -size +40M -delete
timeout 600 python bench.py > gpurun_out/r02_f_bench.json 2> gpurun_out/r02_f_bench.err
head -5 gpurun_out/r02_f_kernel_stats.txt; head -c 400 gpurun_out/r02_f_bench.json
