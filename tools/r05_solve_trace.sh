cd /root/repo; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05_solve_trace; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace -d $OUT/trace -o tr -- python /root/repo/tools/r04b_solve_trace.py > $OUT/run.log 2>&1)
DB=$(find $OUT/trace -name '*.db' | head -1)
python tools/r04b_solve_gaps.py $DB > $OUT/gaps.txt 2>&1
find $OUT -name '*.db' -delete
cat $OUT/gaps.txt
