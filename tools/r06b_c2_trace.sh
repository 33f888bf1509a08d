cd /root/repo; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r06b_c2; rm -rf $OUT; mkdir -p $OUT
for w in gm magsac; do
  (cd /tmp && rocprofv3 --kernel-trace -d $OUT/trace_$w -o tr -- python /root/repo/tools/r06b_c2_trace.py $w > $OUT/c2_$w.log 2>&1)
  DB=$(find $OUT/trace_$w -name '*.db' | head -1)
  python tools/r04b_solve_gaps.py $DB > $OUT/c2_${w}_gaps.txt 2>&1
  grep "^C2" $OUT/c2_$w.log | tail -2; cut -c1-130 $OUT/c2_${w}_gaps.txt | head -42
done
find $OUT -name '*.db' -delete
