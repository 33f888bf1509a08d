# Kernel traces of the two latency-bound configurations (C1 Madrid from its spanning-tree start, C4 as bench.py builds it) on the final tree
cd /root/repo; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r06b_latency; rm -rf $OUT; mkdir -p $OUT
for w in magsac softl1; do
  (cd /tmp && rocprofv3 --kernel-trace -d $OUT/trace_$w -o tr -- python /root/repo/tools/r06b_madrid_trace.py $w > $OUT/madrid_$w.log 2>&1)
  DB=$(find $OUT/trace_$w -name '*.db' | head -1)
  python tools/r04b_solve_gaps.py $DB > $OUT/madrid_${w}_gaps.txt 2>&1
done
(cd /tmp && rocprofv3 --kernel-trace -d $OUT/trace_c4 -o tr -- python /root/repo/tools/r05_c4_trace.py > $OUT/c4.log 2>&1)
DB=$(find $OUT/trace_c4 -name '*.db' | head -1)
python tools/r04b_solve_gaps.py $DB > $OUT/c4_gaps.txt 2>&1
find $OUT -name '*.db' -delete
cat $OUT/madrid_magsac.log $OUT/madrid_magsac_gaps.txt $OUT/c4.log $OUT/c4_gaps.txt | cut -c1-150
