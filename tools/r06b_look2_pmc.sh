# SQ counters of the exact step's kernels on Madrid (k_chol_look2, k_chol_back_group, k_chol_back_update): what a launch issues and waits for
cd /root/repo; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r06b_pmc; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQ_[A-Z_]*MFMA[A-Z_]*\|SQ_LDS_BANK_CONFLICT\|SQ_LDS_[A-Z_]*\|SQ_INSTS_VALU[A-Z_]*" | sort -u > $OUT/avail.txt
for grp in "a:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "b:SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  name=${grp%%:*}; ctr=${grp#*:}
  d=$OUT/$name; mkdir -p $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $d -o pmc -- python /root/repo/tools/r06b_madrid_trace.py magsac > $d/run.log 2>&1)
  f=$(find $d -name '*.db' | head -1)
  if [ -n "$f" ]; then python profiles/summarize_pmc_db.py "$f" k_chol > $OUT/sq_$name.txt 2>&1; fi
  find $d -name '*.db' -delete
done
cat $OUT/avail.txt | tr '\n' ' '; echo; cat $OUT/sq_a.txt $OUT/sq_b.txt | cut -c1-150
