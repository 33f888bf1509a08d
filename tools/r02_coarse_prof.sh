cd /root/repo; export TMPDIR=/tmp
rm -rf gpurun_out/prof_coarse
(cd /tmp && GSFM_PCG_COARSE=64 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_coarse -o c -- python /root/repo/tools/r02_coarse_one.py > /dev/null 2>&1)
DB=$(find gpurun_out/prof_coarse -name '*.db' | head -1)
python profiles/summarize_rocpd.py "$DB" "coherent 100k / 2M, 64 aggregates" | head -16
find gpurun_out/prof_coarse -name '*.db' -delete
