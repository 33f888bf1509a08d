mkdir -p gpurun_out/r05
O=gpurun_out/r05/run5.txt
: > $O
python - >> $O 2>&1 <<'PY'
import sys
sys.path.insert(0, "tests/manual")
import fuzz_forcing
fuzz_forcing.run(53, 6, only=[52], dense=True, oracle_every=1)
fuzz_forcing.run(15, 5, only=[14, 1], dense=True, oracle_every=1)
PY
python tools/r05_c4_components.py 2>&1 | grep -v amdgpu | tail -3 >> $O
python tools/_r05_run3.py 2>&1 | grep -v amdgpu | grep "^C5\|^t4000\|^C2" >> $O
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05/c4prof -- python $GRAFT_REPO_ROOT/tools/r04b_c4_trace.py > $GRAFT_REPO_ROOT/gpurun_out/r05/c4prof.log 2>&1; cd $GRAFT_REPO_ROOT
find gpurun_out/r05/c4prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -25 {}' >> $O
timeout 1500 python -m pytest tests/test_gpu_round5.py tests/test_gpu_fullsize.py tests/test_gpu_round4.py tests/test_gpu_direct_solve.py tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -25 >> $O
