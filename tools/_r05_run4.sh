mkdir -p gpurun_out/r05
O=gpurun_out/r05/run4.txt
: > $O
python - >> $O 2>&1 <<'PY'
import sys
sys.path.insert(0, "tests/manual")
import fuzz_forcing
fuzz_forcing.run(88, 9, only=[87, 1, 35], oracle_every=1)
fuzz_forcing.run(53, 6, only=[52], dense=True, oracle_every=1)
fuzz_forcing.run(15, 5, only=[14, 1], dense=True, oracle_every=1)
fuzz_forcing.run(95, 3, only=[38, 77, 94], oracle_every=1)
PY
python tools/r05_c4_components.py 2>&1 | grep -v amdgpu | tail -3 >> $O
python tools/_r05_run3.py 2>&1 | grep -v amdgpu | grep "^C5\|^t4000\|^C2" >> $O
timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_fullsize.py tests/test_gpu_round4.py -x -q -m gpu 2>&1 | tail -15 >> $O
