mkdir -p gpurun_out/r05
O=gpurun_out/r05/run6.txt
: > $O
python tools/r05_c4_components.py 2>&1 | grep -v amdgpu | tail -2 >> $O
python tools/_r05_run3.py 2>&1 | grep -v amdgpu | grep "^C5\|^t4000\|^C2" >> $O
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -40 >> $O
