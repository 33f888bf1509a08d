set -x
cd /root/repo
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r02_final_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02_final_smoke.txt 2>&1
timeout 600 python bench.py > gpurun_out/r02_final_bench.json 2> gpurun_out/r02_final_bench.err
tail -3 gpurun_out/r02_final_gpu_tests.txt; tail -2 gpurun_out/r02_final_smoke.txt; head -c 600 gpurun_out/r02_final_bench.json
