set -e
mkdir -p gpurun_out/r4e
timeout -k 10 300 python -m pytest tests/test_gpu_groups.py -m gpu -x -q > gpurun_out/r4e/pytest_groups.log 2>&1 || { tail -30 gpurun_out/r4e/pytest_groups.log; exit 1; }
tail -3 gpurun_out/r4e/pytest_groups.log
timeout -k 10 200 python tools/fuzz_filter.py 1500 1200000 > gpurun_out/r4e/fuzz_tiled_rows.log 2>&1; tail -2 gpurun_out/r4e/fuzz_tiled_rows.log
bash tools/profile_final.sh > gpurun_out/profile_final.log 2>&1 || { tail -20 gpurun_out/profile_final.log; exit 1; }
tail -4 gpurun_out/profile_final.log
bash tools/profile_tiled.sh > gpurun_out/profile_tiled.log 2>&1 || { tail -20 gpurun_out/profile_tiled.log; exit 1; }
tail -8 gpurun_out/profile_tiled.log
