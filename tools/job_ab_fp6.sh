set -e
mkdir -p gpurun_out/r4g
timeout -k 10 300 python tools/ab.py build_ab/librt3hip_alignbit.so raytracer-3_amd/librt3hip.so 128 5 > gpurun_out/r4g/ab_headline.log 2>&1 || { tail -20 gpurun_out/r4g/ab_headline.log; exit 1; }
tail -4 gpurun_out/r4g/ab_headline.log
timeout -k 10 300 python tools/ab_cfg.py build_ab/librt3hip_alignbit.so raytracer-3_amd/librt3hip.so > gpurun_out/r4g/ab_cfg.log 2>&1 || { tail -20 gpurun_out/r4g/ab_cfg.log; exit 1; }
tail -4 gpurun_out/r4g/ab_cfg.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_bench_ranks.py --deselect tests/test_bench_contract.py > gpurun_out/r4g/pytest.log 2>&1 || { tail -30 gpurun_out/r4g/pytest.log; exit 1; }
tail -3 gpurun_out/r4g/pytest.log
timeout -k 10 300 python tools/fuzz_filter.py 2000 1300000 > gpurun_out/r4g/fuzz.log 2>&1; tail -2 gpurun_out/r4g/fuzz.log
