# tools/job_validate.sh <tag>: A/B against build_ab/librt3hip_alignbit.so, the whole GPU suite, fuzz (run on the GPU box from the repo root)
set -e
tag=${1:-val}
out=gpurun_out/$tag
mkdir -p $out
{ timeout -k 10 200 python tools/ab.py build_ab/librt3hip_alignbit.so raytracer-3_amd/librt3hip.so 128 5; timeout -k 10 300 python tools/ab_cfg.py build_ab/librt3hip_alignbit.so raytracer-3_amd/librt3hip.so; } 2>&1 | grep -v amdgpu.ids > $out/ab_fp6_decode.log
tail -5 $out/ab_fp6_decode.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 ${FUZZ_SECONDS:-420} python tools/fuzz_filter.py ${FUZZ_SCENES:-6000} ${FUZZ_SEED:-1400000} > $out/fuzz.log 2>&1 || true
tail -2 $out/fuzz.log
