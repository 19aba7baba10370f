# tools/job_ab3.sh libA libB ...: headline A/B of each library against the first (128 spp, 5 rounds)
mkdir -p gpurun_out/ab3
first=$1; shift
for lib in "$@"; do
    timeout -k 10 200 python tools/ab.py $first $lib 128 5 2>&1 | grep -v amdgpu.ids | tail -3
done
