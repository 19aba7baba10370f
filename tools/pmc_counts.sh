#!/bin/bash
# VALU / MFMA instruction counts of the bench kernel at spp 128 for a given library: tools/pmc_counts.sh <tag> [lib path]
export TMPDIR=/tmp
tag=$1
if [ -n "$2" ]; then export RT3_LIB_PATH=$2; fi
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS --output-format csv -d gpurun_out/pmc_${tag} -- python3 bench.py --spp 128 --steps 1 --warmup 0 --cpu-seconds 0 > gpurun_out/pmc_${tag}.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_${tag}/*/*_counter_collection.csv"):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
    print("${tag}: " + "  ".join("%s %.4g" % (k, v) for k, v in sorted(agg.items())))
PY
