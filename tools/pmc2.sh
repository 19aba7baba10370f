#!/bin/bash
# PMC passes for the MFMA trace kernel (bench at spp 128)
export TMPDIR=/tmp
tag=$1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc_${tag}_a -- python3 bench.py --spp 128 --steps 1 --warmup 0 --cpu-seconds 0 > gpurun_out/pmc_${tag}_a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_${tag}_b -- python3 bench.py --spp 128 --steps 1 --warmup 0 --cpu-seconds 0 > gpurun_out/pmc_${tag}_b.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("pmc_${tag}_a","pmc_${tag}_b"):
    for f in glob.glob("gpurun_out/"+d+"/*/*_counter_collection.csv"):
        agg = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "k_trace" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
        for k,v in sorted(agg.items()): print("%-28s %.4g" % (k, v))
tail = open("gpurun_out/pmc_${tag}_a.log").read().strip().splitlines()[-1]
print(tail[:300])
PY
