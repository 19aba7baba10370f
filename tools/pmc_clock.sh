#!/bin/bash
# Clock and unit occupancy of the bench kernel for a given library: tools/pmc_clock.sh <tag> [lib path]
#   GRBM_GUI_ACTIVE / 8 XCDs / kernel time = the clock the kernel ran at; SQ_VALU_MFMA_BUSY_CYCLES / 4 SIMDs / SQ_BUSY_CYCLES...
export TMPDIR=/tmp
tag=$1
if [ -n "$2" ]; then export RT3_LIB_PATH=$2; fi
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmcclk_${tag} -- python3 bench.py --spp 128 --steps 1 --warmup 0 --cpu-seconds 0 --no-extra > gpurun_out/pmcclk_${tag}.log 2>&1
python3 - <<PY
import csv, glob, collections, json
ms = None
for ln in open("gpurun_out/pmcclk_${tag}.log"):
    if ln.startswith("{"):
        ms = json.loads(ln)["roofline"]["kernel_ms"]
for f in glob.glob("gpurun_out/pmcclk_${tag}/*/*_counter_collection.csv"):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
    print("${tag}: kernel %.3f ms  " % ms + "  ".join("%s %.4g" % (k, v) for k, v in sorted(agg.items())))
    if "GRBM_GUI_ACTIVE" in agg:
        cyc = agg["GRBM_GUI_ACTIVE"] / 8.0
        print("${tag}: clock %.3f GHz; matrix pipe busy %.3f of the SIMD cycles; VALU instructions x 4 / SIMD cycles %.3f" % (
            cyc / (ms * 1e6), agg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc, (agg["SQ_INSTS_VALU"] - agg["SQ_INSTS_MFMA"]) * 4.0 / (cyc * 1024.0)))
PY
