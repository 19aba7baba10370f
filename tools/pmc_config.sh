#!/bin/bash
# PMC counters of the trace kernel on a config shape: tools/pmc_config.sh 4|5 [spp]
export TMPDIR=/tmp
c=$1; spp=${2:-2}
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
    d=gpurun_out/pmc_cfg${c}_$(echo $set | cut -c1-12 | tr ' ' '_')
    timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $d -- python3 tools/run_config.py $c $spp > $d.log 2>&1 || { tail -3 $d.log; exit 1; }
    python3 - <<PY
import csv, glob, collections
for f in glob.glob("$d/*/*_counter_collection.csv"):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
    print("  ".join("%s %.4g" % (k, v) for k, v in sorted(agg.items())))
PY
    tail -1 $d.log
done
