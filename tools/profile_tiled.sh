#!/bin/bash
# rocprofv3 evidence for the kernels bench.py's extra_workloads time (run on the GPU box from the repo root, after tools/profile_final.sh):
#   gpurun_out/final/tiled_<name>_kernel_stats.csv   --kernel-trace --stats of `python3 tools/run_config.py <name> <spp>`
#   gpurun_out/final/tiled_pmc.json                   SQ / GRBM counters per kernel from separate --pmc passes (no trace domains)
export TMPDIR=/tmp
out=gpurun_out/final
mkdir -p $out
declare -A SPP=( [4]=16 [5]=32 [r]=1 )
for c in 4 5 r; do
    spp=${SPP[$c]}
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tiled_trace_$c -- python3 tools/run_config.py $c $spp > $out/tiled_$c.log 2>&1 || { tail -3 $out/tiled_$c.log; exit 1; }
    cp $(ls $out/tiled_trace_$c/*/*_kernel_stats.csv | head -1) $out/tiled_${c}_kernel_stats.csv
    tail -1 $out/tiled_$c.log
    i=0
    for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
               "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_VALU_MFMA_COEXEC_CYCLES"; do
        i=$((i+1))
        timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out/tiled_pmc_${c}_$i -- python3 tools/run_config.py $c $spp > $out/tiled_pmc_${c}_$i.log 2>&1 || { echo "pmc pass $c/$i failed"; tail -3 $out/tiled_pmc_${c}_$i.log; exit 1; }
    done
    echo "config $c profiled"
done
python3 - <<'PY'
import csv, glob, json, collections
res = collections.OrderedDict()
for c, pat in (("4", "k_trace_mfma_tiled"), ("5", "k_trace_mfma_tiled"), ("r", "k_mode_r_mfma")):
    agg = collections.OrderedDict()
    for f in sorted(glob.glob("gpurun_out/final/tiled_pmc_%s_*/*/*_counter_collection.csv" % c)):
        per = collections.defaultdict(lambda: [0.0, set()])
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                per[r["Counter_Name"]][0] += float(r["Counter_Value"])
                per[r["Counter_Name"]][1].add(r["Dispatch_Id"])
        for k, (v, d) in per.items():
            agg[k] = {"sum_over_dispatches": v, "dispatches": len(d)}
    res["config_" + c] = agg
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "bench.py"); B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
res["_source_fingerprint"] = B.source_fingerprint()
res["_command"] = "rocprofv3 --pmc <set> -- python3 tools/run_config.py {4 16 | 5 32 | r 1}; kernels matching k_trace_mfma_tiled / k_mode_r_mfma"
json.dump(res, open("gpurun_out/final/tiled_pmc.json", "w"), indent=1)
for c, agg in res.items():
    if isinstance(agg, dict) and "SQ_VALU_MFMA_BUSY_CYCLES" in agg and "GRBM_GUI_ACTIVE" in agg:
        busy = agg["SQ_VALU_MFMA_BUSY_CYCLES"]["sum_over_dispatches"] / 1024.0 / (agg["GRBM_GUI_ACTIVE"]["sum_over_dispatches"] / 8.0)
        print(c, "matrix pipe busy %.3f of the cycles" % busy)
PY
