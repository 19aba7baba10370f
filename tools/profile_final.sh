#!/bin/bash
# Produces the files committed under profiles/ for the current build (run on the GPU box from the repo root):
#   gpurun_out/final/bench_line.json, kernel_stats.csv, pmc_k_trace.json
# rocprofv3 gets `python3 bench.py ...` directly after `--`; counters are collected in their own passes (no trace domains).
export TMPDIR=/tmp
out=gpurun_out/final
export RT3_PROFILE_TAG=${RT3_PROFILE_TAG:-r03}
if [ -n "$RT3_TRACE_ONLY" ]; then mkdir -p $out; else
rm -rf $out && mkdir -p $out
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_VALU_MFMA_COEXEC_CYCLES"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d $out/pmc$i -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-extra > $out/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $out/pmc$i.log; exit 1; }
    echo "pmc pass $i done"
done
python3 - <<'PY'
import csv, glob, json, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/final/pmc*/*/*_counter_collection.csv")):
    per = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Kernel_Name"]:
            per[r["Counter_Name"]][0] += float(r["Counter_Value"])
            per[r["Counter_Name"]][1].add(r["Dispatch_Id"])
    for k, (v, d) in per.items():
        agg[k] = {"sum_over_dispatches": v, "dispatches": len(d)}
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "bench.py"); B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
agg["_source_fingerprint"] = B.source_fingerprint()                 # bench.py reports these counters only for the sources they were collected on
agg["_command"] = "rocprofv3 --pmc <set> -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-extra (one pass per counter set); kernels matching k_trace"
json.dump(agg, open("gpurun_out/final/pmc_k_trace.json", "w"), indent=1)
print(json.dumps({k: v["sum_over_dispatches"] for k, v in agg.items() if isinstance(v, dict)}))
PY
# the counters first, so that the traced run's JSON line reports them (bench.py reads profiles/${RT3_PROFILE_TAG}_bench_pmc_k_trace.json and
# checks its source fingerprint)
cp $out/pmc_k_trace.json profiles/${RT3_PROFILE_TAG}_bench_pmc_k_trace.json
fi   # RT3_TRACE_ONLY=1: only the traced run below, with the committed counter profiles (its JSON line then carries the tiled kernels' counters too)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py > $out/bench_stdout.log 2>$out/bench_stderr.log || exit 1
echo "trace pass done"
grep '^{' $out/bench_stdout.log | tail -1 > $out/bench_line.json
cp $(ls $out/trace/*/*_kernel_stats.csv | head -1) $out/kernel_stats.csv
head -5 $out/kernel_stats.csv
cat $out/bench_line.json | cut -c1-400
