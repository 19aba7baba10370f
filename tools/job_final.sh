# tools/job_final.sh: the round's closing checks on the GPU box (repo root): whole GPU suite, smoke(), the default bench line, 40 repeats of the headline
# frame (lost candidates would show as differing pixels), 400 fuzz seeds judged by the CPU oracle
set -e
out=gpurun_out/final_checks
mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1 || { tail -30 $out/pytest_gpu.log; exit 1; }
tail -2 $out/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
timeout -k 10 400 python bench.py > $out/bench.log 2>$out/bench.err || { tail -20 $out/bench.err; exit 1; }
cut -c1-300 $out/bench.log
timeout -k 10 300 python tools/repeat_headline.py 40 > $out/repeat_headline.log 2>&1 || true
tail -2 $out/repeat_headline.log
timeout -k 10 500 python tools/fuzz_arbiter.py $(seq 2000000 2000399) > $out/fuzz_arbiter.log 2>&1 || true
tail -3 $out/fuzz_arbiter.log
