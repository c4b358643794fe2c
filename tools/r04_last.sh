#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04last; mkdir -p $OUT; cd $R
bash tools/r04_suite.sh ${1:-final3} | tail -6
( time python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04last/bench_default.json").read().strip().splitlines()[-1])
print("B=2 %.2f img/s %.2f ms frac %.3f traffic %.1f MB; fwd %.1f; B=24 %.2f img/s frac %.3f; k512 %.2f; loader %.2f (input %.2f ms); cpu %.4f img/s (%d cores, %.0f s)" % (
    d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"] / 1e6, d["fwd_only"]["value"], d["batch_24"]["value"], d["batch_24"]["roofline"]["frac"],
    d["batch_2_k512"]["value"], d["loader_in_loop"]["value"], d["loader_in_loop"]["input_ms"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["cpu_seconds"]))
PY
