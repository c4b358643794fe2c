#!/bin/bash
# Round-5 call m: the driver's exact N=1 command on the closing tree
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r05m; mkdir -p $OUT; cd $R
( time timeout 170 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; tail -4 $OUT/bench_driver_cmd.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05m/bench_driver_cmd.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "fused", d["accum_fused"]["value"], "b24", d["batch_24"]["value"], "cpu", d["cpu_baseline"]["value"])
PY
