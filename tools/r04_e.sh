#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04e; mkdir -p $OUT; cd $R
( time timeout 1200 python -m pytest tests/test_targets_gpu.py tests/test_backward_gpu.py -q -rf -k "targets or dense or lora_paths or model_grads_lora or reproducible or rotating" ) > $OUT/tests.log 2>&1; grep -E "passed|failed|error" $OUT/tests.log | tail -3
( timeout 900 python bench.py --no-cpu-baseline --no-neighbours --no-k512 --extra-batch 0 --no-fwd-only ) > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04e/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("launches_per_micro_step"), d.get("loader_in_loop"))
PY
