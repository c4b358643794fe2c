#!/bin/bash
# alignbyte check + leaf-stream A/B at 2 and 24 images + the WHOLE suite (run 1 of 2 on this tree)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04f; mkdir -p $OUT; cd $R
python tools/probes/targets_time.py 2>&1 | tail -6 > $OUT/targets_time.txt; cat $OUT/targets_time.txt
B="--no-cpu-baseline --no-neighbours --no-k512 --no-loader --no-fwd-only"
for v in "" "--leaf-stream"; do   # (when this script ran, the side stream was the default and the A/B flag was --no-leaf-stream)
  ( timeout 900 python bench.py $B --extra-batch 24 $v ) > $OUT/bench_ab$v.json 2> $OUT/bench_ab$v.err
  python - "$OUT/bench_ab$v.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "B=2: %.2f ms" % d["ms_per_step"], "B=24: %.2f ms" % d["batch_24"]["ms_per_step"])
PY
done
( time timeout 2400 python -m pytest tests -m gpu -q -rf --durations=8 ) > $OUT/tests.log 2>&1; grep -E "passed|failed|error" $OUT/tests.log | tail -3
