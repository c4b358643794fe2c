#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04b; mkdir -p $OUT; cd $R
( time timeout 1200 python -m pytest tests/test_backward_gpu.py tests/test_dist_gpu.py -q -rf -k "model_grads or trainer_eager or rotating or reproducible or checkpoint or world1" ) > $OUT/tests.log 2>&1; grep -E "passed|failed|error" $OUT/tests.log | tail -3
( time timeout 600 python tools/probes/spread.py grads 6 ) > $OUT/spread_grads.md 2> $OUT/spread_grads.err; tail -25 $OUT/spread_grads.md
( time timeout 1500 python tools/probes/spread.py fulldepth 5 ) > $OUT/spread_fulldepth.md 2> $OUT/spread_fulldepth.err; tail -14 $OUT/spread_fulldepth.md
