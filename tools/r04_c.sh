#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04c; mkdir -p $OUT; cd $R
B="--no-cpu-baseline --no-neighbours --no-k512 --no-loader --extra-batch 0 --no-fwd-only"
( timeout 600 python bench.py $B ) > $OUT/bench_leaf.json 2> $OUT/bench_leaf.err; cut -c1-300 $OUT/bench_leaf.json; tail -2 $OUT/bench_leaf.err
( timeout 600 python bench.py $B --leaf-stream ) > $OUT/bench_withleaf.json 2> $OUT/bench_withleaf.err; cut -c1-300 $OUT/bench_withleaf.json   # (when this script ran, the side stream was the default and the flag was --no-leaf-stream)
( time timeout 1500 python -m pytest tests/test_backward_gpu.py tests/test_dist_gpu.py -q -rf -k "model_grads or trainer_eager or rotating or reproducible or checkpoint or world1 or two_ranks" ) > $OUT/tests.log 2>&1; grep -E "passed|failed|error" $OUT/tests.log | tail -3
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace --stats -d /tmp/prof_tl -o tl -- python $R/bench.py $B --steps 4 --warmup 2 > $OUT/prof_tl.log 2>&1
F=$(find /tmp/prof_tl -name "*_results.db" | head -1)
python $R/tools/timeline.py "$F" > $OUT/step_timeline.csv 2> $OUT/step_timeline.err; tail -3 $OUT/step_timeline.err; wc -l $OUT/step_timeline.csv
python $R/tools/prof_summary.py "$F" "r04c b2 graph" > $OUT/r04c_b2_kernel_stats.md
