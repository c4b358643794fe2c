#!/bin/bash
# Run on the GPU box: bench + rocprofv3 kernel stats.  Usage: tools/gpu_bench.sh <tag> [bench args...]
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python bench.py "$@" > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
echo "bench rc=$?"; tail -3 $OUT/bench_$TAG.err; cat $OUT/bench_$TAG.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o $TAG -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/prof_$TAG.log 2>&1
echo "rocprof rc=$?"
F=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1)
echo "stats file: $F"
[ -n "$F" ] && head -25 "$F"
