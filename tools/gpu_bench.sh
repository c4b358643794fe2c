#!/bin/bash
# Run on the GPU box: bench + rocprofv3 kernel stats (summarised on the box; the raw trace db is too big to ship back).
# Usage: tools/gpu_bench.sh <tag> [bench args...]
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python bench.py "$@" > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
echo "bench rc=$?"; tail -3 $OUT/bench_$TAG.err; cat $OUT/bench_$TAG.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/prof_$TAG.log 2>&1
echo "rocprof rc=$?"
F=$(find /tmp/prof_$TAG -name "*_results.db" | head -1)
[ -n "$F" ] && python $R/tools/prof_summary.py "$F" "$TAG: rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline $*" > $OUT/${TAG}_kernel_stats.md
head -14 $OUT/${TAG}_kernel_stats.md
