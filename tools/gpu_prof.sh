#!/bin/bash
# rocprofv3 kernel-trace summary of a bench.py invocation.  Usage: tools/gpu_prof.sh <tag> [bench args...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG -- python $R/bench.py --no-cpu-baseline "$@" > $OUT/prof_$TAG.log 2>&1
echo "rocprof rc=$?"
F=$(find /tmp/prof_$TAG -name "*_results.db" | head -1)
[ -n "$F" ] && python $R/tools/prof_summary.py "$F" "$TAG: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline $*" > $OUT/${TAG}_kernel_stats.md
head -50 $OUT/${TAG}_kernel_stats.md
tail -2 $OUT/prof_$TAG.log | cut -c1-600
