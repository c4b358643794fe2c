#!/bin/bash
# Round-4 validation on the GPU box (one gpurun call): the WHOLE -m gpu suite (no -x), smoke, a short bench.  Usage: tools/r04_check.sh <tag> [bench args]
TAG=${1:-r04a}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
( time timeout 2400 python -m pytest tests -m gpu -q -rf --durations=15 ) > $OUT/tests.log 2>&1; grep -E "passed|failed|error" $OUT/tests.log | tail -3
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
( time timeout 900 python bench.py --no-cpu-baseline --no-neighbours --no-k512 --no-loader "$@" ) > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; cut -c1-600 $OUT/bench.json
