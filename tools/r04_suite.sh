#!/bin/bash
# The WHOLE -m gpu suite + smoke on the final tree.  usage: tools/r04_suite.sh <tag>
TAG=${1:-final1}; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04suite_$TAG; mkdir -p $OUT; cd $R
git rev-parse HEAD > $OUT/head.txt 2>/dev/null
( time timeout 2400 python -m pytest tests -m gpu -q -rf --durations=6 ) > $OUT/tests.log 2>&1; grep -E "passed|failed|error" $OUT/tests.log | tail -3
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log; grep "flat-1e-3" $OUT/smoke.log | cut -c1-200
