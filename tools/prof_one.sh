#!/bin/bash
# rocprofv3 kernel-trace of one tool invocation, kernel rows matching a regex: tools/prof_one.sh <tag> <regex> <cmd...>
TAG=$1; RE=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/p1_$TAG -o $TAG -- "$@" > /dev/null 2>&1
F=$(find /tmp/p1_$TAG -name "*_results.db" | head -1)
python3 - "$F" "$RE" "$TAG" <<'PY'
import sqlite3, sys, re
cur = sqlite3.connect(sys.argv[1]).cursor()
for n, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if re.search(sys.argv[2], n):
        print(f"{sys.argv[3]:10s} {n[:70]:70s} calls {calls:5d} avg {avg:8.2f} us")
PY
