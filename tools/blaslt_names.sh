cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/bl; rocprofv3 --kernel-trace --stats -d /tmp/bl -o bl -- python $GRAFT_REPO_ROOT/tools/blaslt_names.py > /dev/null 2>&1
F=$(find /tmp/bl -name "*_results.db" | head -1)
python3 - "$F" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for n, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if "Cijk" in n: print(f"{avg:8.1f} us x{calls}  {n[:420]}")
PY
