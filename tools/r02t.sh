#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/bl; rocprofv3 --kernel-trace --stats -d /tmp/bl -o bl --output-format csv -- python tools/blaslt_names.py > /tmp/bl.log 2>&1
f=$(find /tmp/bl -name "*kernel_stats.csv" | head -1)
python - "$f" > gpurun_out/r02t_blaslt.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Calls"], r["AverageNs"], r["Name"][:400])
PY
cat gpurun_out/r02t_blaslt.txt | head -40
