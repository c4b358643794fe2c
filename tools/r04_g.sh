#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04g; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_sg
rocprofv3 --kernel-trace -d /tmp/prof_sg -o sg -- python $R/tools/probes/small_gemm_variants.py > $OUT/groups.txt 2> $OUT/err.txt
F=$(find /tmp/prof_sg -name "*_results.db" | head -1)
cd $R; python tools/probes/small_gemm_report.py "$F" $OUT/groups.txt > $OUT/report.txt 2>&1; cat $OUT/report.txt; grep -c GROUP $OUT/groups.txt; tail -3 $OUT/err.txt
