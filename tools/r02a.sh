#!/bin/bash
# round-2 first GPU call: graph-capture probe, eager B=2 baseline + kernel profile, GEMM table at the B=2 shapes
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python tools/graph_probe.py > $OUT/r02a_graph_probe.txt 2>&1; tail -3 $OUT/r02a_graph_probe.txt
bash tools/gpu_bench.sh r02a_b2 --batch 2 --steps 10 --warmup 3 --no-cpu-baseline
GEMM_SET=b2 WITH_TORCH=1 python tools/gemm_bench.py 2,8 > $OUT/r02a_gemm_b2.txt 2>&1; cat $OUT/r02a_gemm_b2.txt
