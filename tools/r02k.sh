#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_dist_gpu.py -q -x > $OUT/r02k_tests.txt 2>&1; tail -6 $OUT/r02k_tests.txt
timeout 600 python tools/hbm_kernels.py 24 > $OUT/r02k_hbm_b24.json 2> $OUT/r02k_hbm.err; tail -2 $OUT/r02k_hbm.err
REPS=3 bash tools/pmc_traffic.sh r02k_hbm python $R/tools/hbm_kernels.py 24
bash tools/pmc_traffic.sh r02k_bench2 python $R/bench.py --batch 2 --extra-batch 0 --no-cpu-baseline --steps 3 --warmup 1 --no-graph
