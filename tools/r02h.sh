#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_backward_gpu.py -q > $OUT/r02h_tests.txt 2>&1; tail -8 $OUT/r02h_tests.txt
timeout 900 python bench.py --steps 20 --warmup 3 --extra-batch 0 --no-cpu-baseline > $OUT/r02h_bench.json 2> $OUT/r02h_bench.err; echo "bench rc=$?"; tail -3 $OUT/r02h_bench.err; cut -c1-700 $OUT/r02h_bench.json
