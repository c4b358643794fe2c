#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k gemm > $OUT/r02c_test_gemm.txt 2>&1; tail -15 $OUT/r02c_test_gemm.txt
GEMM_SET=b2 WITH_TORCH=1 timeout 600 python tools/gemm_bench.py 0,2,8,9,8:4,9:3,9:2,5 > $OUT/r02c_gemm_b2.txt 2>&1; cat $OUT/r02c_gemm_b2.txt
