#!/bin/bash
# Round-5 call k: forced K-slice plans of the 256 x 256 tile (v8:S) and the 128 x 256 tile (v9:S) on the SAM shapes and llama o / lm_head at 2 images
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r05k; mkdir -p $OUT; cd $R
( GEMM_SET=b2 GEMM_SHAPES=0,1,2,3,4,6,12 timeout 150 python tools/gemm_bench.py 5,8,8:2,8:3,9,9:2,9:4,5 ) > $OUT/sweep.txt 2>&1
grep -v amdgpu.ids $OUT/sweep.txt | cut -c1-220
