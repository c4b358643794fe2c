#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r02r.txt
GEMM_SET=b2 GEMM_SHAPES=0,1,4,7 timeout 300 python tools/gemm_bench.py 8,12,12r1,12r2,12r3,12r7,13,13r7 2>&1 | grep -v amdgpu.ids | cut -c1-130 >> gpurun_out/r02r.txt
cat gpurun_out/r02r.txt
