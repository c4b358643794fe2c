#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r02r.txt
GEMM_SHAPES=0,4,5,7,13 python tools/gemm_bench.py 8q,8qr1,8qr2,8qr3,8qr4,8qr5,8r3 2>&1 | grep -v amdgpu.ids | cut -c1-190 >> gpurun_out/r02r.txt
GEMM_SET=b2 GEMM_SHAPES=0,1,2,3,4,14 python tools/gemm_bench.py 8q,8qr1,8qr2,8qr3,8qr4,8qr5,8r3 2>&1 | grep -v amdgpu.ids | cut -c1-190 >> gpurun_out/r02r.txt
GEMM_SET=b2 GEMM_SHAPES=5,7,11,12 python tools/gemm_bench.py 9q,9qr1,9qr2,9qr3,9qr4,9qr5,9r3 2>&1 | grep -v amdgpu.ids | cut -c1-190 >> gpurun_out/r02r.txt
cat gpurun_out/r02r.txt
