#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
GEMM_SET=dec WITH_TORCH=1 python tools/gemm_bench.py 5,0 2>&1 | grep -v amdgpu | cut -c1-220 > gpurun_out/r02w.txt; cat gpurun_out/r02w.txt
