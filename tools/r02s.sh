#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r02s.txt
for l2 in "" a w aw; do
  echo "== GEMM_L2=$l2" >> gpurun_out/r02s.txt
  GEMM_L2=$l2 GEMM_SHAPES=0,4,5,7,13 python tools/gemm_bench.py 8,9 2>&1 | grep -v amdgpu.ids | tail -n +3 | cut -c1-60 >> gpurun_out/r02s.txt
  GEMM_L2=$l2 GEMM_SET=b2 GEMM_SHAPES=0,2,4,5,7,11 python tools/gemm_bench.py 8,9 2>&1 | grep -v amdgpu.ids | tail -n +3 | cut -c1-60 >> gpurun_out/r02s.txt
done
cat gpurun_out/r02s.txt
