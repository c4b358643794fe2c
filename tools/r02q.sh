#!/bin/bash
# ping-pong GEMM timing experiments (side builds, wrong results by design): which section bounds a phase
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r02q.txt
for e in 0 "$@"; do
  if [ "$e" = 0 ]; then unset LLMSEG_LIB; else export LLMSEG_LIB=$PWD/build_exp/lib_exp$e.so; fi
  echo "== PP_EXP=$e" >> gpurun_out/r02q.txt
  GEMM_SHAPES=${SHAPES:-0,4,13} python tools/gemm_bench.py ${VARS:-8,9} 2>&1 | grep -v amdgpu.ids | tail -n +3 | cut -c1-80 >> gpurun_out/r02q.txt
done
cat gpurun_out/r02q.txt
