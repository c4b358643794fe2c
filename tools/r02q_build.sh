#!/bin/bash
# side builds of the library with the ping-pong GEMM's timing-experiment switches (see PP_EXP in gemm.hip)
cd "$(dirname "$0")/.."
mkdir -p build_exp
for e in "$@"; do
  LLMSEG_OUT=build_exp/lib_exp$e.so bash llmseg_amd/csrc/build.sh -DPP_EXP=$e > /dev/null 2>&1 &
done
wait
ls -la build_exp
