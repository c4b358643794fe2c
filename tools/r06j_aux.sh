#!/bin/bash
# cache-policy bits on the ping-pong kernels' LDS-DMA loads: side libraries (tools/side_gemm.sh) through tools/gemm_bench.py, shipped dispatch
mkdir -p gpurun_out/r06j
for n in base w2 w16 aw2 aw16 w18 w1 a2 w17 base; do
  lib=build_exp/lib_$n.so; [ $n = base ] && lib=llmseg_amd/libllmseg_hip.so
  for set in b2 ""; do
    echo "## lib $n set ${set:-default}" >> gpurun_out/r06j/aux.txt
    LLMSEG_LIB=$PWD/$lib GEMM_SET=$set timeout 300 python tools/gemm_bench.py 5,5 2>&1 | grep -v "amdgpu.ids" >> gpurun_out/r06j/aux.txt
  done
done
tail -40 gpurun_out/r06j/aux.txt
