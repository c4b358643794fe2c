#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python tools/amg_bench.py 64 > gpurun_out/r02w.txt 2>&1; timeout 900 python tools/amg_bench.py 256 >> gpurun_out/r02w.txt 2>&1; timeout 900 python tools/amg_bench.py 1024 >> gpurun_out/r02w.txt 2>&1
grep -v amdgpu gpurun_out/r02w.txt | tail -14 | cut -c1-300
