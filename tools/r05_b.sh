#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_backward_gpu.py -q -k "fused" 2>&1 | tail -25 > gpurun_out/r05b_bwd.log
timeout 1000 python -m pytest tests/test_backward_gpu.py -q -s -k "full_depth_configs2" 2>&1 | tail -160 > gpurun_out/r05b_fulldepth.log
echo "== bwd"; tail -12 gpurun_out/r05b_bwd.log | cut -c1-1500
echo "== fulldepth"; tail -8 gpurun_out/r05b_fulldepth.log | cut -c1-1500
