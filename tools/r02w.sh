#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_generate_gpu.py -q -x > gpurun_out/r02w.txt 2>&1; tail -25 gpurun_out/r02w.txt | cut -c1-600
