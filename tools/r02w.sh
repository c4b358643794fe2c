#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r02w.txt 2>&1; tail -6 gpurun_out/r02w.txt | cut -c1-600
