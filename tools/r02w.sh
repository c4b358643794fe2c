#!/bin/bash
# final-state artefacts: kernel stats (B=2), default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/gpu_prof.sh r02_final_b2 --batch 2 --extra-batch 0 --no-neighbours --steps 10 --warmup 3 > /dev/null 2>&1
( time timeout 900 python bench.py ) > gpurun_out/r02_final_bench.txt 2>&1
tail -4 gpurun_out/r02_final_bench.txt | cut -c1-300
