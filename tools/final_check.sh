#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" ) > gpurun_out/final_smoke.txt 2>&1; tail -1 gpurun_out/final_smoke.txt
( timeout 2400 python -m pytest tests -m gpu -q ) > gpurun_out/final_tests.txt 2>&1; grep -E "passed|failed" gpurun_out/final_tests.txt | tail -2
( time timeout 900 python bench.py ) > gpurun_out/final_bench.txt 2>&1; tail -4 gpurun_out/final_bench.txt | cut -c1-200
bash tools/gpu_prof.sh r02_final_b2 --batch 2 --extra-batch 0 --no-neighbours --steps 10 --warmup 3 > /dev/null 2>&1
