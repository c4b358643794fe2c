#!/bin/bash
# HEAD validation part 2: smoke, whole GPU suite, B=24 PMC traffic + kernel profile
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" ) > gpurun_out/r02u_smoke.txt 2>&1; tail -2 gpurun_out/r02u_smoke.txt | cut -c1-200
( timeout 1800 python -m pytest tests -m gpu -q ) > gpurun_out/r02u_tests.txt 2>&1; tail -6 gpurun_out/r02u_tests.txt
bash tools/pmc_traffic.sh r02x_bench24 python $GRAFT_REPO_ROOT/bench.py --batch 24 --extra-batch 0 --no-cpu-baseline --steps 3 --warmup 1 --no-graph --no-fwd-only > gpurun_out/r02u_pmc24.txt 2>&1; tail -12 gpurun_out/r02u_pmc24.txt | cut -c1-200
bash tools/gpu_prof.sh r02x_b24 --batch 24 --extra-batch 0 --steps 3 --warmup 1 > gpurun_out/r02u_prof24.txt 2>&1; tail -3 gpurun_out/r02u_prof24.txt | cut -c1-300
