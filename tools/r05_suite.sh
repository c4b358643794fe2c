#!/bin/bash
# the whole GPU suite (no -x) + smoke, as the driver runs them; log -> gpurun_out/r05_gpu_tests.log
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -q -m gpu --durations=12 ) > gpurun_out/r05_gpu_tests.log 2>&1
tail -30 gpurun_out/r05_gpu_tests.log | cut -c1-300
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r05_smoke.log 2>&1
tail -4 gpurun_out/r05_smoke.log | cut -c1-300
