#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_backward_gpu.py tests/test_model_gpu.py -q -x > gpurun_out/r02w.txt 2>&1; tail -5 gpurun_out/r02w.txt | cut -c1-600
for i in 1 2; do
timeout 900 python bench.py --extra-batch 0 --no-cpu-baseline --no-neighbours --no-fwd-only 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('run', d['value'], d['ms_per_step'], d['loss'])"
done
