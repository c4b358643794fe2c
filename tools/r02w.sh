#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do
timeout 900 python bench.py --extra-batch 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('run', d['value'], d['ms_per_step'], d['fwd_only']['value'] if 'fwd_only' in d else None, json.dumps(d.get('neighbours',{}).get('everything_mode'))[:200])"
done
