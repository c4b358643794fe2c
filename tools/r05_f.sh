#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --no-fwd-only --extra-batch 0 --steps 20"
for i in 1 2 3; do
  timeout 300 $B 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('tree', round(d['value'],2), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['loss'])"
done
GEMM_SET=b2 python tools/gemm_bench.py 8,9,5,5 2>&1 | cut -c1-110 | head -9
python bench.py --no-cpu-baseline --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --no-fwd-only --batch 24 --extra-batch 0 --steps 6 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('b24', round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"
