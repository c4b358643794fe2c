#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --no-fwd-only --extra-batch 0 --steps 20"
J='import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith("{")][-1]); print(sys.argv[1], round(d["value"],2), round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), round(d["roofline"]["all_gemm_kernels"]["frac"],4), d["roofline"]["kernel"][:40], d["loss"])'
for i in 1 2; do timeout 300 $B 2>/dev/null | python -c "$J" b2; done
GEMM_SET=b2 python tools/gemm_bench.py 5,5 2>&1 | cut -c1-80 | tail -17
python tools/gemm_bench.py 5,5 2>&1 | cut -c1-80 | tail -14
timeout 300 python bench.py --no-cpu-baseline --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --no-fwd-only --batch 24 --extra-batch 0 --steps 6 2>/dev/null | python -c "$J" b24
timeout 300 python bench.py --no-cpu-baseline --no-neighbours --no-k512 --no-loader --no-mix --no-fwd-only --extra-batch 0 --steps 4 2>/dev/null | python -c 'import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith("{")][-1]); print("fused", d["accum_fused"]["value"], d["accum_fused"]["ms_per_optimizer_step"])'
