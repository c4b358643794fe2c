#!/bin/bash
# NOTE: the in-kernel split-K reduction this script A/B-tested was removed again (commit dc2bc25 holds it); LLMSEG_GEMM_FUSED_REDUCE is a no-op on the shipped library.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04h; mkdir -p $OUT; cd $R
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_backward_gpu.py -q -rf -k "gemm or reproducible or full_width or lora_paths" ) > $OUT/tests.log 2>&1; grep -E "passed|failed|error" $OUT/tests.log | tail -3
( GEMM_SET=b2 GEMM_SHAPES=6,8,9,10,15 RACE_REPEATS=300 timeout 300 python tools/gemm_bench.py 5 ) > $OUT/race.txt 2>&1; tail -7 $OUT/race.txt
B="--no-cpu-baseline --no-neighbours --no-k512 --no-loader --extra-batch 0 --no-fwd-only"
for v in 1 0 1 0; do LLMSEG_GEMM_FUSED_REDUCE=$v python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused=$v', d['ms_per_step'], d['launches_per_micro_step']['library_kernels'], d['loss'])"; done
