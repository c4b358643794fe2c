#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_backward_gpu.py -q -x > $OUT/r02j_tests.txt 2>&1; tail -5 $OUT/r02j_tests.txt
for f in "" "--no-overlap"; do
timeout 900 python bench.py --steps 20 --warmup 3 --extra-batch 0 --no-cpu-baseline $f > $OUT/r02j_bench$f.json 2> $OUT/r02j_bench$f.err; echo "bench $f rc=$?"; tail -2 $OUT/r02j_bench$f.err; python -c "
import json,sys; d=json.load(open('$OUT/r02j_bench$f.json')); print('$f', d['value'], d['ms_per_step'], d['fwd_only']['value'], d['config']['graph'], d.get('graph_error'))"
done
timeout 900 python bench.py --steps 10 --warmup 3 --batch 24 --extra-batch 0 --no-cpu-baseline > $OUT/r02j_b24.json 2> $OUT/r02j_b24.err;  python -c "
import json,sys; d=json.load(open('$OUT/r02j_b24.json')); print('b24', d['value'], d['ms_per_step'], d['fwd_only']['value'], d['config']['graph'], d.get('graph_error'))"
