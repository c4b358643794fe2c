#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_backward_gpu.py -q -k "lora_arena or trainer" > $OUT/r02f_tests.txt 2>&1; tail -8 $OUT/r02f_tests.txt
timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/r02f_bench.json 2> $OUT/r02f_bench.err; echo "bench rc=$?"; tail -5 $OUT/r02f_bench.err; cat $OUT/r02f_bench.json
