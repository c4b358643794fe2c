#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04d; mkdir -p $OUT; cd $R
B="--no-cpu-baseline --no-neighbours --no-k512 --no-loader --extra-batch 0 --no-fwd-only"
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_backward_gpu.py tests/test_model_gpu.py -q -rf -k "pointwise or autograd_ops or arena_ops or full_width_llama or reproducible or tiny_train_losses" ) > $OUT/tests.log 2>&1; grep -E "passed|failed|error" $OUT/tests.log | tail -3
( timeout 600 python bench.py $B ) > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json; tail -2 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace --stats -d /tmp/prof_tl -o tl -- python $R/bench.py $B --steps 4 --warmup 2 > $OUT/prof_tl.log 2>&1
F=$(find /tmp/prof_tl -name "*_results.db" | head -1)
python $R/tools/timeline.py "$F" embed_splice 6 > $OUT/step_timeline_graph.csv 2> $OUT/step_timeline_graph.err; tail -3 $OUT/step_timeline_graph.err
python - "$F" <<'PY' > $OUT/schema.txt 2>&1
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for (n, t) in cur.execute("select name, type from sqlite_master where type in ('table','view')").fetchall():
    if 'kernel' in n.lower() or 'dispatch' in n.lower():
        print(t, n, [c[1] for c in cur.execute(f"pragma table_info('{n}')")])
PY
head -20 $OUT/schema.txt
