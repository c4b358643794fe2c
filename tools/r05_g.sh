#!/bin/bash
# Round-5 call g: the residual-stream projection with the next RMSNorm as a second output (llmseg_gemm_args.norm_out): bit checks, the model tests that run the
# Llama trunk, then the 2-image bench with the reduce launch fused (default) and with the two-launch route (LLMSEG_GEMM_NO_NORM_FUSE=1).
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r05g; mkdir -p $OUT; cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py::test_gemm_norm_out tests/test_kernels_gpu.py::test_gemm tests/test_model_gpu.py::test_full_width_llama_layer \
  "tests/test_model_gpu.py::test_tiny_train_losses" tests/test_backward_gpu.py::test_fused_accumulation_window_equals_micro_steps -m gpu -x -q 2>&1 | tail -15 > $OUT/tests.log
cat $OUT/tests.log
B="--no-cpu-baseline --batch 2 --extra-batch 0 --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --no-fwd-only --steps 30 --warmup 5"
for i in 1 2; do
  timeout 200 python bench.py $B 2>/dev/null | grep '^{' > $OUT/bench_fused_$i.json
  LLMSEG_GEMM_NO_NORM_FUSE=1 timeout 200 python bench.py $B 2>/dev/null | grep '^{' > $OUT/bench_two_launch_$i.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05g/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d.get("launches_per_step"))
    except Exception as e: print(f, "ERR", e)
PY
