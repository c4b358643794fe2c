#!/bin/bash
# Round-5 closing call on the tree with llmseg_gemm_args.norm_out: default bench line, 2-image kernel tables (hipGraph and one stream), then the GPU suite minus the
# seven tests tools/r05_g2.sh already ran on this tree (GPU-minute budget), then smoke.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT/r05h; cd $R
( time python bench.py ) > $OUT/r05h/bench_default.json 2> $OUT/r05h/bench_default.err; tail -3 $OUT/r05h/bench_default.err
bash tools/gpu_prof.sh r05_b2 --batch 2 --extra-batch 0 --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --steps 10 --warmup 3 > /dev/null 2>&1
bash tools/gpu_prof.sh r05_b2_1stream --batch 2 --extra-batch 0 --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --no-fwd-only --no-overlap --no-graph --steps 5 --warmup 2 > /dev/null 2>&1
cp $OUT/r05_b2_kernel_stats.md $OUT/r05_b2_1stream_kernel_stats.md $OUT/r05h/ 2>/dev/null
cd $R
( time timeout 560 python -m pytest tests -q -m gpu --durations=10 \
  --deselect tests/test_kernels_gpu.py::test_gemm_norm_out --deselect tests/test_kernels_gpu.py::test_gemm --deselect tests/test_model_gpu.py::test_full_width_llama_layer \
  --deselect "tests/test_model_gpu.py::test_tiny_train_losses" --deselect tests/test_backward_gpu.py::test_fused_accumulation_window_equals_micro_steps \
  --deselect tests/test_backward_gpu.py::test_trainer_eager_and_graph ) > $OUT/r05h/gpu_tests.log 2>&1
tail -22 $OUT/r05h/gpu_tests.log | cut -c1-250
( time timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/r05h/smoke.log 2>&1; tail -4 $OUT/r05h/smoke.log | cut -c1-250
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05h/bench_default.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "all", d["roofline"]["all_gemm_kernels"]["frac"], "model", d["model_mfma_frac"])
print("fused", d["accum_fused"]["value"], "b24", d["batch_24"]["value"], d["batch_24"]["model_mfma_frac"], "mix", d["mix_9_3_1_batch_1"]["value"], "loader", d["loader_in_loop"]["value"], "cpu", d["cpu_baseline"]["value"])
PY
