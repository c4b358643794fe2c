#!/bin/bash
# Round-5 call g2: tests of the norm_out route only (call g measured the bench A/B)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r05g; mkdir -p $OUT; cd $R
timeout 400 python -m pytest tests/test_kernels_gpu.py::test_gemm_norm_out tests/test_kernels_gpu.py::test_gemm tests/test_model_gpu.py::test_full_width_llama_layer \
  "tests/test_model_gpu.py::test_tiny_train_losses" tests/test_backward_gpu.py::test_fused_accumulation_window_equals_micro_steps tests/test_backward_gpu.py::test_trainer_eager_and_graph \
  -m gpu -q --durations=8 2>&1 | tail -25 > $OUT/tests.log
cat $OUT/tests.log
