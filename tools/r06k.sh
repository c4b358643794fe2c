#!/bin/bash
# window gather: kernel + SAM tests, then same-box A/B of LLMSEG_WIN_NO_GATHER
mkdir -p gpurun_out/r06k
timeout 900 python -m pytest tests/test_kernels_gpu.py::test_attention tests/test_model_gpu.py::test_sam_encoder_vs_reference_fixture tests/test_model_gpu.py::test_full_width_sam_blocks "tests/test_model_gpu.py::test_tiny_inference" tests/test_backward_gpu.py::test_window_step_with_batched_frozen_towers -x -q 2>&1 | tail -15 > gpurun_out/r06k/tests.log
cat gpurun_out/r06k/tests.log
bash tools/ab_env.sh "LLMSEG_WIN_NO_GATHER=1" 3 2>&1 | tee gpurun_out/r06k/ab_gather.txt
