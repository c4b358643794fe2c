#!/bin/bash
# Round-5 call l: the three GPU tests the closing two-call suite run skipped (pytest --deselect matches by prefix)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r05l; mkdir -p $OUT; cd $R
timeout 200 python -m pytest tests/test_kernels_gpu.py::test_gemm_hot_shapes tests/test_model_gpu.py::test_tiny_train_losses_ragged_proposal_counts \
  tests/test_model_gpu.py::test_tiny_train_losses_k512 -m gpu -q --durations=3 2>&1 | tail -12 > $OUT/tests.log
cat $OUT/tests.log
