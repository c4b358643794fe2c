#!/bin/bash
# 2 x 4-wave window kernel: parity (test_attention + SAM tests with the switch on), SAM encoder alone, step A/B
mkdir -p gpurun_out/r06n
LLMSEG_WIN_W4=1 timeout 900 python -m pytest tests/test_kernels_gpu.py::test_attention tests/test_model_gpu.py::test_sam_encoder_vs_reference_fixture tests/test_model_gpu.py::test_full_width_sam_blocks -x -q 2>&1 | tail -8 > gpurun_out/r06n/tests.log
cat gpurun_out/r06n/tests.log
for i in 1 2; do python tools/sam_only.py 2; LLMSEG_WIN_W4=1 python tools/sam_only.py 2; done 2>&1 | grep "sam encoder" | tee gpurun_out/r06n/sam_only.txt
python tools/sam_only.py 20 2>&1 | grep "sam enc" | tee -a gpurun_out/r06n/sam_only.txt
LLMSEG_WIN_W4=1 python tools/sam_only.py 20 2>&1 | grep "sam enc" | tee -a gpurun_out/r06n/sam_only.txt
bash tools/ab_env.sh "LLMSEG_WIN_W4=1" 2 2>&1 | tee gpurun_out/r06n/ab_w4.txt
