#!/bin/bash
cd $GRAFT_REPO_ROOT
LLMSEG_LORA_DOWN=$1 timeout 600 python -m pytest tests/test_backward_gpu.py -q -k checkpoint 2>&1 | grep -E "passed|failed|AssertionError:" | cut -c1-300
