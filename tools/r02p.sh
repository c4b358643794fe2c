#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/attn_one.py wint0 24 > gpurun_out/r02p_attn.txt 2>&1
python tools/attn_one.py wint0 2 >> gpurun_out/r02p_attn.txt 2>&1
cat gpurun_out/r02p_attn.txt
