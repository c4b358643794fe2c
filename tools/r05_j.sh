#!/bin/bash
# Round-5 call j: the shipped GEMM dispatch (v5) beside hipBLASLt (torch.addmm) on the 2-image and the 16-image hot shapes -> profiles/r05j_gemm_vs_hipblaslt.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r05j; mkdir -p $OUT; cd $R
( echo "## GEMM_SET=b2 (2 images per micro-step)"; GEMM_SET=b2 WITH_TORCH=1 timeout 100 python tools/gemm_bench.py 5,5
  echo; echo "## default set (16 images per step)"; WITH_TORCH=1 timeout 100 python tools/gemm_bench.py 5,5 ) > $OUT/gemm_vs_hipblaslt.txt 2>&1
cat $OUT/gemm_vs_hipblaslt.txt | cut -c1-200
