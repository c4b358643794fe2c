#!/bin/bash
# Round-3 final measurements on the GPU box (one gpurun call): bench line, kernel-trace summaries, fabric traffic, MFMA occupancy, HBM-kernel table.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
COMMON="--no-cpu-baseline --no-neighbours --no-k512 --no-loader"
( time python bench.py ) > $OUT/r03_final_bench.json 2> $OUT/r03_final_bench.err; tail -3 $OUT/r03_final_bench.err
bash tools/gpu_prof.sh r03_final_b2 --batch 2 --extra-batch 0 --no-neighbours --no-k512 --no-loader --steps 10 --warmup 3 > /dev/null 2>&1
bash tools/gpu_prof.sh r03_final_b2_1stream --batch 2 --extra-batch 0 --no-neighbours --no-k512 --no-loader --no-fwd-only --no-overlap --no-graph --steps 5 --warmup 2 > /dev/null 2>&1
bash tools/gpu_prof.sh r03_final_b24 --batch 24 --extra-batch 0 --no-neighbours --no-k512 --no-loader --steps 5 --warmup 2 > /dev/null 2>&1
PMC_REGEX="gemm_bf16|attn_|splitk" bash tools/pmc_traffic.sh r03_b2 python $R/bench.py $COMMON --no-fwd-only --batch 2 --extra-batch 0 --steps 4 --warmup 2 > $OUT/r03_pmc_b2.txt 2>&1
PMC_REGEX="gemm_bf16|attn_|splitk" bash tools/pmc_traffic.sh r03_b24 python $R/bench.py $COMMON --no-fwd-only --batch 24 --extra-batch 0 --steps 2 --warmup 1 > $OUT/r03_pmc_b24.txt 2>&1
bash tools/pmc_mfma.sh r03_b2 --batch 2 --extra-batch 0 --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/pmc_mfma.sh r03_b24 --batch 24 --extra-batch 0 --steps 2 --warmup 1 > /dev/null 2>&1
python tools/hbm_kernels.py 24 > $OUT/r03_hbm.json 2> $OUT/r03_hbm.err
bash tools/pmc_traffic.sh r03_hbm python $R/tools/hbm_kernels.py 24 > $OUT/r03_pmc_hbm.txt 2>&1
ls -la $OUT | tail -30
