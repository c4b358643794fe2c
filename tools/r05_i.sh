#!/bin/bash
# Round-5 call i: fabric counters of the GEMM / attention / reduce kernels at 2 images on the closing tree (-> profiles/traffic.json batch_2)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT/r05i; cd $R
COMMON="--no-cpu-baseline --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix"
PMC_REGEX="gemm_bf16|attn_|splitk" timeout 230 bash tools/pmc_traffic.sh r05_b2 python $R/bench.py $COMMON --no-fwd-only --batch 2 --extra-batch 0 --steps 4 --warmup 2 > $OUT/r05i/pmc_b2.txt 2>&1
cat $OUT/r05i/pmc_b2.txt | cut -c1-200
cp $OUT/r05_b2_pmc.json $OUT/r05i/
