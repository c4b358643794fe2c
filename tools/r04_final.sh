#!/bin/bash
# Round-4 final measurements on the GPU box (one gpurun call): default bench line, kernel-trace summaries (hipGraph / one stream, 2 and 24 images),
# fabric traffic, MFMA occupancy, HBM-kernel table.  Everything lands in gpurun_out/r04final/.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT/r04final; cd $R
COMMON="--no-cpu-baseline --no-neighbours --no-k512 --no-loader"
( time python bench.py ) > $OUT/r04final/bench_default.json 2> $OUT/r04final/bench_default.err; tail -3 $OUT/r04final/bench_default.err
bash tools/gpu_prof.sh r04_b2 --batch 2 --extra-batch 0 --no-neighbours --no-k512 --no-loader --steps 10 --warmup 3 > /dev/null 2>&1
bash tools/gpu_prof.sh r04_b2_1stream --batch 2 --extra-batch 0 --no-neighbours --no-k512 --no-loader --no-fwd-only --no-overlap --no-graph --steps 5 --warmup 2 > /dev/null 2>&1
bash tools/gpu_prof.sh r04_b24 --batch 24 --extra-batch 0 --no-neighbours --no-k512 --no-loader --steps 5 --warmup 2 > /dev/null 2>&1
bash tools/gpu_prof.sh r04_b24_1stream --batch 24 --extra-batch 0 --no-neighbours --no-k512 --no-loader --no-fwd-only --no-overlap --no-graph --steps 3 --warmup 1 > /dev/null 2>&1
PMC_REGEX="gemm_bf16|attn_|splitk" bash tools/pmc_traffic.sh r04_b2 python $R/bench.py $COMMON --no-fwd-only --batch 2 --extra-batch 0 --steps 4 --warmup 2 > $OUT/r04final/pmc_b2.txt 2>&1
PMC_REGEX="gemm_bf16|attn_|splitk" bash tools/pmc_traffic.sh r04_b24 python $R/bench.py $COMMON --no-fwd-only --batch 24 --extra-batch 0 --steps 2 --warmup 1 > $OUT/r04final/pmc_b24.txt 2>&1
bash tools/pmc_mfma.sh r04_b2 --batch 2 --extra-batch 0 --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/pmc_mfma.sh r04_b24 --batch 24 --extra-batch 0 --steps 2 --warmup 1 > /dev/null 2>&1
python tools/hbm_kernels.py 24 > $OUT/r04final/hbm.json 2> $OUT/r04final/hbm.err
bash tools/pmc_traffic.sh r04_hbm python $R/tools/hbm_kernels.py 24 > $OUT/r04final/pmc_hbm.txt 2>&1
python tools/hbm_kernels.py 2 > $OUT/r04final/hbm_b2.json 2>> $OUT/r04final/hbm.err
cp $OUT/r04_*kernel_stats.md $OUT/r04_*_pmc.json $OUT/r04_*_mfma.md $OUT/r04final/ 2>/dev/null
ls -la $OUT/r04final | tail -30
