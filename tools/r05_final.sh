#!/bin/bash
# Round-5 final measurements on the GPU box (one gpurun call): default bench line, kernel-trace summaries (hipGraph / one stream at 2 images, 24 images),
# fabric traffic (-> profiles/traffic.json), MFMA occupancy at 2 images.  Everything lands in gpurun_out/r05final/.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT/r05final; cd $R
COMMON="--no-cpu-baseline --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix"
( time python bench.py ) > $OUT/r05final/bench_default.json 2> $OUT/r05final/bench_default.err; tail -3 $OUT/r05final/bench_default.err
bash tools/gpu_prof.sh r05_b2 --batch 2 --extra-batch 0 --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --steps 10 --warmup 3 > /dev/null 2>&1
bash tools/gpu_prof.sh r05_b2_1stream --batch 2 --extra-batch 0 --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --no-fwd-only --no-overlap --no-graph --steps 5 --warmup 2 > /dev/null 2>&1
bash tools/gpu_prof.sh r05_b24 --batch 24 --extra-batch 0 --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --steps 5 --warmup 2 > /dev/null 2>&1
PMC_REGEX="gemm_bf16|attn_|splitk" bash tools/pmc_traffic.sh r05_b2 python $R/bench.py $COMMON --no-fwd-only --batch 2 --extra-batch 0 --steps 4 --warmup 2 > $OUT/r05final/pmc_b2.txt 2>&1
PMC_REGEX="gemm_bf16|attn_|splitk" bash tools/pmc_traffic.sh r05_b24 python $R/bench.py $COMMON --no-fwd-only --batch 24 --extra-batch 0 --steps 2 --warmup 1 > $OUT/r05final/pmc_b24.txt 2>&1
bash tools/pmc_mfma.sh r05_b2 --batch 2 --extra-batch 0 --no-accum-fused --no-mix --steps 3 --warmup 1 > /dev/null 2>&1
cp $OUT/r05_*kernel_stats.md $OUT/r05_*_pmc.json $OUT/r05_*_mfma.md $OUT/r05final/ 2>/dev/null
ls -la $OUT/r05final | tail -20
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05final/bench_default.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "all", d["roofline"]["all_gemm_kernels"]["frac"], "model", d["model_mfma_frac"])
print("fused", d["accum_fused"]["value"], "b24", d["batch_24"]["value"], d["batch_24"]["model_mfma_frac"], "mix", d["mix_9_3_1_batch_1"]["value"], "loader", d["loader_in_loop"]["value"], "cpu", d["cpu_baseline"]["value"])
PY
