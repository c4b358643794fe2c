#!/bin/bash
# Round-6 closing measurements on the GPU box (one gpurun call) on the final tree (window gather): the driver's bench command, kernel-trace summaries (hipGraph / one stream at 2 images, window towers, 24 images),
# fabric traffic (-> profiles/traffic.json), MFMA occupancy at 2 images.  Everything lands in gpurun_out/r06p/.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT/r06p; cd $R
COMMON="--no-cpu-baseline --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --no-window-towers"
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/r06p/bench_driver_cmd.json 2> $OUT/r06p/bench_driver_cmd.err; tail -3 $OUT/r06p/bench_driver_cmd.err
bash tools/gpu_prof.sh r06p_b2 --batch 2 --extra-batch 0 --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --no-window-towers --steps 10 --warmup 3 > /dev/null 2>&1
bash tools/gpu_prof.sh r06p_b2_1stream --batch 2 --extra-batch 0 --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --no-window-towers --no-fwd-only --no-overlap --no-graph --steps 5 --warmup 2 > /dev/null 2>&1
bash tools/gpu_prof.sh r06p_window_towers --batch 2 --window-towers-only --steps 20 > /dev/null 2>&1
bash tools/gpu_prof.sh r06p_b24 --batch 24 --extra-batch 0 --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --no-window-towers --steps 5 --warmup 2 > /dev/null 2>&1
PMC_REGEX="gemm_bf16|attn_|splitk|reduce_lora" bash tools/pmc_traffic.sh r06p_b2 python $R/bench.py $COMMON --no-fwd-only --batch 2 --extra-batch 0 --steps 4 --warmup 2 > $OUT/r06p/pmc_b2.txt 2>&1
PMC_REGEX="gemm_bf16|attn_|splitk|reduce_lora" bash tools/pmc_traffic.sh r06p_b24 python $R/bench.py $COMMON --no-fwd-only --batch 24 --extra-batch 0 --steps 2 --warmup 1 > $OUT/r06p/pmc_b24.txt 2>&1
bash tools/pmc_mfma.sh r06p_b2 --batch 2 --extra-batch 0 --no-accum-fused --no-mix --no-window-towers --steps 3 --warmup 1 > /dev/null 2>&1
cp $OUT/r06p_*kernel_stats.md $OUT/r06p_*_pmc.json $OUT/r06p_*_mfma.md $OUT/r06p/ 2>/dev/null
ls -la $OUT/r06p | tail -20
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06p/bench_driver_cmd.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "all", d["roofline"]["all_gemm_kernels"]["frac"], "model", d["model_mfma_frac"])
print("fwd", d["fwd_only"]["value"], "fused", d["accum_fused"]["value"], "window", d["window_towers"]["value"], "b24", d["batch_24"]["value"], d["batch_24"]["model_mfma_frac"],
      "mix", d["mix_9_3_1_batch_1"]["value"], d["mix_9_3_1_batch_1"]["window_towers"]["value"], "loader", d["loader_in_loop"]["value"], "cpu", d["cpu_baseline"]["value"])
PY
