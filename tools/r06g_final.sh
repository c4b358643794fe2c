#!/bin/bash
# Round-6 closing measurements on the GPU box (one gpurun call) on the tree with the Llama-layer fusions: the driver's bench command, an alternating same-box A/B of the
# fusions (environment switches: every fused route back to its pointwise launches), kernel-trace summaries (hipGraph / one stream at 2 images, window towers, 24 images),
# fabric traffic (-> profiles/traffic.json), MFMA occupancy at 2 images.  Everything lands in gpurun_out/r06g/.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT/r06g; cd $R
COMMON="--no-cpu-baseline --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --no-window-towers"
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/r06g/bench_driver_cmd.json 2> $OUT/r06g/bench_driver_cmd.err; tail -3 $OUT/r06g/bench_driver_cmd.err
OFF="LLMSEG_GEMM_NO_FX=1 LLMSEG_GEMM_NO_NB=1 LLMSEG_NO_FINISH_PACK=1 LLMSEG_NO_FUSE_ROPE_BWD=1 LLMSEG_NO_FUSE_ROPE_FWD=1 LLMSEG_NO_FUSE_MLP=1 LLMSEG_NO_FUSE_NORM_BWD=1 LLMSEG_NO_FUSE_DELTA=1 LLMSEG_NO_FUSE_LORA_PARTS=1 LLMSEG_GEMM_NO_DL=1 LLMSEG_NO_LORA_PARTS=1"
echo "# alternating A/B on one box: python bench.py $COMMON --batch 2 --extra-batch 0 --steps 20 --warmup 5  (fused = default; unfused = $OFF)" > $OUT/r06g/ab_fusions.txt
for i in 1 2 3; do
  for mode in fused unfused; do
    if [ $mode = fused ]; then E=""; else E="$OFF"; fi
    env $E python bench.py $COMMON --batch 2 --extra-batch 0 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'img/s  fwd-only', round(d['fwd_only']['value'],2), ' launches', d['launches_per_micro_step']['library_kernels'])" >> $OUT/r06g/ab_fusions.txt
  done
done
cat $OUT/r06g/ab_fusions.txt
bash tools/gpu_prof.sh r06g_b2 --batch 2 --extra-batch 0 --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --no-window-towers --steps 10 --warmup 3 > /dev/null 2>&1
bash tools/gpu_prof.sh r06g_b2_1stream --batch 2 --extra-batch 0 --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --no-window-towers --no-fwd-only --no-overlap --no-graph --steps 5 --warmup 2 > /dev/null 2>&1
bash tools/gpu_prof.sh r06g_window_towers --batch 2 --window-towers-only --steps 20 > /dev/null 2>&1
bash tools/gpu_prof.sh r06g_b24 --batch 24 --extra-batch 0 --no-neighbours --no-k512 --no-loader --no-accum-fused --no-mix --no-window-towers --steps 5 --warmup 2 > /dev/null 2>&1
PMC_REGEX="gemm_bf16|attn_|splitk|reduce_lora" bash tools/pmc_traffic.sh r06g_b2 python $R/bench.py $COMMON --no-fwd-only --batch 2 --extra-batch 0 --steps 4 --warmup 2 > $OUT/r06g/pmc_b2.txt 2>&1
PMC_REGEX="gemm_bf16|attn_|splitk|reduce_lora" bash tools/pmc_traffic.sh r06g_b24 python $R/bench.py $COMMON --no-fwd-only --batch 24 --extra-batch 0 --steps 2 --warmup 1 > $OUT/r06g/pmc_b24.txt 2>&1
bash tools/pmc_mfma.sh r06g_b2 --batch 2 --extra-batch 0 --no-accum-fused --no-mix --no-window-towers --steps 3 --warmup 1 > /dev/null 2>&1
cp $OUT/r06g_*kernel_stats.md $OUT/r06g_*_pmc.json $OUT/r06g_*_mfma.md $OUT/r06g/ 2>/dev/null
ls -la $OUT/r06g | tail -20
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06g/bench_driver_cmd.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "all", d["roofline"]["all_gemm_kernels"]["frac"], "model", d["model_mfma_frac"])
print("fwd", d["fwd_only"]["value"], "fused", d["accum_fused"]["value"], "window", d["window_towers"]["value"], "b24", d["batch_24"]["value"], d["batch_24"]["model_mfma_frac"],
      "mix", d["mix_9_3_1_batch_1"]["value"], d["mix_9_3_1_batch_1"]["window_towers"]["value"], "loader", d["loader_in_loop"]["value"], "cpu", d["cpu_baseline"]["value"])
PY
