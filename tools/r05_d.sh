#!/bin/bash
# bench on the current tree: default line (incl. cpu baseline, mix, fused) + a world-1 RCCL run for the grad_exchange fields + dist tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r05d_bench.json 2> gpurun_out/r05d_bench.err
tail -c 600 gpurun_out/r05d_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05d_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "model_frac", d["model_mfma_frac"])
print("composition:", d["roofline"].get("class_composition"))
for k in ("accum_fused", "mix_9_3_1_batch_1", "loader_in_loop", "batch_2_k512"):
    v = d.get(k) or {}
    print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "ms_per_optimizer_step", "graphs", "sources_drawn", "input_ms", "graph")})
print("batch_24", d.get("batch_24", {}).get("value"), d.get("batch_24", {}).get("model_mfma_frac"), d.get("batch_24", {}).get("roofline", {}).get("frac"))
c = d.get("cpu_baseline", {})
print("cpu", c.get("value"), c.get("cores"), c.get("cpu_seconds"), c.get("fwd_bwd_fp32", {}).get("measured_at_full_depth"), c.get("fwd_fp32", {}).get("s_per_image"))
print(c.get("sample"))
PY
timeout 600 python bench.py --force-dist --no-cpu-baseline --no-neighbours --no-k512 --no-loader --no-mix --no-accum-fused --no-fwd-only --extra-batch 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('force-dist', d['value'], d['grad_exchange'])"
timeout 600 python -m pytest tests/test_dist_gpu.py -q 2>&1 | tail -4
