#!/bin/bash
# round 5, first GPU call: the new tests (collate / init loaders / fused accumulation / segmented dropout / target edge cases / full-depth gradients) + a bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py -q -k "collated or val_sample or from_pretrained" 2>&1 | tail -25 > gpurun_out/r05a_model.log
timeout 600 python -m pytest tests/test_backward_gpu.py -q -k "fused or lora_paths" 2>&1 | tail -25 > gpurun_out/r05a_bwd.log
timeout 300 python -m pytest tests/test_targets_gpu.py -q 2>&1 | tail -12 > gpurun_out/r05a_targets.log
timeout 600 python bench.py --no-cpu-baseline --no-neighbours --no-k512 --no-loader > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err
timeout 1000 python -m pytest tests/test_backward_gpu.py -q -s -k "full_depth_configs2" 2>&1 | tail -160 > gpurun_out/r05a_fulldepth.log
for f in model bwd targets; do echo "== $f"; tail -6 gpurun_out/r05a_$f.log; done
echo "== bench"; tail -c 1500 gpurun_out/r05a_bench.json; tail -3 gpurun_out/r05a_bench.err
echo "== fulldepth"; tail -12 gpurun_out/r05a_fulldepth.log
