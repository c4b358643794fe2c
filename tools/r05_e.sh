#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python bench.py --force-dist --no-cpu-baseline --no-neighbours --no-k512 --no-loader --no-mix --no-accum-fused --no-fwd-only --extra-batch 0 > /tmp/fd.json 2> /tmp/fd.err
tail -c 2500 /tmp/fd.err; tail -c 1500 /tmp/fd.json
timeout 600 python -m pytest tests/test_dist_gpu.py -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -30
