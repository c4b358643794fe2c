"""GPU box, under `rocprofv3 --kernel-trace`: every GEMM variant on the small shapes (CLIP at M = 514, the head), 20 launches each, with a marker kernel
(torch fill of a distinct size) between groups; tools/probes/small_gemm_report.py reads kernel durations per group from the trace."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llmseg_amd import _lib, ops  # noqa: E402

SHAPES = [(514, 3072, 1024), (514, 1024, 1024), (514, 4096, 1024), (514, 1024, 4096), (512, 2048, 256), (512, 256, 2048), (512, 768, 256), (638, 4096, 4096)]
VARS = ["5", "2", "0", "9:0:2", "9:2:2", "9:4:2", "9:8:2", "8"]


def var(spec):
    f = spec.split(":")
    return int(f[0]) | (int(f[1]) << 8 if len(f) > 1 and f[1] else 0) | ((int(f[2]) + 1) << 13 if len(f) > 2 and f[2] else 0)


lib = _lib.load()
mark = torch.zeros(1000, device="cuda")
gi = 0
for (M, N, K) in SHAPES:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for v in VARS:
        lib.llmseg_gemm_set_variant(var(v))
        try:
            ops.gemm(a, w, bias=bias, out=out)
        except RuntimeError:
            print(f"GROUP {gi} {M}x{N}x{K} v{v} unsupported", flush=True)
            gi += 1
            torch.cumsum(mark, 0)
            continue
        torch.cuda.synchronize()
        torch.cumsum(mark, 0)                       # marker: a scan kernel appears nowhere else
        for _ in range(20):
            ops.gemm(a, w, bias=bias, out=out)
        torch.cuda.synchronize()
        print(f"GROUP {gi} {M}x{N}x{K} v{v}", flush=True)
        gi += 1
lib.llmseg_gemm_set_variant(5)
