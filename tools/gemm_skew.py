import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llmseg_amd import _lib, ops
lib = _lib.load()
shapes = [(32768, 3840, 1280), (32768, 1280, 1280), (32768, 5120, 1280), (32768, 1280, 5120), (39200, 3840, 1280), (2552, 4096, 4096)]
skews = [0, 13, 29, 61, 101, 7]
print("shape".ljust(24), " ".join(f"s{s:>5d}" for s in skews))
for M, N, K in shapes:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    best = {s: 0.0 for s in skews}
    for rnd in range(4):                      # interleaved rounds
        for s in skews:
            lib.llmseg_gemm_set_variant(2 + 16 * (s + 1))
            ops.gemm(a, w, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm(a, w, out=out)
            e1.record(); torch.cuda.synchronize()
            best[s] = max(best[s], 2.0 * M * N * K / (e0.elapsed_time(e1) / 10) / 1e9)
    print(f"{M}x{N}x{K}".ljust(24), " ".join(f"{best[s]:6.0f}" for s in skews), flush=True)
