"""Tuning aid (GPU box): time per ROUND of the 256 x 256 tile kernel against the number of K-tiles (fit T = F + k * K/64: F = per-tile fixed cost -- dispatch, prologue fetch,
epilogue --, k = main-loop time per K-tile), beside hipBLASLt (torch.addmm).  tools/gemm_ksweep.py [M N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from llmseg_amd import _lib, ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
lib = _lib.load()
tiles = ((M + 255) // 256) * ((N + 255) // 256)
rounds = (tiles + 255) // 256
print(f"M {M} N {N}: {tiles} tiles of 256 x 256 = {tiles / 256:.2f} rounds of 256 CUs")
print(f"{'K':>6s} {'K-tiles':>8s} {'v8 us':>9s} {'us/round':>9s} {'TF/s':>7s} {'addmm us':>9s} {'TF/s':>7s}")
pts = []
for K in (128, 256, 512, 1024, 1280, 2048, 4096, 5120, 8192):
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)

    def t(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    lib.llmseg_gemm_set_variant(8)
    us = t(lambda: ops.gemm(a, w, bias=bias, out=out))
    ub = t(lambda: torch.addmm(bias, a, w.t(), out=out))
    fl = 2.0 * M * N * K
    pts.append((K // 64, us / rounds))
    print(f"{K:6d} {K // 64:8d} {us:9.1f} {us / rounds:9.2f} {fl / us / 1e6:7.0f} {ub:9.1f} {fl / ub / 1e6:7.0f}", flush=True)
lib.llmseg_gemm_set_variant(5)
(x0, y0), (x1, y1) = pts[2], pts[-1]
k = (y1 - y0) / (x1 - x0)
print(f"fit through K = {pts[2][0] * 64} and {pts[-1][0] * 64}: k = {k:.3f} us per K-tile, F = {y0 - k * x0:.2f} us per round")
