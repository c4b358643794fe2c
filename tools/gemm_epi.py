import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llmseg_amd import ops
def run(M, N, K, iters=60, **kw):
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16) * 0.05
    b = torch.randn(N, device="cuda").to(torch.bfloat16)
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    args = dict(bias=b if kw.get("bias") else None, act=kw.get("act", 0), residual=r if kw.get("res") else None)
    for _ in range(5): ops.gemm(a, w, out=out, **args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm(a, w, out=out, **args)
    e1.record(); torch.cuda.synchronize()
    return 2.0 * M * N * K / (e0.elapsed_time(e1) / iters) / 1e9
for rnd in range(2):
    for (M, N, K) in [(32768, 5120, 1280), (32768, 1280, 5120), (32768, 3840, 1280)]:
        print(f"{M}x{N}x{K}: plain {run(M,N,K):.0f}  bias {run(M,N,K,bias=True):.0f}  bias+gelu {run(M,N,K,bias=True,act=2):.0f}  bias+res {run(M,N,K,bias=True,res=True):.0f}  sustained(400 it) {run(M,N,K,iters=400,bias=True):.0f}", flush=True)
