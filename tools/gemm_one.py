"""Run one GEMM shape a few times (for rocprofv3 --pmc).  args: variant M N K [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llmseg_amd import _lib, ops
v, M, N, K = [int(x) for x in sys.argv[1:5]]
it = int(sys.argv[5]) if len(sys.argv) > 5 else 3
_lib.load().llmseg_gemm_set_variant(v)
a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
b = torch.randn(N, device="cuda").to(torch.bfloat16)
o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(it):
    ops.gemm(a, w, bias=b, out=o)
torch.cuda.synchronize()
