"""Experiment: per-wave section timing of the ping-pong GEMM (variant 10, LLMSEG_GEMM_DBG=129[+knobs])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from llmseg_amd import _lib, ops  # noqa: E402

lib = _lib.load()
M, N, K = 8192, 8192, 8192
a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
lib.llmseg_gemm_set_variant(10)
for _ in range(3):
    ops.gemm(a, w, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.gemm(a, w, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
v = out.view(torch.int32).flatten()[:128].cpu().view(4, 8, 4)
print("dbg", os.environ.get("LLMSEG_GEMM_DBG"), " per-phase averages in clocks (s_memtime ticks): MFMA-issue, closing-barrier wait, L+open-barrier")
for b in range(2):
    for wv in range(8):
        sM, sB, sL, tot = [int(x) for x in v[b, wv]]
        n = K // 64 * 4
        print(f"  wg{b} wave{wv}: M' {sM / n:7.1f}  B {sB / n:7.1f}  L {sL / n:7.1f}   tile ticks {tot}  (kernel {ms * 1e3:.0f} us = {M * N // 65536 // 256} tiles/CU)")
