"""Time the rank-8 down projection (two branches, dropout on) at the 2- and 24-image shapes.  usage: python tools/probes/lora_down_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from llmseg_amd import ops
torch.manual_seed(0)
for M in (638, 7656):
    x = torch.randn(M, 4096, device="cuda").to(torch.bfloat16)
    w, w2 = (torch.randn(8, 4096, device="cuda") * 0.02).to(torch.bfloat16), (torch.randn(8, 4096, device="cuda") * 0.02).to(torch.bfloat16)
    drop = (torch.tensor([7, 0], device="cuda", dtype=torch.int64), 3, 0.05)
    f = lambda: ops.lora_down(x, w, x2=x, w2=w2, drop=drop)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = f()
    e1.record(); torch.cuda.synchronize()
    print(f"M={M}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (launch + finish), checksum {y.float().sum().item():.4f}")
