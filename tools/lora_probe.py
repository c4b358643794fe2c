"""Probe: K-sliced lora_down vs a float reference, call by call in a fresh process (cold-run behaviour)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llmseg_amd import ops
dev = torch.device("cuda", 0); BF = torch.bfloat16
torch.manual_seed(0)
for (M, H) in ((834, 256), (638, 4096)):
    x = (torch.randn(M, H, device=dev) * 0.5).to(BF); aq = (torch.randn(8, H, device=dev) * 0.1).to(BF); av = (torch.randn(8, H, device=dev) * 0.1).to(BF)
    ref = torch.cat([x.float() @ aq.float().t(), x.float() @ av.float().t()], 1)
    d = (torch.randn(M, 3 * H, device=dev) * 0.5).to(BF)
    refb = torch.cat([d[:, :H].float() @ aq.float().t(), d[:, 2 * H:].float() @ av.float().t()], 1) * 2.0
    for i in range(4):
        junk = torch.full((1 << 22,), float("nan"), device=dev)      # poison recycled memory
        del junk
        a2 = torch.empty(M, 64, device=dev, dtype=BF)
        ops.lora_down(x, aq, out=a2, zero_cols=48, x2=x, w2=av)
        t2 = torch.empty(M, 64, device=dev, dtype=BF)
        ops.lora_down(d[:, :H], aq, alpha=2.0, out=t2, zero_cols=48, x2=d[:, 2 * H:], w2=av)
        torch.cuda.synchronize()
        print(M, H, i, "fwd err", float((a2[:, :16].float() - ref).abs().max()), "zeros ok", bool((a2[:, 16:] == 0).all()),
              "bwd err", float((t2[:, :16].float() - refb).abs().max()), "nan", bool(torch.isnan(a2.float()).any() or torch.isnan(t2.float()).any()))
