"""Tuning aid (GPU box): the frozen SAM ViT-H encoder alone (graph replays), ms per pass.  tools/sam_only.py [images]  (LLMSEG_WIN_NO_GATHER=1 for the A/B)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from llmseg_amd.lisa import LISAForCausalLM  # noqa: E402
from llmseg_amd.params import LisaConfig, LlamaConfig  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda", 0)
cfg = LisaConfig(backbone="sam", build_unused_towers=False)
cfg.llama = LlamaConfig(layers=1)
model = LISAForCausalLM(cfg, device=dev).init_random(seed=0)
model.prepare()
img = torch.randn(B, 3, 1024, 1024, device=dev).to(torch.bfloat16)
with torch.no_grad():
    for _ in range(2):
        y = model._sam_encoder_cl(img)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = model._sam_encoder_cl(img)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20 if B <= 4 else 5
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
print(f"sam encoder, {B} images, gather={'off' if os.environ.get('LLMSEG_WIN_NO_GATHER') else 'on'}: {e0.elapsed_time(e1) / n:.3f} ms per pass  checksum {float(y.float().abs().sum()):.6e}")
