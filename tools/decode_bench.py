"""Decode-step timing of the generation path at Llama-7B size, random weights: ms per generated token and the
HBM roofline of the weight stream (every decoder weight + lm_head read once per token: 13.2 GB at bf16).
usage: python tools/decode_bench.py [N=1] [new_tokens=24] [lora_r=8]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from llmseg_amd.lisa import LISAForCausalLM  # noqa: E402
from llmseg_amd.params import LisaConfig, LlamaConfig  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
NEW = int(sys.argv[2]) if len(sys.argv) > 2 else 24
R = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
cfg = LisaConfig(backbone="sam", build_unused_towers=False)
cfg.llama = LlamaConfig(lora_r=R)
m = LISAForCausalLM(cfg, device=dev).init_random(seed=0)
m.prepare()
g = torch.Generator().manual_seed(1)
ids = torch.randint(3, 31999, (N, 64), generator=g)
ids[:, 0] = 1; ids[:, 1] = 32001; ids[:, 2] = -200; ids[:, 3] = 32002
clip = torch.randn(N, 3, 224, 224, generator=g).to(dev, torch.bfloat16)
c = cfg.llama
wbytes = 2.0 * (c.layers * (4 * c.hidden * c.hidden + 3 * c.hidden * c.inter) + c.vocab * c.hidden)
for new in (2, NEW):                       # first call warms up; the difference of two lengths isolates the decode steps
    torch.cuda.synchronize(); t0 = time.perf_counter()
    seq, hid = m.generate(clip, ids, max_new_tokens=new, eos_token_id=None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"N={N} new={new}: {dt * 1e3:.1f} ms total")
    if new == 2:
        t2 = None
torch.cuda.synchronize(); t0 = time.perf_counter(); m.generate(clip, ids, max_new_tokens=2, eos_token_id=None); torch.cuda.synchronize(); t_a = time.perf_counter() - t0
torch.cuda.synchronize(); t0 = time.perf_counter(); m.generate(clip, ids, max_new_tokens=NEW, eos_token_id=None); torch.cuda.synchronize(); t_b = time.perf_counter() - t0
per = (t_b - t_a) / (NEW - 2)
print(f"decode: {per * 1e3:.2f} ms/token at N={N} ({N / per:.1f} tokens/s); weight stream {wbytes / 1e9:.2f} GB/token -> {wbytes / per / 1e12:.2f} TB/s "
      f"({100 * wbytes / per / 8e12:.1f} % of 8 TB/s); prefill+1 token {t_a * 1e3:.1f} ms")
