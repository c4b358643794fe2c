"""Tuning aid (GPU box): one training micro-step under torch.profiler; prints the PyTorch-side (non-library) ops by device time
with input shapes, to find stray copies / casts in the host code."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from llmseg_amd import synthetic  # noqa: E402
from llmseg_amd.lisa import LISAForCausalLM  # noqa: E402
from llmseg_amd.params import LisaConfig, LlamaConfig  # noqa: E402
from llmseg_amd.train import Trainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda", 0)
cfg = LisaConfig(backbone="sam", build_unused_towers=False)
cfg.llama = LlamaConfig(lora_r=8)
model = LISAForCausalLM(cfg, device=dev).init_random(seed=0)
model.prepare()
batch = synthetic.make_batch(B, img_size=1024, L=64, K=256, device=dev, seed=1234)
model.set_trainable()
trainer = Trainer(model, lr=3e-4, grad_accum=10)
for _ in range(2):
    trainer.micro_step(batch)
torch.cuda.synchronize()
STACK = os.environ.get("STACK") is not None          # STACK=1: group by python source location instead of input shapes
kw = {}
if STACK:
    kw["experimental_config"] = torch._C._profiler._ExperimentalConfig(verbose=True)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=STACK, **kw) as prof:
    trainer.micro_step(batch)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.device_time_total)
print(f"{'op':28s} {'calls':>6s} {'dev ms':>9s}  shapes")
for e in rows[:40]:
    print(f"{e.key:28s} {e.count:6d} {e.device_time_total / 1e3:9.2f}  {str(e.input_shapes)[:150]}")

if STACK:
    rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.device_time_total > 0 and e.key.startswith("aten::")]
    agg = {}
    for e in rows:
        where = next((f for f in e.stack if "llmseg_amd" in f and "torch/" not in f), e.stack[0] if e.stack else "?")
        k = where.split("llmseg_amd/")[-1]
        a = agg.setdefault(k, [0, 0.0, set()])
        a[0] += e.count; a[1] += e.device_time_total; a[2].add(e.key.replace("aten::", ""))
    print("\nby source line (device kernels of aten ops):")
    tot = 0
    for k, (n, t, ops_) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        tot += t
        print(f"{t / 1e3:7.3f} ms {n:5d} calls  {k[:70]:70s} {','.join(sorted(ops_))[:60]}")
    print("total listed", tot / 1e3, "ms")
