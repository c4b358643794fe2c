"""Diagnostic: per-tensor difference between the fused accumulation pass and k micro-steps (tests/backward_checks.py::check_fused_accum)."""
import sys
import torch
sys.path.insert(0, ".")
from tests import backward_checks as bc, model_checks as mc
from llmseg_amd.train import Trainer, merge_micro_batches

k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
p_drop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
cfg, m, sd, batch = bc._lora_case("sam", p_drop=p_drop)
names = [n for n, p in m.params.named_parameters() if p.requires_grad]
prm = dict(m.params.named_parameters())
batches = [mc._dev(b) for b in bc._variant_batches(batch, k)]
grab = lambda store: (lambda t, ss: store.update(g={n: prm[n]._g32.detach().clone() for n in names}, ss=float(ss)))
seq = {}
tr = Trainer(m, lr=0.0, grad_accum=k, warmup=1, total_steps=10)
tr.grad_hook = grab(seq)
m.set_dropout_seed(4242, 0)
for b in batches:
    tr.micro_step(b, m.make_plan(**b))
tr.close()
# the same sequential run again: the noise floor of "identical" runs (must be 0: deterministic)
seq2 = {}
tr = Trainer(m, lr=0.0, grad_accum=k, warmup=1, total_steps=10)
tr.grad_hook = grab(seq2)
m.set_dropout_seed(4242, 0)
for b in batches:
    tr.micro_step(b, m.make_plan(**b))
tr.close()
merged = merge_micro_batches(batches)
plan = m.make_plan(**merged, micro_batches=k)
fus = {}
tr = Trainer(m, lr=0.0, grad_accum=1, warmup=1, total_steps=10, fused_accum=k)
tr.grad_hook = grab(fus)
m.set_dropout_seed(4242, 0)
tr.micro_step(merged, plan)
tr.close()
rows = []
gmax = max(float(seq["g"][n].abs().max()) for n in names)
for n in names:
    a, b, a2 = seq["g"][n].double().flatten(), fus["g"][n].double().flatten(), seq2["g"][n].double().flatten()
    rms = float(a.pow(2).mean().sqrt())
    d = float((a - b).pow(2).mean().sqrt())
    cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
    rows.append((d / max(rms, 1e-30), n, rms, d, cos, float((a - a2).abs().max()), float(a.abs().max())))
rows.sort(reverse=True)
print(f"k={k} p_drop={p_drop} ss seq {seq['ss']:.6e} fused {fus['ss']:.6e}  global |g|max {gmax:.3e}")
for r in rows[:25]:
    print(f"rel {r[0]:.3e}  rms {r[2]:.3e}  d {r[3]:.3e}  cos {r[4]:.6f}  seq-vs-seq max {r[5]:.1e}  |g|max {r[6]:.2e}  {r[1]}")
print("...")
for r in rows[-5:]:
    print(f"rel {r[0]:.3e}  rms {r[2]:.3e}  d {r[3]:.3e}  cos {r[4]:.6f}  {r[1]}")
