"""GPU probe: dump every parameter gradient of the tiny dinov2 model (plain .grad path) to a file; run twice under different env switches and diff."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from oracle import cases  # noqa: E402
from tests import model_checks as mc  # noqa: E402

cfg = cases.tiny_lisa_cfg("dinov2")
m, sd = mc.build_pair(cfg)
m.set_trainable()
batch = mc._round_batch(cases.tiny_lisa_batch(img_size=896))
out = m.model_forward(**mc._dev(batch), inference=False)
out["loss"].backward()
g = {n: p.grad.detach().float().cpu() for n, p in m.params.named_parameters() if p.grad is not None}
g["__loss"] = out["loss"].detach().float().cpu()
torch.save(g, sys.argv[1])
if len(sys.argv) > 2:
    o = torch.load(sys.argv[2])
    rows = []
    for n in g:
        d = (g[n] - o[n]).abs().max().item()
        rows.append((d / (o[n].abs().max().item() + 1e-12), n, d, o[n].abs().max().item()))
    for r in sorted(rows, reverse=True)[:25]:
        print("rel %.3e  %-70s maxdiff %.3e |ref| %.3e" % r)
