"""GPU probe: trace every llmseg_amd.ops call (name, tensor arg shapes, checksum of every tensor argument AFTER the call) during the tiny
dinov2 fwd+bwd; write the trace to argv[1]; with argv[2] print the first calls whose checksums differ from that earlier trace."""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from llmseg_amd import ops  # noqa: E402
from oracle import cases  # noqa: E402
from tests import model_checks as mc  # noqa: E402

trace = []


def wrap(name, fn):
    def inner(*a, **k):
        r = fn(*a, **k)
        ts = [x for x in list(a) + list(k.values()) + (list(r) if isinstance(r, (tuple, list)) else [r]) if torch.is_tensor(x)]
        cs = [(tuple(t.shape), float(t.detach().double().abs().sum().cpu()) if t.numel() else 0.0) for t in ts]
        keep = [t.detach().float().cpu() for t in ts] if (name in ("gemm", "add_rows", "norm", "align_reg_loss", "embed_splice", "attention", "swiglu") and sum(t.numel() for t in ts) < 3e6) else None
        trace.append((name, cs, keep))
        return r
    return inner


for n, f in list(vars(ops).items()):
    if isinstance(f, types.FunctionType) and not n.startswith("_") and n not in ("prof_enable", "prof_collect"):
        setattr(ops, n, wrap(n, f))
cfg = cases.tiny_lisa_cfg("dinov2")
m, sd = mc.build_pair(cfg)
m.set_trainable()
batch = mc._round_batch(cases.tiny_lisa_batch(img_size=896))
out = m.model_forward(**mc._dev(batch), inference=False)
nf = len(trace)
out["loss"].backward()
torch.save((nf, trace), sys.argv[1])
print("calls:", nf, len(trace))
if len(sys.argv) > 2:
    nf0, t0 = torch.load(sys.argv[2])
    shown = 0
    for i, (a, b) in enumerate(zip(trace, t0)):
        if a[0] != b[0] or len(a[1]) != len(b[1]):
            print(i, "STRUCTURE", a[0], b[0]); break
        rel = max((abs(x[1] - y[1]) / (abs(y[1]) + 1e-9) for x, y in zip(a[1], b[1])), default=0.0)
        if rel > 1e-5 and i < nf:
            extra = ""
            if a[2] is not None and b[2] is not None:
                extra = " | per-tensor max|diff| / max|ref|: " + " ".join("%.2e/%.2e" % ((x - y).abs().max().item(), y.abs().max().item()) for x, y in zip(a[2], b[2]))
            print(i, "bwd" if i >= nf else "fwd", a[0], "rel %.3e" % rel, [s for s, _ in a[1]], extra)
            shown += 1
            if shown > 40:
                break
