"""GPU probe: run the tiny dinov2 model fwd+bwd with EVERY ops.gemm call checked against an fp32 torch product of the same operands."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from llmseg_amd import ops  # noqa: E402
from oracle import cases  # noqa: E402
from tests import model_checks as mc  # noqa: E402

orig = ops.gemm
bad = []


def checked(a, w, bias=None, act=ops.ACT_NONE, residual=None, gamma=None, out=None, alpha=1.0, out_f32=False, trans_a=False, trans_w=False,
            a2=None, w2=None, accumulate=False, **kw):
    prev = out.detach().float().clone() if (out is not None and accumulate) else None
    res_c = residual.detach().float().clone() if residual is not None else None
    r = orig(a, w, bias=bias, act=act, residual=residual, gamma=gamma, out=out, alpha=alpha, out_f32=out_f32, trans_a=trans_a, trans_w=trans_w,
             a2=a2, w2=w2, accumulate=accumulate, **kw)
    if a2 is None and gamma is None and not kw:
        A = a.float().t() if trans_a else a.float()
        W = w.float() if trans_w else w.float().t()
        ref = alpha * (A @ W)
        if bias is not None:
            ref = ref + bias.float()
        if act == ops.ACT_RELU:
            ref = ref.relu()
        elif act == ops.ACT_SIGMOID:
            ref = ref.sigmoid()
        elif act != ops.ACT_NONE:
            return r
        if res_c is not None:
            ref = ref + res_c
        if prev is not None:
            ref = ref + prev
        e = (r.float() - ref).abs().max().item()
        tol = 2.0 ** -7 * ref.abs().max().item() + 1e-6
        if not e <= tol:
            bad.append((tuple(a.shape), tuple(w.shape), trans_a, trans_w, accumulate, e, tol))
            print("MISMATCH", bad[-1], flush=True)
    return r


ops.gemm = checked
cfg = cases.tiny_lisa_cfg("dinov2")
m, sd = mc.build_pair(cfg)
m.set_trainable()
batch = mc._round_batch(cases.tiny_lisa_batch(img_size=896))
out = m.model_forward(**mc._dev(batch), inference=False)
out["loss"].backward()
torch.cuda.synchronize()
print("checked; mismatches:", len(bad))
