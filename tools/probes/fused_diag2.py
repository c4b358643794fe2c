"""Where do the fused pass and the k micro-steps part?  Forward values and gradients at the head's boundary, item by item."""
import sys
import torch
sys.path.insert(0, ".")
from tests import backward_checks as bc, model_checks as mc
from llmseg_amd.train import GradArena, merge_micro_batches
from llmseg_amd import autograd as ag

k = 3
cfg, m, sd, batch = bc._lora_case("sam", p_drop=0.0)
batches = [mc._dev(b) for b in bc._variant_batches(batch, k)]
arena = GradArena(m)
store = {}
orig_head = m._mask_head


def run(tag, b, plan):
    rec = store.setdefault(tag, {"emb": [], "iou": [], "text": [], "pooled": [], "d_emb": [], "d_iou": [], "d_text": [], "d_pooled": []})

    def head(pooled, text, F, stacked=False):
        if pooled.requires_grad:
            pooled.register_hook(lambda g: rec["d_pooled"].append(g.detach().float().clone()))
        text.register_hook(lambda g: rec["d_text"].append(g.detach().float().clone()))
        iou, emb = orig_head(pooled, text, F, stacked)
        rec["emb"].append(emb.detach().float().clone()); rec["iou"].append(iou.detach().float().clone())
        rec["text"].append(text.detach().float().clone()); rec["pooled"].append(pooled.detach().float().clone())
        emb.register_hook(lambda g: rec["d_emb"].append(g.detach().float().clone()))
        iou.register_hook(lambda g: rec["d_iou"].append(g.detach().float().clone()))
        return iou, emb
    m._mask_head = head
    arena.zero_()
    out = m.model_forward(**b, inference=False, plan=plan)
    out["loss"].backward()
    torch.cuda.synchronize()
    m.__dict__.pop("_mask_head", None)
    return {kk: float(v) for kk, v in out.items() if torch.is_tensor(v) and v.numel() == 1}


losses = [run("seq", b, m.make_plan(**b)) for b in batches]
merged = merge_micro_batches(batches)
lf = run("fus", merged, m.make_plan(**merged, micro_batches=k))
print("losses seq", [round(l["loss"], 4) for l in losses], "sum", sum(l["loss"] for l in losses), "fused", lf["loss"])
cat = lambda xs: torch.cat([x.reshape(-1, x.shape[-1]) if x.dim() > 1 else x.reshape(-1, 1) for x in xs], 0)
for key in ("pooled", "text", "emb", "iou", "d_emb", "d_iou", "d_text"):
    a, b = cat(store["seq"][key]), cat(store["fus"][key])
    print(f"{key:8s} shapes {tuple(a.shape)} {tuple(b.shape)}", end=" ")
    if a.shape == b.shape:
        d = (a - b)
        print(f"max|a| {a.abs().max():.3e} max diff {d.abs().max():.3e} rel rms {float(d.pow(2).mean().sqrt() / (a.pow(2).mean().sqrt() + 1e-30)):.3e}")
        if key in ("d_emb", "d_iou", "d_text"):
            n_items = 9
            per = a.shape[0] // n_items
            print("   per item rel rms:", [f"{float((d[i*per:(i+1)*per]).pow(2).mean().sqrt() / (a[i*per:(i+1)*per].pow(2).mean().sqrt() + 1e-30)):.2e}" for i in range(n_items)])
    else:
        print()
