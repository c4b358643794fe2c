"""GPU: the FULL-DEPTH model at BASELINE configs[1] size -- Llama-7B (32 layers, LoRA r = 8) + CLIP-L/14 + SAM ViT-H (32 blocks), one
1024 x 1024 image, 64-token prompt, K = 256 proposals, inference -- HIP vs the fp32 oracle on the host cores (reference
`model/LISA.py:225-474`).  Measures what the tiny-depth and single-layer tests cannot: error growth over 32 bf16 layers.

Weights: random-init on the device (bf16), the SAME tensors handed to the oracle (a lazy dict widens one tensor at a time, so the host
holds 15 GB of bf16 instead of 31 GB of fp32).  Scale of every tolerance: what the oracle itself loses when it runs in bf16 on the CPU
(the reference's own arithmetic), eps_cpu.  ~1 min of host time on the 128-core box.  VERDICT r2 item 1a."""
import time

import torch

from llmseg_amd import lisa as hip_lisa, params as hp, synthetic
from oracle import lisa as olisa, llama as ol, sam_encoder as osam, vit as ovit

BF = torch.bfloat16
# tolerance = max(floor, K_CPU x the bf16-CPU oracle's own error): HIP and the bf16-CPU oracle are two independent draws of bf16 rounding noise
# pushed through 32 + 32 layers and ReLU heads (heavy tails); across three kernel revisions of the SAME arithmetic (different fp32 summation
# orders only) the HIP max error of pred_iou moved between 3.1e-3 and 5.0e-3 against the oracle's 3.1e-3, so 1.5 x was a coin toss.
# Round 6: the inference scores come from the fp32-activation head (csrc/head_f32.hip) on the bf16 trunk outputs -- the bf16 head was ~40 % of the
# score error (profiles/r04b_spread_fulldepth.md) -- and the multiplier is back from 2.5 to 2.0.
K_CPU = 2.0


class _LazyState(dict):
    """bf16 host tensors, widened to `dtype` on access."""

    def __init__(self, src, dtype):
        super().__init__(src)
        self.dtype = dtype

    def __getitem__(self, k):
        return dict.__getitem__(self, k).to(self.dtype)

    def get(self, k, default=None):
        return self[k] if k in self else default


def _e(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()


def check_full_depth_inference(K=256, L=64, with_bf16_cpu=True, log=print, seed=0, raw=None):
    """seed: weights (init_random(5 + seed)) and batch (4321 + seed) of this draw; raw: a list that receives (name, HIP err, bf16-CPU err) per
    output (tools/probes/spread.py builds the multi-seed table that justifies K_CPU from it)."""
    dev = "cuda"
    hcfg = hp.LisaConfig(backbone="sam", build_unused_towers=False)
    hcfg.llama = hp.LlamaConfig(lora_r=8)
    m = hip_lisa.LISAForCausalLM(hcfg, device=dev).init_random(seed=5 + seed)
    m.eval()
    ocfg = olisa.LisaCfg(llama=ol.LlamaCfg(lora_r=8), clip=ovit.VitCfg(eps=1e-5, img=224), sam=osam.SamCfg(), backbone="sam")
    batch = synthetic.make_batch(1, img_size=1024, L=L, K=K, device=dev, seed=4321 + seed, soft=True)
    inf = dict(images=batch["images"], images_clip=batch["images_clip"], input_ids=batch["input_ids"], labels=None,
               attention_masks=batch["attention_masks"], offset=batch["offset"], sam_segs_list=batch["sam_segs_list"])
    with torch.no_grad():
        got = m.model_forward(**inf, inference=True, return_aux=True)
    torch.cuda.synchronize()
    host = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    cpu = lambda t, dt: t.detach().cpu().to(dt) if (torch.is_tensor(t) and t.is_floating_point()) else (t.cpu() if torch.is_tensor(t) else t)

    def run(dt):
        b = {k: ([cpu(t, dt) for t in v] if isinstance(v, list) else cpu(v, dt)) for k, v in inf.items()}
        t0 = time.perf_counter()
        with torch.no_grad():
            out = olisa.model_forward(_LazyState(host, dt), ocfg, **b, inference=True, return_aux=True)
        return out, time.perf_counter() - t0
    ref, t_ref = run(torch.float32)
    log(f"full depth: fp32 oracle forward {t_ref:.1f} s on {torch.get_num_threads()} threads")
    lo = None
    if with_bf16_cpu:
        lo, t_lo = run(BF)
        log(f"full depth: bf16 oracle forward {t_lo:.1f} s")
    res = []

    def add(name, g_, r_, l_, floor, rel=True):
        scale = max(1.0, r_.abs().max().item()) if rel else 1.0
        lo_e = _e(l_, r_) if l_ is not None else 0.0
        if raw is not None:
            raw.append((name, _e(g_, r_), lo_e))
        res.append((f"full-depth {name} (|ref| {r_.abs().max().item():.3g}, bf16-CPU err {lo_e:.2e}, flat-1e-3 {'met' if _e(g_, r_) <= 1e-3 else 'NOT met'})",
                    _e(g_, r_), max(floor * scale, K_CPU * lo_e)))
    L_ = lambda k: None if lo is None else lo[k]
    B, C, g, _ = ref["feats"].shape
    rf = ref["feats"].permute(0, 2, 3, 1).reshape(B * g * g, C)
    lf = None if lo is None else lo["feats"].permute(0, 2, 3, 1).reshape(B * g * g, C)
    add("SAM ViT-H features (32 blocks)", got["feats"].view(B * g * g, C), rf, lf, 3e-2)
    add("Llama hidden (32 layers, post-norm)", got["hidden"], ref["hidden"], L_("hidden"), 3e-2)
    add("logits", got["logits"], ref["logits"], L_("logits"), 3e-2)
    add("pred_embedding", got["pred_embeddings"][0], ref["pred_embeddings"][0], None if lo is None else lo["pred_embeddings"][0], 3e-2)
    add("pred_similarity", got["pred_similarity"][0], ref["pred_similarity"][0], None if lo is None else lo["pred_similarity"][0], 1e-3, rel=False)
    add("pred_iou", got["pred_iou"][0], ref["pred_iou"][0], None if lo is None else lo["pred_iou"][0], 1e-3, rel=False)
    # A/B: how much of the score error is the bf16 HEAD and how much the bf16 TRUNK?  The oracle's fp32 upsample + mask
    # pooling + mask-selection head + cosine evaluated on the HIP path's own trunk outputs (SAM features, [SEG] embedding): whatever error
    # remains against the all-fp32 oracle comes from the trunk alone.  Informational lines (tolerance = the same bound as the HIP head's).
    from oracle import mask_head as omh
    with torch.no_grad():
        hf = got["feats"].detach().float().cpu().view(B, g, g, C).permute(0, 3, 1, 2)
        up = omh.upsample_feats(hf, 256)
        pooled = omh.mask_pooling(up[0], inf["sam_segs_list"][0].detach().float().cpu())
        pe = got["pred_embeddings"][0].detach().float().cpu()
        iou32, emb32 = omh.mask_head(_LazyState(host, torch.float32), "model.", pooled, pe)
        sim32 = omh.cosine_scores(pe, emb32[0])
    m.fp32_head = False
    with torch.no_grad():
        got16 = m.model_forward(**inf, inference=True)          # the bf16 MFMA head of rounds 1-5 on the same trunk (A/B)
    m.fp32_head = True
    for nm, a32, key, idx in (("pred_similarity", sim32, "pred_similarity", 4), ("pred_iou", iou32[0].t(), "pred_iou", 5)):
        e32, eh, e16 = _e(a32, ref[key][0]), res[idx][1], _e(got16[key][0], ref[key][0])
        if raw is not None:
            raw.append((nm + " [bf16 head on the same trunk]", e16, res[idx][1]))
        # round 6: the HIP path RUNS the fp32 head; the oracle's fp32 pooling + head + cosine on the HIP trunk outputs must reproduce its scores
        res.append((f"full-depth {nm}: the HIP fp32 head vs the oracle's fp32 pool / head / cosine on the same HIP trunk outputs "
                    f"(errors vs the all-fp32 oracle: fp32 head {eh:.2e}, host fp32 head on the HIP trunk {e32:.2e}, bf16 head {e16:.2e})", _e(a32, got[key][0]), 5e-5))
    # next-token agreement over the text positions: arg-max of the HIP logits vs the fp32 oracle, with the bf16-CPU oracle's own
    # agreement as the yardstick (random weights give flat logits, so ties flip easily: the yardstick, not 100 %, is the bar)
    am_r, am_g = ref["logits"].float().argmax(-1), got["logits"].float().cpu().argmax(-1)
    agree = (am_r == am_g).float().mean().item()
    agree_lo = (am_r == lo["logits"].float().argmax(-1)).float().mean().item() if lo is not None else 0.9
    res.append((f"full-depth next-token arg-max agreement with the fp32 oracle = {agree:.4f} (bf16-CPU oracle: {agree_lo:.4f}); shown as 1 - agreement",
                1.0 - agree, max(0.02, K_CPU * (1.0 - agree_lo))))
    # the proposal `validate` would select (arg-max similarity): random weights put all 256 similarities within a few 1e-3 of each other,
    # so "the same index" is not a meaningful bar; the bar is that the HIP pick is, under the fp32 oracle, as good as the oracle's own
    # pick to within the similarity tolerance above (regret), with the bf16-CPU oracle's regret printed beside it
    rs = ref["pred_similarity"][0].flatten().float()
    pick = got["pred_similarity"][0].float().cpu().flatten().argmax().item()
    regret = (rs.max() - rs[pick]).item()
    regret_lo = (rs.max() - rs[lo["pred_similarity"][0].flatten().float().argmax().item()]).item() if lo is not None else 0.0
    res.append((f"full-depth regret of the selected proposal under the fp32 oracle (same index: {pick == rs.argmax().item()}; bf16-CPU oracle's regret {regret_lo:.2e})",
                regret, 2.0 * res[4][2]))
    del m
    torch.cuda.empty_cache()
    return res


class _GradState(dict):
    """bf16 host tensors widened to `dtype` on access; the trainable ones are persistent LEAVES (requires_grad) so that autograd through the
    oracle leaves their gradients in `.leaves[name].grad`."""

    def __init__(self, src, dtype, trainable):
        super().__init__(src)
        self.dtype = dtype
        self.leaves = {n: dict.__getitem__(self, n).to(dtype).clone().requires_grad_(True) for n in trainable}

    def __getitem__(self, k):
        return self.leaves[k] if k in self.leaves else dict.__getitem__(self, k).to(self.dtype)

    def get(self, k, default=None):
        return self[k] if k in self else default


def check_full_depth_gradients(K=256, L=64, B=2, log=print, seed=0, table=None, dump=None, hip_only=False, force_gates=True):
    """BASELINE configs[2] at ITS OWN depth: 32-layer Llama-7B with LoRA r = 8 + dropout 0.05 on q / v, CLIP-L, 32-block SAM
    ViT-H, B = 2 images of 1024 x 1024, 64-token prompts, K = 256 proposals -- ONE training micro-step into the fp32 gradient arena -- against
    autograd through `oracle.lisa.model_forward` on the host with the same dropout masks (reference `model/LISA.py:225-474`, `training.py:546`).
    Yardstick: the same oracle run in bf16 on the CPU (the reference's own arithmetic).  Policy: `tests/backward_checks.py::grad_err` (flipped
    ReLU gates excluded row-wise, RMS / max <= max(3 %, 4 x bf16-CPU)) + the distribution bound over all compared tensors.
    Compared: LoRA A / B (q and v) of layers 0 / 15 / 31, the touched rows of `embed_tokens` (all other rows must be exactly zero), `lm_head`
    on the label rows + every 97th row, `text_hidden_fcs`, every trainable `lisa_*` tensor.  table: a list that receives markdown rows."""
    from llmseg_amd.train import GradArena
    from tests.backward_checks import GateTrace, grad_err, ratio_summary
    dev = "cuda"
    hcfg = hp.LisaConfig(backbone="sam", build_unused_towers=False)
    hcfg.llama = hp.LlamaConfig(lora_r=8, lora_dropout=0.05)
    m = hip_lisa.LISAForCausalLM(hcfg, device=dev).init_random(seed=11 + seed)
    m.train()
    m.set_trainable()
    ocfg = olisa.LisaCfg(llama=ol.LlamaCfg(lora_r=8, lora_dropout=0.05), clip=ovit.VitCfg(eps=1e-5, img=224), sam=osam.SamCfg(), backbone="sam")
    batch = synthetic.make_batch(B, img_size=1024, L=L, K=K, device=dev, seed=977 + seed, soft=True)
    names = [n for n, p in m.params.named_parameters() if p.requires_grad]
    host = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    drop = (0x5EED1234, 3)
    gt = GateTrace()
    cpu = lambda t, dt: t.detach().cpu().to(dt) if (torch.is_tensor(t) and t.is_floating_point()) else (t.cpu() if torch.is_tensor(t) else t)
    keys = ("images", "images_clip", "input_ids", "labels", "attention_masks", "offset", "sam_segs_list", "sam_ious_list", "sam_iops_list")

    def run(side, dt):
        sd = _GradState(host, dt, names)
        b = {k: ([cpu(t, dt if k == "sam_segs_list" else torch.float64 if t.dtype == torch.float64 else torch.float32) for t in batch[k]]
                 if isinstance(batch[k], list) else cpu(batch[k], dt)) for k in keys}
        t0 = time.perf_counter()

        def fb():
            o = olisa.model_forward(sd, ocfg, **b, inference=False, dropout_state=drop)
            o["loss"].backward()
            return o
        out = gt.oracle(side, fb)
        log(f"full depth gradients: {side} oracle ({dt}) forward + backward {time.perf_counter() - t0:.1f} s on {torch.get_num_threads()} threads")
        return {k: float(out[k].detach()) for k in ("loss", "ce_loss", "align_loss", "regression_loss")}, {n: sd.leaves[n].grad for n in names}

    # the HIP micro-step first (its memory is released before the host holds two sets of gradients)
    arena = GradArena(m)
    m.set_dropout_seed(*drop)
    plan = m.make_plan(**batch)
    gt.hip_begin(m)
    out = m.model_forward(**batch, inference=False, plan=plan)
    out["loss"].backward()
    torch.cuda.synchronize()
    gt.hip_collect(m)
    gt.hip_end(m)
    prm = dict(m.params.named_parameters())
    hip_loss = {k: float(out[k].detach()) for k in ("loss", "ce_loss", "align_loss", "regression_loss")}
    ids = batch["input_ids"].cpu()
    touched = torch.unique(ids[ids >= 0])
    lab = batch["labels"].cpu()
    lm_rows = torch.unique(torch.cat([lab[lab >= 0], torch.arange(0, hcfg.llama.vocab, 97)]))
    pick = [n for n in names if (".lora_" in n and any(f"layers.{i}." in n for i in (0, 15, 31))) or "text_hidden_fcs" in n or ".lisa_" in n]
    hip_g = {n: prm[n]._g32.detach().float().cpu() for n in pick}
    emb_g = prm["model.embed_tokens.weight"]._g32.detach()
    untouched = torch.ones(emb_g.shape[0], dtype=torch.bool, device=emb_g.device)
    untouched[touched.to(emb_g.device)] = False
    emb_clean = float(emb_g[untouched].abs().max())
    hip_g["model.embed_tokens.weight"] = emb_g[touched.to(emb_g.device)].float().cpu()
    hip_g["lm_head.weight"] = prm["lm_head.weight"]._g32.detach()[lm_rows.to(emb_g.device)].float().cpu()
    arena.detach()
    del m, arena, prm, out, emb_g
    torch.cuda.empty_cache()
    if dump is not None:                               # (tools/probes/fd_seed_diag.py: the tensors themselves, for cross-comparisons between runs)
        dump["hip"] = {n: hip_g[n].clone() for n in pick}
        dump["hip_loss"] = hip_loss
    if hip_only:
        return []

    # Round 6: both oracle runs differentiate the piecewise-linear branch the HIP forward took (its recorded ReLU gates are imposed: `oracle.mask_head.force_gates`).
    # Why: at this size the 256 proposal rows of an image differ by less than one bf16 step (row spread 1.6e-3 of their magnitude), so a head unit whose
    # pre-activation is within rounding noise of zero flips for ALL rows of the image at once; every such coherent flip moves all upstream head gradients
    # by several per cent.  Seeds 3 and 4 of the spread (profiles/r06_spread_fulldepth_grads_seeds3-5.md) failed that way at 10 x the bf16-CPU draw, while
    # the head backward ALONE -- HIP vs fp32 autograd on the HIP path's own head inputs and upstream gradients -- is within 1.2-5 % and TWICE as close as
    # torch's bf16 autograd on the same inputs, and the loss kernel's gradients match float64 to 1e-6 (profiles/r06_head_bwd_diag_seed{0,4}.md).  Either
    # subgradient at a kink is valid; what the comparison has to pin is the arithmetic on the branch taken.  force_gates=False restores the round-5 form.
    from oracle import mask_head as omh_
    if force_gates:
        omh_.force_gates({n: torch.cat(v, 0) for n, v in gt.gates["hip"].items()})
    try:
        ref_loss, ref_g = run("ref", torch.float32)
        if force_gates:
            omh_.force_gates({n: torch.cat(v, 0) for n, v in gt.gates["hip"].items()})
        lo_loss, lo_g = run("lo", BF)
    finally:
        omh_.force_gates(None)
    flips = gt.flipped_rows()
    if dump is not None:
        dump["ref"] = {n: ref_g[n].detach().float().clone() for n in pick}
        dump["lo"] = {n: lo_g[n].detach().float().clone() for n in pick}
        dump["ref_loss"], dump["lo_loss"] = ref_loss, lo_loss
        dump["flips"] = {k: v.clone() for k, v in flips.items()}
    res = []
    for k in ("ce_loss", "align_loss", "regression_loss", "loss"):
        r = ref_loss[k]
        res.append((f"full-depth train {k} (ref {r:.4f}, bf16-CPU err {abs(lo_loss[k] - r):.2e})", abs(hip_loss[k] - r), max(5e-3 * max(1.0, abs(r)), K_CPU * abs(lo_loss[k] - r))))
    res.append(("full-depth embed_tokens gradient outside the touched rows (must be exactly 0)", emb_clean, 0.0))
    sub = {"model.embed_tokens.weight": touched, "lm_head.weight": lm_rows}
    stats = []
    for n in pick + ["model.embed_tokens.weight", "lm_head.weight"]:
        r, l = ref_g[n], lo_g[n].float()
        if n in sub:
            r, l = r[sub[n]], l[sub[n]]
        st = []
        ratio, desc = grad_err(hip_g[n], r, l, floor=3e-4, skip_rows=GateTrace.rows_for(n, flips), stats=st)
        stats.append(st[0])
        tag = f" ({len(sub[n])} rows)" if n in sub else ""
        res.append((f"full-depth arena grad {n}{tag}: {desc}; shown as err / tol", ratio, 1.0))
        if table is not None:
            s = st[0]
            table.append(f"| `{n}`{tag} | {s[4]:.2e} | {s[0]:.2e} | {s[1]:.2e} | {s[0] / max(s[1], 1e-30):.2f} | {s[2]:.2e} | {s[3]:.2e} | {s[2] / max(s[3], 1e-30):.2f} | {s[6]} | {ratio:.2f} |")
    res += ratio_summary(stats, "full-depth arena grads")
    return res
