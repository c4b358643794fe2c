"""GPU box: WHERE does the head-gradient error of a failing full-depth seed come from?  (profiles/r06_spread_fulldepth_grads_seeds3-5.md: seeds 3 and 4
fail the per-tensor policy with HIP errors ~10 x one bf16-CPU draw on every mask-selection-head tensor; HIP = 0.96-0.99 x ref + a residual with cos 0.99.)

The full-depth training micro-step of tests/fulldepth_checks.py::check_full_depth_gradients(seed) is run with the head's inputs, outputs and the
gradients arriving at its outputs recorded (hooks without arithmetic).  Then, on the host:
  (1) the loss kernel alone: autograd through oracle.losses at the HIP path's OWN (emb, text, pred_iou) vs the kernel's d_e / d_t / d_pred;
  (2) the head backward alone: autograd through oracle.mask_head in fp32 -- and again in bf16 -- on the HIP path's own head inputs (bf16 pooled rows,
      bf16 text rows) with the HIP path's own upstream gradients, vs the head-parameter gradients the HIP backward left in the arena;
so a difference in (2) is the head backward's arithmetic (its bf16 intermediates), not anything upstream or downstream of it.
    python tools/probes/head_bwd_diag.py <seed>"""
import sys

import torch

sys.path.insert(0, ".")
from llmseg_amd import autograd as ag, lisa as hip_lisa, params as hp, synthetic      # noqa: E402
from llmseg_amd.train import GradArena                                                 # noqa: E402
from oracle import losses as olosses, mask_head as omh                                 # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev, BF = "cuda", torch.bfloat16
hcfg = hp.LisaConfig(backbone="sam", build_unused_towers=False)
hcfg.llama = hp.LlamaConfig(lora_r=8, lora_dropout=0.05)
m = hip_lisa.LISAForCausalLM(hcfg, device=dev).init_random(seed=11 + seed)
m.train(); m.set_trainable()
batch = synthetic.make_batch(2, img_size=1024, L=64, K=256, device=dev, seed=977 + seed, soft=True)
arena = GradArena(m)
m.set_dropout_seed(0x5EED1234, 3)
plan = m.make_plan(**batch)
rec = {}
orig_head = m._mask_head


def head(pooled, text, F, stacked=False):
    rec["pooled"], rec["text"], rec["stacked"] = pooled.detach().clone(), text.detach().clone(), stacked
    iou, emb = orig_head(pooled, text, F, stacked)
    rec["iou"], rec["emb"] = iou.detach().clone(), emb.detach().clone()
    iou.register_hook(lambda g: rec.__setitem__("d_iou", g.detach().clone()))
    emb.register_hook(lambda g: rec.__setitem__("d_emb", g.detach().clone()))
    return iou, emb


m._mask_head = head
orig_fwd = ag.AlignRegFn.forward


def ar_forward(ctx, e, t, pred, gt_iou, gt_iop):
    from llmseg_amd import ops
    out, d_e, d_t, d_p = ops.align_reg_loss(e, t, gt_iou, pred, gt_iop, want_grads=True)
    rec["ar"] = tuple(x.detach().clone() for x in (e, t, pred, gt_iou, gt_iop, out, d_e, d_t, d_p))
    ctx.save_for_backward(d_e, d_t, d_p)
    return out


ag.AlignRegFn.forward = staticmethod(ar_forward)
out = m.model_forward(**batch, inference=False, plan=plan)
out["loss"].backward()
torch.cuda.synchronize()
prm = dict(m.params.named_parameters())
head_names = [n for n in prm if ".lisa_" in n and prm[n].requires_grad and hasattr(prm[n], "_g32")]
hip_g = {n: prm[n]._g32.detach().float().cpu().clone() for n in head_names}
sd = {n: p.detach().float().cpu() for n, p in m.state_dict().items() if ".lisa_" in n}
rms = lambda x: float(x.double().pow(2).mean().sqrt())
cosv = lambda a, b: float((a.double().flatten() @ b.double().flatten()) / (a.double().norm() * b.double().norm() + 1e-300))
print(f"# head backward in isolation, full depth, seed {seed}\n")
print("losses (HIP):", {k: float(v) for k, v in out.items() if torch.is_tensor(v) and v.numel() == 1})

# (1) the loss kernel
e, t, pred, gi, gp, o, d_e, d_t, d_p = [x.float().cpu() for x in rec["ar"]]
R, K, D = e.shape
print(f"\n## (1) align / IoP loss kernel at the HIP path's own inputs ({R} items, K = {K}, D = {D}): autograd through oracle.losses in float64\n")
print("| item | align (kernel / oracle) | IoP (kernel / oracle) | d_e rel err | d_t rel err | d_pred rel err | sum_k dKL/dcos (must be 0) | rms cos | spread of cos over k | |e| mean |")
print("|---|---|---|---|---|---|---|---|---|---|")
for i in range(R):
    ei, ti, pi = e[i].double().requires_grad_(True), t[i:i + 1].double().requires_grad_(True), pred[i].double().reshape(K, 1).requires_grad_(True)
    la = olosses.softmax_align(ei, ti, gi[i].double().reshape(K, 1))
    lr = olosses.iop_regression(pi, gp[i].double().reshape(K, 1))
    ge, gt_ = torch.autograd.grad(la, (ei, ti))
    (gpd,) = torch.autograd.grad(lr, (pi,))
    cs = (ei / ei.norm(dim=-1, keepdim=True)) @ (ti / ti.norm()).t()
    print(f"| {i} | {float(o[i, 0]):.6f} / {float(la):.6f} | {float(o[i, 1]):.6f} / {float(lr):.6f} | {rms(d_e[i] - ge) / rms(ge):.2e} | {rms(d_t[i] - gt_[0]) / rms(gt_):.2e} | "
          f"{rms(d_p[i] - gpd.flatten()) / rms(gpd):.2e} | - | {rms(cs):.4f} | {float(cs.std()):.2e} | {float(ei.norm(dim=-1).mean()):.3f} |")

# (2) the head backward
pooled, text = rec["pooled"].float().cpu(), rec["text"].float().cpu()
d_iou, d_emb = rec["d_iou"].float().cpu(), rec["d_emb"].float().cpu()
Cn = text.shape[0]
Kh = pooled.shape[0] // Cn
print(f"\n## (2) head backward alone: {Cn} stacked (image, conversation) blocks of K = {Kh} rows; upstream gradients d_iou rms {rms(d_iou):.2e}, d_emb rms {rms(d_emb):.2e}\n")
print(f"row spread of the head input: rms over k of (pooled_k - mean_k pooled) / rms(pooled) = {rms(pooled.view(Cn, Kh, -1) - pooled.view(Cn, Kh, -1).mean(1, keepdim=True)) / rms(pooled):.2e} (bf16 resolution 3.9e-3)\n")


def oracle_grads(dt):
    w = {n: v.to(dt).clone().requires_grad_(True) for n, v in sd.items()}
    tot = 0.0
    for c in range(Cn):                         # the HIP pass stacks the blocks; the oracle's head takes one image (K rows) and its conversation rows
        iou, emb = omh.mask_head(w, "model.", pooled[c * Kh:(c + 1) * Kh].to(dt), text[c:c + 1].to(dt))
        tot = tot + (iou.reshape(-1).float() * d_iou[c * Kh:(c + 1) * Kh]).sum() + (emb.reshape(Kh, -1).float() * d_emb[c * Kh:(c + 1) * Kh]).sum()
    tot.backward()
    return {n: (w[n].grad.float() if w[n].grad is not None else torch.zeros_like(sd[n])) for n in w}


g32, g16 = oracle_grads(torch.float32), oracle_grads(BF)
print("| head tensor | rms(fp32 autograd) | HIP arena: rel err | cos | fit scale | bf16 torch autograd on the same inputs: rel err | cos |")
print("|---|---|---|---|---|---|---|")
for n in head_names:
    r = g32[n]
    if rms(r) < 1e-7:
        continue
    h, l = hip_g[n].reshape(r.shape), g16[n].reshape(r.shape)
    fit = float((h.double().flatten() @ r.double().flatten()) / (r.double().flatten() @ r.double().flatten()))
    print(f"| {n} | {rms(r):.2e} | {rms(h - r) / rms(r):.2e} | {cosv(h, r):.5f} | {fit:.4f} | {rms(l - r) / rms(r):.2e} | {cosv(l, r):.5f} |")
