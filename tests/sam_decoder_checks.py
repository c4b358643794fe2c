"""GPU: SAM prompt encoder (text path) + mask decoder + post-processing (llmseg_amd/sam_decoder.py) against the oracle restatement
(oracle/sam_decoder.py, pinned bit-exactly against the imported reference modules by tests/golden/sam_decoder.pt)."""
import torch

from llmseg_amd import lisa as hip_lisa, ops
from oracle import cases, sam_decoder as osd
from tests import model_checks as mc

DEV = "cuda"
BF = torch.bfloat16


def _model():
    cfg = cases.tiny_lisa_cfg("sam")
    hcfg = mc.to_hip_cfg(cfg)
    hcfg.sam_decoder = True
    m = hip_lisa.LISAForCausalLM(hcfg, device=DEV).init_random(seed=1)
    sd = {k: v.to(BF).float() for k, v in cases.sam_decoder_state().items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected[:3]
    return m, sd


def check_sam_decoder():
    m, sd = _model()
    emb, text = cases.sam_decoder_case(b=2)
    emb, text = emb.to(BF).float(), text.to(BF).float()
    res = []
    with torch.no_grad():
        r_low, r_iou = osd.decode_masks(sd, emb, text)
        lo_sd = {k: v.to(BF) for k, v in sd.items()}
        l_low, l_iou = osd.decode_masks(lo_sd, emb.to(BF), text.to(BF))                  # the reference's own bf16 arithmetic on the CPU
        f_cl = emb[0].reshape(256, 4096).t().contiguous().to(DEV, BF)
        low, iou = m.sam_decode(f_cl, text.to(DEV, BF))
        # nested row order -> raster, for the comparison only
        raster = low.view(2, 64, 64, 2, 2, 2, 2).permute(0, 1, 3, 5, 2, 4, 6).reshape(2, 256, 256).cpu()
        e_lo = (l_low.float()[:, 0] - r_low[:, 0]).abs().max().item()
        scale = max(1.0, r_low.abs().max().item())
        res.append((f"sam decoder low-res masks (bf16-CPU err {e_lo:.2e}, |ref| {r_low.abs().max().item():.2e})", (raster - r_low[:, 0]).abs().max().item(),
                    max(2e-2 * scale, 1.5 * e_lo)))
        e_io = (l_iou.float() - r_iou).abs().max().item()
        res.append((f"sam decoder iou prediction (bf16-CPU err {e_io:.2e})", (iou.float().cpu() - r_iou).abs().max().item(), max(2e-2, 1.5 * e_io)))
        # post-processing alone, fp32 in / fp32 out: raster input (nested = 0) and the decoder's own nested output
        for inp, orig in (((683, 1024), (427, 640)), ((1024, 1024), (1024, 1024)), ((1024, 768), (1365, 1024)), ((512, 1024), (7, 13))):
            ref = osd.postprocess_masks(r_low, inp, orig)[:, 0]
            got = ops.sam_postprocess(r_low[:, 0].reshape(2, 65536).contiguous().to(DEV), inp, orig, nested=False).cpu()
            res.append((f"sam postprocess {inp} -> {orig} (fp32)", (got - ref).abs().max().item(), 2e-6 * scale))
        nested = r_low[:, 0].view(2, 64, 2, 2, 64, 2, 2).permute(0, 1, 4, 2, 5, 3, 6).reshape(2, 65536).contiguous()
        got = ops.sam_postprocess(nested.to(DEV), (683, 1024), (427, 640), nested=True).cpu()
        res.append(("sam postprocess nested row order", (got - osd.postprocess_masks(r_low, (683, 1024), (427, 640))[:, 0]).abs().max().item(), 2e-6 * scale))
        # binary masks of the whole mask path agree except where the logit is within the bf16 error of zero
        pm = ops.sam_postprocess(low, (683, 1024), (427, 640)).cpu()
        rm = osd.postprocess_masks(r_low, (683, 1024), (427, 640))[:, 0]
        flip = ((pm > 0) != (rm > 0)) & (rm.abs() > max(2e-2 * scale, 1.5 * e_lo))
        res.append(("sam decoder + postprocess: sign flips outside the bf16 band", float(flip.sum()), 0.0))
    return res


def check_evaluate():
    """evaluate() end to end on the tiny configuration (Llama 2 layers, SAM dim 160 / depth 2 at 1024 x 1024): token choices where
    decidable, mask logits against the oracle with the bf16-CPU error as the scale, the no-[SEG] sequence, the resize bookkeeping."""
    from oracle import generate as ogen
    cfg = cases.tiny_lisa_cfg("sam")
    sd = cases.tiny_lisa_state(cfg)
    sd.update(cases.sam_decoder_state())
    sd = {k: v.to(BF).float() for k, v in sd.items()}
    hcfg = mc.to_hip_cfg(cfg)
    hcfg.sam_decoder = True
    m = hip_lisa.LISAForCausalLM(hcfg, device=DEV)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing[:3], unexpected[:3])
    batch = mc._round_batch(cases.tiny_lisa_batch(img_size=cfg.sam.img))
    clip, images, ids = batch["images_clip"][:2], batch["images"][:2], batch["input_ids"][:2].clone()
    ids[1][ids[1] == cfg.seg_token_idx] = 77                               # sequence 1 carries no [SEG] token in its prompt
    resize, orig = [(683, 1024), (1024, 768)], [(427, 640), (96, 72)]
    res = []
    with torch.no_grad():
        r_ids, r_masks, aux = ogen.evaluate(sd, cfg, clip, images, ids, resize, orig, max_new_tokens=3, eos_token_id=None)
        lo_sd = {k: v.to(BF) for k, v in sd.items()}
        l_ids, l_masks, _ = ogen.evaluate(lo_sd, cfg, clip.to(BF), images.to(BF), ids, resize, orig, max_new_tokens=3, eos_token_id=None)
        g_ids, g_masks = m.evaluate(clip.to(DEV), images.to(DEV), ids.to(DEV), resize, orig, max_new_tokens=3, eos_token_id=None)
    assert len(g_masks) == 2 and g_masks[0].shape == r_masks[0].shape and g_masks[1].shape == r_masks[1].shape, [t.shape for t in g_masks]
    same = bool((g_ids.cpu() == r_ids).all())
    res.append(("evaluate: generated ids equal the oracle's (3 new tokens)", 0.0 if same else 1.0, 0.0))
    n_seg = [int((r_ids[n, 1:] == cfg.seg_token_idx).sum()) for n in range(2)]
    res.append(("evaluate: one mask per [SEG] token", float(abs(g_masks[0].shape[0] - n_seg[0]) + abs(g_masks[1].shape[0] - n_seg[1])), 0.0))
    if same and bool((l_ids == r_ids).all()):
        for n in range(2):
            if r_masks[n].numel():
                lo_e = (l_masks[n].float() - r_masks[n]).abs().max().item()
                scale = max(1.0, r_masks[n].abs().max().item())
                res.append((f"evaluate: mask logits image {n} (bf16-CPU err {lo_e:.2e}, |ref| {r_masks[n].abs().max().item():.2e})",
                            (g_masks[n].cpu() - r_masks[n]).abs().max().item(), max(3e-2 * scale, 1.5 * lo_e)))
    return res
