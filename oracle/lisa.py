"""Oracle: `LISAForCausalLM.model_forward` end to end (test infrastructure).

Follows reference `model/LISA.py:225-474` (orchestration, [SEG] gather, loss
reduction), `model/llava/model/llava_arch.py:93-96,98-347` (encode_images and
the live splice branch :185-208,242-251,327-345) and
`model/llava/model/language_model/llava_llama.py:55-135` (lm_head + shifted CE,
which `hidden_states` is returned).  Pinned end-to-end against the imported
reference on a reduced-width config (oracle/make_goldens.py).

`backbone="dinov2"` is what the reference runs (LISA.py:244-245);
`backbone="sam"` is the line the reference has commented out (LISA.py:242:
`image_embeddings = self.get_visual_embs(images)`), i.e. the SAM ViT-H encoder
feeding the 256-channel feature map straight into upsample + mask pooling --
the configuration BASELINE.json's north_star benchmarks at 1024x1024.
"""
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F

from . import llama as _llama, losses, mask_head, sam_encoder as _sam, vit as _vit

IMAGE_TOKEN_INDEX = -200
IGNORE_INDEX = -100


@dataclass
class LisaCfg:
    llama: _llama.LlamaCfg = field(default_factory=_llama.LlamaCfg)
    clip: _vit.VitCfg = field(default_factory=lambda: _vit.VitCfg(eps=1e-5, img=224))
    dino: _vit.VitCfg = field(default_factory=lambda: _vit.VitCfg(eps=1e-6, img=518))
    sam: _sam.SamCfg = field(default_factory=_sam.SamCfg)
    out_dim: int = 256
    seg_token_idx: int = 32000
    select_layer: int = -2
    backbone: str = "dinov2"
    ce_loss_weight: float = 1.0
    align_loss_weight: float = 1.0
    regression_loss_weight: float = 1.0

    @property
    def n_img_tokens(self):
        return (self.clip.img // self.clip.patch) ** 2


def encode_images(sd, cfg, images_clip):
    f = _vit.clip_vision_features(sd, "model.vision_tower.vision_tower.", images_clip, cfg.clip, cfg.select_layer)
    return F.linear(f, sd["model.mm_projector.weight"], sd["model.mm_projector.bias"])


def splice(sd, cfg, input_ids, attention_mask, labels, image_features):
    """Replace every IMAGE_TOKEN_INDEX by that sequence's projected CLIP tokens
    (llava_arch.py:185-208 -- the im_start/im_end ids stay ordinary embedded tokens),
    IGNORE_INDEX over the image span, left-extend the mask (:327-345).
    All sequences must hold the same number of image tokens (the reference's own
    `seg_token_mask` hack assumes exactly one, LISA.py:262-266)."""
    emb_w = sd["model.embed_tokens.weight"]
    new_e, new_l = [], []
    img_i = 0
    for n in range(input_ids.shape[0]):
        ids = input_ids[n]
        pos = (ids == IMAGE_TOKEN_INDEX).nonzero().flatten().tolist()
        if not pos:
            new_e.append(F.embedding(ids, emb_w))
            if labels is not None:
                new_l.append(labels[n])
            img_i += 1
            continue
        pe, pl, prev = [], [], 0
        for p in pos:
            pe += [F.embedding(ids[prev:p], emb_w), image_features[img_i]]
            if labels is not None:
                pl += [labels[n, prev:p], torch.full((image_features[img_i].shape[0],), IGNORE_INDEX,
                                                     dtype=labels.dtype, device=labels.device)]
            img_i += 1
            prev = p + 1
        pe.append(F.embedding(ids[prev:], emb_w))
        new_e.append(torch.cat(pe, 0))
        if labels is not None:
            pl.append(labels[n, prev:])
            new_l.append(torch.cat(pl, 0))
    assert all(e.shape == new_e[0].shape for e in new_e), "ragged image-token counts not restated"
    embeds = torch.stack(new_e, 0)
    new_labels = torch.stack(new_l, 0) if labels is not None else None
    extra = embeds.shape[1] - input_ids.shape[1]
    mask = torch.cat([torch.ones(attention_mask.shape[0], extra, dtype=attention_mask.dtype,
                                 device=attention_mask.device), attention_mask], 1)
    return embeds, mask, new_labels


def llava_forward(sd, cfg, images_clip, attention_mask, input_ids, labels=None, dropout_state=None):
    """-> (loss | None, logits [N,T,V], final-norm hidden [N,T,H])."""
    feats = encode_images(sd, cfg, images_clip)
    embeds, mask, new_labels = splice(sd, cfg, input_ids, attention_mask, labels, feats)
    hs = _llama.llama_model(sd, "model.", embeds, mask, cfg.llama, dropout_state)
    logits = F.linear(hs[-1], sd["lm_head.weight"])
    loss = _llama.shifted_ce(logits, new_labels, logits.shape[-1]) if labels is not None else None
    return loss, logits, hs[-1]


def seg_token_mask(cfg, input_ids):
    """LISA.py:254-266: [SEG] positions shifted by one, +255 for the image-token expansion."""
    m = input_ids[:, 1:] == cfg.seg_token_idx
    m = torch.cat([m, torch.zeros(m.shape[0], 1, dtype=torch.bool, device=m.device)], 1)
    return torch.cat([torch.zeros(m.shape[0], cfg.n_img_tokens - 1, dtype=torch.bool, device=m.device), m], 1)


def visual_features(sd, cfg, images):
    """-> [B, 256, g, g] feature map fed to upsample + mask pooling."""
    if cfg.backbone == "sam":
        outs = [_sam.sam_image_encoder(sd, "model.visual_model.image_encoder.", images[i:i + 1], cfg.sam)
                for i in range(images.shape[0])]                                  # LISA.py:173-184
        return torch.cat(outs, 0)
    toks = torch.cat([_vit.dinov2_patch_tokens(sd, "model.visual_model_dinov2.", images[i:i + 1], cfg.dino)
                      for i in range(images.shape[0])], 0)                        # LISA.py:186-199
    B, P, D = toks.shape
    g = int(P ** 0.5)
    fm = toks.permute(0, 2, 1).reshape(B, D, g, g)
    return F.conv2d(fm, sd["model.lisa_dino_conv.weight"], sd["model.lisa_dino_conv.bias"])   # LISA.py:245


def model_forward(sd, cfg: LisaCfg, images, images_clip, input_ids, labels, attention_masks, offset,
                  sam_segs_list, sam_ious_list=None, sam_iops_list=None, masks_list=None, inference=False,
                  return_aux=False, dropout_state=None):
    """dropout_state = (seed, offset) turns the LoRA dropout on (training-mode forward, training.py:91); None = eval."""
    feats = visual_features(sd, cfg, images)
    B = feats.shape[0]
    assert B == len(offset) - 1
    segmask = seg_token_mask(cfg, input_ids)

    if inference:
        assert images_clip.shape[0] == 1                                           # LISA.py:271
        clip_in = images_clip.expand(input_ids.shape[0], -1, -1, -1)
        ce, logits, hidden = llava_forward(sd, cfg, clip_in, attention_masks, input_ids, None)
    else:
        reps = (offset[1:] - offset[:-1]).tolist()
        clip_in = torch.cat([images_clip[i:i + 1].expand(r, -1, -1, -1) for i, r in enumerate(reps)], 0)
        ce, logits, hidden = llava_forward(sd, cfg, clip_in, attention_masks, input_ids, labels, dropout_state)

    pre = F.linear(hidden, sd["model.text_hidden_fcs.0.0.weight"], sd["model.text_hidden_fcs.0.0.bias"])
    h = F.relu(pre)
    g_forced = mask_head.forced("model.text_hidden_fcs.0.0", pre[segmask])         # test hook: gates imposed on the [SEG] rows (the only rows read below)
    if g_forced is not None:
        h = h.clone()
        h[segmask] = pre[segmask] * g_forced
    if mask_head.TRACE is not None:                                                # test hook: gates of the rows that are gathered below
        mask_head.TRACE.setdefault("model.text_hidden_fcs.0.0", []).append(h.detach()[segmask])
    h = F.linear(h, sd["model.text_hidden_fcs.0.2.weight"], sd["model.text_hidden_fcs.0.2.bias"])
    pred = h[segmask]                                                              # [sum C, D]
    seg_off = torch.cat([torch.zeros(1, dtype=torch.long), segmask.int().sum(-1).cumsum(-1).cpu()])[offset.cpu()]
    pred_embeddings = [pred[seg_off[i]:seg_off[i + 1]] for i in range(B)]

    up = mask_head.upsample_feats(feats, 256)
    ious, embs = [], []
    for b in range(B):
        pooled = mask_head.mask_pooling(up[b], sam_segs_list[b])
        iou, emb = mask_head.mask_head(sd, "model.", pooled, pred_embeddings[b])
        ious.append(iou)
        embs.append(emb)

    if inference:
        sims = [mask_head.cosine_scores(pred_embeddings[b], embs[b][0]) for b in range(B)]
        out = {"pred_similarity": sims, "gt_masks": masks_list, "pred_iou": [i[0].t() for i in ious]}
        if return_aux:
            out.update(logits=logits, hidden=hidden, feats=feats, pred_embeddings=pred_embeddings)
        return out

    align, reg, valid = 0.0, 0.0, 0
    for b in range(B):
        R = pred_embeddings[b].shape[0]
        if R == 0:
            raise ValueError("number of rounds = 0")
        a_r, r_r = 0.0, 0.0
        for r in range(R):
            gi = sam_ious_list[b][r].unsqueeze(1).to(ious[b].dtype)
            gp = sam_iops_list[b][r].unsqueeze(1).to(ious[b].dtype)
            a_r = a_r + losses.softmax_align(embs[b][r], pred_embeddings[b][r:r + 1], gi)
            r_r = r_r + losses.iop_regression(ious[b][r], gp)
        valid += 1
        align = align + a_r / (R + 1e-8)
        reg = reg + r_r / (R + 1e-8)
    if valid > 0:
        align, reg = align / valid, reg / valid
    ce = ce * cfg.ce_loss_weight
    align = align * cfg.align_loss_weight
    reg = reg * cfg.regression_loss_weight
    out = {"loss": ce + align + reg, "ce_loss": ce, "align_loss": align, "regression_loss": reg}
    if return_aux:
        out.update(logits=logits, hidden=hidden, feats=feats)
    return out
