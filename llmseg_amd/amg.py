"""SAM "everything" mode on the device: `SamAutomaticMaskGenerator.generate` for its default single-crop
configuration (reference `model/segment_anything/automatic_mask_generator.py:127-324`, `utils/amg.py`, `predictor.py:166-258`,
`utils/transforms.py:36-50,103-113`), starting from the image embedding the path's SAM encoder produces.

MI355X-first shape: the reference walks the 1024 grid points in batches of 64 (a 16 GB-GPU memory knob), upsamples all 192 logit
maps of a batch to the original resolution, and filters there.  Here every prompt of the image goes through the decoder in a few large
batches (288 GB: default 256 prompts), the 3072 low-resolution logit maps stay resident (805 MB), and the original resolution is only
ever *evaluated*, never stored: one pass reduces each candidate to its stability counts, area and box (`llmseg_sam_mask_stats`, which
also applies the predicted-IoU filter), a one-workgroup greedy NMS ranks the survivors (`llmseg_nms`), and a second pass writes the
binary masks of the records that are returned (`llmseg_sam_binarize`).  Output masks are uint8 on the device -- the input format of the
target computation (`llmseg_amd/targets.py::proposals_and_targets`), so proposal generation feeds the path without a CPU / RLE round trip.
"""
import math

import numpy as np
import torch

from . import ops

BF16 = torch.bfloat16
PFX = "model.visual_model."


def build_point_grid(n):
    """amg.py:179-187"""
    off = 1 / (2 * n)
    one = np.linspace(off, 1 - off, n)
    return np.stack([np.tile(one[None, :], (n, 1)), np.tile(one[:, None], (1, n))], -1).reshape(-1, 2)


def preprocess_shape(h, w, long_side=1024):
    """transforms.py:103-113"""
    sc = long_side * 1.0 / max(h, w)
    return int(h * sc + 0.5), int(w * sc + 0.5)


class AmgMixin:
    @torch.no_grad()
    def embed_points(self, points_1024):
        """prompt_encoder.py:77-97,231-241 for one positive point per prompt + the padding point: points_1024 fp32 [b, 2] (x, y) in the
        1024-frame -> sparse prompt tokens bf16 [b, 2, 256]."""
        P, S = self.params, self._samdec()
        b = points_1024.shape[0]
        c = (points_1024.to(self.device_, torch.float32) + 0.5) / 1024.0                       # input_image_size = (1024, 1024)
        c = 2 * c - 1
        c = 2 * math.pi * (c[:, :1] * S["G"][0] + c[:, 1:] * S["G"][1])          # the 2-term contraction spelled out: no BLAS call on the path
        e = torch.cat([c.sin(), c.cos()], -1) + P[PFX + "prompt_encoder.point_embeddings.1.weight"].float()
        pad = P[PFX + "prompt_encoder.not_a_point_embed.weight"].float().expand(b, -1)
        return torch.stack([e, pad], 1).to(BF16)

    @torch.no_grad()
    def generate_proposals(self, feats_cl, input_size, original_size, points_per_side=32, points_per_batch=256, pred_iou_thresh=0.88,
                           stability_score_thresh=0.95, stability_score_offset=1.0, box_nms_thresh=0.7, mask_threshold=0.0, return_aux=False):
        """feats_cl bf16 [4096, 256]: the image's SAM embedding (channels-last rows); input_size = (h, w) of the resized image inside the
        1024 frame, original_size = (H, W).  -> dict(masks uint8 [K, H, W], boxes int64 [K, 4] XYXY, iou_preds fp32 [K],
        stability_score fp32 [K], points fp64 [K, 2], areas int64 [K]) in the reference's record order (NMS order)."""
        H, W = int(original_size[0]), int(original_size[1])
        nh, nw = preprocess_shape(H, W, self.config.sam.img)
        grid = build_point_grid(points_per_side) * np.array([[W, H]])                           # original-image pixels (x, y)
        tp = grid.copy()
        tp[:, 0] *= nw / W
        tp[:, 1] *= nh / H
        pts = torch.as_tensor(tp, dtype=torch.float32, device=self.device_)
        lows, ious = [], []
        for i in range(0, len(grid), points_per_batch):
            low, iou = self.sam_decode(feats_cl, None, sparse=self.embed_points(pts[i:i + points_per_batch]), multimask_output=True)
            lows.append(low.view(-1, 65536))
            ious.append(iou.float().reshape(-1))
        low = torch.cat(lows, 0) if len(lows) > 1 else lows[0]                                  # [3 n_points, 65536], candidate c = 3 point + mask
        iou = torch.cat(ious, 0).contiguous()
        st = ops.sam_mask_stats(low, iou, pred_iou_thresh, input_size, (H, W), self.config.sam.img, mask_threshold, stability_score_offset)
        stab = st[:, 0].float() / st[:, 1].float()                                              # int32 / int32 as torch divides them (amg.py:176)
        ok = (iou > pred_iou_thresh) & (stab >= stability_score_thresh)
        empty = (st[:, 5] < st[:, 3]) | (st[:, 6] < st[:, 4])
        boxes = torch.where(empty[:, None], torch.zeros_like(st[:, 3:7]), st[:, 3:7])
        cand = ok.nonzero().flatten()
        if cand.numel() == 0:
            z = torch.zeros((0,), device=self.device_)
            return dict(masks=torch.empty((0, H, W), device=self.device_, dtype=torch.uint8), boxes=boxes[:0].long(), iou_preds=z, stability_score=z,
                        points=torch.zeros((0, 2), dtype=torch.float64), areas=z.long())
        order = cand[torch.argsort(iou[cand], descending=True, stable=True)].to(torch.int32).contiguous()
        keep = ops.nms(boxes.float().contiguous(), order, box_nms_thresh)
        sel = order[keep.bool()].long()
        masks = ops.sam_binarize(low, sel, input_size, (H, W), self.config.sam.img, mask_threshold)
        out = dict(masks=masks, boxes=boxes[sel].long(), iou_preds=iou[sel], stability_score=stab[sel],
                   points=torch.as_tensor(grid)[(sel // 3).cpu()], areas=st[sel, 2].long())
        if return_aux:                                       # tests: every candidate's low-resolution logits / predicted IoU / statistics
            out.update(low=low, iou_all=iou, stats=st, selected=sel)
        return out


def to_records(out, original_size, output_mode="coco_rle"):
    """`SamAutomaticMaskGenerator.generate`'s return value (automatic_mask_generator.py:160-197) from `generate_proposals`' tensors: a list of
    dicts {segmentation, area, bbox (XYWH), predicted_iou, point_coords, stability_score, crop_box}; segmentation = the COCO RLE the
    reference's preparation scripts store (prepare_datasets/prepare_ReasonSeg.py:90-97) or the binary mask."""
    from .targets import rle_encode_masks
    H, W = int(original_size[0]), int(original_size[1])
    masks = out["masks"].cpu()
    boxes = out["boxes"].cpu().tolist()
    seg = rle_encode_masks(masks) if output_mode == "coco_rle" else [m.numpy().astype(bool) for m in masks]
    recs = []
    for k in range(masks.shape[0]):
        x0, y0, x1, y1 = boxes[k]
        recs.append({"segmentation": seg[k], "area": int(out["areas"][k]), "bbox": [x0, y0, x1 - x0, y1 - y0],
                     "predicted_iou": float(out["iou_preds"][k]), "point_coords": [out["points"][k].tolist()],
                     "stability_score": float(out["stability_score"][k]), "crop_box": [0, 0, W, H]})
    return recs
