"""Oracle: SAM "everything" mode -- `SamAutomaticMaskGenerator.generate` for its default single-crop configuration (test infrastructure;
reference `model/segment_anything/automatic_mask_generator.py:127-324`, `utils/amg.py`, `predictor.py:166-258`, `utils/transforms.py:36-50,103-113`;
SURVEY.md 8f N1).  Starts from the image embedding (the encoder is row A6 of the path) and the sizes `set_image` records.

  point grid 32 x 32 (amg.py:179-187) scaled to the image, mapped to the 1024-frame (transforms.py:36-50)
  per batch of 64 points: prompt encoder (one positive point + the padding point, prompt_encoder.py:77-97,140-186), mask decoder with
    multimask_output (3 masks per point), postprocess_masks to the original size (logits), then
    predicted-IoU filter (> 0.88), stability score = |mask > +1| / |mask > -1| (amg.py:156-176) filter (>= 0.95), binarise (> 0),
    boxes (amg.py:303-346), crop-edge filter (amg.py:78-88; a no-op for the single full-image crop), uncompressed RLE (amg.py:107-135)
  box NMS at IoU 0.7 ranked by predicted IoU: `torchvision.ops.boxes.batched_nms` with one category -- third party, NOT installed here:
    restated from its documented semantics (greedy, descending score, suppress IoU > threshold, areas (x2 - x1)(y2 - y1)); PARITY
    UNPINNED for this step, everything before it is pinned against the imported reference by oracle/make_goldens.py::gold_amg.
"""
import numpy as np
import torch

from . import sam_decoder as sdec


def build_point_grid(n):
    off = 1 / (2 * n)
    one = np.linspace(off, 1 - off, n)
    return np.stack([np.tile(one[None, :], (n, 1)), np.tile(one[:, None], (1, n))], -1).reshape(-1, 2)


def preprocess_shape(h, w, long_side=1024):
    sc = long_side * 1.0 / max(h, w)
    return int(h * sc + 0.5), int(w * sc + 0.5)


def stability_score(masks, thr=0.0, off=1.0):
    inter = (masks > (thr + off)).sum(-1, dtype=torch.int16).sum(-1, dtype=torch.int32)
    union = (masks > (thr - off)).sum(-1, dtype=torch.int16).sum(-1, dtype=torch.int32)
    return inter / union


def masks_to_boxes(masks):
    """amg.py:303-346 on [C, H, W] bool -> [C, 4] XYXY (inclusive max coordinates), zeros for an empty mask."""
    if masks.numel() == 0:
        return torch.zeros((masks.shape[0], 4))
    h, w = masks.shape[-2:]
    in_h, _ = masks.max(-1)
    ch = in_h * torch.arange(h)[None, :]
    bottom, _ = ch.max(-1)
    top, _ = (ch + h * (~in_h)).min(-1)
    in_w, _ = masks.max(-2)
    cw = in_w * torch.arange(w)[None, :]
    right, _ = cw.max(-1)
    left, _ = (cw + w * (~in_w)).min(-1)
    empty = (right < left) | (bottom < top)
    return torch.stack([left, top, right, bottom], -1) * (~empty).unsqueeze(-1)


def mask_to_rle(masks):
    """amg.py:107-135: column-major runs, first run counts zeros."""
    b, h, w = masks.shape
    t = masks.permute(0, 2, 1).flatten(1)
    out = []
    for i in range(b):
        ch = (t[i, 1:] ^ t[i, :-1]).nonzero().flatten()
        idx = torch.cat([torch.tensor([0]), ch + 1, torch.tensor([h * w])])
        counts = [] if t[i, 0] == 0 else [0]
        counts.extend((idx[1:] - idx[:-1]).tolist())
        out.append({"size": [h, w], "counts": counts})
    return out


def nms(boxes, scores, thr):
    """torchvision.ops.nms semantics (restated, unpinned): indices kept, by decreasing score."""
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes.float()
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    dead = torch.zeros(len(b), dtype=torch.bool)
    keep = []
    for oi in range(len(order)):
        i = int(order[oi])
        if dead[i]:
            continue
        keep.append(i)
        rest = order[oi + 1:]
        xx1, yy1 = torch.maximum(b[i, 0], b[rest, 0]), torch.maximum(b[i, 1], b[rest, 1])
        xx2, yy2 = torch.minimum(b[i, 2], b[rest, 2]), torch.minimum(b[i, 3], b[rest, 3])
        inter = (xx2 - xx1).clamp(min=0) * (yy2 - yy1).clamp(min=0)
        iou = inter / (area[i] + area[rest] - inter)
        dead[rest[iou > thr]] = True
    return torch.tensor(keep, dtype=torch.long)


def process_batch(sd, image_embedding, points, input_size, original_size, pred_iou_thresh=0.88, stability_score_thresh=0.95,
                  stability_score_offset=1.0):
    """automatic_mask_generator.py:264-324 for the full-image crop.  points [n, 2] float64 (x, y) in ORIGINAL image pixels."""
    h, w = original_size
    nh, nw = preprocess_shape(h, w)
    tp = points.copy().astype(float)
    tp[..., 0] *= nw / w
    tp[..., 1] *= nh / h
    in_points = torch.as_tensor(tp)
    sparse = sdec.embed_points(sd, in_points[:, None, :].float(), torch.ones((len(tp), 1)))
    low, iou = sdec.decode_masks(sd, image_embedding, None, sparse=sparse, multimask_output=True)
    masks = sdec.postprocess_masks(low, input_size, original_size).flatten(0, 1)
    iou = iou.flatten(0, 1)
    pts = torch.as_tensor(points.repeat(3, axis=0))
    keep = iou > pred_iou_thresh
    masks, iou, pts = masks[keep], iou[keep], pts[keep]
    stab = stability_score(masks, 0.0, stability_score_offset)
    keep = stab >= stability_score_thresh
    masks, iou, pts, stab = masks[keep], iou[keep], pts[keep], stab[keep]
    binm = masks > 0.0
    boxes = masks_to_boxes(binm)
    return dict(masks=binm, iou_preds=iou, points=pts, stability_score=stab, boxes=boxes)


def generate(sd, image_embedding, input_size, original_size, points_per_side=32, points_per_batch=64, pred_iou_thresh=0.88,
             stability_score_thresh=0.95, stability_score_offset=1.0, box_nms_thresh=0.7):
    """-> dict(masks bool [K, H, W], boxes [K, 4] XYXY, iou_preds [K], stability_score [K], points [K, 2], rles): records in NMS order."""
    h, w = original_size
    grid = build_point_grid(points_per_side) * np.array([[w, h]])
    parts = [process_batch(sd, image_embedding, grid[i:i + points_per_batch], input_size, original_size, pred_iou_thresh,
                           stability_score_thresh, stability_score_offset) for i in range(0, len(grid), points_per_batch)]
    data = {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}
    keep = nms(data["boxes"].float(), data["iou_preds"], box_nms_thresh)
    data = {k: v[keep] for k, v in data.items()}
    data["rles"] = mask_to_rle(data["masks"])
    return data
