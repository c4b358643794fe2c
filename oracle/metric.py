"""Oracle: the per-image body of `validate_threshold` (reference training.py:712-770) restated on CPU tensors.
Test infrastructure.  The I/U step is `losses.intersection_and_union` (pinned to the reference fixture)."""
import torch
import torch.nn.functional as F

from .losses import intersection_and_union


def union_resize_iou(segs_hwk, pred_iou_row, gt, threshold=0.5, out_size=1024):
    ids = [i for i in range(pred_iou_row.shape[0]) if pred_iou_row[i] > threshold]
    pred = torch.zeros(segs_hwk.shape[:2], dtype=torch.float32)
    for i in ids:
        pred += segs_hwk[:, :, i].float()
    pred = (pred > 0).float()[None, None]
    pred = F.interpolate(pred, size=(out_size, out_size), mode="nearest")[0, 0]
    g = F.interpolate(gt.float()[None, None], size=(out_size, out_size), mode="nearest")[0, 0]
    i, u, t = intersection_and_union(pred.long(), g.long(), 2, 255)
    acc = i / (u + 1e-8)
    acc[u == 0] += 1.0
    return i, u, t, acc


def argmax_iou(segs_hwk, pred_similarity_row, gt):
    """Per-image body of `validate` (reference training.py:625-660): the arg-max-similarity proposal, nearest-resized to the ground
    truth's shape, 2-class I/U (ignore 255), per-image accuracy with the no-object convention."""
    k = int(torch.argmax(pred_similarity_row))
    pred = segs_hwk[:, :, k].float()[None, None]
    if pred.shape[-2:] != gt.shape:
        pred = F.interpolate(pred, size=tuple(gt.shape), mode="nearest")
    i, u, t = intersection_and_union(pred[0, 0].long(), gt.long(), 2, 255)
    acc = i / (u + 1e-8)
    acc[u == 0] += 1.0
    return i, u, t, acc
