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


def _selected_iou(segs_hwk, ids, gt):
    """Union of the proposals `ids` at the proposals' resolution, nearest-resized to the ground truth's shape when they differ
    (reference training.py:912-930 / 1011-1030), 2-class I/U with ignore 255 (the reference calls intersectionAndUnionGPU without an
    ignore label change: 255 is its default), per-image accuracy with the no-object convention."""
    pred = torch.zeros(segs_hwk.shape[:2], dtype=torch.float32)
    for i in ids:
        pred += segs_hwk[:, :, int(i)].float()
    pred = (pred > 0).float()[None, None]
    if pred.shape[-2:] != gt.shape:
        pred = F.interpolate(pred, size=tuple(gt.shape), mode="nearest")
    i, u, t = intersection_and_union(pred[0, 0].long(), gt.long(), 2, 255)
    acc = i / (u + 1e-8)
    acc[u == 0] += 1.0
    return i, u, t, acc


def iou_iop_iou(segs_hwk, pred_similarity_row, pred_iop_row, gt, threshold=0.5):
    """Per-image body of `validate_iou_iop` (reference training.py:899-935): the arg-max-similarity proposal plus every proposal whose
    predicted IoP exceeds the threshold."""
    k = int(torch.argmax(pred_similarity_row))
    ids = [k] + [i for i in range(pred_iop_row.shape[0]) if pred_iop_row[i] > threshold and i != k]
    return _selected_iou(segs_hwk, ids, gt)


def top_iou_iou(segs_hwk, pred_similarity_row, pred_iop_row, gt, threshold=0.5, top=5):
    """Per-image body of `validate_threshold_from_topIoU` (reference training.py:1000-1030): of the `top` (5) most similar proposals, those
    whose predicted IoP exceeds the threshold (possibly none: an empty prediction)."""
    K = min(top, pred_similarity_row.shape[-1])
    ids = [int(i) for i in torch.topk(pred_similarity_row, K, dim=0).indices if pred_iop_row[int(i)] > threshold]
    return _selected_iou(segs_hwk, ids, gt)
