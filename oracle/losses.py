"""Oracle: loss functions on/near the path (test infrastructure).

Follows reference `model/loss.py`: softmax_align_loss :50-80, iou_regression_loss
:82-94 (both called from `model/LISA.py:448-449`); dice_loss :4-27 and
sigmoid_ce_loss :30-47 (named by BASELINE.json's north_star; no caller in the
reference).  Pinned against the imported reference functions.
"""
import torch
import torch.nn.functional as F


def softmax_align(prop, target, gt_iou, tau=0.05):
    """prop [K,D], target [1,D], gt_iou [K,1] -> scalar KL(gt || sim), summed."""
    p = prop / prop.norm(dim=-1, keepdim=True)
    t = target / target.norm(dim=-1, keepdim=True)
    sim = torch.softmax((p @ t.t()) / tau, 0)
    gt = torch.softmax(gt_iou / tau, 0)
    return F.kl_div(sim.log(), gt, reduction="sum")


def iop_regression(pred, gt):
    """pred, gt [K,1] -> mean((p-g)^2 * exp(g-1)) * 50."""
    return ((pred - gt) ** 2 * torch.exp(gt - 1.0)).mean() * 50.0


def dice(inputs, targets, num_masks, scale=1000, eps=1e-6):
    x = inputs.sigmoid().flatten(1, 2)
    y = targets.flatten(1, 2)
    num = 2 * (x / scale * y).sum(-1)
    den = (x / scale).sum(-1) + (y / scale).sum(-1)
    return (1 - (num + eps) / (den + eps)).sum() / (num_masks + 1e-8)


def sigmoid_ce(inputs, targets, num_masks):
    l = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    return l.flatten(1, 2).mean(1).sum() / (num_masks + 1e-8)


def intersection_and_union(output, target, K=2, ignore_index=255):
    """utils/utils.py:119-132 restated with integer bincounts (exact; histc on ints is the same count)."""
    output = output.reshape(-1).clone()
    target = target.reshape(-1)
    output[target == ignore_index] = ignore_index
    inter = output[output == target]
    cnt = lambda t: torch.bincount(t[(t >= 0) & (t < K)].long(), minlength=K).float()
    ai, ao, at = cnt(inter), cnt(output), cnt(target)
    return ai, ao + at - ai, at
