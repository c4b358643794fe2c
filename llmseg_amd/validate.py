"""gIoU / cIoU validation loops (reference `training.py`: `validate` :605-687, `validate_threshold` :690-870, `validate_iou_iop` :872-967,
`validate_threshold_from_topIoU` :969-1078 -- they differ only in WHICH proposals form the prediction): per validation image run
`model_forward(inference=True)`, keep the proposals whose predicted IoP exceeds the threshold, score their union against the
ground truth at 1024 x 1024.  The per-image body after the model call is ONE kernel (`llmseg_union_resize_iou`); the meters are
integer sums, reduced across ranks with three `all_reduce`s like the reference's `AverageMeter.all_reduce` (utils/utils.py:76-97).

Each sample dict carries the `model_forward` kwargs plus `origin_segs` (uint8 [H, W, K], the reader's layout,
`utils/sam_mask_reader.py`) and `gt_mask` (uint8 [H', W'], 255 = ignore)."""
import torch
import torch.distributed as dist

from . import ops


def sample_from_collated(collated):
    """A collated validation batch of ONE image (`collate_fn_new` -> `dict_to_cuda`, as `training.py:700-711` feeds the model) -> the sample
    dict of the loops below: the `model_forward` kwargs + `origin_segs` (`origin_segs_list[0]`, training.py:722) + `gt_mask`
    (`masks_list[0][0]`, training.py:719)."""
    from .collate import model_kwargs
    assert collated["images"].shape[0] == 1, "the reference validates one image per step (val_batch_size = 1)"
    kw = {k: v for k, v in model_kwargs(collated).items() if k != "inference"}
    dev = collated["images"].device
    kw["origin_segs"] = torch.as_tensor(collated["origin_segs_list"][0]).to(device=dev, dtype=torch.uint8)
    kw["gt_mask"] = collated["masks_list"][0][0].to(device=dev, dtype=torch.uint8)
    return kw


def _meters(dev):
    return (torch.zeros(2, device=dev, dtype=torch.float64), torch.zeros(2, device=dev, dtype=torch.float64),
            torch.zeros(2, device=dev, dtype=torch.float64), torch.zeros(1, device=dev, dtype=torch.float64))


def _finish(inter, union, acc, count):
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in (inter, union, acc, count):
            dist.all_reduce(t)
    giou = (acc / count.clamp(min=1))[1].item()
    ciou = (inter / (union + 1e-10))[1].item()
    return {"giou": giou, "ciou": ciou, "images": int(count.item())}


@torch.no_grad()
def validate(model, samples):
    """The arg-max variant (reference `training.py:605-687` `validate`): the proposal with the highest cosine similarity to the [SEG]
    embedding IS the prediction; it is nearest-resized to the ground truth's resolution and scored there."""
    dev = next(model.parameters()).device
    inter, union, acc, count = _meters(dev)
    for s in samples:
        kw = {k: v for k, v in s.items() if k not in ("origin_segs", "gt_mask")}
        out = model.model_forward(**kw, inference=True)
        sim = out["pred_similarity"][0][0]
        select = torch.zeros_like(sim, dtype=torch.uint8)
        select[torch.argmax(sim)] = 1                                                   # training.py:627-634
        iu = ops.union_resize_iou(s["origin_segs"], select, s["gt_mask"], out_size=None).double()
        i, u = iu[0:2], iu[2:4]
        a = i / (u + 1e-8)
        a = torch.where(u == 0, a + 1.0, a)                                             # no-object target (training.py:653)
        inter += i; union += u; acc += a; count += 1
    return _finish(inter, union, acc, count)


@torch.no_grad()
def validate_threshold(model, samples, threshold=0.5, out_size=1024):
    dev = next(model.parameters()).device
    inter, union, acc, count = _meters(dev)
    for s in samples:
        kw = {k: v for k, v in s.items() if k not in ("origin_segs", "gt_mask")}
        out = model.model_forward(**kw, inference=True)
        select = (out["pred_iou"][0][0] > threshold).to(torch.uint8)                    # training.py:712-718
        iu = ops.union_resize_iou(s["origin_segs"], select, s["gt_mask"], out_size=out_size).double()
        i, u = iu[0:2], iu[2:4]
        a = i / (u + 1e-8)
        a = torch.where(u == 0, a + 1.0, a)                                             # no-object target (training.py:768)
        inter += i; union += u; acc += a; count += 1
    return _finish(inter, union, acc, count)


def _selected_loop(model, samples, choose):
    """Shared body of the two variants below: `choose(similarity [K], pred_iop [K]) -> uint8 [K]` selection mask; the union of the selected
    proposals is scored at the ground truth's own resolution (nearest resize when the shapes differ)."""
    dev = next(model.parameters()).device
    inter, union, acc, count = _meters(dev)
    for s in samples:
        kw = {k: v for k, v in s.items() if k not in ("origin_segs", "gt_mask")}
        out = model.model_forward(**kw, inference=True)
        select = choose(out["pred_similarity"][0][0], out["pred_iou"][0][0])
        iu = ops.union_resize_iou(s["origin_segs"], select, s["gt_mask"], out_size=None).double()
        i, u = iu[0:2], iu[2:4]
        a = i / (u + 1e-8)
        a = torch.where(u == 0, a + 1.0, a)
        inter += i; union += u; acc += a; count += 1
    return _finish(inter, union, acc, count)


@torch.no_grad()
def validate_iou_iop(model, samples, threshold=0.5):
    """Reference `training.py:872-967`: the arg-max-similarity proposal plus every proposal whose predicted IoP exceeds the threshold."""
    def choose(sim, iop):
        select = (iop > threshold).to(torch.uint8)
        select[torch.argmax(sim)] = 1
        return select
    return _selected_loop(model, samples, choose)


@torch.no_grad()
def validate_threshold_from_topIoU(model, samples, threshold=0.5, top=5):
    """Reference `training.py:969-1078`: of the 5 most similar proposals, those whose predicted IoP exceeds the threshold (possibly none)."""
    def choose(sim, iop):
        idx = torch.topk(sim, min(top, sim.shape[-1]), dim=0).indices
        select = torch.zeros_like(iop, dtype=torch.uint8)
        select[idx] = (iop[idx] > threshold).to(torch.uint8)
        return select
    return _selected_loop(model, samples, choose)
