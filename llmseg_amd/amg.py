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

Beyond the default configuration (`generate_masks`): crop layers (automatic_mask_generator.py:199-262; utils/amg.py:189-264) and
`postprocess_small_regions` (automatic_mask_generator.py:326-372).  The image stays on the device as uint8 HWC; a crop is an origin + the
image's row stride handed to the resize kernel (`llmseg_image_resize_u8` = Pillow's BILINEAR resize bit for bit, what `set_image` applies,
predictor.py:34-60), `llmseg_sam_preprocess` normalises and pads (modeling/sam.py:174-186), the path's SAM encoder embeds the crop, and the
single-crop pipeline above runs on the crop's frame with the crop-edge filter (utils/amg.py:78-88) in front of its NMS.  Masks are un-cropped
into the full frame on the device; cross-crop NMS ranks by 1 / crop area; small holes / islands are removed by union-find connected components
(`llmseg_mask_small_regions`) and boxes recomputed by `llmseg_mask_boxes`.
"""
import math
from itertools import product

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


def build_all_layer_point_grids(n_per_side, n_layers, scale_per_layer):
    """utils/amg.py:189-197"""
    return [build_point_grid(int(n_per_side / (scale_per_layer ** i))) for i in range(n_layers + 1)]


def generate_crop_boxes(im_size, n_layers, overlap_ratio):
    """utils/amg.py:200-234 -> (crop boxes XYXY, layer index per box); box 0 is the whole image."""
    im_h, im_w = im_size
    short = min(im_h, im_w)
    boxes, layers = [[0, 0, im_w, im_h]], [0]
    for i_layer in range(n_layers):
        n = 2 ** (i_layer + 1)
        overlap = int(overlap_ratio * short * (2 / n))
        cw = int(math.ceil((overlap * (n - 1) + im_w) / n))
        ch = int(math.ceil((overlap * (n - 1) + im_h) / n))
        xs = [int((cw - overlap) * i) for i in range(n)]
        ys = [int((ch - overlap) * i) for i in range(n)]
        for x0, y0 in product(xs, ys):
            boxes.append([x0, y0, min(x0 + cw, im_w), min(y0 + ch, im_h)])
            layers.append(i_layer + 1)
    return boxes, layers


PIXEL_MEAN, PIXEL_STD = (123.675, 116.28, 103.53), (58.395, 57.12, 57.375)      # modeling/sam.py:27-28


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
                           stability_score_thresh=0.95, stability_score_offset=1.0, box_nms_thresh=0.7, mask_threshold=0.0, return_aux=False,
                           point_grid=None, crop_box=None, full_size=None, edge_atol=20.0):
        """feats_cl bf16 [4096, 256]: the image's SAM embedding (channels-last rows); input_size = (h, w) of the resized image inside the
        1024 frame, original_size = (H, W).  -> dict(masks uint8 [K, H, W], boxes int64 [K, 4] XYXY, iou_preds fp32 [K],
        stability_score fp32 [K], points fp64 [K, 2], areas int64 [K]) in the reference's record order (NMS order).
        Crop layers: `original_size` is the crop's (h, w), `point_grid` the layer's unit grid, `crop_box` (XYXY) / `full_size` (H, W) switch on the
        crop-edge filter (utils/amg.py:78-88); everything returned is in the crop's frame."""
        H, W = int(original_size[0]), int(original_size[1])
        nh, nw = preprocess_shape(H, W, self.config.sam.img)
        grid = (build_point_grid(points_per_side) if point_grid is None else point_grid) * np.array([[W, H]])      # original-image pixels (x, y)
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
        if crop_box is not None:                             # a side within 20 px of the crop's side but not of the image's
            cb = torch.tensor([float(v) for v in crop_box], device=self.device_)
            ob = torch.tensor([0.0, 0.0, float(full_size[1]), float(full_size[0])], device=self.device_)
            ub = boxes.float() + torch.tensor([crop_box[0], crop_box[1], crop_box[0], crop_box[1]], device=self.device_, dtype=torch.float32)
            near = ((ub - cb).abs() <= edge_atol) & ~((ub - ob).abs() <= edge_atol)
            ok = ok & ~near.any(1)
        cand = ok.nonzero().flatten()
        if cand.numel() == 0:
            z = torch.zeros((0,), device=self.device_)
            out = dict(masks=torch.empty((0, H, W), device=self.device_, dtype=torch.uint8), boxes=boxes[:0].long(), iou_preds=z, stability_score=z,
                       points=torch.zeros((0, 2), dtype=torch.float64), areas=z.long())
            if return_aux:
                out.update(low=low, iou_all=iou, stats=st, selected=cand)
            return out
        order = cand[torch.argsort(iou[cand], descending=True, stable=True)].to(torch.int32).contiguous()
        keep = ops.nms(boxes.float().contiguous(), order, box_nms_thresh)
        sel = order[keep.bool()].long()
        masks = ops.sam_binarize(low, sel, input_size, (H, W), self.config.sam.img, mask_threshold)
        out = dict(masks=masks, boxes=boxes[sel].long(), iou_preds=iou[sel], stability_score=stab[sel],
                   points=torch.as_tensor(grid)[(sel // 3).cpu()], areas=st[sel, 2].long())
        if return_aux:                                       # tests: every candidate's low-resolution logits / predicted IoU / statistics
            out.update(low=low, iou_all=iou, stats=st, selected=sel)
        return out

    @torch.no_grad()
    def set_image(self, image, crop_box=None):
        """`SamPredictor.set_image` (predictor.py:34-91) for the window `crop_box` (XYXY; None = all) of image uint8 [H, W, 3] (RGB, on the device):
        -> (embedding bf16 [grid^2, 256] channels-last rows, input_size (h, w) inside the square frame, the window's (h, w))."""
        s = self.config.sam
        H, W = image.shape[:2]
        x0, y0, x1, y1 = (0, 0, W, H) if crop_box is None else crop_box
        ch, cw = y1 - y0, x1 - x0
        nh, nw = preprocess_shape(ch, cw, s.img)
        rs = ops.image_resize_u8(image, nh, nw, None if crop_box is None else (x0, y0, x1, y1))
        return self._sam_encoder_cl(ops.sam_preprocess(rs, s.img, PIXEL_MEAN, PIXEL_STD)), (nh, nw), (ch, cw)

    @torch.no_grad()
    def generate_masks(self, image, points_per_side=32, points_per_batch=256, pred_iou_thresh=0.88, stability_score_thresh=0.95,
                       stability_score_offset=1.0, box_nms_thresh=0.7, crop_n_layers=0, crop_nms_thresh=0.7, crop_overlap_ratio=512 / 1500,
                       crop_n_points_downscale_factor=1, min_mask_region_area=0, mask_threshold=0.0, encode=None, return_aux=False):
        """`SamAutomaticMaskGenerator._generate_masks` + `postprocess_small_regions` (automatic_mask_generator.py:199-262,326-372): image uint8
        [H, W, 3] on the device -> dict(masks uint8 [K, H, W], boxes int64 [K, 4] XYXY, iou_preds, stability_score, points fp64 [K, 2], areas int64
        [K], crop_boxes int64 [K, 4] XYXY) in the reference's record order.  `encode(image, crop_box)` replaces `set_image` in tests."""
        assert image.dtype == torch.uint8 and image.dim() == 3 and image.shape[2] == 3 and image.is_contiguous()
        H, W = int(image.shape[0]), int(image.shape[1])
        crop_boxes, layer_idxs = generate_crop_boxes((H, W), crop_n_layers, crop_overlap_ratio)
        grids = build_all_layer_point_grids(points_per_side, crop_n_layers, crop_n_points_downscale_factor)
        parts, aux = [], []
        for cb, li in zip(crop_boxes, layer_idxs):
            x0, y0, x1, y1 = cb
            feats, inp, csize = (encode or self.set_image)(image, cb)
            whole = x0 == 0 and y0 == 0 and x1 == W and y1 == H
            d = self.generate_proposals(feats, inp, csize, points_per_batch=points_per_batch, pred_iou_thresh=pred_iou_thresh,
                                        stability_score_thresh=stability_score_thresh, stability_score_offset=stability_score_offset,
                                        box_nms_thresh=box_nms_thresh, mask_threshold=mask_threshold, point_grid=grids[li], crop_box=cb,
                                        full_size=(H, W), return_aux=return_aux)
            k = d["masks"].shape[0]
            if not whole:                                     # uncrop_masks / uncrop_boxes_xyxy / uncrop_points (utils/amg.py:237-264)
                full = torch.zeros((k, H, W), device=self.device_, dtype=torch.uint8)
                full[:, y0:y1, x0:x1] = d["masks"]
                d["masks"] = full
            d["boxes"] = d["boxes"] + torch.tensor([x0, y0, x0, y0], device=d["boxes"].device)
            d["points"] = d["points"].reshape(-1, 2) + torch.tensor([[x0, y0]], dtype=torch.float64)
            d["crop_boxes"] = torch.tensor(cb, dtype=torch.int64).repeat(k, 1)
            if return_aux:
                aux.append({n: d.pop(n, None) for n in ("low", "iou_all", "stats", "selected")})
            parts.append(d)
        keys = ("masks", "boxes", "iou_preds", "stability_score", "points", "areas", "crop_boxes")
        data = {n: torch.cat([p[n].to(p["masks"].device) if n not in ("points", "crop_boxes") else p[n] for p in parts], 0) for n in keys}
        if len(crop_boxes) > 1 and data["masks"].shape[0]:   # duplicates between crops: prefer masks from smaller crops
            c = data["crop_boxes"].to(torch.float32)
            scores = 1 / ((c[:, 2] - c[:, 0]) * (c[:, 3] - c[:, 1]))
            order = torch.argsort(scores, descending=True, stable=True).to(torch.int32)
            keep = ops.nms(data["boxes"].float().contiguous(), order.to(self.device_), crop_nms_thresh)
            sel = order.to(self.device_)[keep.bool()].long()
            data = {n: v[sel if v.is_cuda else sel.cpu()] for n, v in data.items()}
        if min_mask_region_area > 0 and data["masks"].shape[0]:
            masks = data["masks"].contiguous()
            changed = ops.mask_small_regions_(masks, min_mask_region_area)
            boxes, areas = ops.mask_boxes(masks)
            scores = (changed == 0).float()                   # NMS prefers masks that needed no clean-up
            order = torch.argsort(scores, descending=True, stable=True).to(torch.int32)
            keep = ops.nms(boxes.float().contiguous(), order, max(box_nms_thresh, crop_nms_thresh))
            sel = order[keep.bool()].long()
            data["masks"] = masks
            data["boxes"] = torch.where(changed.bool()[:, None], boxes.long(), data["boxes"])
            data["areas"] = areas.long()
            data = {n: v[sel if v.is_cuda else sel.cpu()] for n, v in data.items()}
        if return_aux:
            data["aux"] = aux
        return data


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
                     "stability_score": float(out["stability_score"][k]),
                     "crop_box": [0, 0, W, H] if "crop_boxes" not in out else
                     [int(out["crop_boxes"][k][0]), int(out["crop_boxes"][k][1]), int(out["crop_boxes"][k][2] - out["crop_boxes"][k][0]), int(out["crop_boxes"][k][3] - out["crop_boxes"][k][1])]})
    return recs
