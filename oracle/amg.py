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

Beyond the default configuration (`generate_crops`): crop layers (automatic_mask_generator.py:199-262, utils/amg.py:189-264), `set_image` per crop
(predictor.py:34-91: Pillow BILINEAR resize -- oracle/pil_resize.py, pinned against Pillow -- then `Sam.preprocess`, modeling/sam.py:174-186), the
crop-edge filter (utils/amg.py:78-88), cross-crop NMS preferring small crops, and `postprocess_small_regions` (automatic_mask_generator.py:326-372,
utils/amg.py:267-291).  Pinned against the imported generator by oracle/make_goldens.py::gold_amg_crops with the image encoder replaced by a
seeded stand-in on BOTH sides; the two NMS steps and the connected-component labelling (cv2 absent: scipy.ndimage.label, 8-connectivity,
raster label order) are restated from their documented semantics -- parity unpinned for those steps, pinned for everything around them.
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


def post_decoder(low, iou, points, input_size, original_size, pred_iou_thresh=0.88, stability_score_thresh=0.95, stability_score_offset=1.0,
                 crop_box=None, full_size=None):
    """automatic_mask_generator.py:283-322, everything after `predict_torch`'s decoder call: low fp32 [n, 3, 256, 256] mask logits and iou
    [n, 3] of n point prompts; points [n, 2] float64 in pixels of the crop (`original_size` = the crop's (h, w)); masks come back in the full
    image's frame (`uncrop_masks`), boxes and points in the crop's."""
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
    if crop_box is not None:
        fh, fw = full_size
        keep = ~is_box_near_crop_edge(boxes, crop_box, [0, 0, fw, fh])
        binm, iou, pts, stab, boxes = binm[keep], iou[keep], pts[keep], stab[keep], boxes[keep]
        x0, y0, x1, y1 = crop_box
        if not (x0 == 0 and y0 == 0 and x1 == fw and y1 == fh):                                    # utils/amg.py:255-264
            binm = torch.nn.functional.pad(binm, (x0, fw - (x1 - x0) - x0, y0, fh - (y1 - y0) - y0), value=0)
    return dict(masks=binm, iou_preds=iou, points=pts, stability_score=stab, boxes=boxes)


def process_batch(sd, image_embedding, points, input_size, original_size, pred_iou_thresh=0.88, stability_score_thresh=0.95,
                  stability_score_offset=1.0, crop_box=None, full_size=None):
    """automatic_mask_generator.py:264-324.  points [n, 2] float64 (x, y) in pixels of the CROP (the full image when crop_box is None)."""
    h, w = original_size
    nh, nw = preprocess_shape(h, w)
    tp = points.copy().astype(float)
    tp[..., 0] *= nw / w
    tp[..., 1] *= nh / h
    in_points = torch.as_tensor(tp)
    sparse = sdec.embed_points(sd, in_points[:, None, :].float(), torch.ones((len(tp), 1)))
    low, iou = sdec.decode_masks(sd, image_embedding, None, sparse=sparse, multimask_output=True)
    return post_decoder(low, iou, points, input_size, original_size, pred_iou_thresh, stability_score_thresh, stability_score_offset, crop_box, full_size)


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


def build_all_layer_point_grids(n_per_side, n_layers, scale_per_layer):
    """utils/amg.py:189-197"""
    return [build_point_grid(int(n_per_side / (scale_per_layer ** i))) for i in range(n_layers + 1)]


def generate_crop_boxes(im_size, n_layers, overlap_ratio):
    """utils/amg.py:200-234 -> (crop boxes XYXY, layer index per box); box 0 is the whole image."""
    import math
    from itertools import product
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


def is_box_near_crop_edge(boxes, crop_box, orig_box, atol=20.0):
    """utils/amg.py:78-88: a side within `atol` of the crop's side but not of the image's."""
    cb = torch.as_tensor(crop_box, dtype=torch.float)
    ob = torch.as_tensor(orig_box, dtype=torch.float)
    b = (boxes + torch.tensor([[crop_box[0], crop_box[1], crop_box[0], crop_box[1]]])).float()
    near_crop = torch.isclose(b, cb[None, :], atol=atol, rtol=0)
    near_img = torch.isclose(b, ob[None, :], atol=atol, rtol=0)
    return torch.any(near_crop & ~near_img, dim=1)


PIXEL_MEAN, PIXEL_STD = (123.675, 116.28, 103.53), (58.395, 57.12, 57.375)      # modeling/sam.py:27-28


def set_image(image, encode_fn, img_size=1024):
    """predictor.py:34-91: image uint8 [h, w, 3] numpy (RGB) -> (features, input_size, original_size)."""
    from . import pil_resize
    rs = pil_resize.apply_image(image, img_size)
    x = torch.as_tensor(rs).permute(2, 0, 1).contiguous()[None]
    x = (x - torch.tensor(PIXEL_MEAN).view(-1, 1, 1)) / torch.tensor(PIXEL_STD).view(-1, 1, 1)
    h, w = x.shape[-2:]
    x = torch.nn.functional.pad(x, (0, img_size - w, 0, img_size - h))
    return encode_fn(x), (h, w), tuple(image.shape[:2])


def remove_small_regions(mask, area_thresh, mode):
    """utils/amg.py:267-291 with scipy.ndimage.label (8-connectivity) in place of cv2.connectedComponentsWithStats (absent: unpinned step)."""
    import numpy as np
    from scipy import ndimage
    holes = mode == "holes"
    working = (holes ^ mask).astype(np.uint8)
    regions, n = ndimage.label(working, structure=np.ones((3, 3), np.int32))
    sizes = np.bincount(regions.ravel(), minlength=n + 1)[1:]
    small = [i + 1 for i, sz in enumerate(sizes) if sz < area_thresh]
    if len(small) == 0:
        return mask, False
    fill = [0] + small
    if not holes:
        fill = [i for i in range(n + 1) if i not in fill]
        if len(fill) == 0:
            fill = [int(np.argmax(sizes)) + 1]
    return np.isin(regions, fill), True


def postprocess_small_regions(data, min_area, nms_thresh):
    """automatic_mask_generator.py:326-372 on dict(masks bool [K, H, W], boxes, ...): -> filtered dict (NMS order: unchanged masks first)."""
    if data["masks"].shape[0] == 0:
        return data
    new, scores = [], []
    for m in data["masks"].numpy():
        m, ch = remove_small_regions(m, min_area, "holes")
        unchanged = not ch
        m, ch = remove_small_regions(m, min_area, "islands")
        unchanged = unchanged and not ch
        new.append(torch.as_tensor(m)[None])
        scores.append(float(unchanged))
    masks = torch.cat(new, 0)
    boxes = masks_to_boxes(masks)
    keep = nms(boxes.float(), torch.as_tensor(scores), nms_thresh)
    data = dict(data)
    data["masks"], data["boxes"] = masks, torch.where(torch.as_tensor(scores)[:, None] == 0.0, boxes.to(data["boxes"].dtype), data["boxes"])
    return {k: v[keep] for k, v in data.items()}


def finish_crop(d, crop_box, box_nms_thresh):
    """automatic_mask_generator.py:247-262 after the batches of one crop: NMS inside the crop, boxes / points back to the image frame."""
    x0, y0 = crop_box[0], crop_box[1]
    keep = nms(d["boxes"].float(), d["iou_preds"], box_nms_thresh)
    d = {k: v[keep] for k, v in d.items()}
    d["boxes"] = d["boxes"] + torch.tensor([[x0, y0, x0, y0]])
    d["points"] = d["points"] + torch.tensor([[x0, y0]])
    d["crop_boxes"] = torch.tensor([crop_box for _ in range(d["masks"].shape[0])]).reshape(-1, 4)
    return d


def merge_crops(parts, n_crops, box_nms_thresh, crop_nms_thresh, min_mask_region_area):
    """automatic_mask_generator.py:205-221 + 150-157: cross-crop NMS (smaller crops first), then the small-region clean-up."""
    data = {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}
    if n_crops > 1:
        c = data["crop_boxes"]
        scores = 1 / ((c[:, 2] - c[:, 0]) * (c[:, 3] - c[:, 1]))          # torchvision box_area
        keep = nms(data["boxes"].float(), scores, crop_nms_thresh)
        data = {k: v[keep] for k, v in data.items()}
    if min_mask_region_area > 0:
        data = postprocess_small_regions(data, min_mask_region_area, max(box_nms_thresh, crop_nms_thresh))
    return data


def generate_crops(sd, encode_fn, image, points_per_side=32, points_per_batch=64, pred_iou_thresh=0.88, stability_score_thresh=0.95,
                   stability_score_offset=1.0, box_nms_thresh=0.7, crop_n_layers=0, crop_nms_thresh=0.7, crop_overlap_ratio=512 / 1500,
                   crop_n_points_downscale_factor=1, min_mask_region_area=0):
    """`SamAutomaticMaskGenerator.generate` up to the record list (automatic_mask_generator.py:127-262): image uint8 [H, W, 3] numpy ->
    dict(masks bool [K, H, W], boxes [K, 4] XYXY, iou_preds, stability_score, points [K, 2], crop_boxes [K, 4] XYXY)."""
    H, W = image.shape[:2]
    crop_boxes, layer_idxs = generate_crop_boxes((H, W), crop_n_layers, crop_overlap_ratio)
    grids = build_all_layer_point_grids(points_per_side, crop_n_layers, crop_n_points_downscale_factor)
    parts = []
    for cb, li in zip(crop_boxes, layer_idxs):
        x0, y0, x1, y1 = cb
        feats, inp, csize = set_image(image[y0:y1, x0:x1, :], encode_fn)
        pts = grids[li] * np.array(csize)[None, ::-1]
        bs = [process_batch(sd, feats, pts[i:i + points_per_batch], inp, csize, pred_iou_thresh, stability_score_thresh, stability_score_offset,
                            crop_box=cb, full_size=(H, W)) for i in range(0, len(pts), points_per_batch)]
        parts.append(finish_crop({k: torch.cat([b[k] for b in bs], 0) for k in bs[0]}, cb, box_nms_thresh))
    return merge_crops(parts, len(crop_boxes), box_nms_thresh, crop_nms_thresh, min_mask_region_area)
