"""GPU: SAM everything mode (llmseg_amd/amg.py) against the oracle (oracle/amg.py; pinned against the imported
reference generator by tests/golden/amg.pt up to the NMS step, whose torchvision implementation is absent)."""
import numpy as np
import torch

from llmseg_amd import ops
from oracle import amg as oamg, cases, sam_decoder as osd
from tests import sam_decoder_checks as sc

DEV = "cuda"
BF = torch.bfloat16


def _raster(low):        # nested row order [n, 65536] -> [n, 256, 256]
    n = low.shape[0]
    return low.view(n, 64, 64, 2, 2, 2, 2).permute(0, 1, 3, 5, 2, 4, 6).reshape(n, 256, 256)


def check_amg():
    m, sd = sc._model()
    emb = cases.amg_embedding_case().to(BF).float()
    thr = cases.amg_thresholds()
    inp, orig = (683, 1024), (427, 640)
    res = []
    with torch.no_grad():
        # 1. point prompts + multimask decoder vs the oracle
        grid = oamg.build_point_grid(8) * np.array([[orig[1], orig[0]]])
        nh, nw = oamg.preprocess_shape(*orig)
        tp = grid.copy(); tp[:, 0] *= nw / orig[1]; tp[:, 1] *= nh / orig[0]
        pts = torch.as_tensor(tp[:16], dtype=torch.float32)
        r_sparse = osd.embed_points(sd, pts[:, None, :], torch.ones((16, 1)))
        g_sparse = m.embed_points(pts.to(DEV))
        res.append(("amg point prompt tokens", (g_sparse.float().cpu() - r_sparse).abs().max().item(), 2e-2))
        r_low, r_iou = osd.decode_masks(sd, emb, None, sparse=r_sparse, multimask_output=True)
        lo_sd = {k: v.to(BF) for k, v in sd.items()}
        l_low, l_iou = osd.decode_masks(lo_sd, emb.to(BF), None, sparse=r_sparse.to(BF), multimask_output=True)
        f_cl = emb[0].reshape(256, 4096).t().contiguous().to(DEV, BF)
        g_low, g_iou = m.sam_decode(f_cl, None, sparse=g_sparse, multimask_output=True)
        e_lo = (l_low.float() - r_low).abs().max().item()
        scale = max(1.0, r_low.abs().max().item())
        res.append((f"amg multimask low-res logits (bf16-CPU err {e_lo:.2e})", (_raster(g_low.view(-1, 65536)).cpu().view(16, 3, 256, 256) - r_low).abs().max().item(),
                    max(2e-2 * scale, 1.5 * e_lo)))
        e_io = (l_iou.float() - r_iou).abs().max().item()
        res.append((f"amg multimask iou (bf16-CPU err {e_io:.2e})", (g_iou.float().cpu() - r_iou).abs().max().item(), max(2e-2, 1.5 * e_io)))
        # 2. per-candidate statistics / binarisation at the original resolution, fp32 in: against torch on the oracle's own logits
        flat = r_low.flatten(0, 1)                                                     # [48, 256, 256]
        post = osd.postprocess_masks(r_low, inp, orig).flatten(0, 1)
        iou_f = r_iou.flatten()
        st = ops.sam_mask_stats(flat.reshape(48, 65536).contiguous().to(DEV), iou_f.contiguous().to(DEV), -1e9, inp, orig, 1024, 0.0, thr["stability_score_offset"],
                                nested=False).cpu()
        off = thr["stability_score_offset"]
        ref_st = torch.stack([(post > off).flatten(1).sum(1), (post > -off).flatten(1).sum(1), (post > 0).flatten(1).sum(1)], 1)
        res.append(("amg stability / area counts (pixels that differ, of 273k per mask)", float((st[:, :3] - ref_st).abs().max()), 3.0))
        rb = oamg.masks_to_boxes(post > 0)
        nonempty = ref_st[:, 2] > 0
        res.append(("amg boxes (pixels)", float((st[nonempty][:, 3:7] - rb[nonempty]).abs().max()), 1.0))
        sel = torch.arange(0, 48, 5, dtype=torch.int32)
        bm = ops.sam_binarize(flat.reshape(48, 65536).contiguous().to(DEV), sel.to(DEV), inp, orig, nested=False).cpu()
        res.append(("amg binarised masks (pixels that differ)", float((bm.bool() != (post[sel.long()] > 0)).flatten(1).sum(1).max()), 3.0))
        # predicted-IoU filter inside the statistics kernel: skipped rows keep their initial value
        st2 = ops.sam_mask_stats(flat.reshape(48, 65536).contiguous().to(DEV), iou_f.contiguous().to(DEV), 0.3, inp, orig, 1024, 0.0, off, nested=False).cpu()
        skipped = ~(iou_f > 0.3)
        res.append(("amg iou filter in the statistics pass", float((st2[skipped][:, :3].abs().sum() + (st2[~skipped] - st[~skipped]).abs().sum())), 0.0))
        # 3. NMS kernel vs the restated torchvision semantics on random boxes (incl. duplicates and ties)
        g = torch.Generator().manual_seed(5)
        xy = torch.randint(0, 300, (500, 2), generator=g).float()
        wh = torch.randint(5, 200, (500, 2), generator=g).float()
        boxes = torch.cat([xy, xy + wh], 1)
        boxes[100:120] = boxes[0:20]
        scores = torch.rand(500, generator=g)
        scores[200:210] = scores[0]
        for t in (0.3, 0.7):
            ref_keep = oamg.nms(boxes, scores, t)
            order = torch.argsort(scores, descending=True, stable=True).to(torch.int32)
            kf = ops.nms(boxes.to(DEV).contiguous(), order.to(DEV), t).cpu().bool()
            res.append((f"amg nms thr {t}: kept set", 0.0 if torch.equal(order[kf].long(), ref_keep) else 1.0, 0.0))
        # 4. whole pipeline: the records must be exactly what the reference's post-decoder steps (restated) make of the HIP decoder's outputs
        out = m.generate_proposals(f_cl, inp, orig, points_per_side=8, points_per_batch=24, return_aux=True, **thr)
        low_r = _raster(out["low"]).cpu().view(64, 3, 256, 256)
        iou_all = out["iou_all"].cpu()
        masks = osd.postprocess_masks(low_r, inp, orig).flatten(0, 1)
        keep = iou_all > thr["pred_iou_thresh"]
        idx = keep.nonzero().flatten()
        stab = oamg.stability_score(masks[idx], 0.0, thr["stability_score_offset"])
        k2 = stab >= thr["stability_score_thresh"]
        idx, stab = idx[k2], stab[k2]
        binm = masks[idx] > 0
        bx = oamg.masks_to_boxes(binm)
        kn = oamg.nms(bx.float(), iou_all[idx], thr["box_nms_thresh"])
        exp_sel = idx[kn]
        got_sel = out["selected"].cpu()
        same = got_sel.shape == exp_sel.shape and bool((got_sel == exp_sel).all())
        res.append((f"amg pipeline: selected candidates ({len(exp_sel)} records expected, {len(got_sel)} produced)", 0.0 if same else 1.0, 0.0))
        if same:
            res.append(("amg pipeline: masks (pixels that differ)", float((out["masks"].cpu().bool() != binm[kn]).flatten(1).sum(1).max()), 3.0))
            res.append(("amg pipeline: boxes", float((out["boxes"].cpu() - bx[kn]).abs().max()), 1.0))
            res.append(("amg pipeline: stability", float((out["stability_score"].cpu() - stab[kn]).abs().max()), 1e-4))
            res.append(("amg pipeline: areas", float((out["areas"].cpu() - binm[kn].flatten(1).sum(1)).abs().max()), 3.0))
            res.append(("amg pipeline: points", float((out["points"] - torch.as_tensor(oamg.build_point_grid(8) * np.array([[orig[1], orig[0]]]))[exp_sel // 3]).abs().max()), 0.0))
        res.append(("amg pipeline: at least 3 records", float(max(0, 3 - len(got_sel))), 0.0))
    return res


def check_amg_crops():
    """Everything mode beyond the default single crop (llmseg_amd/amg.py::generate_masks + csrc/image.hip) against the oracle
    (oracle/amg.py::generate_crops, pinned against the imported generator by tests/golden/amg_crops.pt; oracle/pil_resize.py, pinned against Pillow)."""
    from oracle import pil_resize
    m, sd = sc._model()
    thr = cases.amg_thresholds()
    img = cases.amg_image_case()
    H, W = img.shape[:2]
    d_img = torch.as_tensor(img).to(DEV)
    res = []
    with torch.no_grad():
        # 1. the resize kernel = Pillow's BILINEAR resize, bit for bit: whole image, every crop window (pointer + stride, no copy), odd sizes
        bad = 0
        boxes, layers = oamg.generate_crop_boxes((H, W), 2, 512 / 1500)
        for (x0, y0, x1, y1) in boxes:
            nh, nw = oamg.preprocess_shape(y1 - y0, x1 - x0)
            got = ops.image_resize_u8(d_img, nh, nw, (x0, y0, x1, y1)).cpu().numpy()
            bad += int((got != pil_resize.resize_bilinear_u8(img[y0:y1, x0:x1], nh, nw)).sum())
        res.append((f"amg resize == Pillow on {len(boxes)} crop windows (bytes that differ)", float(bad), 0.0))
        rng = np.random.default_rng(3)
        bad = 0
        for (h, w, oh, ow) in [(333, 517, 100, 91), (1, 7, 3, 20), (50, 50, 50, 80), (80, 50, 50, 50), (64, 48, 64, 48), (1500, 2000, 768, 1024), (240, 320, 1024, 768)]:
            a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            bad += int((ops.image_resize_u8(torch.as_tensor(a).to(DEV), oh, ow).cpu().numpy() != pil_resize.resize_bilinear_u8(a, oh, ow)).sum())
        res.append(("amg resize == Pillow on up- / down-scaling and one-axis cases (bytes that differ)", float(bad), 0.0))
        try:
            from PIL import Image
            a = rng.integers(0, 256, (427, 640, 3), dtype=np.uint8)
            ref = np.array(Image.fromarray(a).resize((1024, 683), Image.BILINEAR))
            res.append(("amg resize == PIL.Image.resize itself (bytes that differ)", float((ops.image_resize_u8(torch.as_tensor(a).to(DEV), 683, 1024).cpu().numpy() != ref).sum()), 0.0))
        except ImportError:
            pass
        # 2. Sam.preprocess
        rs = pil_resize.apply_image(img)
        x = torch.as_tensor(rs).permute(2, 0, 1).contiguous()[None]
        x = (x - torch.tensor(oamg.PIXEL_MEAN).view(-1, 1, 1)) / torch.tensor(oamg.PIXEL_STD).view(-1, 1, 1)
        x = torch.nn.functional.pad(x, (0, 1024 - x.shape[-1], 0, 1024 - x.shape[-2]))
        got = ops.sam_preprocess(torch.as_tensor(rs).to(DEV), 1024, oamg.PIXEL_MEAN, oamg.PIXEL_STD).float().cpu()
        res.append(("amg preprocess (normalise + zero pad) vs torch, bf16 output", (got - x.to(BF).float()).abs().max().item(), 1.6e-2))
        res.append(("amg preprocess: padding is exactly zero", got[..., rs.shape[0]:, :].abs().sum().item() + got[..., :, rs.shape[1]:].abs().sum().item(), 0.0))
        # 3. small-region clean-up and boxes on random masks with holes, islands, diagonal contacts, an empty and a full mask
        g = torch.Generator().manual_seed(9)
        K, h, w = 12, 97, 130
        base = torch.rand((K, 1, h // 8 + 2, w // 8 + 2), generator=g)
        masks = (torch.nn.functional.interpolate(base, (h, w), mode="bilinear")[:, 0] > 0.5)
        masks ^= torch.rand((K, h, w), generator=g) > 0.97                                  # speckle: 1-pixel holes and islands
        masks[0] = False; masks[1] = True; masks[2] = False; masks[2, 5, 5] = True; masks[2, 40, 41:43] = True      # empty / full / only small islands
        for min_area in (4, 30):
            dm = masks.to(torch.uint8).to(DEV).contiguous()
            changed = ops.mask_small_regions_(dm, min_area).cpu().bool()
            bad, badc = 0, 0
            for k in range(K):
                f, c1 = oamg.remove_small_regions(masks[k].numpy(), min_area, "holes")
                f, c2 = oamg.remove_small_regions(f, min_area, "islands")
                bad += int((torch.as_tensor(f) != dm[k].cpu().bool()).sum())
                badc += int(bool(c1 or c2) != bool(changed[k]))
            res.append((f"amg small regions (min area {min_area}): pixels that differ from the oracle", float(bad), 0.0))
            res.append((f"amg small regions (min area {min_area}): changed flags that differ", float(badc), 0.0))
            bx, ar = ops.mask_boxes(dm)
            res.append((f"amg boxes of the cleaned masks (min area {min_area})", float((bx.cpu() - oamg.masks_to_boxes(dm.cpu().bool())).abs().max()), 0.0))
            res.append((f"amg areas of the cleaned masks (min area {min_area})", float((ar.cpu() - dm.cpu().flatten(1).sum(1)).abs().max()), 0.0))
        # 4. whole pipeline, crop layers + clean-up, with the oracle's stand-in encoder feeding both sides: the records must be exactly what the
        #    reference's post-decoder steps (restated) make of the HIP decoder's outputs, crop by crop, then across crops
        enc = cases.amg_standin_encoder()

        def encode(image, cb):
            x0, y0, x1, y1 = cb
            feats, inp, csize = oamg.set_image(img[y0:y1, x0:x1, :], enc)
            return feats[0].reshape(256, 4096).t().contiguous().to(DEV, BF), inp, csize
        for min_area in (0, 12):
            kw = dict(points_per_side=8, crop_n_layers=1, crop_n_points_downscale_factor=2, min_mask_region_area=min_area)
            out = m.generate_masks(d_img, points_per_batch=24, encode=encode, return_aux=True, **kw, **thr)
            cbs, lis = oamg.generate_crop_boxes((H, W), 1, 512 / 1500)
            grids = oamg.build_all_layer_point_grids(8, 1, 2)
            parts = []
            for cb, li, aux in zip(cbs, lis, out["aux"]):
                x0, y0, x1, y1 = cb
                csize = (y1 - y0, x1 - x0)
                inp = oamg.preprocess_shape(*csize)
                pts = grids[li] * np.array(csize)[None, ::-1]
                n = len(pts)
                low = _raster(aux["low"]).cpu().view(n, 3, 256, 256)
                d = oamg.post_decoder(low, aux["iou_all"].cpu().view(n, 3), pts, inp, csize, thr["pred_iou_thresh"], thr["stability_score_thresh"],
                                      thr["stability_score_offset"], crop_box=cb, full_size=(H, W))
                parts.append(oamg.finish_crop(d, cb, thr["box_nms_thresh"]))
            exp = oamg.merge_crops(parts, len(cbs), thr["box_nms_thresh"], 0.7, min_area)
            tag = f"amg crops (min area {min_area})"
            same = out["masks"].shape[0] == exp["masks"].shape[0]
            res.append((f"{tag}: {exp['masks'].shape[0]} records expected, {out['masks'].shape[0]} produced", 0.0 if same else 1.0, 0.0))
            if same:
                res.append((f"{tag}: crop boxes", float((out["crop_boxes"] - exp["crop_boxes"]).abs().max()), 0.0))
                res.append((f"{tag}: masks (pixels that differ, worst record)", float((out["masks"].cpu().bool() != exp["masks"]).flatten(1).sum(1).max()), 3.0))
                res.append((f"{tag}: boxes", float((out["boxes"].cpu() - exp["boxes"]).abs().max()), 1.0))
                res.append((f"{tag}: points", float((out["points"] - exp["points"]).abs().max()), 0.0))
                res.append((f"{tag}: predicted IoU", float((out["iou_preds"].cpu() - exp["iou_preds"]).abs().max()), 1e-6))
                res.append((f"{tag}: areas", float((out["areas"].cpu() - exp["masks"].flatten(1).sum(1)).abs().max()), 3.0))
            res.append((f"{tag}: records from at least 3 crops", float(max(0, 3 - len({tuple(c) for c in out['crop_boxes'].tolist()}))), 0.0))
        # 5. set_image on the device (resize -> preprocess -> the path's SAM encoder) runs and returns the sizes the reference records
        S = m.config.sam.img
        feats, inp, csize = m.set_image(d_img, (149, 99, 400, 300))
        ok = inp == oamg.preprocess_shape(201, 251, S) and csize == (201, 251) and feats.shape == (m.config.sam.grid ** 2, m.config.sam.out_chans) and bool(torch.isfinite(feats.float()).all())
        res.append(("amg set_image on a crop window: sizes / finite embedding", 0.0 if ok else 1.0, 0.0))
        # ... and equals the encoder applied to the oracle's own resize + preprocess of that window (same kernels downstream: bit-identical)
        rs = pil_resize.apply_image(img[99:300, 149:400], S)
        x = torch.as_tensor(rs).permute(2, 0, 1).contiguous()[None]
        x = (x - torch.tensor(oamg.PIXEL_MEAN).view(-1, 1, 1)) / torch.tensor(oamg.PIXEL_STD).view(-1, 1, 1)
        x = torch.nn.functional.pad(x, (0, S - x.shape[-1], 0, S - x.shape[-2]))
        ref_feats = m._sam_encoder_cl(x.to(DEV, BF))
        res.append(("amg set_image == encoder(oracle resize + preprocess)", (feats.float() - ref_feats.float()).abs().max().item(), 0.0))
    return res
