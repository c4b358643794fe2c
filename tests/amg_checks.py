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
