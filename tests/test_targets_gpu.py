"""GPU: proposal decode + IoU / IoP targets + proposal maps (llmseg_rle_decode / llmseg_mask_targets / llmseg_resize_aa through the C ABI)
against the oracle restatement of the reference's CPU data path.  Integer work must be bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _proposals(rng, h, w, k):
    from oracle import targets as ot
    recs = []
    for i in range(k):
        m = np.zeros((h, w), np.uint8)
        y0, x0 = rng.integers(0, h - 8), rng.integers(0, w - 8)
        m[y0:y0 + rng.integers(4, h - y0), x0:x0 + rng.integers(4, w - x0)] = 1
        m &= (rng.random((h, w)) > 0.15).astype(np.uint8)
        recs.append({"segmentation": ot.rle_encode(m), "area": int(m.sum()), "bbox": [int(x0), int(y0), 1, 1]})
    z = np.zeros((h, w), np.uint8)
    recs.append({"segmentation": ot.rle_encode(z), "area": 0, "bbox": [0, 0, 0, 0]})                 # empty proposal: IoP = 0 / 0
    return recs


@pytest.mark.parametrize("h,w,hg,wg,k", [(683, 1024, 1365, 2048, 60), (300, 420, 300, 420, 7), (1024, 1024, 512, 384, 12), (97, 130, 194, 260, 3)])
def test_decode_targets_and_maps_vs_oracle(h, w, hg, wg, k):
    from llmseg_amd import targets as ht
    from oracle import targets as ot
    rng = np.random.default_rng(h * 7 + k)
    recs = _proposals(rng, h, w, k)
    gts = [(rng.random((hg, wg)) > 0.6).astype(np.uint8), np.zeros((hg, wg), np.uint8)]            # the second: empty ground truth
    ref = ot.extract_sam_segs(recs, top=50)
    got = ht.proposals_and_targets(recs, gts, DEV, top=50)
    K = min(50, len(recs))
    assert got["segs_origin"].shape == (K, h, w)
    # decode: bit-exact, both layouts
    assert torch.equal(got["segs_origin"].cpu(), torch.from_numpy(ref["segs_origin"]).permute(2, 0, 1))
    ms = sorted(recs, key=lambda m: m["area"], reverse=True)[:50]
    hwk = ht.decode_rles([m["segmentation"] for m in ms], DEV, hwk=True)
    assert torch.equal(hwk.cpu(), torch.from_numpy(ref["segs_origin"]))
    # targets: IEEE-double identical (nan == nan for the empty proposal / empty union)
    for c, gt in enumerate(gts):
        iou, iop = ot.compute_all_iou_iop(ref["segs_origin"], gt)
        assert np.array_equal(got["sam_ious"][c].cpu().numpy(), iou, equal_nan=True), (got["sam_ious"][c].cpu().numpy() - iou)
        assert np.array_equal(got["sam_iops"][c].cpu().numpy(), iop, equal_nan=True)
    # proposal maps: the float64 antialiased resize rounded to bf16 -- identical up to (rare) double-rounding ties: <= 1 bf16 ulp, < 0.01 % of pixels
    maps = ot.proposal_maps(ref["segs_square"]).float()
    d = (got["sam_segs"].float().cpu() - maps).abs()
    assert d.max().item() <= 2.0 ** -8, d.max().item()
    assert (d > 0).float().mean().item() < 1e-4, (d > 0).float().mean().item()


def test_targets_feed_model_forward_contract():
    """dtype / shape contract of `collate_fn_new` (utils/dataset.py:33-170): segs bf16 [K, 256, 256], ious / iops float64 [C, K]."""
    from llmseg_amd import targets as ht
    rng = np.random.default_rng(3)
    recs = _proposals(rng, 120, 90, 5)
    out = ht.proposals_and_targets(recs, [(rng.random((120, 90)) > 0.5).astype(np.uint8)], DEV)
    assert out["sam_segs"].dtype == torch.bfloat16 and out["sam_segs"].shape == (6, 256, 256)
    assert out["sam_ious"].dtype == torch.float64 and out["sam_ious"].shape == (1, 6) and out["sam_iops"].shape == (1, 6)


def test_dense_proposals_equal_the_rle_route():
    """Everything-mode output feeds the target computation directly: dense uint8 masks + areas (llmseg_amd/amg.py) must give exactly what the
    reference's file route gives (masks -> COCO RLE records -> sort by area / top 50 -> decode -> targets -> 256 x 256 maps)."""
    from llmseg_amd import amg, targets as ht
    rng = np.random.default_rng(11)
    h, w, k = 240, 320, 60
    m = np.zeros((k, h, w), np.uint8)
    for i in range(k):
        y0, x0 = rng.integers(0, h - 8), rng.integers(0, w - 8)
        m[i, y0:y0 + rng.integers(4, h - y0), x0:x0 + rng.integers(4, w - x0)] = 1
    masks = torch.from_numpy(m).to(DEV)
    areas = masks.flatten(1).sum(1)
    out = dict(masks=masks, boxes=torch.zeros((k, 4), dtype=torch.long), iou_preds=torch.ones(k), stability_score=torch.ones(k),
               points=torch.zeros((k, 2), dtype=torch.float64), areas=areas)
    recs = amg.to_records(out, (h, w))
    gts = [(rng.random((h, w)) > 0.5).astype(np.uint8)]
    a = ht.proposals_and_targets(recs, gts, DEV, top=50)
    b = ht.proposals_and_targets_dense(masks, areas, gts, top=50)
    # equal areas may be ordered differently by the two sorts: compare as sets of (mask, iou, iop, map) keyed by the mask bytes
    key = lambda segs: [bytes(s.cpu().numpy().tobytes()) for s in segs]
    ka, kb = key(a["segs_origin"]), key(b["segs_origin"])
    assert sorted(ka) == sorted(kb)
    perm = [kb.index(x) for x in ka]
    assert np.array_equal(a["sam_ious"].cpu().numpy(), b["sam_ious"][:, perm].cpu().numpy(), equal_nan=True)
    assert np.array_equal(a["sam_iops"].cpu().numpy(), b["sam_iops"][:, perm].cpu().numpy(), equal_nan=True)
    assert torch.equal(a["sam_segs"], b["sam_segs"][perm])


# (1500, 2000): ~17 taps -> the scalar-weight branch (taps > 12) and the RING = 32 instantiation; (2100, 640): H > 2048 >= W -> the clamp of the
# zero-padded square's columns beyond the image (both were compiled and shipped but never bit-checked); n_gt = 0: maps only
@pytest.mark.parametrize("h,w,hg,wg,k,n_gt", [(97, 130, 194, 260, 9, 2), (300, 420, 150, 210, 20, 5), (256, 96, 256, 96, 6, 1), (64, 1024, 64, 1024, 5, 3),
                                              (1500, 2000, 750, 1000, 4, 2), (2100, 640, 525, 160, 3, 1), (1030, 1700, 515, 850, 3, 6)])
def test_one_pass_kernel_equals_the_three_kernel_route(h, w, hg, wg, k, n_gt):
    """llmseg_proposal_targets (one pass through the order index) against gather + llmseg_mask_targets + llmseg_resize_aa, bit for bit: widths that are
    not multiples of 16 (scalar staging path), tall / wide images (zero padding below / right of the image), several ground truths incl. an empty one and
    more than the 4 one launch takes, an `order` that permutes and drops proposals."""
    from llmseg_amd import targets as ht
    rng = np.random.default_rng(h + 3 * w + k)
    m = (rng.random((k + 3, h, w)) > 0.6).astype(np.uint8)
    m[1] = 0                                                                     # an empty proposal: IoP = 0 / 0
    m[2] *= 255                                                                  # 0 / 255 masks count like 0 / 1
    masks = torch.from_numpy(m).to(DEV)
    order = torch.from_numpy(rng.permutation(k + 3)[:k].astype(np.int64)).to(DEV)
    gts = [torch.from_numpy((rng.random((hg, wg)) > 0.5).astype(np.uint8)).to(DEV) for _ in range(n_gt)]
    gts[-1] = torch.zeros_like(gts[-1])
    fused = ht.proposal_targets_fused(masks, order, gts)
    assert fused is not None
    maps, ious, iops, cnts = fused
    segs = masks[order].contiguous()
    assert torch.equal(maps, ht.resize_square_aa(segs, 256))
    for c, g in enumerate(gts):
        iou, iop, cnt = ht.mask_targets(segs, g)
        assert torch.equal(cnts[c], cnt), c
        assert np.array_equal(ious[c].cpu().numpy(), iou.cpu().numpy(), equal_nan=True) and np.array_equal(iops[c].cpu().numpy(), iop.cpu().numpy(), equal_nan=True)
