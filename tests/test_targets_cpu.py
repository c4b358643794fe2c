"""CPU: the oracle of the proposal-decode / target path (oracle/targets.py) pinned against what is installed here -- scipy.ndimage.zoom
(what skimage.transform.resize(order=0) calls), torch's antialiased F.interpolate (what the reference itself calls) -- and the host-side
tables of the product (llmseg_amd/targets.py: RLE string parser, nearest-index rule, resampling taps) against the oracle."""
import numpy as np
import scipy.ndimage as ndi
import torch
import torch.nn.functional as F

from llmseg_amd import targets as ht
from oracle import targets as ot


def _masks(rng, h, w, k):
    out = []
    for i in range(k):
        m = np.zeros((h, w), np.uint8)
        y0, x0 = rng.integers(0, h - 4), rng.integers(0, w - 4)
        m[y0:y0 + rng.integers(2, h - y0), x0:x0 + rng.integers(2, w - x0)] = 1
        m &= (rng.random((h, w)) > 0.1).astype(np.uint8)
        out.append(m)
    out.append(np.zeros((h, w), np.uint8))            # an empty proposal (its IoP is 0 / 0)
    out.append(np.ones((h, w), np.uint8))             # one run only
    return out


def test_rle_codec_round_trip_and_product_parser():
    rng = np.random.default_rng(1)
    for h, w in ((37, 53), (64, 64), (5, 300)):
        for m in _masks(rng, h, w, 4):
            r = ot.rle_encode(m)
            assert (ot.rle_decode(r) == m).all()
            cn = ht.rle_counts(r)
            flat, pos, v = np.zeros(h * w, np.uint8), 0, 0
            for n in cn:
                flat[pos:pos + n] = v
                pos += n
                v ^= 1
            assert pos == h * w and (flat.reshape(w, h).T == m).all()
    # hand-built run lists: 2 x 3 image, column-major pixels 0 0 | 1 1 | 1 0  ->  counts [2, 3, 1]
    m = np.array([[0, 1, 1], [0, 1, 0]], np.uint8)
    assert (ot.rle_decode({"size": [2, 3], "counts": [2, 3, 1]}) == m).all()
    assert list(ht.rle_counts(ot.rle_encode(m))) == [2, 3, 1]
    # a count > 31 needs two 5-bit groups; a negative delta needs the sign bit
    big = np.zeros((40, 3), np.uint8)
    big[35:, 0] = 1
    big[:2, 1] = 1
    big[39, 2] = 1                         # runs: 35 zeros, 5 + 2 ones, 38 + 39 zeros, 1 one (4th count is stored as 1 - 7 = -6)
    assert (ot.rle_decode(ot.rle_encode(big)) == big).all() and list(ht.rle_counts(ot.rle_encode(big))) == [35, 7, 77, 1]


def test_nearest_rule_is_scipy_zoom():
    rng = np.random.default_rng(2)
    for hin, win, H, W in ((300, 420, 1024, 683), (97, 130, 1024, 1024), (1365, 2048, 682, 1024), (512, 512, 512, 512), (333, 500, 1000, 1501)):
        gt = (rng.random((hin, win)) > 0.5).astype(np.uint8)
        z = ndi.zoom(gt.astype(np.float64), (H / hin, W / win), order=0, mode="mirror", grid_mode=True)
        mine = ot.resize_nearest(gt, H, W)
        assert z.shape == mine.shape and (z == mine).all()
        assert (gt[ht.nearest_index(H, hin)][:, ht.nearest_index(W, win)] == mine).all()


def test_aa_taps_are_torch_antialias():
    for S, O in ((1024, 256), (683, 256), (300, 256), (200, 256), (1365, 256)):
        first, count, w = ht.aa_taps(S, O)
        x = torch.rand(1, 1, S, S, dtype=torch.float64, generator=torch.Generator().manual_seed(S))
        ref = F.interpolate(x, size=(O, O), mode="bilinear", align_corners=False, antialias=True)[0, 0]
        Wm = np.zeros((O, S))
        for i in range(O):
            Wm[i, first[i]:first[i] + count[i]] = w[i, :count[i]]
        mine = torch.from_numpy(Wm) @ x[0, 0] @ torch.from_numpy(Wm).T
        assert (mine - ref).abs().max().item() < 1e-14


def test_targets_known_answers():
    seg = np.zeros((4, 4, 2), np.uint8)
    seg[:2, :, 0] = 1                     # top half
    seg[:, :2, 1] = 1                     # left half
    gt = np.zeros((8, 8), np.uint8)
    gt[:4, :4] = 1                        # top-left quadrant at twice the resolution
    iou, iop = ot.compute_all_iou_iop(seg, gt)
    assert np.allclose(iou, [4 / 8, 4 / 8]) and np.allclose(iop, [4 / 8, 4 / 8])
    iou, iop = ot.compute_all_iou_iop(np.zeros((4, 4, 1), np.uint8), gt)
    assert iou[0] == 0.0 and np.isnan(iop[0])


def test_rle_encode_masks_matches_the_restated_pycocotools_codec():
    """llmseg_amd.targets.rle_encode_masks (vectorised run extraction) against oracle.targets.rle_encode (maskApi restated loop), incl. a mask
    that starts with ones, an empty and a full mask, long runs (delta coding with negative deltas), and decode(encode(m)) == m."""
    import numpy as np
    import torch
    from llmseg_amd import targets as ht
    from oracle import targets as ot
    g = torch.Generator().manual_seed(3)
    H, W = 37, 53
    m = (torch.rand(6, H, W, generator=g) > 0.6).to(torch.uint8)
    m[1] = 0
    m[2] = 1
    m[3, 0, 0] = 1
    m[4, :, :20] = 1; m[4, :, 20:] = 0
    m[5] = (torch.rand(H, W, generator=g) > 0.97).to(torch.uint8)
    got = ht.rle_encode_masks(m)
    for k in range(6):
        ref = ot.rle_encode(m[k].numpy())
        assert got[k] == ref, (k, got[k], ref)
        assert np.array_equal(ot.rle_decode(got[k]), m[k].numpy())
        assert np.array_equal(ht.rle_counts(got[k]), ht.rle_counts({"size": [H, W], "counts": ref["counts"]}))


def test_amg_records_have_the_reference_fields():
    import torch
    from llmseg_amd import amg
    masks = torch.zeros((2, 8, 10), dtype=torch.uint8)
    masks[0, 2:5, 3:7] = 1
    masks[1, 0, 0] = 1
    out = dict(masks=masks, boxes=torch.tensor([[3, 2, 6, 4], [0, 0, 0, 0]]), iou_preds=torch.tensor([0.93, 0.9]), stability_score=torch.tensor([0.97, 0.96]),
               points=torch.tensor([[4.5, 3.5], [0.5, 0.5]], dtype=torch.float64), areas=torch.tensor([12, 1]))
    recs = amg.to_records(out, (8, 10))
    assert set(recs[0]) == {"segmentation", "area", "bbox", "predicted_iou", "point_coords", "stability_score", "crop_box"}
    assert recs[0]["bbox"] == [3, 2, 3, 2] and recs[0]["area"] == 12 and recs[0]["crop_box"] == [0, 0, 10, 8] and recs[0]["segmentation"]["size"] == [8, 10]
    assert recs[1]["point_coords"] == [[0.5, 0.5]]
    # crop layers: the record's crop_box is the XYWH form of the crop the mask came from (automatic_mask_generator.py:187)
    out["crop_boxes"] = torch.tensor([[2, 1, 9, 7], [0, 0, 10, 8]])
    recs = amg.to_records(out, (8, 10), output_mode="binary_mask")
    assert recs[0]["crop_box"] == [2, 1, 7, 6] and recs[1]["crop_box"] == [0, 0, 10, 8] and recs[0]["segmentation"].dtype == bool and recs[0]["segmentation"].sum() == 12


def test_device_table_cache_is_bounded():
    """the device tables of the target computation are an LRU of 64 entries, not an ever-growing dict; one locked
    lookup-or-build per call, keyed by device AND stream."""
    import threading
    from llmseg_amd import targets as ht
    d = ht._LRU(3)
    built = []
    mk = lambda i: (lambda: built.append(i) or i * i)
    for i in range(5):
        assert d.get(i, mk(i)) == i * i
    assert list(d._d) == [2, 3, 4] and len(d) == 3
    assert d.get(2, mk(2)) == 4 and built == [0, 1, 2, 3, 4]          # a hit builds nothing and refreshes the entry ...
    d.get(9, mk(9))
    assert list(d._d) == [4, 2, 9]                                     # ... so the oldest UNUSED one (3) was evicted
    errs = []

    def hammer(seed):
        try:
            for i in range(2000):
                k = (i * 7 + seed) % 11
                assert d.get(k, lambda k=k: k * k) == k * k
        except Exception as e:                                         # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=hammer, args=(s,)) for s in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs and len(d) <= 3
    assert ht._where("cpu") == ("cpu", 0)
    a = ht._dev_nearest(7, 5, "cpu")
    assert a is ht._dev_nearest(7, 5, "cpu") and a.tolist() == ht.nearest_index(7, 5).tolist()


def test_oracle_equals_the_references_own_target_functions(golden):
    """tests/golden/targets_ref.pt was recorded from the imported `SAM_Mask_Reader.extract_sam_segs` + `compute_all_iou` / `compute_all_iop`
    (oracle/make_goldens.py::gold_targets); the oracle must reproduce it exactly (float64 quotients of integer counts)."""
    from oracle import cases
    g = golden("targets_ref.pt")
    masks, gt = cases.target_case()
    recs = [{"segmentation": ot.rle_encode(m), "area": int(m.sum()), "bbox": [0, 0, 1, 1 + i]} for i, m in enumerate(masks)]
    mine = ot.extract_sam_segs(recs)
    assert [b[3] for b in mine["bbox"]] == g["order_bbox_h"].tolist()                  # area order (stable on ties), top 50
    assert mine["segs_square"].shape[0] == mine["segs_square"].shape[1] == max(masks.shape[1:])
    assert np.array_equal(mine["segs_square"].sum((0, 1)), g["square_sum"].numpy())
    ious, iops = ot.compute_all_iou_iop(mine["segs_origin"], gt)
    assert np.array_equal(ious, g["ious"].numpy()) and np.array_equal(iops, g["iops"].numpy())
    sub = mine["segs_origin"][:, :, :3].copy()
    sub[:, :, 1] = 0
    ious, iops = ot.compute_all_iou_iop(sub, gt)
    assert np.array_equal(ious, g["empty_ious"].numpy(), equal_nan=True) and np.array_equal(iops, g["empty_iops"].numpy(), equal_nan=True)
