"""CPU: the everything-mode oracle (oracle/amg.py) against the fixture made with the imported reference generator (tests/golden/amg.pt)."""
import torch

from oracle import amg as oamg, cases


def test_amg_oracle_matches_reference_fixture(golden):
    g = golden("amg.pt")
    sd = cases.sam_decoder_state()
    emb = cases.amg_embedding_case()
    with torch.no_grad():
        d = oamg.generate(sd, emb, g["input_size"], g["original_size"], points_per_side=8, points_per_batch=16, **cases.amg_thresholds())
    assert [r["counts"] for r in d["rles"]] == g["rle_counts"]
    assert torch.equal(d["boxes"], g["boxes"]) and torch.equal(d["points"], g["points"])
    assert (d["iou_preds"] - g["iou_preds"]).abs().max() < 1e-5 and (d["stability_score"] - g["stability_score"]).abs().max() < 1e-6
    assert torch.equal(d["masks"].flatten(1).sum(1), g["areas"])


def test_amg_helpers():
    m = torch.zeros((3, 6, 8), dtype=torch.bool)
    m[0, 2:4, 3:7] = True
    m[2, 0, 0] = True
    assert oamg.masks_to_boxes(m).tolist() == [[3, 2, 6, 3], [0, 0, 0, 0], [0, 0, 0, 0]]
    r = oamg.mask_to_rle(m)
    assert r[1]["counts"] == [48] and r[2]["counts"] == [0, 1, 47] and sum(r[0]["counts"]) == 48 and r[0]["counts"][0] == 3 * 6 + 2
    b = torch.tensor([[0., 0., 10., 10.], [1., 1., 11., 11.], [20., 20., 30., 30.], [0., 0., 10., 10.]])
    assert oamg.nms(b, torch.tensor([0.9, 0.8, 0.7, 0.9]), 0.5).tolist() == [0, 2]
    assert oamg.preprocess_shape(427, 640) == (683, 1024) and oamg.build_point_grid(2).tolist() == [[0.25, 0.25], [0.75, 0.25], [0.25, 0.75], [0.75, 0.75]]


def test_amg_crop_layers_oracle_matches_reference_fixture(golden):
    """Crop layers + small-region clean-up (oracle/amg.py::generate_crops) against the records the imported reference generator produced on the
    same seeded image with the same stand-in encoder (tests/golden/amg_crops.pt, written by oracle/make_goldens.py::gold_amg_crops)."""
    g = golden("amg_crops.pt")
    sd, img, enc = cases.sam_decoder_state(), cases.amg_image_case(), cases.amg_standin_encoder()
    for tag, min_area in (("crops", 0), ("crops_clean", 12)):
        with torch.no_grad():
            d = oamg.generate_crops(sd, enc, img, points_per_side=8, points_per_batch=16, crop_n_layers=1, crop_n_points_downscale_factor=2,
                                    min_mask_region_area=min_area, **cases.amg_thresholds())
        e = g[tag]
        assert [r["counts"] for r in oamg.mask_to_rle(d["masks"])] == e["rle_counts"], tag
        assert torch.equal(d["boxes"].long(), e["boxes"]) and torch.equal(d["crop_boxes"].long(), e["crop_boxes"]) and torch.equal(d["points"], e["points"])
        assert (d["iou_preds"] - e["iou_preds"]).abs().max() < 1e-5 and (d["stability_score"] - e["stability_score"]).abs().max() < 1e-6
        assert len({tuple(c) for c in e["crop_boxes"].tolist()}) >= 3                      # records really come from several crops
    assert not torch.equal(g["crops"]["areas"], g["crops_clean"]["areas"])                 # and the clean-up really changed some masks


def test_pil_resize_restatement_is_pillow(golden):
    """oracle/pil_resize.py against Pillow itself (the library `ResizeLongestSide.apply_image` ends in) and against the committed fixture."""
    import numpy as np
    from oracle import pil_resize
    img = cases.amg_image_case()
    small = pil_resize.resize_bilinear_u8(img, 77, 131)
    g = golden("amg_crops.pt")
    assert int(small.astype(np.int64).sum()) == g["resize_77x131_sum"] and torch.equal(torch.as_tensor(small[::9, ::11].copy()), g["resize_77x131_sample"])
    Image = __import__("pytest").importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    for (h, w, oh, ow) in [(201, 251, 820, 1024), (300, 400, 768, 1024), (64, 48, 1024, 768), (333, 517, 100, 91), (1, 7, 3, 20), (50, 50, 50, 80)]:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(np.array(Image.fromarray(a).resize((ow, oh), Image.BILINEAR)), pil_resize.resize_bilinear_u8(a, oh, ow)), (h, w, oh, ow)


def test_crop_boxes_and_small_regions_helpers():
    import numpy as np
    from llmseg_amd import amg
    for size, n in (((300, 400), 2), ((427, 640), 1), ((1024, 1024), 2), ((33, 1000), 1)):
        assert amg.generate_crop_boxes(size, n, 512 / 1500) == oamg.generate_crop_boxes(size, n, 512 / 1500)
    b, l = oamg.generate_crop_boxes((300, 400), 1, 512 / 1500)
    assert b == [[0, 0, 400, 300], [0, 0, 251, 201], [0, 99, 251, 300], [149, 0, 400, 201], [149, 99, 400, 300]] and l == [0, 1, 1, 1, 1]
    assert [len(g) for g in amg.build_all_layer_point_grids(8, 2, 2)] == [64, 16, 4]
    near = oamg.is_box_near_crop_edge(torch.tensor([[0, 0, 100, 100], [30, 30, 100, 100], [0, 0, 249, 100], [30, 30, 240, 120]]), [0, 0, 251, 201], [0, 0, 400, 300])
    assert near.tolist() == [False, False, True, True]
    m = np.zeros((12, 12), bool)
    m[1:7, 1:7] = True; m[3, 3] = False; m[9, 9] = True; m[10, 10] = True                 # a 1-pixel hole, a 2-pixel diagonal island (8-connected)
    f, ch = oamg.remove_small_regions(m, 2, "holes")
    assert ch and f[3, 3] and f.sum() == m.sum() + 1
    f2, ch2 = oamg.remove_small_regions(f, 3, "islands")
    assert ch2 and not f2[9, 9] and not f2[10, 10] and f2.sum() == 36
    f3, ch3 = oamg.remove_small_regions(f, 2, "islands")
    assert not ch3 and f3 is f
    only = np.zeros((6, 6), bool); only[0, 0] = True; only[3, 3:5] = True
    f4, ch4 = oamg.remove_small_regions(only, 9, "islands")                                  # every island is small: the largest stays
    assert ch4 and f4.sum() == 2 and f4[3, 3]
