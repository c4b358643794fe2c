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
