"""CPU: the SAM decoder oracle (oracle/sam_decoder.py) against the fixture made from the imported reference modules
(tests/golden/sam_decoder.pt: PromptEncoder + MaskDecoder + Sam.postprocess_masks of /root/reference)."""
import torch

from oracle import cases, sam_decoder as osd


def test_sam_decoder_oracle_matches_reference_fixture(golden):
    g = golden("sam_decoder.pt")
    sd = cases.sam_decoder_state()
    emb, text = cases.sam_decoder_case()
    with torch.no_grad():
        low, iou = osd.decode_masks(sd, emb, text)
        post = osd.postprocess_masks(low, g["input_size"], g["original_size"])
    assert (low[:, 0, ::4, ::4] - g["low_res_sub"]).abs().max() < 1e-5
    assert abs(float(low.double().sum()) - float(g["low_res_sum"])) < 1e-3 * max(1.0, abs(float(g["low_res_sum"])))
    assert (iou - g["iou"]).abs().max() < 1e-5
    assert (post[:, 0, ::7, ::9] - g["post_sub"]).abs().max() < 1e-5
    assert post.shape[-2:] == tuple(g["original_size"])
