"""CPU: key mapping and directory layout of the reference's DeepSpeed checkpoints (README.md:118-130, training.py:404-421,460-477)."""
import os

import torch

from llmseg_amd import checkpoint as ck


def test_reference_key_mapping():
    p = "base_model.model."
    assert ck.reference_key(p + "model.layers.3.self_attn.q_proj.lora_A.default.weight") == "model.layers.3.self_attn.q_proj.lora_A.default.weight"
    assert ck.reference_key(p + "lm_head.weight") == "lm_head.weight"
    assert ck.reference_key("module." + p + "model.text_hidden_fcs.0.0.bias") == "model.text_hidden_fcs.0.0.bias"
    assert ck.reference_key(p + "model.layers.0.self_attn.rotary_emb.inv_freq") is None
    assert ck.reference_key(p + "model.visual_model.mask_decoder.iou_token.weight") is None
    assert ck.reference_key(p + "model.visual_model.prompt_encoder.pe_layer.positional_encoding_gaussian_matrix") is None
    assert ck.reference_key(p + "model.visual_model.image_encoder.blocks.7.attn.rel_pos_h") == "model.visual_model.image_encoder.blocks.7.attn.rel_pos_h"
    assert ck.reference_key(p + "model.vision_tower.vision_tower.vision_model.encoder.layers.0.self_attn.k_proj.bias").endswith("self_attn.k_proj.bias")


def test_resolve_layout(tmp_path):
    d = tmp_path / "ckpt_model"
    (d / "global_step5000").mkdir(parents=True)
    (d / "latest").write_text("global_step5000\n")
    f, tag = ck.resolve(str(d))
    assert tag == "global_step5000" and f == os.path.join(str(d), "global_step5000", "mp_rank_00_model_states.pt")
    f2, tag2 = ck.resolve(os.path.join(str(d), "global_step5000"))
    assert f2 == f and tag2 == tag
