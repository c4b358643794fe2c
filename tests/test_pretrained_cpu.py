"""CPU: the pretrained-init loaders (llmseg_amd/pretrained.py; reference training.py:139-243, build_sam.py:98-107) on tiny synthetic
checkpoints written by the test in the authors' formats: HF sharded safetensors + index.json, sharded .bin + index.json, a SAM .pth,
an HF CLIP directory, a DINOv2 hub state dict.  The HIP model itself has no CPU path, so the loaders run against its parameter tree
(`ParamTree`) held by a stand-in object -- they only touch `.params`, `.config`, `._invalidate_derived()` and `.set_trainable()`."""
import json
import os

import pytest
import torch

from llmseg_amd import params as hp
from llmseg_amd import pretrained as pt
from llmseg_amd.trainable import TrainableMixin


class _Holder(TrainableMixin):
    def __init__(self, cfg):
        self.config = cfg
        self.shapes = hp.lisa_shapes(cfg)
        self.params = hp.ParamTree(self.shapes, torch.device("cpu"), torch.bfloat16, hp.fused_groups(cfg))
        for p in self.params.parameters():
            p.data.fill_(7.0)                              # "uninitialised": every loaded / initialised tensor must overwrite this
        self.invalidated = 0

    def _invalidate_derived(self):
        self.invalidated += 1


def _cfg(backbone="sam", lora_r=4, vocab=67, sam_decoder=False):
    return hp.LisaConfig(llama=hp.LlamaConfig(hidden=32, inter=48, layers=2, heads=2, vocab=vocab, lora_r=lora_r),
                         clip=hp.VitConfig(dim=16, layers=2, heads=2, mlp=32, patch=14, img=28),
                         dino=hp.VitConfig(dim=1024, layers=1, heads=2, mlp=32, patch=14, img=28, eps=1e-6),
                         sam=hp.SamConfig(img=64, patch=16, dim=32, depth=2, heads=2, window=2, global_idx=(1,), out_chans=256),
                         backbone=backbone, sam_decoder=sam_decoder)


def _rand(shapes, seed):
    g = torch.Generator().manual_seed(seed)
    return {k: torch.randn(s, generator=g) for k, s in shapes.items()}


def _llava_tensors(cfg, file_vocab):
    lc = hp.LlamaConfig(**{**cfg.llama.__dict__, "vocab": file_vocab, "lora_r": 0})
    sd = _rand(hp.llama_shapes(lc), 1)
    sd["model.mm_projector.weight"], sd["model.mm_projector.bias"] = torch.randn(cfg.llama.hidden, cfg.clip.dim), torch.randn(cfg.llama.hidden)
    sd["model.layers.0.self_attn.rotary_emb.inv_freq"] = torch.ones(8)       # a buffer 4.29 checkpoints carry: must be skipped
    return sd


def _write_llava(d, cfg, file_vocab, fmt):
    sd = _llava_tensors(cfg, file_vocab)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "config.json"), "w") as fh:
        json.dump(dict(hidden_size=cfg.llama.hidden, intermediate_size=cfg.llama.inter, num_hidden_layers=cfg.llama.layers,
                       num_attention_heads=cfg.llama.heads, vocab_size=file_vocab, rms_norm_eps=1e-6, mm_vision_select_layer=-2), fh)
    names = sorted(sd)
    shards = [names[0::2], names[1::2]]
    wm = {}
    for i, part in enumerate(shards):
        if fmt == "safetensors":
            from safetensors.torch import save_file
            f = f"model-0000{i + 1}-of-00002.safetensors"
            save_file({k: sd[k].contiguous() for k in part}, os.path.join(d, f))
        else:
            f = f"pytorch_model-0000{i + 1}-of-00002.bin"
            torch.save({k: sd[k] for k in part}, os.path.join(d, f))
        wm.update({k: f for k in part})
    with open(os.path.join(d, "model.safetensors.index.json" if fmt == "safetensors" else "pytorch_model.bin.index.json"), "w") as fh:
        json.dump({"metadata": {}, "weight_map": wm}, fh)
    return sd


def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("fmt", ["safetensors", "bin"])
def test_llava_shards_and_vocab_resize(tmp_path, fmt):
    cfg = _cfg(vocab=67)
    src = _write_llava(str(tmp_path / "llava"), cfg, file_vocab=64, fmt=fmt)
    m = _Holder(cfg)
    rep = pt.load_llava(m, str(tmp_path / "llava"))
    assert rep["short"] == {"model.embed_tokens.weight": 64, "lm_head.weight": 64} and m.invalidated == 1
    assert "model.layers.0.self_attn.rotary_emb.inv_freq" in rep["ignored"]
    P = m.params
    assert torch.equal(P["model.layers.1.mlp.up_proj.weight"], _bf(src["model.layers.1.mlp.up_proj.weight"]))
    # fused q|k|v backing tensor sees the per-projection loads (views)
    assert torch.equal(P["model.layers.0.qkv"][32:64], _bf(src["model.layers.0.self_attn.k_proj.weight"]))
    assert torch.equal(P["model.embed_tokens.weight"][:64], _bf(src["model.embed_tokens.weight"])) and bool((P["model.embed_tokens.weight"][64:] == 7).all())
    filled = pt.resize_token_embeddings(m, rep["short"], mode="normal", seed=3)
    assert filled == {"model.embed_tokens.weight": (64, 67), "lm_head.weight": (64, 67)}
    new = P["lm_head.weight"][64:].float()
    assert bool((new != 7).all()) and new.abs().max() < 0.2 and torch.equal(P["lm_head.weight"][:64], _bf(src["lm_head.weight"]))     # N(0, 0.02) rows, old rows kept
    pt.resize_token_embeddings(m, rep["short"], mode="mean")
    assert torch.allclose(P["model.embed_tokens.weight"][65].float(), _bf(src["model.embed_tokens.weight"]).float().mean(0), atol=2e-2)
    assert pt.config_from_hf(str(tmp_path / "llava")).llama.vocab == 64


def test_missing_shard_and_missing_tensor_fail_loudly(tmp_path):
    cfg = _cfg()
    d = str(tmp_path / "llava")
    _write_llava(d, cfg, 64, "safetensors")
    os.remove(os.path.join(d, "model-00002-of-00002.safetensors"))
    with pytest.raises(FileNotFoundError):
        pt.load_llava(_Holder(cfg), d)
    d2 = str(tmp_path / "single")
    os.makedirs(d2)
    sd = _llava_tensors(cfg, 64)
    del sd["model.layers.1.mlp.down_proj.weight"]
    torch.save(sd, os.path.join(d2, "pytorch_model.bin"))
    with pytest.raises(KeyError, match="down_proj"):
        pt.load_llava(_Holder(cfg), d2)
    sd = _llava_tensors(cfg, 64)
    sd["model.norm.weight"] = torch.ones(5)
    torch.save(sd, os.path.join(d2, "pytorch_model.bin"))
    with pytest.raises(ValueError, match="shape"):
        pt.load_llava(_Holder(cfg), d2)


def test_sam_clip_dinov2_key_prefixes(tmp_path):
    cfg = _cfg(sam_decoder=False)
    m = _Holder(cfg)
    sam = _rand(hp.sam_shapes(cfg.sam, pfx="image_encoder."), 2)
    sam.update({"prompt_encoder.no_mask_embed.weight": torch.zeros(1, 256), "mask_decoder.iou_token.weight": torch.zeros(1, 256)})
    torch.save(sam, str(tmp_path / "sam_vit_h_4b8939.pth"))
    rep = pt.load_sam(m, str(tmp_path / "sam_vit_h_4b8939.pth"))
    assert sorted(rep["ignored"]) == ["mask_decoder.iou_token.weight", "prompt_encoder.no_mask_embed.weight"]     # model built without the decoder
    k = "model.visual_model.image_encoder.blocks.1.attn.rel_pos_h"
    assert torch.equal(m.params[k], _bf(sam["image_encoder.blocks.1.attn.rel_pos_h"]))
    # HF CLIPModel export: vision_model.* is taken, the text tower and the buffers are ignored
    clip = {k[len("vision_tower."):]: v for k, v in _rand(hp.clip_shapes(cfg.clip, pfx="vision_tower."), 3).items()}
    clip.update({"text_model.embeddings.token_embedding.weight": torch.zeros(4, 4), "logit_scale": torch.zeros(()),
                 "vision_model.embeddings.position_ids": torch.arange(5)[None]})
    os.makedirs(str(tmp_path / "clip"))
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in clip.items()}, str(tmp_path / "clip" / "model.safetensors"))
    rep = pt.load_clip(m, str(tmp_path / "clip"))
    assert "logit_scale" in rep["ignored"] and "vision_model.embeddings.position_ids" in rep["ignored"]
    k = "model.vision_tower.vision_tower.vision_model.encoder.layers.1.self_attn.v_proj.bias"
    assert torch.equal(m.params[k], _bf(clip["vision_model.encoder.layers.1.self_attn.v_proj.bias"]))
    fused = "model.vision_tower.vision_tower.vision_model.encoder.layers.1.self_attn.qkv.bias"
    assert torch.equal(m.params[fused][32:], _bf(clip["vision_model.encoder.layers.1.self_attn.v_proj.bias"]))
    dino = _rand(hp.dinov2_shapes(cfg.dino, pfx=""), 4)
    rep = pt.load_dinov2(m, dino)
    assert torch.equal(m.params["model.visual_model_dinov2.blocks.0.ls1.gamma"], _bf(dino["blocks.0.ls1.gamma"])) and not rep["missing"]
    del dino["norm.bias"]
    with pytest.raises(KeyError, match="norm.bias"):
        pt.load_dinov2(_Holder(cfg), dino)


def test_fresh_modules_lora_and_trainable_set(tmp_path):
    cfg = _cfg(backbone="sam", lora_r=4)
    m = _Holder(cfg)
    names = pt.init_lisa_modules(m, seed=0)
    P = m.params
    assert all(".lisa_" in n or ".text_hidden_fcs." in n for n in names) and len(names) > 60
    w = P["model.text_hidden_fcs.0.0.weight"].float()
    assert w.abs().max() <= 1 / 32 ** 0.5 + 1e-3 and w.std() > 0.05                         # U(+-1/sqrt(fan_in)), fan_in = 32
    assert bool((P["model.lisa_norm_final_attn.weight"] == 1).all()) and bool((P["model.lisa_attention_layers.1.norm3.bias"] == 0).all())
    assert P["model.lisa_dino_conv.weight"].float().abs().max() <= 1 / 1024 ** 0.5 + 1e-3
    assert P["model.lisa_iou_head.2.bias"].float().abs().max() <= 1 / 128 ** 0.5 + 1e-3
    lora = pt.init_lora(m, seed=1)
    assert len(lora) == 2 * 2 * 2                                                             # layers x {q, v} x {A, B}
    assert bool((P["model.layers.1.self_attn.v_proj.lora_B.default.weight"] == 0).all())
    a = P["model.layers.0.self_attn.q_proj.lora_A.default.weight"].float()
    assert a.abs().max() <= 1 / 32 ** 0.5 + 1e-3 and a.std() > 0.05
    with pytest.raises(ValueError):
        pt.init_lora(_Holder(_cfg(lora_r=0)))
    m.set_trainable()
    on = {n for n, p in m.params.named_parameters() if p.requires_grad}
    assert "lm_head.weight" in on and "model.embed_tokens.weight" in on and "model.text_hidden_fcs.0.2.bias" in on
    assert "model.layers.0.self_attn.q_proj.lora_A.default.weight" in on and "model.lisa_embedding_head.2.weight" in on
    assert not any(k.startswith(("model.visual_model.", "model.vision_tower.", "model.mm_projector.")) for k in on)     # training.py:173-176: frozen
    assert "model.layers.0.self_attn.q_proj.weight" not in on and "model.layers.0.mlp.up_proj.weight" not in on          # LoRA: base weights frozen


def test_load_pretrained_order_and_requirements(tmp_path):
    cfg = _cfg(backbone="sam", lora_r=4, vocab=67)
    d = str(tmp_path / "llava")
    _write_llava(d, cfg, 64, "safetensors")
    with pytest.raises(ValueError, match="CLIP"):
        pt.load_pretrained(_Holder(cfg), d)
    clip = {k[len("vision_tower."):]: v for k, v in _rand(hp.clip_shapes(cfg.clip, pfx="vision_tower."), 3).items()}
    os.makedirs(str(tmp_path / "clip"))
    torch.save(clip, str(tmp_path / "clip" / "pytorch_model.bin"))
    with pytest.raises(ValueError, match="vision_pretrained"):
        pt.load_pretrained(_Holder(cfg), d, clip_dir=str(tmp_path / "clip"))
    torch.save(_rand(hp.sam_shapes(cfg.sam, pfx="image_encoder."), 2), str(tmp_path / "sam.pth"))
    m = _Holder(cfg)
    rep = pt.load_pretrained(m, d, sam_ckpt=str(tmp_path / "sam.pth"), clip_dir=str(tmp_path / "clip"), seed=5)
    assert set(rep) == {"llava", "clip", "sam", "fresh", "lora", "resized"} and rep["resized"]["lm_head.weight"] == (64, 67)
    left = [n for n, p in m.params.named_parameters() if bool((p == 7).all()) and not n.startswith("model.visual_model_dinov2.")]
    assert not left, left[:5]                                                                  # nothing on the SAM-backbone path is left uninitialised
    assert any(p.requires_grad for p in m.params.parameters())
