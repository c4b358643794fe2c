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
    # SAM decoder tensors are model-dependent (LisaConfig.sam_decoder): the key maps through, the model's own key set decides
    assert ck.reference_key(p + "model.visual_model.mask_decoder.iou_token.weight") == "model.visual_model.mask_decoder.iou_token.weight"
    assert ck.reference_key(p + "model.visual_model.pixel_mean") is None
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


class _Fake:
    """Duck-typed stand-in for the HIP model (which has no CPU path): `.params.named_parameters()` + `.load_state_dict`."""

    def __init__(self, names):
        self.params = torch.nn.ParameterDict()
        self._names = names
        self.t = {n: torch.zeros(2) for n in names}
        self.params.named_parameters = lambda: [(n, torch.nn.Parameter(v, requires_grad=False)) for n, v in self.t.items()]

    def load_state_dict(self, sd, strict=False):
        for k, v in sd.items():
            self.t[k] = v.clone()
        return [n for n in self.t if n not in sd], []


def _write(tmp_path, keys):
    d = tmp_path / "global_step3"
    d.mkdir(parents=True, exist_ok=True)
    torch.save({"module": {ck.PEFT_PREFIX + k: torch.ones(2) for k in keys}, "global_steps": 3}, str(d / "mp_rank_00_model_states.pt"))
    (tmp_path / "latest").write_text("global_step3")
    return str(tmp_path)


def test_decoder_keys_follow_the_model(tmp_path):
    """prompt-encoder / mask-decoder tensors load when the model has them, are ignored when it has not; a model
    tensor the file lacks is reported and (unless it is a LoRA matrix) warned about."""
    import pytest
    dec = "model.visual_model.mask_decoder.iou_token.weight"
    base = ["lm_head.weight", "model.layers.0.self_attn.q_proj.lora_A.default.weight"]
    path = _write(tmp_path, base + [dec, "model.layers.0.self_attn.rotary_emb.inv_freq"])
    with_dec, without = _Fake(base + [dec]), _Fake(base)
    info = ck.load_reference_checkpoint(with_dec, path, strict=True)
    assert torch.equal(with_dec.t[dec], torch.ones(2)) and not info["missing"]
    assert [k for k in info["ignored"] if "inv_freq" not in k] == []
    info = ck.load_reference_checkpoint(without, path)
    assert dec in info["ignored"] and not info["missing"]
    path2 = _write(tmp_path / "b", base[:1])
    with pytest.warns(RuntimeWarning, match="non-LoRA"):
        info = ck.load_reference_checkpoint(_Fake(base + [dec]), path2)
    assert dec in info["missing"]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        info = ck.load_reference_checkpoint(_Fake(base), path2)      # only a LoRA matrix is missing: no warning
    assert info["missing"] == [base[1]]
    with pytest.raises(KeyError):
        ck.load_reference_checkpoint(_Fake(base), path2, strict=True)


def test_zero2_optimizer_partitions_are_read(tmp_path):
    """N4: the Adam state of a DeepSpeed ZeRO-2 bf16 TRAINING checkpoint -- flat fp32 groups cut into equal per-rank partitions, padded to
    2 x world, `param_shapes` in the model-states file -- lands in the trainer's masters / moments.  The writer below follows the published layout
    (deepspeed absent: unpinned); 3 ranks, a parameter the model does not train in the file, padding at the end."""
    import types
    names = ["model.embed_tokens.weight", "model.layers.0.self_attn.q_proj.lora_A.default.weight", "model.text_hidden_fcs.0.0.bias"]
    shapes = {"base_model.model." + names[0]: torch.Size([5, 4]), "base_model.model." + names[1]: torch.Size([2, 4]),
              "base_model.model.model.gone.weight": torch.Size([3]), "base_model.model." + names[2]: torch.Size([7])}
    g = torch.Generator().manual_seed(0)
    full = {k: {n: torch.randn(shp, generator=g) for n, shp in shapes.items()} for k in ("master", "m", "v")}
    world = 3
    flat = {k: torch.cat([t.reshape(-1) for t in full[k].values()]) for k in full}
    total = (flat["master"].numel() + 2 * world - 1) // (2 * world) * (2 * world)
    flat = {k: torch.cat([v, torch.zeros(total - v.numel())]) for k, v in flat.items()}
    per = total // world
    d = tmp_path / "global_step40"
    d.mkdir()
    torch.save({"module": {}, "param_shapes": [shapes], "global_steps": 40}, str(d / "mp_rank_00_model_states.pt"))
    for r in range(world):
        sl = slice(r * per, (r + 1) * per)
        torch.save({"optimizer_state_dict": {"zero_stage": 2, "partition_count": [world], "single_partition_of_fp32_groups": [flat["master"][sl].clone()],
                                              "base_optimizer_state": {"state": {0: {"step": 40, "exp_avg": flat["m"][sl].clone(), "exp_avg_sq": flat["v"][sl].clone()}}}}},
                   str(d / f"bf16_zero_pp_rank_{r}_mp_rank_00_optim_states.pt"))
    prm = {n: torch.nn.Parameter(torch.zeros(shapes["base_model.model." + n])) for n in names}
    model = types.SimpleNamespace(params=types.SimpleNamespace(named_parameters=lambda: list(prm.items())))
    opt = types.SimpleNamespace(master=[torch.zeros_like(p) for p in prm.values()], m=[torch.zeros_like(p) for p in prm.values()],
                                v=[torch.zeros_like(p) for p in prm.values()], t=0)
    tr = types.SimpleNamespace(opt=opt)
    rep = ck.load_zero2_optimizer_states(str(d), model, tr)
    assert rep["restored"] == names and rep["skipped"] == ["base_model.model.model.gone.weight"] and rep["step"] == 40 and not rep["not_in_file"] and opt.t == 40
    for i, n in enumerate(names):
        for dst, k in ((opt.master, "master"), (opt.m, "m"), (opt.v, "v")):
            assert torch.equal(dst[i], full[k]["base_model.model." + n]), (n, k)
    import pytest
    torch.save({"module": {}}, str(d / "mp_rank_00_model_states.pt"))
    with pytest.raises(ValueError, match="param_shapes"):
        ck.load_zero2_optimizer_states(str(d), model, tr)
