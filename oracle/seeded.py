"""Oracle: platform-exact seeded tensors and state-dict layouts (test infrastructure).

A counter-based integer hash (splitmix64 finaliser on int64 lanes) gives the
same bytes on every machine/torch build, so fixtures only need to store
(seed, config, expected outputs) -- never the weights themselves.

`*_shapes()` enumerate the reference's state-dict names for each module on the
path; oracle/make_goldens.py checks them against the imported reference with a
strict `load_state_dict`.
"""
import zlib

import torch

from .lisa import LisaCfg
from .llama import LlamaCfg
from .sam_encoder import SamCfg
from .vit import VitCfg

_M64 = (1 << 64) - 1


def _wrap(v):
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


_C1, _C2, _G = _wrap(0xBF58476D1CE4E5B9), _wrap(0x94D049BB133111EB), _wrap(0x9E3779B97F4A7C15)


def _srl(x, n):
    # logical right shift on int64 lanes
    return (x >> n) & ((1 << (64 - n)) - 1)


def uniform(shape, seed, lo=-1.0, hi=1.0):
    """Deterministic U[lo,hi) fp32 tensor (24-bit mantissa grid), identical on every platform."""
    n = 1
    for s in shape:
        n *= s
    x = torch.arange(n, dtype=torch.int64) * _G + _wrap(seed * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D)
    x = (x ^ _srl(x, 30)) * _C1
    x = (x ^ _srl(x, 27)) * _C2
    x = x ^ _srl(x, 31)
    u = _srl(x, 40).to(torch.float32) * (1.0 / (1 << 24))
    return (u * (hi - lo) + lo).reshape(shape)


def _key_seed(seed, name):
    return (seed << 32) ^ zlib.crc32(name.encode())


def llama_shapes(c: LlamaCfg, pfx="model."):
    H, I = c.hidden, c.inter
    s = {pfx + "embed_tokens.weight": (c.vocab, H), pfx + "norm.weight": (H,), "lm_head.weight": (c.vocab, H)}
    for i in range(c.layers):
        p = f"{pfx}layers.{i}."
        for n in "qkvo":
            s[p + f"self_attn.{n}_proj.weight"] = (H, H)
        s[p + "mlp.gate_proj.weight"] = (I, H)
        s[p + "mlp.up_proj.weight"] = (I, H)
        s[p + "mlp.down_proj.weight"] = (H, I)
        s[p + "input_layernorm.weight"] = (H,)
        s[p + "post_attention_layernorm.weight"] = (H,)
        if c.lora_r > 0:
            for n in "qv":
                s[p + f"self_attn.{n}_proj.lora_A.default.weight"] = (c.lora_r, H)
                s[p + f"self_attn.{n}_proj.lora_B.default.weight"] = (H, c.lora_r)
    return s


def clip_shapes(c: VitCfg, pfx="model.vision_tower.vision_tower."):
    D, g = c.dim, c.img // c.patch
    v = pfx + "vision_model."
    s = {v + "embeddings.class_embedding": (D,),
         v + "embeddings.patch_embedding.weight": (D, 3, c.patch, c.patch),
         v + "embeddings.position_embedding.weight": (g * g + 1, D),
         v + "pre_layrnorm.weight": (D,), v + "pre_layrnorm.bias": (D,),
         v + "post_layernorm.weight": (D,), v + "post_layernorm.bias": (D,)}
    for i in range(c.layers):
        p = f"{v}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (D, D)
            s[p + f"self_attn.{n}.bias"] = (D,)
        for n in ("layer_norm1", "layer_norm2"):
            s[p + n + ".weight"] = (D,)
            s[p + n + ".bias"] = (D,)
        s[p + "mlp.fc1.weight"], s[p + "mlp.fc1.bias"] = (c.mlp, D), (c.mlp,)
        s[p + "mlp.fc2.weight"], s[p + "mlp.fc2.bias"] = (D, c.mlp), (D,)
    return s


def dinov2_shapes(c: VitCfg, pfx="model.visual_model_dinov2."):
    D, g = c.dim, c.img // c.patch
    s = {pfx + "cls_token": (1, 1, D), pfx + "pos_embed": (1, g * g + 1, D), pfx + "mask_token": (1, D),
         pfx + "patch_embed.proj.weight": (D, 3, c.patch, c.patch), pfx + "patch_embed.proj.bias": (D,),
         pfx + "norm.weight": (D,), pfx + "norm.bias": (D,)}
    for i in range(c.layers):
        p = f"{pfx}blocks.{i}."
        for n in ("norm1", "norm2"):
            s[p + n + ".weight"], s[p + n + ".bias"] = (D,), (D,)
        s[p + "attn.qkv.weight"], s[p + "attn.qkv.bias"] = (3 * D, D), (3 * D,)
        s[p + "attn.proj.weight"], s[p + "attn.proj.bias"] = (D, D), (D,)
        s[p + "ls1.gamma"], s[p + "ls2.gamma"] = (D,), (D,)
        s[p + "mlp.fc1.weight"], s[p + "mlp.fc1.bias"] = (c.mlp, D), (c.mlp,)
        s[p + "mlp.fc2.weight"], s[p + "mlp.fc2.bias"] = (D, c.mlp), (D,)
    return s


def sam_shapes(c: SamCfg, pfx="model.visual_model.image_encoder."):
    D, g, hd = c.dim, c.grid, c.dim // c.heads
    M = int(D * c.mlp_ratio)
    s = {pfx + "pos_embed": (1, g, g, D),
         pfx + "patch_embed.proj.weight": (D, 3, c.patch, c.patch), pfx + "patch_embed.proj.bias": (D,),
         pfx + "neck.0.weight": (c.out_chans, D, 1, 1),
         pfx + "neck.1.weight": (c.out_chans,), pfx + "neck.1.bias": (c.out_chans,),
         pfx + "neck.2.weight": (c.out_chans, c.out_chans, 3, 3),
         pfx + "neck.3.weight": (c.out_chans,), pfx + "neck.3.bias": (c.out_chans,)}
    for i in range(c.depth):
        p = f"{pfx}blocks.{i}."
        sz = g if i in c.global_idx else c.window
        for n in ("norm1", "norm2"):
            s[p + n + ".weight"], s[p + n + ".bias"] = (D,), (D,)
        s[p + "attn.qkv.weight"], s[p + "attn.qkv.bias"] = (3 * D, D), (3 * D,)
        s[p + "attn.proj.weight"], s[p + "attn.proj.bias"] = (D, D), (D,)
        s[p + "attn.rel_pos_h"], s[p + "attn.rel_pos_w"] = (2 * sz - 1, hd), (2 * sz - 1, hd)
        s[p + "mlp.lin1.weight"], s[p + "mlp.lin1.bias"] = (M, D), (M,)
        s[p + "mlp.lin2.weight"], s[p + "mlp.lin2.bias"] = (D, M), (D,)
    return s


def head_shapes(hidden, out_dim=256, dino_dim=1024, pfx="model."):
    D = out_dim
    s = {pfx + "text_hidden_fcs.0.0.weight": (hidden, hidden), pfx + "text_hidden_fcs.0.0.bias": (hidden,),
         pfx + "text_hidden_fcs.0.2.weight": (D, hidden), pfx + "text_hidden_fcs.0.2.bias": (D,),
         pfx + "lisa_dino_conv.weight": (D, dino_dim, 1, 1), pfx + "lisa_dino_conv.bias": (D,)}

    def attn(p):
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + n + ".weight"], s[p + n + ".bias"] = (D, D), (D,)

    for i in range(2):
        p = f"{pfx}lisa_attention_layers.{i}."
        attn(p + "self_attn.")
        attn(p + "cross_attn_token_to_image.")
        attn(p + "cross_attn_image_to_token.")
        for n in ("norm1", "norm2", "norm3", "norm4"):
            s[p + n + ".weight"], s[p + n + ".bias"] = (D,), (D,)
        s[p + "mlp.lin1.weight"], s[p + "mlp.lin1.bias"] = (2048, D), (2048,)
        s[p + "mlp.lin2.weight"], s[p + "mlp.lin2.bias"] = (D, 2048), (D,)
    attn(pfx + "lisa_final_attn.")
    s[pfx + "lisa_norm_final_attn.weight"], s[pfx + "lisa_norm_final_attn.bias"] = (D,), (D,)
    s[pfx + "lisa_iou_head.0.weight"], s[pfx + "lisa_iou_head.0.bias"] = (128, D), (128,)
    s[pfx + "lisa_iou_head.2.weight"], s[pfx + "lisa_iou_head.2.bias"] = (1, 128), (1,)
    s[pfx + "lisa_embedding_head.0.weight"], s[pfx + "lisa_embedding_head.0.bias"] = (2048, D), (2048,)
    s[pfx + "lisa_embedding_head.2.weight"], s[pfx + "lisa_embedding_head.2.bias"] = (D, 2048), (D,)
    return s


def lisa_shapes(c: LisaCfg):
    s = {}
    s.update(llama_shapes(c.llama))
    s["model.mm_projector.weight"], s["model.mm_projector.bias"] = (c.llama.hidden, c.clip.dim), (c.llama.hidden,)
    s.update(clip_shapes(c.clip))
    s.update(dinov2_shapes(c.dino))
    s.update(sam_shapes(c.sam))
    s.update(head_shapes(c.llama.hidden, c.out_dim, c.dino.dim))
    return s


def fill_state_dict(shapes, seed, dtype=torch.float32):
    """Seeded fill with scales that keep activations O(1) and exercise every term:
    norm/LayerScale weights ~ 1 +- 0.25, biases/pos tables small but non-zero (the reference
    zero-inits rel_pos/pos_embed/LoRA-B, which would hide those paths)."""
    sd = {}
    for name, shp in shapes.items():
        ks = _key_seed(seed, name)
        last = name.rsplit(".", 1)[-1]
        if ("norm" in name or "layer_norm" in name or "layrnorm" in name or name.endswith(".gamma")
                or ".neck.1." in name or ".neck.3." in name) and last in ("weight", "gamma"):
            t = uniform(shp, ks, 0.75, 1.25)
        elif last == "bias":
            t = uniform(shp, ks, -0.1, 0.1)
        elif len(shp) >= 2 and last == "weight":
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            if "embed_tokens" in name or "position_embedding" in name:
                t = uniform(shp, ks, -0.5, 0.5)
            else:
                t = uniform(shp, ks, -1.0, 1.0) * (1.7 / fan_in ** 0.5)
        else:   # class_embedding, cls_token, pos_embed, mask_token, rel_pos_*
            t = uniform(shp, ks, -0.3, 0.3)
        sd[name] = t.to(dtype)
    return sd
