"""Configuration dataclasses and the parameter layout of the hot path.

State-dict key names are the reference's (LLaVA-Llama / HF CLIP (transformers 4.29 naming) / DINOv2 hub /
SAM image encoder / LISA heads -- `model/LISA.py:35-121`), so a reference checkpoint loads 1:1.
`ParamTree` materialises the names as a nested `torch.nn.Module` tree (PyTorch as the memory/state-dict
plumbing); the compute code reads the tensors through a flat dict.
"""
from dataclasses import dataclass, field

import torch
import torch.nn as nn


@dataclass
class LlamaConfig:
    hidden: int = 4096
    inter: int = 11008
    layers: int = 32
    heads: int = 32
    vocab: int = 32004
    eps: float = 1e-6
    theta: float = 10000.0
    lora_r: int = 0
    lora_alpha: float = 16.0
    lora_dropout: float = 0.0      # reference: 0.05 (training.py:91); active in training-mode forwards under autograd only

    @property
    def head_dim(self):
        return self.hidden // self.heads


@dataclass
class VitConfig:
    dim: int = 1024
    layers: int = 24
    heads: int = 16
    mlp: int = 4096
    patch: int = 14
    img: int = 224
    eps: float = 1e-5


@dataclass
class SamConfig:
    img: int = 1024
    patch: int = 16
    dim: int = 1280
    depth: int = 32
    heads: int = 16
    mlp_ratio: float = 4.0
    out_chans: int = 256
    window: int = 14
    global_idx: tuple = (7, 15, 23, 31)
    eps: float = 1e-6

    @property
    def grid(self):
        return self.img // self.patch


@dataclass
class LisaConfig:
    llama: LlamaConfig = field(default_factory=LlamaConfig)
    clip: VitConfig = field(default_factory=lambda: VitConfig(eps=1e-5, img=224))
    dino: VitConfig = field(default_factory=lambda: VitConfig(eps=1e-6, img=518))
    sam: SamConfig = field(default_factory=SamConfig)
    out_dim: int = 256
    seg_token_idx: int = 32000
    select_layer: int = -2
    backbone: str = "dinov2"          # "dinov2" = what the reference runs (LISA.py:244-245); "sam" = LISA.py:242
    ce_loss_weight: float = 1.0
    align_loss_weight: float = 1.0
    regression_loss_weight: float = 1.0
    build_unused_towers: bool = True   # keep both vision backbones' parameters, as the reference does
    sam_decoder: bool = False          # SAM prompt encoder (text path) + mask decoder: only `evaluate()` reaches them (LISA.py:523-557)

    @property
    def n_img_tokens(self):
        return (self.clip.img // self.clip.patch) ** 2


def llama_shapes(c, pfx="model."):
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


def clip_shapes(c, pfx="model.vision_tower.vision_tower."):
    D, g = c.dim, c.img // c.patch
    v = pfx + "vision_model."
    s = {v + "embeddings.class_embedding": (D,), v + "embeddings.patch_embedding.weight": (D, 3, c.patch, c.patch),
         v + "embeddings.position_embedding.weight": (g * g + 1, D),
         v + "pre_layrnorm.weight": (D,), v + "pre_layrnorm.bias": (D,),
         v + "post_layernorm.weight": (D,), v + "post_layernorm.bias": (D,)}
    for i in range(c.layers):
        p = f"{v}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"], s[p + f"self_attn.{n}.bias"] = (D, D), (D,)
        for n in ("layer_norm1", "layer_norm2"):
            s[p + n + ".weight"], s[p + n + ".bias"] = (D,), (D,)
        s[p + "mlp.fc1.weight"], s[p + "mlp.fc1.bias"] = (c.mlp, D), (c.mlp,)
        s[p + "mlp.fc2.weight"], s[p + "mlp.fc2.bias"] = (D, c.mlp), (D,)
    return s


def dinov2_shapes(c, pfx="model.visual_model_dinov2."):
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


def sam_shapes(c, pfx="model.visual_model.image_encoder."):
    D, g, hd = c.dim, c.grid, c.dim // c.heads
    M = int(D * c.mlp_ratio)
    s = {pfx + "pos_embed": (1, g, g, D), pfx + "patch_embed.proj.weight": (D, 3, c.patch, c.patch),
         pfx + "patch_embed.proj.bias": (D,), pfx + "neck.0.weight": (c.out_chans, D, 1, 1),
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


def sam_decoder_shapes(pfx="model.visual_model.", D=256, mlp=2048):
    """Prompt encoder (the tensors the text-prompt path reads) + mask decoder under the reference's names
    (model/segment_anything/modeling/{prompt_encoder,mask_decoder,transformer}.py; sizes from build_sam.py:56-102)."""
    s = {pfx + "prompt_encoder.pe_layer.positional_encoding_gaussian_matrix": (2, D // 2),
         pfx + "prompt_encoder.no_mask_embed.weight": (1, D),
         pfx + "prompt_encoder.not_a_point_embed.weight": (1, D),
         pfx + "prompt_encoder.point_embeddings.0.weight": (1, D), pfx + "prompt_encoder.point_embeddings.1.weight": (1, D),
         pfx + "prompt_encoder.point_embeddings.2.weight": (1, D), pfx + "prompt_encoder.point_embeddings.3.weight": (1, D),
         pfx + "mask_decoder.iou_token.weight": (1, D), pfx + "mask_decoder.mask_tokens.weight": (4, D)}

    def attn(p, inner):
        for n in ("q_proj", "k_proj", "v_proj"):
            s[p + n + ".weight"], s[p + n + ".bias"] = (inner, D), (inner,)
        s[p + "out_proj.weight"], s[p + "out_proj.bias"] = (D, inner), (D,)
    t = pfx + "mask_decoder.transformer."
    for i in range(2):
        p = f"{t}layers.{i}."
        attn(p + "self_attn.", D)
        attn(p + "cross_attn_token_to_image.", D // 2)
        attn(p + "cross_attn_image_to_token.", D // 2)
        for n in ("norm1", "norm2", "norm3", "norm4"):
            s[p + n + ".weight"], s[p + n + ".bias"] = (D,), (D,)
        s[p + "mlp.lin1.weight"], s[p + "mlp.lin1.bias"] = (mlp, D), (mlp,)
        s[p + "mlp.lin2.weight"], s[p + "mlp.lin2.bias"] = (D, mlp), (D,)
    attn(t + "final_attn_token_to_image.", D // 2)
    s[t + "norm_final_attn.weight"], s[t + "norm_final_attn.bias"] = (D,), (D,)
    m = pfx + "mask_decoder."
    s[m + "output_upscaling.0.weight"], s[m + "output_upscaling.0.bias"] = (D, D // 4, 2, 2), (D // 4,)
    s[m + "output_upscaling.1.weight"], s[m + "output_upscaling.1.bias"] = (D // 4,), (D // 4,)
    s[m + "output_upscaling.3.weight"], s[m + "output_upscaling.3.bias"] = (D // 4, D // 8, 2, 2), (D // 8,)
    for i in range(4):
        for j, (o, k) in enumerate(((D, D), (D, D), (D // 8, D))):
            s[f"{m}output_hypernetworks_mlps.{i}.layers.{j}.weight"], s[f"{m}output_hypernetworks_mlps.{i}.layers.{j}.bias"] = (o, k), (o,)
    for j, (o, k) in enumerate(((D, D), (D, D), (4, D))):
        s[f"{m}iou_prediction_head.layers.{j}.weight"], s[f"{m}iou_prediction_head.layers.{j}.bias"] = (o, k), (o,)
    return s


def lisa_shapes(c: LisaConfig):
    s = {}
    s.update(llama_shapes(c.llama))
    s["model.mm_projector.weight"], s["model.mm_projector.bias"] = (c.llama.hidden, c.clip.dim), (c.llama.hidden,)
    s.update(clip_shapes(c.clip))
    if c.backbone == "dinov2" or c.build_unused_towers:
        s.update(dinov2_shapes(c.dino))
    if c.backbone == "sam" or c.build_unused_towers:
        s.update(sam_shapes(c.sam))
    s.update(head_shapes(c.llama.hidden, c.out_dim, c.dino.dim))
    if c.sam_decoder:
        s.update(sam_decoder_shapes())
    return s


class _Node(nn.Module):
    pass


class ParamTree(nn.Module):
    """Registers `shapes` as parameters of a nested module tree so that state_dict() keys == the dotted names.

    Fused GEMM operands (Llama q|k|v and gate|up, CLIP q|k|v, head attention projections) are allocated as ONE
    tensor and the per-projection parameters are views into it: state-dict I/O stays per-projection while the
    kernels see a single [N_total, K] weight."""

    def __init__(self, shapes, device, dtype, fused_groups=()):
        super().__init__()
        self.flat = {}
        backing = {}
        for gname, members in fused_groups:
            members = [m for m in members if m in shapes]
            if len(members) < 2:
                continue
            tail = shapes[members[0]][1:]
            rows = [shapes[m][0] for m in members]
            buf = torch.empty((sum(rows),) + tuple(tail), device=device, dtype=dtype)
            self.flat[gname] = buf
            r = 0
            for m, n in zip(members, rows):
                backing[m] = buf[r:r + n]
                r += n
        for name, shp in shapes.items():
            t = backing[name] if name in backing else torch.empty(shp, device=device, dtype=dtype)
            prm = nn.Parameter(t, requires_grad=False)
            node = self
            parts = name.split(".")
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            node.register_parameter(parts[-1], prm)
            self.flat[name] = prm

    def __getitem__(self, name):
        t = self.flat[name]
        return t.data if isinstance(t, nn.Parameter) else t

    def get(self, name, default=None):
        return self[name] if name in self.flat else default


def fused_groups(c: LisaConfig):
    g = []
    for i in range(c.llama.layers):
        p = f"model.layers.{i}."
        g.append((p + "qkv", [p + f"self_attn.{n}_proj.weight" for n in "qkv"]))
        g.append((p + "gate_up", [p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight"]))
    for i in range(c.clip.layers):
        p = f"model.vision_tower.vision_tower.vision_model.encoder.layers.{i}.self_attn."
        g.append((p + "qkv.weight", [p + f"{n}_proj.weight" for n in "qkv"]))
        g.append((p + "qkv.bias", [p + f"{n}_proj.bias" for n in "qkv"]))
    heads = [f"model.lisa_attention_layers.{i}.{a}." for i in range(2)
             for a in ("self_attn", "cross_attn_token_to_image", "cross_attn_image_to_token")] + ["model.lisa_final_attn."]
    for p in heads:
        g.append((p + "qkv.weight", [p + f"{n}_proj.weight" for n in "qkv"]))
        g.append((p + "qkv.bias", [p + f"{n}_proj.bias" for n in "qkv"]))
    return g


@torch.no_grad()
def init_random_(tree: ParamTree, shapes, seed=0):
    """Random init on the device (synthetic benchmark weights: there are no checkpoints offline).  Scales keep
    activations O(1); tables the reference zero-inits (rel_pos, pos_embed, LoRA-B) are randomised so every path runs."""
    dev = next(iter(tree.parameters())).device
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    for name, shp in shapes.items():
        t = tree[name]
        last = name.rsplit(".", 1)[-1]
        is_norm = ("norm" in name or ".neck.1." in name or ".neck.3." in name or last == "gamma")
        if is_norm and last in ("weight", "gamma"):
            t.copy_(1.0 + 0.1 * torch.randn(shp, device=dev, generator=gen))
        elif last == "bias":
            t.copy_(0.02 * torch.randn(shp, device=dev, generator=gen))
        elif len(shp) >= 2 and last == "weight":
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            std = 0.02 if ("embed_tokens" in name or "position_embedding" in name) else 1.0 / fan_in ** 0.5
            t.normal_(0.0, std, generator=gen)
        else:
            t.normal_(0.0, 0.02, generator=gen)
