"""Oracle: the two plain pre-LN ViT-L/14 towers on the path (test infrastructure).

CLIP ViT-L/14 vision tower -- third-party `transformers==4.29.0` `CLIPVisionModel`,
reference call site `model/llava/model/multimodal_encoder/clip_encoder.py:41-60`
(`hidden_states[select_layer][:, 1:]`).  Pinned against the installed HF eager model.

DINOv2 ViT-L/14 -- third-party torch.hub `facebookresearch/dinov2` `dinov2_vitl14`,
un-vendored and unpinned; reference call site `model/LISA.py:48,186-199`
(`forward_features(x)['x_norm_patchtokens']`).  PARITY UNPINNED: the published
architecture is restated with the hub's state-dict names; the structural stand-in
for validation is HF `Dinov2Model`.  Position-embedding resize is
bicubic, align_corners=False, to the exact target grid.
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class VitCfg:
    dim: int = 1024
    layers: int = 24
    heads: int = 16
    mlp: int = 4096
    patch: int = 14
    img: int = 224          # pretrain grid = img // patch
    eps: float = 1e-5


def _attention(q, k, v, heads):
    B, N, D = q.shape
    hd = D // heads
    q = q.view(B, N, heads, hd).transpose(1, 2) * hd ** -0.5
    k = k.view(B, N, heads, hd).transpose(1, 2)
    v = v.view(B, N, heads, hd).transpose(1, 2)
    p = torch.softmax(q @ k.transpose(-1, -2), -1)
    return (p @ v).transpose(1, 2).reshape(B, N, D)


def clip_vision_features(sd, pfx, images, cfg: VitCfg, select_layer=-2):
    """pfx e.g. 'model.vision_tower.vision_tower.'; returns [B, (img/patch)^2, dim] (CLS dropped)."""
    e = pfx + "vision_model.embeddings."
    x = F.conv2d(images, sd[e + "patch_embedding.weight"], stride=cfg.patch)   # no bias
    B, D = x.shape[:2]
    x = x.flatten(2).transpose(1, 2)
    cls = sd[e + "class_embedding"].to(x.dtype).expand(B, 1, D)
    x = torch.cat([cls, x], 1) + sd[e + "position_embedding.weight"][None]
    x = F.layer_norm(x, (D,), sd[pfx + "vision_model.pre_layrnorm.weight"],
                     sd[pfx + "vision_model.pre_layrnorm.bias"], cfg.eps)
    # hidden_states = [embeddings, out_0, ..., out_{L-1}]; select_layer indexes that list
    n_run = cfg.layers + 1 + select_layer if select_layer < 0 else select_layer
    for i in range(n_run):
        p = f"{pfx}vision_model.encoder.layers.{i}."
        h = F.layer_norm(x, (D,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], cfg.eps)
        q = F.linear(h, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"])
        k = F.linear(h, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
        v = F.linear(h, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
        a = _attention(q, k, v, cfg.heads)
        x = x + F.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        h = F.layer_norm(x, (D,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], cfg.eps)
        h = F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
        h = h * torch.sigmoid(1.702 * h)                                      # quick_gelu
        x = x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x[:, 1:]


def dinov2_pos_embed(pos_embed, grid_hw):
    """[1, 1+M*M, D] -> [1, 1+h*w, D], bicubic on the patch part (fp32)."""
    M = int(math.isqrt(pos_embed.shape[1] - 1))
    h, w = grid_hw
    if (h, w) == (M, M):
        return pos_embed
    dt = pos_embed.dtype
    pe = pos_embed.float()
    patch = pe[:, 1:].reshape(1, M, M, -1).permute(0, 3, 1, 2)
    patch = F.interpolate(patch, size=(h, w), mode="bicubic", align_corners=False)
    patch = patch.permute(0, 2, 3, 1).reshape(1, h * w, -1)
    return torch.cat([pe[:, :1], patch], 1).to(dt)


def dinov2_patch_tokens(sd, pfx, images, cfg: VitCfg):
    """pfx e.g. 'model.visual_model_dinov2.'; returns x_norm_patchtokens [B, h*w, dim]."""
    x = F.conv2d(images, sd[pfx + "patch_embed.proj.weight"], sd[pfx + "patch_embed.proj.bias"],
                 stride=cfg.patch)
    B, D, gh, gw = x.shape
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd[pfx + "cls_token"].to(x.dtype).expand(B, 1, D), x], 1)
    x = x + dinov2_pos_embed(sd[pfx + "pos_embed"], (gh, gw)).to(x.dtype)
    for i in range(cfg.layers):
        p = f"{pfx}blocks.{i}."
        h = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], cfg.eps)
        q, k, v = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).chunk(3, -1)
        a = F.linear(_attention(q, k, v, cfg.heads), sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        x = x + a * sd[p + "ls1.gamma"]
        h = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], cfg.eps)
        h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        x = x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"]) * sd[p + "ls2.gamma"]
    x = F.layer_norm(x, (D,), sd[pfx + "norm.weight"], sd[pfx + "norm.bias"], cfg.eps)
    return x[:, 1:]
