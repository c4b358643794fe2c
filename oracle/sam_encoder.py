"""Oracle: SAM ViT image encoder (test infrastructure).

Follows reference `model/segment_anything/modeling/image_encoder.py`:
forward :110-125, Block :177-193, Attention :235-260, window partition/unpartition
:263-318, get_rel_pos :321-351, add_decomposed_rel_pos :354-392, PatchEmbed :395-426,
LayerNorm2d `common.py:31-43`; ViT-H config `build_sam.py:15-22,63-81`.
Pinned against the imported reference module (oracle/make_goldens.py).

Layout here is channels-last token grids [B, H, W, C] throughout; the neck's
convs/LayerNorm2d are applied channels-last and the result is returned in the
reference's [B, C, H, W].
"""
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class SamCfg:
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


def _rel_table(size, rel_pos):
    """get_rel_pos for q_size == k_size == size (image_encoder.py:321-351).

    Table length is 2*size-1 for every SAM config, so the interpolation branch
    (:335-342) is restated only for completeness.
    """
    L = 2 * size - 1
    if rel_pos.shape[0] != L:
        r = F.interpolate(rel_pos.float().t()[None], size=L, mode="linear")[0].t().to(rel_pos.dtype)
    else:
        r = rel_pos
    idx = torch.arange(size)[:, None] - torch.arange(size)[None, :] + (size - 1)
    return r[idx]                                     # [size_q, size_k, hd]


def _attention(sd, p, x, heads):
    """x [B', H, W, C] (B' = windows or images) -> same shape."""
    Bp, H, W, C = x.shape
    hd = C // heads
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"]).view(Bp, H * W, 3, heads, hd)
    q, k, v = qkv.permute(2, 0, 3, 1, 4)              # each [B', heads, HW, hd]
    s = (q * hd ** -0.5) @ k.transpose(-1, -2)
    # decomposed rel-pos uses the UNSCALED q (image_encoder.py:244-249)
    Rh = _rel_table(H, sd[p + "rel_pos_h"])
    Rw = _rel_table(W, sd[p + "rel_pos_w"])
    rq = q.reshape(Bp, heads, H, W, hd)
    bh = torch.einsum("bnhwc,hkc->bnhwk", rq, Rh)
    bw = torch.einsum("bnhwc,wkc->bnhwk", rq, Rw)
    s = (s.view(Bp, heads, H, W, H, W) + bh[..., :, None] + bw[..., None, :]).view(Bp, heads, H * W, H * W)
    o = torch.softmax(s, -1) @ v
    o = o.transpose(1, 2).reshape(Bp, H, W, C)
    return F.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


def _to_windows(x, ws):
    B, H, W, C = x.shape
    ph, pw = (-H) % ws, (-W) % ws
    x = F.pad(x, (0, 0, 0, pw, 0, ph))                 # zero rows/cols AFTER norm1: they act as keys
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C).transpose(2, 3).reshape(-1, ws, ws, C)
    return x, (Hp, Wp)


def _from_windows(w, ws, pad_hw, hw):
    Hp, Wp = pad_hw
    H, W = hw
    B = w.shape[0] // ((Hp // ws) * (Wp // ws))
    x = w.view(B, Hp // ws, Wp // ws, ws, ws, -1).transpose(2, 3).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W]


def sam_block(sd, p, x, cfg: SamCfg, window):
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], cfg.eps)
    if window > 0:
        H, W = h.shape[1:3]
        h, pad_hw = _to_windows(h, window)
        h = _attention(sd, p + "attn.", h, cfg.heads)
        h = _from_windows(h, window, pad_hw, (H, W))
    else:
        h = _attention(sd, p + "attn.", h, cfg.heads)
    x = x + h
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], cfg.eps)
    h = F.gelu(F.linear(h, sd[p + "mlp.lin1.weight"], sd[p + "mlp.lin1.bias"]))
    return x + F.linear(h, sd[p + "mlp.lin2.weight"], sd[p + "mlp.lin2.bias"])


def _ln2d_cl(x, w, b, eps):
    # LayerNorm2d over channels == layer_norm over the last dim in channels-last
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def sam_image_encoder(sd, pfx, images, cfg: SamCfg):
    """pfx e.g. 'model.visual_model.image_encoder.'; images [B,3,img,img] -> [B,out_chans,g,g]."""
    x = F.conv2d(images, sd[pfx + "patch_embed.proj.weight"], sd[pfx + "patch_embed.proj.bias"],
                 stride=cfg.patch).permute(0, 2, 3, 1)
    x = x + sd[pfx + "pos_embed"]
    for i in range(cfg.depth):
        x = sam_block(sd, f"{pfx}blocks.{i}.", x, cfg, 0 if i in cfg.global_idx else cfg.window)
    x = F.linear(x, sd[pfx + "neck.0.weight"].flatten(1))                       # 1x1 conv, no bias
    x = _ln2d_cl(x, sd[pfx + "neck.1.weight"], sd[pfx + "neck.1.bias"], cfg.eps)
    x = F.conv2d(x.permute(0, 3, 1, 2), sd[pfx + "neck.2.weight"], padding=1)   # 3x3, no bias
    x = _ln2d_cl(x.permute(0, 2, 3, 1), sd[pfx + "neck.3.weight"], sd[pfx + "neck.3.bias"], cfg.eps)
    return x.permute(0, 3, 1, 2)
