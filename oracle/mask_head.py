"""Oracle: mask-selection head (test infrastructure).

Follows reference `model/transformer.py` (Attention :286-341,
LISA_TwoWayAttentionBlock :215-283, MLPBlock :13-26) and the head wiring in
`model/LISA.py` (:91-121 construction, :201-218 mask_pooling, :350-391 per-image
flow, :394-408 inference scoring).  Pinned against the imported reference.
"""
import math

import torch
import torch.nn.functional as F


# Test hook (no arithmetic): when a dict, every ReLU on the path appends its OUTPUT, flattened to [samples, units], under the name of the
# Linear that feeds it -- the gradient tests compare gate patterns (output > 0) between the HIP path and this restatement to tell a flipped
# ReLU gate (pre-activation within rounding noise of zero) from a wrong gradient (tests/backward_checks.py::grad_err).
TRACE = None


# Test hook (round 6): gates imposed from outside.  {name of the feeding Linear: bool / float [samples, units]} (all calls of a forward pass stacked
# in call order, as TRACE records them): where set, ReLU(x) is evaluated as x * gate -- the SAME piecewise-linear branch another evaluation of the
# model took -- so that a gradient comparison measures arithmetic and not which side of a kink a pre-activation within rounding noise of zero fell
# on.  At full depth the 256 proposal rows of an image are nearly identical (their spread is below one bf16 step), so a unit near zero flips for ALL
# rows at once: one coherent flip moves every head gradient by several per cent (tests/fulldepth_checks.py, profiles/r06_head_bwd_diag_seed4.md).
FORCE = None
_FORCE_POS = {}


def force_gates(gates):
    """gates: None (off) or {name: tensor [samples, units]}; resets the per-name cursors."""
    global FORCE
    FORCE = gates
    _FORCE_POS.clear()


def forced(name, x2d):
    """-> gate rows for the next x2d.shape[0] samples of `name`, or None."""
    if FORCE is None or name not in FORCE:
        return None
    pos = _FORCE_POS.get(name, 0)
    g = FORCE[name][pos:pos + x2d.shape[0]]
    assert g.shape == x2d.shape, (name, tuple(g.shape), tuple(x2d.shape), pos)
    _FORCE_POS[name] = pos + x2d.shape[0]
    return g.to(x2d.dtype)


def _relu(name, x):
    g = forced(name, x.reshape(-1, x.shape[-1]))
    y = F.relu(x) if g is None else x * g.reshape(x.shape)
    if TRACE is not None:
        TRACE.setdefault(name, []).append(y.detach().reshape(-1, y.shape[-1]))
    return y


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def _ln(sd, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def mh_attention(sd, p, q, k, v, heads=8):
    """transformer.py:319-341 (downsample_rate=1)."""
    q, k, v = _lin(sd, p + "q_proj", q), _lin(sd, p + "k_proj", k), _lin(sd, p + "v_proj", v)
    B, Nq, D = q.shape
    hd = D // heads
    sp = lambda t: t.view(B, t.shape[1], heads, hd).transpose(1, 2)
    q, k, v = sp(q), sp(k), sp(v)
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1)
    o = (a @ v).transpose(1, 2).reshape(B, Nq, D)
    return _lin(sd, p + "out_proj", o)


def two_way_block(sd, p, queries, keys):
    """transformer.py:255-283: post-LN; queries = mask feats [C,K,D], keys = text [C,1,D]."""
    queries = _ln(sd, p + "norm1", queries + mh_attention(sd, p + "self_attn.", queries, queries, queries))
    queries = _ln(sd, p + "norm2", queries + mh_attention(sd, p + "cross_attn_token_to_image.", queries, keys, keys))
    m = _lin(sd, p + "mlp.lin2", _relu(p + "mlp.lin1", _lin(sd, p + "mlp.lin1", queries)))
    queries = _ln(sd, p + "norm3", queries + m)
    keys = _ln(sd, p + "norm4", keys + mh_attention(sd, p + "cross_attn_image_to_token.", keys, queries, queries))
    return queries, keys


def upsample_feats(feats, size=256):
    """LISA.py:350-354: fp32 bilinear (align_corners=False) then back to the input dtype."""
    dt = feats.dtype
    return F.interpolate(feats.float(), size=(size, size), mode="bilinear", align_corners=False).to(dt)


def mask_pooling(feats_chw, segs):
    """LISA.py:201-218: feats [D,h,w], segs [K,h,w] -> [K,D]."""
    e, w = feats_chw.flatten(1), segs.flatten(1)
    return (w @ e.t()) / (w.sum(-1, keepdim=True) + 1e-8)


def mask_head(sd, pfx, segs_feature, text_feature):
    """LISA.py:363-391.  segs_feature [K,D], text_feature [C,D] ->
    (pred_iou [C,K,1], seg embeddings [C,K,D])."""
    C = text_feature.shape[0]
    t = text_feature.unsqueeze(1)
    s = segs_feature.unsqueeze(0)
    if C > 0:
        s = s.expand(C, -1, -1)
    for i in range(2):
        s, t = two_way_block(sd, f"{pfx}lisa_attention_layers.{i}.", s, t)
    s = _ln(sd, pfx + "lisa_norm_final_attn", s + mh_attention(sd, pfx + "lisa_final_attn.", s, t, t))
    iou = torch.sigmoid(_lin(sd, pfx + "lisa_iou_head.2", _relu(pfx + "lisa_iou_head.0", _lin(sd, pfx + "lisa_iou_head.0", s))))
    emb = _lin(sd, pfx + "lisa_embedding_head.2", _relu(pfx + "lisa_embedding_head.0", _lin(sd, pfx + "lisa_embedding_head.0", s)))
    return iou, emb


def cosine_scores(pred_embedding, seg_emb):
    """LISA.py:398-403: pred_embedding [1,D], seg_emb [K,D] -> [1,K]."""
    a = pred_embedding / pred_embedding.norm(dim=-1, keepdim=True)
    b = seg_emb / seg_emb.norm(dim=-1, keepdim=True)
    return a @ b.t()
