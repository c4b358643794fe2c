"""Oracle: SAM prompt encoder (text-embedding prompt) + mask decoder + mask post-processing as `LISAForCausalLM.evaluate` runs them
(test infrastructure; reference `model/LISA.py:523-557`).

Restated modules (all under /root/reference/model/segment_anything/modeling):
  prompt_encoder.py:140-186  PromptEncoder.forward with points = boxes = masks = None, text_embeds [b, 1, 256]:
                             sparse = text_embeds, dense = no_mask_embed broadcast over the 64 x 64 grid
  prompt_encoder.py:67-76, 189-229  get_dense_pe / PositionEmbeddingRandom: sin | cos of 2 pi (2 xy - 1) G, pixel centres (i + 0.5) / 64
  mask_decoder.py:75-166     MaskDecoder.forward / predict_masks (multimask_output = False -> mask token 0, IoU column 0)
  transformer.py:16-245      TwoWayTransformer (depth 2, 8 heads, mlp 2048, attention_downsample_rate 2), TwoWayAttentionBlock, Attention
  sam.py:137-172             postprocess_masks: bilinear to 1024 x 1024 (align_corners False) in fp32, crop to input_size, bilinear to original_size
Config: build_sam.py:56-102 (prompt_embed_dim 256, image_embedding_size 64, transformer depth 2 / 8 heads / mlp 2048).
PARITY: pinned by oracle/make_goldens.py::gold_sam_decoder against the imported reference modules (tests/golden/sam_decoder.pt).
"""
import math

import torch
import torch.nn.functional as F

PFX = "model.visual_model."


def decoder_shapes(pfx=PFX, D=256, mlp=2048):
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


def dense_pe(sd, pfx=PFX, g=64):
    """prompt_encoder.py:67-76,204-229 -> [D, g, g]"""
    G = sd[pfx + "prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"].float()
    grid = torch.ones((g, g), dtype=torch.float32)
    y = (grid.cumsum(0) - 0.5) / g
    x = (grid.cumsum(1) - 0.5) / g
    c = 2 * torch.stack([x, y], -1) - 1
    c = 2 * math.pi * (c @ G)
    return torch.cat([c.sin(), c.cos()], -1).permute(2, 0, 1)


def _attention(sd, p, q, k, v, heads=8):
    """transformer.py:185-245"""
    q = F.linear(q, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"])
    k = F.linear(k, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"])
    v = F.linear(v, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"])
    sep = lambda x: x.reshape(x.shape[0], x.shape[1], heads, x.shape[2] // heads).transpose(1, 2)
    q, k, v = sep(q), sep(k), sep(v)
    a = torch.softmax((q @ k.permute(0, 1, 3, 2)) / math.sqrt(q.shape[-1]), -1)
    o = (a @ v).transpose(1, 2)
    o = o.reshape(o.shape[0], o.shape[1], -1)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def _ln(sd, p, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def two_way_transformer(sd, t, image_embedding, image_pe, point_embedding):
    """transformer.py:62-108,140-182.  image_embedding / image_pe [b, C, h, w], point_embedding [b, n, C] -> (queries, keys)"""
    keys = image_embedding.flatten(2).permute(0, 2, 1)
    kpe = image_pe.flatten(2).permute(0, 2, 1)
    queries = point_embedding
    for i in range(2):
        p = f"{t}layers.{i}."
        if i == 0:                                                       # skip_first_layer_pe
            queries = _attention(sd, p + "self_attn.", queries, queries, queries)
        else:
            q = queries + point_embedding
            queries = queries + _attention(sd, p + "self_attn.", q, q, queries)
        queries = _ln(sd, p + "norm1", queries)
        queries = _ln(sd, p + "norm2", queries + _attention(sd, p + "cross_attn_token_to_image.", queries + point_embedding, keys + kpe, keys))
        m = F.linear(F.relu(F.linear(queries, sd[p + "mlp.lin1.weight"], sd[p + "mlp.lin1.bias"])), sd[p + "mlp.lin2.weight"], sd[p + "mlp.lin2.bias"])
        queries = _ln(sd, p + "norm3", queries + m)
        keys = _ln(sd, p + "norm4", keys + _attention(sd, p + "cross_attn_image_to_token.", keys + kpe, queries + point_embedding, queries))
    queries = queries + _attention(sd, t + "final_attn_token_to_image.", queries + point_embedding, keys + kpe, keys)
    return _ln(sd, t + "norm_final_attn", queries), keys


def _mlp(sd, p, x, n=3):
    for j in range(n):
        x = F.linear(x, sd[f"{p}layers.{j}.weight"], sd[f"{p}layers.{j}.bias"])
        if j < n - 1:
            x = F.relu(x)
    return x


def embed_points(sd, points, labels, pfx=PFX, input_image_size=(1024, 1024)):
    """prompt_encoder.py:77-97 with pad = True (no boxes): points [b, n, 2] (x, y) in the 1024-frame, labels [b, n] -> [b, n + 1, 256]."""
    G = sd[pfx + "prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"].float()
    pts = torch.cat([points.float() + 0.5, torch.zeros((points.shape[0], 1, 2))], 1)
    lab = torch.cat([labels.float(), -torch.ones((labels.shape[0], 1))], 1)
    c = pts.clone()
    c[:, :, 0] = c[:, :, 0] / input_image_size[1]
    c[:, :, 1] = c[:, :, 1] / input_image_size[0]
    c = 2 * math.pi * ((2 * c - 1) @ G)
    e = torch.cat([c.sin(), c.cos()], -1)
    e[lab == -1] = 0.0
    e[lab == -1] += sd[pfx + "prompt_encoder.not_a_point_embed.weight"]
    e[lab == 0] += sd[pfx + "prompt_encoder.point_embeddings.0.weight"]
    e[lab == 1] += sd[pfx + "prompt_encoder.point_embeddings.1.weight"]
    return e


def decode_masks(sd, image_embedding, text_embeds, pfx=PFX, sparse=None, multimask_output=False):
    """LISA.py:531-547 for ONE image: image_embedding [1, 256, 64, 64], text_embeds [b, 256] (the [SEG] embeddings of that image).
    -> (low_res_masks [b, 1, 256, 256], iou_predictions [b, 1]) with multimask_output = False.
    sparse [b, n, 256] (e.g. `embed_points`) replaces the text prompt; multimask_output = True -> mask tokens 1..3 (mask_decoder.py:97-104)."""
    m = pfx + "mask_decoder."
    if sparse is None:
        sparse = text_embeds[:, None, :]
    b = sparse.shape[0]
    dense = sd[pfx + "prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(b, -1, 64, 64)
    pe = dense_pe(sd, pfx)[None].to(image_embedding.dtype)
    out_tokens = torch.cat([sd[m + "iou_token.weight"], sd[m + "mask_tokens.weight"]], 0)[None].expand(b, -1, -1)
    tokens = torch.cat([out_tokens, sparse], 1)
    src = torch.repeat_interleave(image_embedding, b, 0) + dense
    pos = torch.repeat_interleave(pe, b, 0)
    hs, src = two_way_transformer(sd, m + "transformer.", src, pos, tokens)
    src = src.transpose(1, 2).reshape(b, 256, 64, 64)
    u = F.conv_transpose2d(src, sd[m + "output_upscaling.0.weight"], sd[m + "output_upscaling.0.bias"], stride=2)
    mu = u.mean(1, keepdim=True)                                          # LayerNorm2d (common.py:31-43), eps 1e-6
    va = (u - mu).pow(2).mean(1, keepdim=True)
    u = sd[m + "output_upscaling.1.weight"][:, None, None] * ((u - mu) / torch.sqrt(va + 1e-6)) + sd[m + "output_upscaling.1.bias"][:, None, None]
    u = F.gelu(u)
    u = F.gelu(F.conv_transpose2d(u, sd[m + "output_upscaling.3.weight"], sd[m + "output_upscaling.3.bias"], stride=2))
    hyper = torch.stack([_mlp(sd, f"{m}output_hypernetworks_mlps.{i}.", hs[:, 1 + i, :]) for i in range(4)], 1)
    masks = (hyper @ u.reshape(b, 32, 256 * 256)).reshape(b, 4, 256, 256)
    iou = _mlp(sd, m + "iou_prediction_head.", hs[:, 0, :])
    sl = slice(1, None) if multimask_output else slice(0, 1)
    return masks[:, sl], iou[:, sl]


def postprocess_masks(masks, input_size, original_size, img_size=1024):
    """sam.py:137-172"""
    x = F.interpolate(masks.float(), (img_size, img_size), mode="bilinear", align_corners=False)
    x = x[..., : input_size[0], : input_size[1]]
    return F.interpolate(x, original_size, mode="bilinear", align_corners=False)
