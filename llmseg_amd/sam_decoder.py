"""SAM prompt encoder (text-embedding prompt) + mask decoder + mask post-processing: the mask half of `LISAForCausalLM.evaluate`
(reference `model/LISA.py:523-557`; `model/segment_anything/modeling/prompt_encoder.py:140-186`, `mask_decoder.py:75-166`,
`transformer.py:16-245`, `sam.py:137-172`; sizes `build_sam.py:56-102`).

Everything is token-major bf16 rows through the path's own kernels:
  * the two-way transformer's projections are GEMMs, its attentions `llmseg_attn_fwd` with separate strided q / k / v.  The cross
    attentions run at internal width 128 = 8 heads x 16 (`attention_downsample_rate = 2`); the kernel's smallest head is 32 wide, so
    the projection weights are zero-padded per head at weight-preparation time (q.k and the softmax are unchanged by zero columns, the
    padded v / out_proj columns contribute zeros) and the softmax scale stays 1 / sqrt(16);
  * a stride-2 2x2 transposed convolution of a [HW, Cin] row matrix is ONE GEMM against the [4 Cout, Cin] re-laid weight; its output
    rows, viewed as [4 HW, Cout], are the up-sampled pixels in a nested (y, x, dy, dx) order.  LayerNorm2d, GELU and the second
    transposed convolution act per pixel, so no pixel-shuffle pass is needed: the order is undone for free by the index arithmetic of
    the post-processing kernel (`llmseg_sam_postprocess`, nested = 1);
  * multimask_output = False: only mask token 0's hyper-network and IoU column 0 are evaluated.
"""
import math

import torch

from . import ops

BF16 = torch.bfloat16
PFX = "model.visual_model."


def _pad_heads_rows(w, b, heads=8, hd=16, to=32):
    """q/k/v projection [heads*hd, D] -> [heads*to, D] with zero rows after every head's hd rows (bias alike)."""
    D = w.shape[1]
    wp = torch.zeros((heads, to, D), device=w.device, dtype=w.dtype)
    wp[:, :hd] = w.view(heads, hd, D)
    bp = torch.zeros((heads, to), device=w.device, dtype=w.dtype)
    bp[:, :hd] = b.view(heads, hd)
    return wp.reshape(heads * to, D).contiguous(), bp.reshape(-1).contiguous()


def _pad_heads_cols(w, heads=8, hd=16, to=32):
    """out_proj [D, heads*hd] -> [D, heads*to] with zero columns after every head's hd columns."""
    D = w.shape[0]
    wp = torch.zeros((D, heads, to), device=w.device, dtype=w.dtype)
    wp[:, :, :hd] = w.view(D, heads, hd)
    return wp.reshape(D, heads * to).contiguous()


class SamDecoderMixin:
    @torch.no_grad()
    def _samdec(self):
        """One-time weight re-layouts of the decoder (cached; dropped by `_invalidate_derived`)."""
        d = self.prepare()
        if "samdec" in d:
            return d["samdec"]
        P = self.params
        assert PFX + "mask_decoder.iou_token.weight" in P.flat, "construct the model with LisaConfig(sam_decoder=True) to use evaluate()"
        dev = self.device_
        s = {}
        # dense positional encoding of the 64 x 64 grid (prompt_encoder.py:67-76,204-229), token-major [4096, 256]
        G = P[PFX + "prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"].float()
        ar = (torch.arange(64, device=dev, dtype=torch.float32) + 0.5) / 64
        xy = 2 * torch.stack([ar[None, :].expand(64, 64), ar[:, None].expand(64, 64)], -1) - 1           # (x, y) per pixel
        c = 2 * math.pi * (xy[..., :1] * G[0] + xy[..., 1:] * G[1])                # the 2-term contraction spelled out: no BLAS call on the path
        s["pos"] = torch.cat([c.sin(), c.cos()], -1).reshape(4096, 256).to(BF16).contiguous()
        t = PFX + "mask_decoder.transformer."
        for p in [f"{t}layers.{i}.{a}." for i in range(2) for a in ("cross_attn_token_to_image", "cross_attn_image_to_token")] + [t + "final_attn_token_to_image."]:
            for n in ("q_proj", "k_proj", "v_proj"):
                s[p + n + ".w"], s[p + n + ".b"] = _pad_heads_rows(P[p + n + ".weight"], P[p + n + ".bias"])
            s[p + "out_proj.w"] = _pad_heads_cols(P[p + "out_proj.weight"])
        m = PFX + "mask_decoder."
        for k, key in ((0, "up0"), (3, "up3")):
            w = P[f"{m}output_upscaling.{k}.weight"]                                       # ConvTranspose2d: [Cin, Cout, 2, 2]
            s[key + ".w"] = w.permute(2, 3, 1, 0).reshape(4 * w.shape[1], w.shape[0]).contiguous()      # rows (dy, dx, co)
            s[key + ".b"] = P[f"{m}output_upscaling.{k}.bias"].repeat(4).contiguous()
        s["out_tokens"] = torch.cat([P[m + "iou_token.weight"], P[m + "mask_tokens.weight"]], 0).contiguous()     # [5, 256]
        s["G"] = G
        d["samdec"] = s
        return s

    def _sd_attn(self, p, q_in, k_in, v_in, b, nq, nk, padded):
        """transformer.py:185-245 on row matrices: q_in [b*nq, 256], k_in / v_in [b*nk, 256] -> out_proj(attention) [b*nq, 256]."""
        P, S = self.params, self._samdec()
        if padded:
            w = lambda n: (S[p + n + ".w"], S[p + n + ".b"])
            wo = S[p + "out_proj.w"]
            scale = 1.0 / math.sqrt(16)
        else:
            w = lambda n: (P[p + n + ".weight"], P[p + n + ".bias"])
            wo = P[p + "out_proj.weight"]
            scale = 1.0 / math.sqrt(32)
        q = ops.gemm(q_in, w("q_proj")[0], bias=w("q_proj")[1])
        k = ops.gemm(k_in, w("k_proj")[0], bias=w("k_proj")[1])
        v = ops.gemm(v_in, w("v_proj")[0], bias=w("v_proj")[1])
        D = 256
        o = torch.empty((b * nq, D), device=q.device, dtype=BF16)
        ops.attention(q, k, v, o, batch=b, heads=8, Nq=nq, Nk=nk, head_dim=32, q_strides=(nq * D, 32, D), k_strides=(nk * D, 32, D),
                      v_strides=(nk * D, 32, D), o_strides=(nq * D, 32, D), scale=scale)
        return o, wo, P[p + "out_proj.bias"]

    @torch.no_grad()
    def sam_decode(self, feats_cl, text_embeds, sparse=None, multimask_output=False):
        """feats_cl bf16 [4096, 256]: one image's SAM embedding, channels-last rows; text_embeds bf16 [b, 256]: its [SEG] embeddings
        (or `sparse` bf16 [b, n, 256]: any sparse prompt tokens, e.g. `embed_points`).
        -> (low_res fp32 [b, 65536] mask logits in nested row order, iou bf16 [b, 1]); with multimask_output (mask tokens 1..3,
        mask_decoder.py:97-104): low_res [b, 3, 65536], iou [b, 3]."""
        P, S = self.params, self._samdec()
        if sparse is None:
            sparse = text_embeds.to(BF16)[:, None, :]
        b = sparse.shape[0]
        D, NT, NI = 256, 5 + sparse.shape[1], 4096
        m = PFX + "mask_decoder."
        t = m + "transformer."
        ln = lambda x, name, eps=1e-5: ops.norm(x, P[name + ".weight"], P[name + ".bias"], eps=eps, rms=False)
        pe_q = torch.cat([S["out_tokens"][None].expand(b, -1, -1), sparse.to(BF16)], 1).reshape(b * NT, D).contiguous()   # point_embedding
        keys = ops.add_rows(feats_cl.repeat(b, 1) if b > 1 else feats_cl.contiguous(), P[PFX + "prompt_encoder.no_mask_embed.weight"])   # + dense prompt
        queries = pe_q
        for i in range(2):
            p = f"{t}layers.{i}."
            if i == 0:                                                       # skip_first_layer_pe: queries REPLACED by the attention output
                o, wo, bo = self._sd_attn(p + "self_attn.", queries, queries, queries, b, NT, NT, False)
                queries = ops.gemm(o, wo, bias=bo)
            else:
                q = ops.add_rows(queries, pe_q, out=torch.empty_like(queries))
                o, wo, bo = self._sd_attn(p + "self_attn.", q, q, queries, b, NT, NT, False)
                queries = ops.gemm(o, wo, bias=bo, residual=queries)
            queries = ln(queries, p + "norm1")
            kpe = ops.add_rows(keys, S["pos"])                               # keys + key_pe (row % 4096)
            q = ops.add_rows(queries, pe_q, out=torch.empty_like(queries))
            o, wo, bo = self._sd_attn(p + "cross_attn_token_to_image.", q, kpe, keys, b, NT, NI, True)
            queries = ln(ops.gemm(o, wo, bias=bo, residual=queries), p + "norm2")
            h = ops.gemm(queries, P[p + "mlp.lin1.weight"], bias=P[p + "mlp.lin1.bias"], act=ops.ACT_RELU)
            queries = ln(ops.gemm(h, P[p + "mlp.lin2.weight"], bias=P[p + "mlp.lin2.bias"], residual=queries), p + "norm3")
            q = ops.add_rows(queries, pe_q, out=torch.empty_like(queries))
            o, wo, bo = self._sd_attn(p + "cross_attn_image_to_token.", kpe, q, queries, b, NI, NT, True)
            keys = ln(ops.gemm(o, wo, bias=bo, residual=keys), p + "norm4")
        kpe = ops.add_rows(keys, S["pos"])
        q = ops.add_rows(queries, pe_q, out=torch.empty_like(queries))
        o, wo, bo = self._sd_attn(t + "final_attn_token_to_image.", q, kpe, keys, b, NT, NI, True)
        hs = ln(ops.gemm(o, wo, bias=bo, residual=queries), t + "norm_final_attn").view(b, NT, D)
        # upscaling: two GEMM-form transposed convolutions; rows stay in nested pixel order
        u = ops.gemm(keys, S["up0.w"], bias=S["up0.b"]).view(b * NI * 4, D // 4)
        u = ops.act_(ln(u, m + "output_upscaling.1", 1e-6), ops.ACT_GELU)
        u = ops.gemm(u, S["up3.w"], bias=S["up3.b"], act=ops.ACT_GELU).view(b, NI * 16, D // 8)

        def mlp(x, pfx):
            for j in range(3):
                x = ops.gemm(x, P[f"{pfx}layers.{j}.weight"], bias=P[f"{pfx}layers.{j}.bias"], act=ops.ACT_RELU if j < 2 else ops.ACT_NONE)
            return x
        iou = mlp(hs[:, 0].contiguous(), m + "iou_prediction_head.")
        if not multimask_output:
            hyper = mlp(hs[:, 1].contiguous(), m + "output_hypernetworks_mlps.0.")       # [b, 32]: mask token 0
            low = torch.empty((b, NI * 16), device=keys.device, dtype=torch.float32)
            for bi in range(b):
                ops.gemm(u[bi], hyper[bi:bi + 1], out=low[bi].view(NI * 16, 1), out_f32=True)
            return low, iou[:, 0:1]
        hyper = torch.stack([mlp(hs[:, 1 + i].contiguous(), f"{m}output_hypernetworks_mlps.{i}.") for i in (1, 2, 3)], 1).contiguous()   # [b, 3, 32]
        low = torch.empty((b, 3, NI * 16), device=keys.device, dtype=torch.float32)
        # masks[b] = hyper[b] (3 x 32) . up[b]^T (32 x 65536): one strided-batched launch over the prompts
        ops.gemm_batched(hyper, u, low, M=3, N=NI * 16, K=D // 8, lda=D // 8, ldw=D // 8, ldc=NI * 16, batch=b, sA=3 * (D // 8), sW=NI * 16 * (D // 8),
                         sC=3 * NI * 16)
        return low, iou[:, 1:4].contiguous()

    @torch.no_grad()
    def evaluate(self, images_clip, images, input_ids, resize_list, original_size_list, max_new_tokens=32, tokenizer=None,
                 eos_token_id=2, pad_token_id=0):
        """`LISAForCausalLM.evaluate` (LISA.py:477-559): greedy generation, [SEG] embeddings, SAM image embedding, one mask per [SEG]
        token through the prompt encoder + mask decoder, masks resized to the original image.  -> (output_ids, [fp32 [n_seg, H, W]])."""
        assert self.config.backbone == "sam", "evaluate() decodes masks from the SAM image embedding"
        output_ids, hidden = self.generate(images_clip, input_ids, max_new_tokens=max_new_tokens, eos_token_id=eos_token_id, pad_token_id=pad_token_id)
        pred_embeddings = self.seg_embeddings(output_ids, hidden)
        g2 = self.config.sam.grid ** 2
        assert g2 == 4096 and self.config.sam.out_chans == 256, "the mask decoder is built for the 64 x 64 x 256 embedding (build_sam.py:63-66)"
        feats = self._sam_encoder_cl(images.to(self.device_, BF16))                      # [B * 4096, 256] channels-last rows
        pred_masks = []
        for i, pe in enumerate(pred_embeddings):
            H, W = int(original_size_list[i][0]), int(original_size_list[i][1])
            if pe.shape[0] == 0:
                pred_masks.append(torch.empty((0, H, W), device=self.device_, dtype=torch.float32))
                continue
            low, _ = self.sam_decode(feats[i * g2:(i + 1) * g2], pe)
            pred_masks.append(ops.sam_postprocess(low, resize_list[i], (H, W), img_size=self.config.sam.img))
        return output_ids, pred_masks
