"""Host-side mirror of the reference's `LISAForCausalLM` interface (reference `model/LISA.py:144-474`) on top of the
gfx950 kernels.  Same constructor kwargs, same `forward(**collate_dict)` / `model_forward` arguments and return keys,
same `get_visual_embs` / `get_dinov2_visual_embs` / `mask_pooling` helpers, same state-dict key names.

Everything dense runs in libllmseg_hip.so through `ops`; torch is used for device memory, integer index tables and
the state dict.  There is no CPU path: tensors must live on a HIP device.

Layout choices (MI355X-first, differ from the reference's eager code on purpose):
  * activations are token-major [rows, channels] bf16 everywhere (ViT grids are channels-last), so every Linear /
    1x1 conv / patch-embed is ONE TN GEMM with bias/activation/LayerScale/residual fused in its epilogue;
  * images of a batch are processed together (the reference loops per image, LISA.py:176-199);
  * SAM's window partition / un-partition are folded into the LayerNorm store and the attention store (row maps),
    the zero padding rows of the window buffer are written once and still act as keys (image_encoder.py:278-282);
  * q.R^T of the decomposed relative position is a strided-batched fp32-out GEMM, the bias add happens inside the
    fused attention kernel;
  * bilinear upsample + mask pooling is one pass over the proposals (llmseg_upsample_maskpool);
  * `text_hidden_fcs` runs on the gathered [SEG] rows only (identical result, ~300x fewer rows -- LISA.py:318-323).
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F  # only F.interpolate for the one-time DINOv2 pos-embed resize at weight-prep time

from . import ops
from .params import LisaConfig, ParamTree, fused_groups, init_random_, lisa_shapes
from .amg import AmgMixin
from .generate import GenerateMixin
from .sam_decoder import SamDecoderMixin
from .trainable import TrainableMixin

IMAGE_TOKEN_INDEX = -200
IGNORE_INDEX = -100
BF16 = torch.bfloat16
WIN_GATHER = os.environ.get("LLMSEG_WIN_NO_GATHER") is None      # A/B switch: SAM windows partitioned by the norm1 launch (row map) + q|k|v over the padded windows


def _pad_rows(t, rows):
    out = torch.zeros((rows,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    out[: t.shape[0]] = t
    return out


class LISAForCausalLM(TrainableMixin, GenerateMixin, SamDecoderMixin, AmgMixin, nn.Module):
    def __init__(self, config: LisaConfig, device="cuda", **kwargs):
        super().__init__()
        # reference kwargs (model/LISA.py:150-161, training.py:140-150)
        config.ce_loss_weight = kwargs.pop("ce_loss_weight", config.ce_loss_weight)
        config.align_loss_weight = kwargs.pop("align_loss_weight", config.align_loss_weight)
        config.regression_loss_weight = kwargs.pop("regression_loss_weight", config.regression_loss_weight)
        config.seg_token_idx = kwargs.pop("seg_token_idx", config.seg_token_idx)
        config.out_dim = kwargs.pop("out_dim", config.out_dim)
        # the init half of the surface (training.py:140-171): kept and ACTED on by `initialize_vision_modules` / `initialize_lisa_modules` /
        # `from_pretrained` (llmseg_amd/pretrained.py) -- round 4 dropped them silently
        if kwargs.pop("train_mask_decoder", False):
            raise NotImplementedError("train_mask_decoder=True: LLM-Seg never trains SAM's mask decoder (training.py:60 default False; the "
                                      "decoder is only reached by evaluate()), and this package keeps it frozen")
        self.vision_pretrained = kwargs.pop("vision_pretrained", None)      # path of the SAM ViT-H checkpoint (build_sam.py:98-107)
        self.vision_tower = kwargs.pop("vision_tower", None)                # HF CLIP ViT-L/14: a LOCAL directory here (no network)
        self.use_mm_start_end = kwargs.pop("use_mm_start_end", True)        # read by the collate (utils/dataset.py:73-83), stored for it
        if kwargs:
            raise TypeError(f"unexpected keyword arguments {sorted(kwargs)} (reference surface: model/LISA.py:150-161)")
        self.config = config
        self.seg_token_idx = config.seg_token_idx
        self.device_ = torch.device(device)
        assert self.device_.type == "cuda", "llmseg_amd has no CPU path (the HIP kernels are the product)"
        self.shapes = lisa_shapes(config)
        self.params = ParamTree(self.shapes, self.device_, BF16, fused_groups(config))
        self._derived = None
        self._maps = {}

    # the reference's state-dict keys start at the top module ("model.layers...", "lm_head.weight")
    def state_dict(self, *a, **k):
        return {n[len("params."):]: v for n, v in super().state_dict(*a, **k).items()}

    def load_state_dict(self, sd, strict=True):
        own = {n: p for n, p in self.params.named_parameters()}
        missing = [n for n in own if n not in sd]
        unexpected = [n for n in sd if n not in own]
        if strict and (missing or unexpected):
            raise KeyError(f"missing {missing[:5]} unexpected {unexpected[:5]}")
        with torch.no_grad():
            for n, p in own.items():
                if n in sd:
                    p.copy_(sd[n].to(device=p.device, dtype=p.dtype))
        self._invalidate_derived()
        return missing, unexpected

    def init_random(self, seed=0):
        init_random_(self.params, self.shapes, seed)
        self._invalidate_derived()
        return self

    def _invalidate_derived(self):
        """Weights changed wholesale (checkpoint load / re-init): drop every tensor derived from them (re-laid-out conv / position
        tables, transposed copies of frozen weights) and tell an attached optimizer to re-read its fp32 master copies."""
        self._derived = None
        self.__dict__.pop("_wt_cache", None)
        for hook in self.__dict__.get("_weight_hooks", []):
            hook()

    def get_model(self):
        return self

    def get_vision_tower(self):
        """The reference returns the CLIP module (llava_arch.py:90-91); here the tower's parameters live under this prefix of `params`."""
        return self.params._modules["model"]._modules["vision_tower"]

    # ------------------------------------------------------------------------------------------ init half (training.py:139-243)
    def initialize_vision_modules(self, config=None):
        """`LlavaMetaModel.initialize_vision_modules` (llava_arch.py:43-82, called at training.py:168): build the CLIP tower and load its
        weights.  The tower's tensors already exist (ParamTree); this loads them from `vision_tower` when that is a local HF directory."""
        import os
        from . import pretrained
        if not (isinstance(self.vision_tower, str) and os.path.isdir(self.vision_tower)):
            raise FileNotFoundError(f"vision_tower={self.vision_tower!r} is not a local directory: there is no network here, download "
                                    "openai/clip-vit-large-patch14 and pass its path")
        return pretrained.load_clip(self, self.vision_tower)

    def initialize_lisa_modules(self, config=None, seed=0, dinov2_state=None):
        """`LisaMetaModel.initialize_lisa_modules` (LISA.py:35-121, called at training.py:171): SAM ViT-H from `vision_pretrained`
        (None leaves it as initialised, as `build_sam_vit_h(None)` does), DINOv2 from `dinov2_state` (the hub download of LISA.py:48 is
        impossible offline: pass the state dict or a file), and the freshly initialised text_hidden_fcs / lisa_* modules."""
        from . import pretrained
        rep = {}
        if self.vision_pretrained is not None:
            rep["sam"] = pretrained.load_sam(self, self.vision_pretrained)
        if dinov2_state is not None:
            rep["dinov2"] = pretrained.load_dinov2(self, dinov2_state)
        rep["fresh"] = pretrained.init_lisa_modules(self, seed)
        return rep

    @classmethod
    def from_pretrained(cls, version, torch_dtype=BF16, device="cuda", dinov2_state=None, lora_r=8, lora_alpha=16, lora_dropout=0.05,
                        backbone="dinov2", seed=0, new_rows="normal", vocab_size=None, sam_decoder=False, towers=None, **model_args):
        """`init_LISA_model` (training.py:139-243) in one call: config.json of the HF LLaVA directory `version` -> model at the tokenizer's
        final vocabulary (`vocab_size` = len(tokenizer) after training.py:130-135 added [SEG] / <im_start> / <im_end>; default: the file's + 1,
        i.e. 32004 for the LLaVA-Lightning v1-1 checkpoints whose 32003 rows already hold the two image tags) -> weights
        (`pretrained.load_pretrained`) -> LoRA -> resized embeddings -> trainable set.  `model_args`: the reference's kwargs
        (`vision_pretrained`, `vision_tower`, `seg_token_idx`, loss weights, ...)."""
        from . import pretrained
        assert torch_dtype == BF16, "the HIP path computes in bf16 (training.py:151-156 `--precision bf16`)"
        cfg = pretrained.config_from_hf(version, backbone=backbone, sam_decoder=sam_decoder, **(towers or {}))     # towers: {"clip" / "sam" / "dino": config} when not ViT-L / ViT-H / ViT-L
        file_vocab = cfg.llama.vocab
        if vocab_size is None:
            import warnings
            warnings.warn(f"from_pretrained: vocab_size not given -- assuming len(tokenizer) = the file's {file_vocab} + 1 ([SEG] added, training.py:130-135).  A checkpoint "
                          "that already holds [SEG] (a LISA / LLM-Seg export) needs vocab_size = its own row count and the matching seg_token_idx.", RuntimeWarning, stacklevel=2)
        cfg.llama.vocab = int(vocab_size) if vocab_size is not None else file_vocab + 1          # pass len(tokenizer); default: the file's + [SEG]
        cfg.llama.lora_r, cfg.llama.lora_alpha, cfg.llama.lora_dropout = lora_r, lora_alpha, lora_dropout
        m = cls(cfg, device=device, **model_args)
        m.load_report = pretrained.load_pretrained(m, version, sam_ckpt=m.vision_pretrained, clip_dir=m.vision_tower, dinov2_sd=dinov2_state,
                                                   seed=seed, new_rows=new_rows)
        return m

    # ------------------------------------------------------------------------------------------ derived weights
    @torch.no_grad()
    def prepare(self):
        """One-time weight re-layouts (not part of the timed path): patch-embed / 3x3-conv weights as GEMM operands,
        class/position tables, padded relative-position tables, RoPE tables."""
        if self._derived is not None:
            return self._derived
        c, P, dev = self.config, self.params, self.device_
        d = {}
        # CLIP
        vp = "model.vision_tower.vision_tower.vision_model."
        kp = 3 * c.clip.patch ** 2
        kpad = (kp + 7) // 8 * 8
        d["clip.patch_w"] = F.pad(P[vp + "embeddings.patch_embedding.weight"].reshape(c.clip.dim, kp), (0, kpad - kp)).contiguous()
        pos = P[vp + "embeddings.position_embedding.weight"].float().clone()
        pos[0] += P[vp + "embeddings.class_embedding"].float()
        d["clip.pos"] = pos.to(BF16)
        d["clip.kpad"] = kpad
        # DINOv2
        if "model.visual_model_dinov2.cls_token" in P.flat:
            dp = "model.visual_model_dinov2."
            kp = 3 * c.dino.patch ** 2
            kpad = (kp + 7) // 8 * 8
            d["dino.patch_w"] = F.pad(P[dp + "patch_embed.proj.weight"].reshape(c.dino.dim, kp), (0, kpad - kp)).contiguous()
            d["dino.kpad"] = kpad
            d["dino.pos_cache"] = {}
            d["dino.conv_w"] = P["model.lisa_dino_conv.weight"].reshape(c.out_dim, c.dino.dim).contiguous()
        # SAM
        if "model.visual_model.image_encoder.pos_embed" in P.flat:
            sp = "model.visual_model.image_encoder."
            s = c.sam
            hd = s.dim // s.heads
            d["sam.patch_w"] = P[sp + "patch_embed.proj.weight"].reshape(s.dim, 3 * s.patch ** 2).contiguous()
            d["sam.pos"] = P[sp + "pos_embed"].reshape(s.grid * s.grid, s.dim).contiguous()
            d["sam.neck0_w"] = P[sp + "neck.0.weight"].reshape(s.out_chans, s.dim).contiguous()
            d["sam.neck2_w"] = P[sp + "neck.2.weight"].permute(0, 2, 3, 1).reshape(s.out_chans, 9 * s.out_chans).contiguous()
            for i in range(s.depth):
                sz = s.grid if i in s.global_idx else s.window
                ld = (2 * sz - 1 + 3) // 4 * 4 if i in s.global_idx else 32     # window tables feed the fused kernel: 32 MFMA rows
                d[f"sam.relh.{i}"] = _pad_rows(P[f"{sp}blocks.{i}.attn.rel_pos_h"], ld)
                d[f"sam.relw.{i}"] = _pad_rows(P[f"{sp}blocks.{i}.attn.rel_pos_w"], ld)
        self._derived = d
        return d

    def _rope(self, T):
        key = ("rope", T)
        if key not in self._maps:
            c = self.config.llama
            inv = 1.0 / (c.theta ** (torch.arange(0, c.head_dim, 2, dtype=torch.float32, device=self.device_) / c.head_dim))
            ang = torch.outer(torch.arange(T, dtype=torch.float32, device=self.device_), inv)
            sin = ang.sin().contiguous()
            self._maps[key] = (ang.cos().contiguous(), sin, (-sin).contiguous())      # (cos, sin, -sin): forward and inverse rotation
        return self._maps[key]

    def _window_maps(self, B, g, ws):
        """Row maps of SAM's window_partition / window_unpartition (image_encoder.py:263-318) for B images."""
        key = ("win", B, g, ws)
        if key not in self._maps:
            gp = (g + ws - 1) // ws * ws
            nw = gp // ws
            yy, xx = torch.meshgrid(torch.arange(g), torch.arange(g), indexing="ij")
            win = (yy // ws) * nw + (xx // ws)
            pos = (yy % ws) * ws + (xx % ws)
            part1 = (win * ws * ws + pos).reshape(-1)                                   # token -> row within one image
            per_img = nw * nw * ws * ws
            part = torch.cat([part1 + b * per_img for b in range(B)]).to(torch.int32)
            unpart = torch.full((B * per_img,), -1, dtype=torch.int32)
            unpart[part.long()] = torch.arange(B * g * g, dtype=torch.int32)
            self._maps[key] = (part.to(self.device_), unpart.to(self.device_), nw * nw, per_img)
        return self._maps[key]

    # --------------------------------------------------------------------------------------------------- towers
    def _vit_block(self, x, p, heads, batch, n_tok, eps, act, names, gamma=(None, None)):
        """One pre-LN ViT block on token-major x [batch*n_tok, D] (CLIP / DINOv2 naming differs, `names` maps it)."""
        P = self.params
        D = x.shape[1]
        h = ops.norm(x, P[p + names["ln1"] + ".weight"], P[p + names["ln1"] + ".bias"], eps=eps)
        qkv = ops.gemm(h, P[p + names["qkv"] + ".weight"], bias=P[p + names["qkv"] + ".bias"])
        a = ops.attention_packed(qkv, batch, n_tok, heads, D // heads)
        x = ops.gemm(a, P[p + names["proj"] + ".weight"], bias=P[p + names["proj"] + ".bias"], residual=x, gamma=gamma[0])
        h = ops.norm(x, P[p + names["ln2"] + ".weight"], P[p + names["ln2"] + ".bias"], eps=eps)
        h = ops.gemm(h, P[p + names["fc1"] + ".weight"], bias=P[p + names["fc1"] + ".bias"], act=act)
        return ops.gemm(h, P[p + names["fc2"] + ".weight"], bias=P[p + names["fc2"] + ".bias"], residual=x, gamma=gamma[1])

    _CLIP_NAMES = dict(ln1="layer_norm1", qkv="self_attn.qkv", proj="self_attn.out_proj", ln2="layer_norm2", fc1="mlp.fc1", fc2="mlp.fc2")
    _DINO_NAMES = dict(ln1="norm1", qkv="attn.qkv", proj="attn.proj", ln2="norm2", fc1="mlp.fc1", fc2="mlp.fc2")

    def encode_images(self, images_clip):
        """CLIP ViT-L/14 penultimate-layer patch tokens -> mm_projector (clip_encoder.py:41-60, llava_arch.py:93-96).
        Returns the projected tokens for ALL 1+P rows per image, [N*(P+1), H]; row 0 of each image (CLS) is unused."""
        c, P, d = self.config, self.params, self.prepare()
        N = images_clip.shape[0]
        n_tok = c.n_img_tokens + 1
        cols = ops.patchify(images_clip, c.clip.patch, d["clip.kpad"], rows_per_img=n_tok, row_off=1)
        x = ops.gemm(cols, d["clip.patch_w"])
        x = ops.add_rows(x, d["clip.pos"])
        vp = "model.vision_tower.vision_tower.vision_model."
        x = ops.norm(x, P[vp + "pre_layrnorm.weight"], P[vp + "pre_layrnorm.bias"], eps=c.clip.eps)
        n_run = c.clip.layers + 1 + c.select_layer if c.select_layer < 0 else c.select_layer
        for i in range(n_run):
            x = self._vit_block(x, f"{vp}encoder.layers.{i}.", c.clip.heads, N, n_tok, c.clip.eps, ops.ACT_QUICKGELU, self._CLIP_NAMES)
        return ops.gemm(x, P["model.mm_projector.weight"], bias=P["model.mm_projector.bias"])

    def _dino_pos(self, gh, gw):
        d, P, c = self.prepare(), self.params, self.config.dino
        key = (gh, gw)
        if key not in d["dino.pos_cache"]:
            dp = "model.visual_model_dinov2."
            pe = P[dp + "pos_embed"].float()
            M = int(math.isqrt(pe.shape[1] - 1))
            patch = pe[:, 1:]
            if (gh, gw) != (M, M):   # one-time weight transform: bicubic resize of the learned table (fp32)
                patch = F.interpolate(patch.reshape(1, M, M, -1).permute(0, 3, 1, 2), size=(gh, gw), mode="bicubic",
                                      align_corners=False).permute(0, 2, 3, 1).reshape(1, gh * gw, -1)
            tab = torch.cat([pe[:, :1] + P[dp + "cls_token"].float(), patch + P[dp + "patch_embed.proj.bias"].float()], 1)[0]
            d["dino.pos_cache"][key] = tab.to(BF16).contiguous()
        return d["dino.pos_cache"][key]

    def _dinov2_tokens(self, images):
        """DINOv2 ViT-L/14 x_norm tokens (incl. CLS row) [B*(1+gh*gw), D] (LISA.py:186-199; hub arithmetic restated)."""
        c, P, d = self.config.dino, self.params, self.prepare()
        B, _, Hh, Ww = images.shape
        gh, gw = Hh // c.patch, Ww // c.patch
        n_tok = gh * gw + 1
        cols = ops.patchify(images, c.patch, d["dino.kpad"], rows_per_img=n_tok, row_off=1)
        x = ops.gemm(cols, d["dino.patch_w"])                      # conv bias is folded into the position table
        x = ops.add_rows(x, self._dino_pos(gh, gw))
        dp = "model.visual_model_dinov2."
        for i in range(c.layers):
            p = f"{dp}blocks.{i}."
            x = self._vit_block(x, p, c.heads, B, n_tok, c.eps, ops.ACT_GELU, self._DINO_NAMES, gamma=(P[p + "ls1.gamma"], P[p + "ls2.gamma"]))
        return ops.norm(x, P[dp + "norm.weight"], P[dp + "norm.bias"], eps=c.eps), n_tok, (gh, gw)

    def get_dinov2_visual_embs(self, pixel_values):
        """Reference API (LISA.py:186-199): [B, D, gh, gw] view of the patch tokens."""
        x, n_tok, (gh, gw) = self._dinov2_tokens(pixel_values.to(BF16))
        B = pixel_values.shape[0]
        return x.view(B, n_tok, -1)[:, 1:].permute(0, 2, 1).reshape(B, -1, gh, gw)

    def _sam_encoder_cl(self, images):
        """SAM ViT image encoder, channels-last output [B*g*g, out_chans] (image_encoder.py:110-125)."""
        s, P, d = self.config.sam, self.params, self.prepare()
        B = images.shape[0]
        g, D, nh = s.grid, s.dim, s.heads
        hd = D // nh
        sp = "model.visual_model.image_encoder."
        cols = ops.patchify(images, s.patch, 3 * s.patch ** 2)
        x = ops.gemm(cols, d["sam.patch_w"], bias=P[sp + "patch_embed.proj.bias"])
        x = ops.add_rows(x, d["sam.pos"])
        part, unpart, n_win, per_img = self._window_maps(B, g, s.window)
        winbuf = None
        gather = hd == 80 and s.window == 14 and WIN_GATHER and P[sp + "blocks.0.attn.qkv.bias"].dtype == BF16
        for i in range(s.depth):
            p = f"{sp}blocks.{i}."
            glob = i in s.global_idx
            sz = g if glob else s.window
            ld = d[f"sam.relh.{i}"].shape[0]
            if glob:
                h = ops.norm(x, P[p + "norm1.weight"], P[p + "norm1.bias"], eps=s.eps)
                batch, n_tok = B, g * g
            elif gather:
                # window_partition, its zero padding and window_unpartition happen inside the attention kernel (llmseg_attn_args.win_grid): q|k|v of the real tokens only --
                # the reference pads AFTER norm1 (image_encoder.py:178-183), so a padded token's q|k|v row is the projection's bias, which the kernel reads instead
                h = ops.norm(x, P[p + "norm1.weight"], P[p + "norm1.bias"], eps=s.eps)
                batch, n_tok = B * n_win, s.window * s.window
            else:
                if winbuf is None:
                    winbuf = torch.zeros((B * per_img, D), device=x.device, dtype=BF16)   # padding rows stay zero
                h = ops.norm(x, P[p + "norm1.weight"], P[p + "norm1.bias"], eps=s.eps, out=winbuf, row_map=part)
                batch, n_tok = B * n_win, s.window * s.window
            qkv = ops.gemm(h, P[p + "attn.qkv.weight"], bias=P[p + "attn.qkv.bias"])
            rows = batch * n_tok
            a = torch.empty((B * g * g, D), device=x.device, dtype=BF16)
            if (not glob) and gather:
                ops.attention_packed(qkv, batch, n_tok, nh, hd, out=a, rel_tab_h=d[f"sam.relh.{i}"], rel_tab_w=d[f"sam.relw.{i}"],
                                     grid_hw=(sz, sz), win_pad=(g, P[p + "attn.qkv.bias"]))
            elif (not glob) and hd == 80 and sz == 14:
                # fused: q.R^T is computed inside the attention kernel from the (32-row padded) tables
                ops.attention_packed(qkv, batch, n_tok, nh, hd, out=a, rel_tab_h=d[f"sam.relh.{i}"], rel_tab_w=d[f"sam.relw.{i}"],
                                     grid_hw=(sz, sz), o_row_map=unpart)
            else:
                rel = torch.empty((2, nh, rows, ld), device=x.device, dtype=torch.float32)
                for j, tab in enumerate((d[f"sam.relh.{i}"], d[f"sam.relw.{i}"])):
                    ops.gemm_batched(qkv, tab, rel[j], M=rows, N=2 * sz - 1, K=hd, lda=3 * D, ldw=hd, ldc=ld, batch=nh, sA=hd, sW=0,
                                     sC=rows * ld, out_f32=True)
                ops.attention_packed(qkv, batch, n_tok, nh, hd, out=a, rel_h=rel[0], rel_w=rel[1], rel_ld=ld, grid_hw=(sz, sz),
                                     o_row_map=None if glob else unpart)
            x = ops.gemm(a, P[p + "attn.proj.weight"], bias=P[p + "attn.proj.bias"], residual=x)
            h = ops.norm(x, P[p + "norm2.weight"], P[p + "norm2.bias"], eps=s.eps)
            h = ops.gemm(h, P[p + "mlp.lin1.weight"], bias=P[p + "mlp.lin1.bias"], act=ops.ACT_GELU)
            x = ops.gemm(h, P[p + "mlp.lin2.weight"], bias=P[p + "mlp.lin2.bias"], residual=x)
        y = ops.gemm(x, d["sam.neck0_w"])
        y = ops.norm(y, P[sp + "neck.1.weight"], P[sp + "neck.1.bias"], eps=s.eps)
        y = ops.gemm(ops.im2col3x3(y, B, g, g, s.out_chans), d["sam.neck2_w"])
        return ops.norm(y, P[sp + "neck.3.weight"], P[sp + "neck.3.bias"], eps=s.eps)

    def get_visual_embs(self, pixel_values):
        """Reference API (LISA.py:173-184): SAM image embeddings [B, out_chans, g, g] (a view of the channels-last result)."""
        s = self.config.sam
        y = self._sam_encoder_cl(pixel_values.to(BF16))
        return y.view(pixel_values.shape[0], s.grid, s.grid, s.out_chans).permute(0, 3, 1, 2)

    def mask_pooling(self, image_embeddings, weight_maps):
        """Reference API (LISA.py:201-218) on an already-upsampled [D,h,w] map: identity interpolation (g == S)."""
        Dd, hh, ww = image_embeddings.shape
        assert hh == ww
        feat_cl = image_embeddings.permute(1, 2, 0).reshape(hh * ww, Dd).contiguous().to(BF16)
        return ops.upsample_maskpool(feat_cl, weight_maps.to(BF16).contiguous(), hh, hh)
