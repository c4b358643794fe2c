"""The init half of the reference's Python surface: start a finetune from the authors' inputs (reference `training.py:139-243`
`init_LISA_model`): an HF LLaVA directory (`LISAForCausalLM.from_pretrained(args.version, ...)`, `:157`), the SAM ViT-H checkpoint
(`--vision_pretrained`, `model/segment_anything/build_sam.py:98-107`), the HF CLIP ViT-L/14 directory (`--vision-tower`,
`model/llava/model/multimodal_encoder/clip_encoder.py:19-27`), a DINOv2 ViT-L/14 hub state dict (`model/LISA.py:48`), then
`resize_token_embeddings(len(tokenizer))` (`:229`), LoRA on the Llama q/v projections (`:183-227`) and the trainable set (`:231-241`).

Everything here is host-side file parsing + `copy_` into the device parameters (one-time, not on the timed path).  Tensors keep the
reference's key names (llmseg_amd/params.py), so loading is a key-prefix question:

    source                              keys in the file                         keys here
    HF LLaVA dir (safetensors / bin,    model.layers.*, model.embed_tokens.*,     unchanged (rows beyond the file's vocabulary: see
      single file or index.json shards)   model.norm.*, lm_head.*, model.mm_projector.*    `resize_token_embeddings`)
    sam_vit_h_4b8939.pth                image_encoder.*, prompt_encoder.*,         model.visual_model.<key>
                                          mask_decoder.*
    HF CLIP dir                         vision_model.* (+ text_model.*, ...)      model.vision_tower.vision_tower.vision_model.*
    DINOv2 hub state dict               cls_token, pos_embed, blocks.*, ...       model.visual_model_dinov2.<key>

Fresh modules (`text_hidden_fcs`, `lisa_*`: `LISA.py:54-121`) get torch's default initialisers, LoRA A kaiming-uniform(a = sqrt 5) / B
zeros (peft 0.4.0; absent here: PARITY UNPINNED, formulas from its published source).  No network: every path is a local file.
"""
import json
import math
import os
import re

import torch

SAM_PREFIX = "model.visual_model."
CLIP_PREFIX = "model.vision_tower.vision_tower."
DINO_PREFIX = "model.visual_model_dinov2."
_SKIP = (re.compile(r"\.rotary_emb\.inv_freq$"), re.compile(r"\.position_ids$"), re.compile(r"^pixel_(mean|std)$"))


# ------------------------------------------------------------------------------------------------ file readers
def _read_file(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    try:
        sd = torch.load(path, map_location="cpu", weights_only=True)
    except Exception:                                  # noqa: BLE001 -- old checkpoints pickle more than tensors
        # TRUST ASSUMPTION: a full unpickle executes code from the file.  Only reached for files the tensors-only reader rejects (the authors' 2023
        # `.bin` / `.pth` exports); point the loaders at files you would also hand to the reference's own `torch.load` (build_sam.py:105, HF `from_pretrained`).
        sd = torch.load(path, map_location="cpu", weights_only=False)
    for k in ("state_dict", "model", "module"):        # common wrappers
        if isinstance(sd, dict) and k in sd and isinstance(sd[k], dict) and not torch.is_tensor(sd[k]):
            sd = sd[k]
    return sd


def hf_weight_files(path):
    """An HF model directory (or a single weight file) -> ordered list of weight files.  Sharded checkpoints are described by
    `model.safetensors.index.json` / `pytorch_model.bin.index.json` ({"weight_map": {tensor name: shard file}}); safetensors wins when both
    formats are present, as in `from_pretrained`."""
    if os.path.isfile(path):
        return [path]
    for index, single in (("model.safetensors.index.json", "model.safetensors"), ("pytorch_model.bin.index.json", "pytorch_model.bin")):
        if os.path.exists(os.path.join(path, index)):
            with open(os.path.join(path, index)) as fh:
                wm = json.load(fh)["weight_map"]
            files = sorted(set(wm.values()))
            missing = [f for f in files if not os.path.exists(os.path.join(path, f))]
            if missing:
                raise FileNotFoundError(f"{index} names shards that are not in {path}: {missing[:3]}")
            return [os.path.join(path, f) for f in files]
        if os.path.exists(os.path.join(path, single)):
            return [os.path.join(path, single)]
    raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin (or their index.json) under {path}")


def iter_weights(path):
    """Yield (name, tensor) of every tensor of a checkpoint, one shard in host memory at a time."""
    for f in hf_weight_files(path):
        for k, v in _read_file(f).items():
            if torch.is_tensor(v):
                yield k, v


def hf_config(path):
    with open(os.path.join(path, "config.json")) as fh:
        return json.load(fh)


def config_from_hf(path, **overrides):
    """`LisaConfig` whose Llama part follows an HF LLaVA / Llama `config.json` (what `from_pretrained` reads before building modules)."""
    from .params import LisaConfig, LlamaConfig
    c = hf_config(path)
    if c.get("num_key_value_heads", c["num_attention_heads"]) != c["num_attention_heads"]:
        raise ValueError("grouped-query attention is not on the LLM-Seg path (LLaVA-Llama-7B/13B v1 are multi-head)")
    ll = LlamaConfig(hidden=c["hidden_size"], inter=c["intermediate_size"], layers=c["num_hidden_layers"], heads=c["num_attention_heads"],
                     vocab=c["vocab_size"], eps=c.get("rms_norm_eps", 1e-6), theta=c.get("rope_theta", 10000.0))
    cfg = LisaConfig(llama=ll, select_layer=c.get("mm_vision_select_layer", -2))
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


# ------------------------------------------------------------------------------------------------ copy into the model
def _own(model):
    return dict(model.params.named_parameters())


@torch.no_grad()
def _copy_in(model, items, rename, what, allow_extra_rows=(), skip_mismatched=()):
    """Copy (name, tensor) pairs into the model.  `rename(name) -> key | None`.  -> dict(loaded, ignored, mismatched).
    `allow_extra_rows`: parameters that may have MORE rows than the file (vocabulary grown by added tokens): the file's rows are copied,
    the remaining row indices are returned under `short` for `resize_token_embeddings`.  `skip_mismatched`: key prefixes whose tensors are
    optional extras of the file -- a shape mismatch there is recorded under `ignored` (with a warning) instead of raising."""
    own = _own(model)
    loaded, ignored, short = [], [], {}
    for name, t in items:
        key = rename(name)
        if key is None or key not in own:
            ignored.append(name)
            continue
        p = own[key]
        if tuple(t.shape) != tuple(p.shape):
            if key in allow_extra_rows and t.dim() == p.dim() and t.shape[1:] == p.shape[1:] and t.shape[0] < p.shape[0]:
                p[: t.shape[0]].copy_(t.to(device=p.device, dtype=p.dtype))
                short[key] = int(t.shape[0])
                loaded.append(key)
                continue
            if any(key.startswith(pre) for pre in skip_mismatched):
                import warnings
                warnings.warn(f"{what}: {name} has shape {tuple(t.shape)}, the model's {key} is {tuple(p.shape)}: skipped (optional tensor of the file)", RuntimeWarning, stacklevel=3)
                ignored.append(name)
                continue
            raise ValueError(f"{what}: {name} has shape {tuple(t.shape)}, the model's {key} is {tuple(p.shape)}")
        p.copy_(t.to(device=p.device, dtype=p.dtype))
        loaded.append(key)
    return {"loaded": loaded, "ignored": ignored, "short": short}


def _finish(model, res, expect_prefixes, what, allow_missing=()):
    """Every model tensor under `expect_prefixes` must have been loaded (a silently half-initialised tower is the failure this guards)."""
    got = set(res["loaded"])
    want = [k for k in _own(model) if any(k.startswith(p) for p in expect_prefixes) and not any(a in k for a in allow_missing)]
    res["missing"] = [k for k in want if k not in got]
    if res["missing"]:
        raise KeyError(f"{what}: the file lacks {len(res['missing'])} tensors of the model, e.g. {res['missing'][:4]}")
    model._invalidate_derived()
    return res


def load_llava(model, llava_dir):
    """The language side of `LISAForCausalLM.from_pretrained(args.version)` (training.py:157): Llama stack, embeddings, lm_head, mm_projector
    from an HF directory (sharded safetensors / bin).  The file's vocabulary may be smaller than the model's (tokens added afterwards):
    those rows are listed in the result's `short` and filled by `resize_token_embeddings`.  CLIP weights inside a LLaVA checkpoint
    (`model.vision_tower.*`, present in some exports) are loaded too when the shapes fit."""
    def rename(n):
        return None if any(p.search(n) for p in _SKIP) else n
    res = _copy_in(model, iter_weights(llava_dir), rename, "LLaVA checkpoint", allow_extra_rows=("model.embed_tokens.weight", "lm_head.weight"),
                   skip_mismatched=("model.vision_tower.",))
    return _finish(model, res, ("model.layers.", "model.embed_tokens.", "model.norm.", "lm_head.", "model.mm_projector."), "LLaVA checkpoint",
                   allow_missing=(".lora_",))


def load_sam(model, sam_ckpt):
    """`build_sam_vit_h(checkpoint)` (build_sam.py:98-107: `sam.load_state_dict(torch.load(f), strict=False)`): `sam_vit_h_4b8939.pth` keys
    under `model.visual_model.`.  Prompt encoder / mask decoder tensors load when the model was built with `LisaConfig(sam_decoder=True)`."""
    def rename(n):
        return None if any(p.search(n) for p in _SKIP) else SAM_PREFIX + n
    res = _copy_in(model, iter_weights(sam_ckpt), rename, "SAM checkpoint")
    return _finish(model, res, (SAM_PREFIX,), "SAM checkpoint")


def load_clip(model, clip_dir):
    """`CLIPVisionModel.from_pretrained(vision_tower)` (clip_encoder.py:19-27): the `vision_model.*` tensors of an HF CLIP directory (a full
    `CLIPModel` export also carries `text_model.*`, `visual_projection`, `logit_scale`: ignored) under the reference's prefix."""
    def rename(n):
        if any(p.search(n) for p in _SKIP) or not n.startswith("vision_model."):
            return None
        return CLIP_PREFIX + n
    res = _copy_in(model, iter_weights(clip_dir), rename, "CLIP checkpoint")
    return _finish(model, res, (CLIP_PREFIX,), "CLIP checkpoint")


def load_dinov2(model, state):
    """`torch.hub.load('facebookresearch/dinov2', 'dinov2_vitl14')` (LISA.py:48): its state dict (a dict or a file holding one; hub names:
    cls_token, pos_embed, mask_token, patch_embed.proj.*, blocks.N.{norm1,attn.qkv,attn.proj,ls1.gamma,norm2,mlp.fc1,mlp.fc2,ls2.gamma}, norm.*)."""
    items = state.items() if isinstance(state, dict) else iter_weights(state)
    res = _copy_in(model, items, lambda n: DINO_PREFIX + n, "DINOv2 state dict")
    return _finish(model, res, (DINO_PREFIX,), "DINOv2 state dict")


# ------------------------------------------------------------------------------------------------ fresh modules
def _uniform_(t, bound, gen):
    t.copy_(((torch.rand(t.shape, generator=gen, dtype=torch.float32) * 2 - 1) * bound).to(t.dtype))


@torch.no_grad()
def init_lisa_modules(model, seed=0):
    """`initialize_lisa_modules` (LISA.py:35-121) for the modules it CREATES: `text_hidden_fcs`, `lisa_dino_conv`, `lisa_attention_layers`,
    `lisa_final_attn`, `lisa_norm_final_attn`, `lisa_iou_head`, `lisa_embedding_head` with torch's default initialisers -- Linear / Conv2d:
    weight kaiming-uniform(a = sqrt 5) = U(+-1 / sqrt(fan_in)), bias U(+-1 / sqrt(fan_in)); LayerNorm: ones / zeros.  Drawn on the host from
    `seed` (the reference draws from torch's global generator: the values are unpinned by construction)."""
    gen = torch.Generator().manual_seed(seed)
    own = _own(model)
    fresh = [k for k in own if ".text_hidden_fcs." in k or ".lisa_" in k]
    for k in fresh:
        p = own[k]
        if k.endswith(".weight") and p.dim() >= 2:
            fan_in = p[0].numel()
            _uniform_(p, 1.0 / math.sqrt(fan_in), gen)
        elif k.endswith(".weight"):                                   # LayerNorm
            p.fill_(1.0)
        elif "norm" in k:
            p.zero_()
        else:                                                         # bias of a Linear / Conv: fan_in of its weight
            w = own[k[: -len("bias")] + "weight"]
            _uniform_(p, 1.0 / math.sqrt(w[0].numel()), gen)
    model._invalidate_derived()
    return fresh


@torch.no_grad()
def init_lora(model, seed=0):
    """peft 0.4.0 `Linear.reset_lora_parameters`: A kaiming-uniform(a = sqrt 5) = U(+-1 / sqrt(in_features)), B zeros -- the adapter starts as
    the identity.  (The model must have been built with `LlamaConfig.lora_r > 0`: training.py:183-227 wraps every Llama q_proj / v_proj,
    `--lora_target_modules q_proj,v_proj`, and nothing else -- the name filter there excludes the vision towers and every lisa_* module.)"""
    gen = torch.Generator().manual_seed(seed)
    own = _own(model)
    names = [k for k in own if ".lora_A." in k or ".lora_B." in k]
    if not names:
        raise ValueError("the model has no LoRA tensors (LlamaConfig.lora_r == 0)")
    for k in names:
        if ".lora_A." in k:
            _uniform_(own[k], 1.0 / math.sqrt(own[k].shape[1]), gen)
        else:
            own[k].zero_()
    model._invalidate_derived()
    return names


@torch.no_grad()
def resize_token_embeddings(model, short, mode="normal", seed=0, std=0.02):
    """`model.resize_token_embeddings(len(tokenizer))` (training.py:229) for a model ALLOCATED at the final vocabulary: fill the rows the
    checkpoint did not have (`short`: {key: rows present}, from `load_llava`) -- `[SEG]`, `<im_start>`, `<im_end>` (training.py:130-135).
    mode "normal" = transformers 4.29 (`_get_resized_embeddings` / `_get_resized_lm_head`: new matrix through `_init_weights` = N(0,
    initializer_range = 0.02), old rows copied); "mean" = the mean of the old rows (what later transformers releases centre their draw on).
    Applies to `model.embed_tokens.weight` and `lm_head.weight` alike."""
    gen = torch.Generator().manual_seed(seed)
    own = _own(model)
    filled = {}
    for key, n_old in short.items():
        p = own[key]
        n_new = p.shape[0] - n_old
        if mode == "normal":
            rows = torch.randn((n_new, p.shape[1]), generator=gen) * std
        elif mode == "mean":
            rows = p[:n_old].float().mean(0, keepdim=True).cpu().expand(n_new, -1)
        else:
            raise ValueError(mode)
        p[n_old:].copy_(rows.to(device=p.device, dtype=p.dtype))
        filled[key] = (n_old, p.shape[0])
    model._invalidate_derived()
    return filled


def load_pretrained(model, llava_dir, sam_ckpt=None, clip_dir=None, dinov2_sd=None, seed=0, new_rows="normal"):
    """`init_LISA_model`'s weight side in the reference's order: LLaVA weights -> CLIP tower (`initialize_vision_modules`) -> SAM + DINOv2 +
    fresh LISA modules (`initialize_lisa_modules`) -> LoRA -> `resize_token_embeddings` -> trainable set.  A tower whose file is None is
    required only when the model uses it: `config.backbone` decides between SAM and DINOv2 (the reference builds both: pass both to
    mirror it).  -> report dict per source."""
    rep = {"llava": load_llava(model, llava_dir)}
    if clip_dir is None and not any(k.startswith(CLIP_PREFIX) for k in rep["llava"]["loaded"]):
        raise ValueError("no CLIP weights: pass clip_dir (the reference loads openai/clip-vit-large-patch14 separately, training.py:47)")
    if clip_dir is not None:
        rep["clip"] = load_clip(model, clip_dir)
    has = lambda pfx: any(k.startswith(pfx) for k in _own(model))
    if sam_ckpt is not None:
        rep["sam"] = load_sam(model, sam_ckpt)
    elif model.config.backbone == "sam" and has(SAM_PREFIX):
        raise ValueError("backbone 'sam' needs vision_pretrained (the SAM ViT-H checkpoint, training.py:48)")
    if dinov2_sd is not None:
        rep["dinov2"] = load_dinov2(model, dinov2_sd)
    elif model.config.backbone == "dinov2" and has(DINO_PREFIX):
        raise ValueError("backbone 'dinov2' needs the DINOv2 ViT-L/14 hub state dict (no network here: torch.hub cannot fetch it)")
    rep["fresh"] = init_lisa_modules(model, seed)
    if model.config.llama.lora_r > 0:
        rep["lora"] = init_lora(model, seed + 1)
    rep["resized"] = resize_token_embeddings(model, rep["llava"]["short"], mode=new_rows, seed=seed + 2)
    model.set_trainable()
    return rep
