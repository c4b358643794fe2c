"""Import the read-only reference (/root/reference) as a parity oracle.  BUILD CONTAINER ONLY.

Test infrastructure: used by oracle/make_goldens.py to pin the restatement and to
produce tests/golden/*.  Never imported by the product, never shipped to the GPU box
(/root/reference does not exist there), copies no reference source.

The shims only neutralise imports and device calls the
hot path never needs on CPU:
  * cv2 / skimage / torchvision stubs while `model.segment_anything` imports,
  * transformers>=5 registry collision for the "llava" model type, dead MPT family,
  * `.cuda()` / `empty_cache()` -> no-ops,
  * CLIP `from_pretrained` and `torch.hub.load` (no network) -> random-init stand-ins
    of a caller-chosen size.
"""
import importlib
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference"
_state = {"ready": False}


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m


def setup(clip_cfg_kwargs=None, dino_cfg_kwargs=None):
    """Install shims and import the reference packages.  Idempotent."""
    if _state["ready"]:
        return
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    _stub("cv2")
    _stub("skimage")
    _stub("skimage.transform", resize=None)
    tv = ["torchvision", "torchvision.ops", "torchvision.ops.boxes",
          "torchvision.transforms", "torchvision.transforms.functional"]
    _stub(tv[0]); _stub(tv[1]); _stub(tv[2], batched_nms=None, box_area=None)
    _stub(tv[3]); _stub(tv[4], resize=None, to_pil_image=None)
    importlib.import_module("model.segment_anything")
    for t in tv:
        del sys.modules[t]

    from transformers import (AutoConfig, AutoModelForCausalLM, CLIPVisionConfig, CLIPVisionModel,
                              Dinov2Config, Dinov2Model)
    AutoConfig.register = staticmethod(lambda *a, **k: None)
    AutoModelForCausalLM.register = classmethod(lambda cls, *a, **k: None)
    _stub("model.llava.model.language_model.llava_mpt", LlavaMPTConfig=object, LlavaMPTForCausalLM=object)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None

    ck = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
              image_size=224, patch_size=14)
    ck.update(clip_cfg_kwargs or {})
    clip_cfg = CLIPVisionConfig(**ck)
    clip_cfg._attn_implementation = "eager"
    CLIPVisionConfig.from_pretrained = classmethod(lambda cls, *a, **k: clip_cfg)
    CLIPVisionModel.from_pretrained = classmethod(lambda cls, *a, **k: CLIPVisionModel(clip_cfg))
    import model.llava.model.multimodal_encoder.clip_encoder as ce
    ce.CLIPImageProcessor = types.SimpleNamespace(from_pretrained=lambda *a, **k: None)

    dk = dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, image_size=518, patch_size=14,
              mlp_ratio=4)
    dk.update(dino_cfg_kwargs or {})

    class DinoStandIn(nn.Module):
        """Same I/O contract as hub dinov2_vitl14.forward_features (LISA.py:192-193)."""

        def __init__(self):
            super().__init__()
            cfg = Dinov2Config(**dk)
            cfg._attn_implementation = "eager"
            self.m = Dinov2Model(cfg)

        def forward_features(self, x):
            return {"x_norm_patchtokens": self.m(pixel_values=x).last_hidden_state[:, 1:]}

    torch.hub.load = lambda *a, **k: DinoStandIn()
    _state.update(ready=True, clip_cfg=clip_cfg, dino_kwargs=dk)


def setup_utils():
    """Import the reference's `utils` package (datasets, collate, target functions).  Its modules import cv2, pycocotools, h5py,
    skimage, matplotlib at the top: none is installed, none is called by `collate_fn_new`; two ARE called by the target functions and
    are spelled with what they run on:
      * `skimage.transform.resize(gt, (H, W), anti_aliasing=False, preserve_range=True, order=0)` (utils/utils.py:240,260) ==
        `scipy.ndimage.zoom(float64 image, out / in, order=0, mode="mirror", grid_mode=True)` -- skimage 0.22's own implementation of that call;
      * `pycocotools.mask.decode(list of RLE dicts) -> uint8 [H, W, K]` (utils/sam_mask_reader.py:87) by `oracle.targets.rle_decode`
        (maskApi.c restated; PARITY UNPINNED for the codec -- what this import pins is the reader's ordering / top-50 / padding).
    -> (utils.dataset, utils.utils, utils.sam_mask_reader) modules."""
    setup()
    import numpy as np
    import scipy.ndimage as ndi
    from . import targets as otargets

    def sk_resize(image, output_shape, order=None, mode="reflect", cval=0, clip=True, preserve_range=False, anti_aliasing=None, **_):
        assert order == 0 and not anti_aliasing and preserve_range, "only the call the reference makes is spelled out"
        img = np.asarray(image).astype(np.float64)
        return ndi.zoom(img, [o / i for o, i in zip(output_shape, img.shape)], order=0, mode="mirror", cval=cval, grid_mode=True)

    def coco_decode(rles):
        if isinstance(rles, dict):
            return otargets.rle_decode(rles)
        return np.stack([otargets.rle_decode(r) for r in rles], -1)

    _stub("skimage.transform", resize=sk_resize)
    sys.modules["skimage"].transform = sys.modules["skimage.transform"]
    _stub("skimage.io")
    _stub("pycocotools")
    _stub("pycocotools.mask", decode=coco_decode)
    sys.modules["pycocotools"].mask = sys.modules["pycocotools.mask"]
    _stub("pycocotools.coco", COCO=object)
    _stub("h5py")
    _stub("matplotlib"); _stub("matplotlib.pyplot")
    _stub("matplotlib.patches", Polygon=object, Rectangle=object)
    _stub("matplotlib.collections", PatchCollection=object)
    for m in ("utils.utils", "utils.sam_mask_reader"):          # imported earlier with the None stubs? re-import against the working ones
        sys.modules.pop(m, None)
    D = importlib.import_module("utils.dataset")
    return D, importlib.import_module("utils.utils"), importlib.import_module("utils.sam_mask_reader")


class _ClipHiddenStates(nn.Module):
    """transformers-5.x records `hidden_states` through forward hooks that get duplicated when the
    tower is re-entered from several call paths (observed: 7 entries for a 3-layer model), which
    breaks `hidden_states[-2]`.  This wrapper rebuilds the 4.29 list -- [embeddings(+pre-LN),
    out_0, ..., out_{L-1}] -- by calling the HF sub-modules in order; the arithmetic stays HF's."""

    def __init__(self, hf):
        super().__init__()
        self.hf = hf

    @property
    def dtype(self):
        return self.hf.dtype

    @property
    def device(self):
        return self.hf.device

    @property
    def config(self):
        return self.hf.config

    def forward(self, images, output_hidden_states=True):
        vm = getattr(self.hf, "vision_model", self.hf)
        h = vm.pre_layrnorm(vm.embeddings(images))
        hs = [h]
        for layer in vm.encoder.layers:
            o = layer(h, None)
            h = o[0] if isinstance(o, (tuple, list)) else o
            hs.append(h)
        return types.SimpleNamespace(hidden_states=tuple(hs))


def build_lisa(llama_kwargs, seg_token_idx=32000):
    """Construct the reference LISAForCausalLM the way training.py:140-171 does (minus from_pretrained)."""
    setup()
    from model.LISA import LISAForCausalLM
    from model.llava.model.language_model.llava_llama import LlavaConfig
    cfg = LlavaConfig(**llama_kwargs)
    cfg._attn_implementation = "eager"
    cfg.mm_vision_select_layer = -2
    cfg.mm_hidden_size = _state["clip_cfg"].hidden_size
    cfg.mm_vision_select_feature = "patch"
    cfg.pretrain_mm_mlp_adapter = None
    m = LISAForCausalLM(cfg, train_mask_decoder=False, out_dim=256, seg_token_idx=seg_token_idx,
                        vision_pretrained=None, vision_tower="openai/clip-vit-large-patch14",
                        use_mm_start_end=True)
    cfg.vision_tower = cfg.mm_vision_tower
    m.get_model().initialize_vision_modules(m.get_model().config)
    m.get_model().initialize_lisa_modules(m.get_model().config)
    vt = m.get_model().get_vision_tower()
    vt.vision_tower = _ClipHiddenStates(vt.vision_tower)
    return m


def hf_dino_to_hub_names(hf_sd, pfx_hf, pfx_hub, layers):
    """Map HF Dinov2Model state-dict names onto the hub names the oracle/product use."""
    out = {}
    g = lambda k: hf_sd[pfx_hf + k]
    out[pfx_hub + "cls_token"] = g("embeddings.cls_token")
    out[pfx_hub + "pos_embed"] = g("embeddings.position_embeddings")
    out[pfx_hub + "mask_token"] = g("embeddings.mask_token")
    out[pfx_hub + "patch_embed.proj.weight"] = g("embeddings.patch_embeddings.projection.weight")
    out[pfx_hub + "patch_embed.proj.bias"] = g("embeddings.patch_embeddings.projection.bias")
    out[pfx_hub + "norm.weight"], out[pfx_hub + "norm.bias"] = g("layernorm.weight"), g("layernorm.bias")
    for i in range(layers):
        h, u = f"encoder.layer.{i}.", f"{pfx_hub}blocks.{i}."
        for a, b in (("norm1", "norm1"), ("norm2", "norm2")):
            out[u + b + ".weight"], out[u + b + ".bias"] = g(h + a + ".weight"), g(h + a + ".bias")
        out[u + "attn.qkv.weight"] = torch.cat([g(h + f"attention.attention.{n}.weight") for n in ("query", "key", "value")], 0)
        out[u + "attn.qkv.bias"] = torch.cat([g(h + f"attention.attention.{n}.bias") for n in ("query", "key", "value")], 0)
        out[u + "attn.proj.weight"], out[u + "attn.proj.bias"] = g(h + "attention.output.dense.weight"), g(h + "attention.output.dense.bias")
        out[u + "ls1.gamma"], out[u + "ls2.gamma"] = g(h + "layer_scale1.lambda1"), g(h + "layer_scale2.lambda1")
        out[u + "mlp.fc1.weight"], out[u + "mlp.fc1.bias"] = g(h + "mlp.fc1.weight"), g(h + "mlp.fc1.bias")
        out[u + "mlp.fc2.weight"], out[u + "mlp.fc2.bias"] = g(h + "mlp.fc2.weight"), g(h + "mlp.fc2.bias")
    return out


def hub_to_hf_dino_names(hub_sd, pfx_hub, pfx_hf, layers):
    """Inverse of the above (splits the fused qkv)."""
    out = {}
    g = lambda k: hub_sd[pfx_hub + k]
    out[pfx_hf + "embeddings.cls_token"] = g("cls_token")
    out[pfx_hf + "embeddings.position_embeddings"] = g("pos_embed")
    out[pfx_hf + "embeddings.mask_token"] = g("mask_token")
    out[pfx_hf + "embeddings.patch_embeddings.projection.weight"] = g("patch_embed.proj.weight")
    out[pfx_hf + "embeddings.patch_embeddings.projection.bias"] = g("patch_embed.proj.bias")
    out[pfx_hf + "layernorm.weight"], out[pfx_hf + "layernorm.bias"] = g("norm.weight"), g("norm.bias")
    for i in range(layers):
        h, u = f"{pfx_hf}encoder.layer.{i}.", f"blocks.{i}."
        for n in ("norm1", "norm2"):
            out[h + n + ".weight"], out[h + n + ".bias"] = g(u + n + ".weight"), g(u + n + ".bias")
        qw, kw, vw = g(u + "attn.qkv.weight").chunk(3, 0)
        qb, kb, vb = g(u + "attn.qkv.bias").chunk(3, 0)
        for n, w, b in (("query", qw, qb), ("key", kw, kb), ("value", vw, vb)):
            out[h + f"attention.attention.{n}.weight"], out[h + f"attention.attention.{n}.bias"] = w, b
        out[h + "attention.output.dense.weight"], out[h + "attention.output.dense.bias"] = g(u + "attn.proj.weight"), g(u + "attn.proj.bias")
        out[h + "layer_scale1.lambda1"], out[h + "layer_scale2.lambda1"] = g(u + "ls1.gamma"), g(u + "ls2.gamma")
        out[h + "mlp.fc1.weight"], out[h + "mlp.fc1.bias"] = g(u + "mlp.fc1.weight"), g(u + "mlp.fc1.bias")
        out[h + "mlp.fc2.weight"], out[h + "mlp.fc2.bias"] = g(u + "mlp.fc2.weight"), g(u + "mlp.fc2.bias")
    return out
