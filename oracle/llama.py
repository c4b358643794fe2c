"""Oracle: Llama decoder stack as the reference runs it (test infrastructure).

Third-party arithmetic: `transformers==4.29.0` (reference `requirements.txt:276`),
call sites `model/llava/model/language_model/llava_llama.py:21-22,93-102`.
LoRA: `peft==0.4.0` (reference `requirements.txt:198`, wiring `training.py:183-227`)
-- not installed here: PARITY UNPINNED, published formula restated.

Functional style: `sd` is a flat dict name -> tensor using the reference's
state-dict names under `pfx` (e.g. "model.").
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class LlamaCfg:
    hidden: int = 4096
    inter: int = 11008
    layers: int = 32
    heads: int = 32
    vocab: int = 32004
    eps: float = 1e-6
    theta: float = 10000.0
    lora_r: int = 0          # 0 = no LoRA
    lora_alpha: float = 16.0
    lora_dropout: float = 0.0   # reference: 0.05 (training.py:91); applied only when a dropout state is passed (training mode)

    @property
    def head_dim(self):
        return self.hidden // self.heads


def rmsnorm(x, w, eps):
    # HF LlamaRMSNorm: variance in fp32, normalise, cast back, then scale.
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return w * y.to(x.dtype)


def rope_tables(T, hd, theta, device):
    # HF LlamaRotaryEmbedding: inv_freq over even dims, table = cat(freqs, freqs).
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32, device=device) / hd))
    ang = torch.outer(torch.arange(T, dtype=torch.float32, device=device), inv)
    ang = torch.cat([ang, ang], -1)
    return ang.cos(), ang.sin()


def apply_rope(x, cos, sin):
    # x [N, heads, T, hd]; rotate_half = cat(-x2, x1)
    h = x.shape[-1] // 2
    rot = torch.cat([-x[..., h:], x[..., :h]], -1)
    return (x.float() * cos + rot.float() * sin).to(x.dtype)


def lora_linear(x, sd, name, cfg, drop=None):
    """y = x W^T (+ (alpha/r) B A dropout(x) when LoRA tensors exist; dropout = identity at eval).
    drop = (seed, offset, stream) of the counter-based mask (oracle/dropout.py) or None."""
    y = F.linear(x, sd[name + ".weight"])
    a = sd.get(name + ".lora_A.default.weight")
    if a is not None and cfg.lora_r > 0:
        b = sd[name + ".lora_B.default.weight"]
        xd = x
        if drop is not None and cfg.lora_dropout > 0:
            from . import dropout as _do
            xd = _do.apply(x.reshape(-1, x.shape[-1]), drop[0], drop[1], drop[2], cfg.lora_dropout).reshape(x.shape)
        y = y + (cfg.lora_alpha / cfg.lora_r) * F.linear(F.linear(xd, a), b)
    return y


def additive_mask(attention_mask, T, dtype, device):
    """Causal + key-padding additive mask, finfo.min where masked (HF 4.29 _prepare_decoder_attention_mask)."""
    neg = torch.finfo(dtype).min
    causal = torch.full((T, T), neg, dtype=dtype, device=device).triu(1)
    m = causal[None, None].expand(attention_mask.shape[0], 1, T, T).clone()
    pad = ~attention_mask.bool()
    m = m.masked_fill(pad[:, None, None, :], neg)
    return m


def decoder_layer(sd, p, h, mask, cos, sin, cfg, drop=None):
    """drop = (seed, offset, layer index) -> dropout streams 2 layer (q_proj) and 2 layer + 1 (v_proj)."""
    N, T, H = h.shape
    nh, hd = cfg.heads, cfg.head_dim
    x = rmsnorm(h, sd[p + "input_layernorm.weight"], cfg.eps)
    dq = None if drop is None else (drop[0], drop[1], 2 * drop[2])
    dv = None if drop is None else (drop[0], drop[1], 2 * drop[2] + 1)
    q = lora_linear(x, sd, p + "self_attn.q_proj", cfg, dq).view(N, T, nh, hd).transpose(1, 2)
    k = lora_linear(x, sd, p + "self_attn.k_proj", cfg).view(N, T, nh, hd).transpose(1, 2)
    v = lora_linear(x, sd, p + "self_attn.v_proj", cfg, dv).view(N, T, nh, hd).transpose(1, 2)
    q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
    s = q @ k.transpose(-1, -2) / math.sqrt(hd) + mask
    s = torch.max(s, torch.tensor(torch.finfo(s.dtype).min, dtype=s.dtype))
    pr = torch.softmax(s, -1, dtype=torch.float32).to(q.dtype)
    o = (pr @ v).transpose(1, 2).reshape(N, T, H)
    h = h + F.linear(o, sd[p + "self_attn.o_proj.weight"])
    x = rmsnorm(h, sd[p + "post_attention_layernorm.weight"], cfg.eps)
    g = F.linear(x, sd[p + "mlp.gate_proj.weight"])
    u = F.linear(x, sd[p + "mlp.up_proj.weight"])
    return h + F.linear(F.silu(g) * u, sd[p + "mlp.down_proj.weight"])


def llama_model(sd, pfx, inputs_embeds, attention_mask, cfg, dropout_state=None):
    """Returns the HF `hidden_states` tuple: input of each layer, then the final-norm output.
    dropout_state = (seed, offset): LoRA dropout active (training mode), None = eval."""
    N, T, _ = inputs_embeds.shape
    cos, sin = rope_tables(T, cfg.head_dim, cfg.theta, inputs_embeds.device)
    mask = additive_mask(attention_mask, T, inputs_embeds.dtype, inputs_embeds.device)
    h = inputs_embeds
    hs = []
    for i in range(cfg.layers):
        hs.append(h)
        h = decoder_layer(sd, f"{pfx}layers.{i}.", h, mask, cos, sin, cfg, None if dropout_state is None else (dropout_state[0], dropout_state[1], i))
    hs.append(rmsnorm(h, sd[pfx + "norm.weight"], cfg.eps))
    return hs


def shifted_ce(logits, labels, vocab):
    # llava_llama.py:108-118: shift, flatten, CrossEntropyLoss() (mean over labels != -100)
    sl = logits[..., :-1, :].reshape(-1, vocab)
    tl = labels[..., 1:].reshape(-1)
    return F.cross_entropy(sl, tl, ignore_index=-100)
