"""Generation half of `LISAForCausalLM.evaluate` (reference `model/LISA.py:477-521`; `prepare_inputs_for_generation`,
`model/llava/model/language_model/llava_llama.py:137-163`): greedy decoding with a KV cache, the hidden state of every fed token,
and the `[SEG]` embeddings `text_hidden_fcs` makes of them.

MI355X-first shape of the loop:
  * prefill = the path's own batched forward (`TrainableMixin._llama` with the no-grad kernels), which leaves RoPE-rotated K and V of
    every layer in its packed q|k|v buffer; they are copied once into the cache [layer][N, Tmax, H] (bf16, 0.5 MB per token and sequence
    at Llama-7B: 288 GB of HBM hold any batch the path sees);
  * a decode step feeds ONE token per sequence: RMSNorm -> q|k|v GEMM (+LoRA) on N rows -> RoPE at the step's position -> K, V rows
    appended to the cache -> attention of the single query over the cache (`llmseg_attn_fwd`, Nq = 1, strided K / V) -> o_proj -> MLP.
    At N <= 8 rows every GEMM is a weight stream (13.5 GB per token): HBM-bound, not MFMA-bound;
  * the reference generates WITHOUT a cache in its shipped configuration (`use_cache = False`) and reads the hidden states of its last
    forward; the cache yields the same tensor step by step (oracle/generate.py explains the equivalence and pins it).
HF greedy-search rules restated from `transformers==4.29.0 generation/utils.py::greedy_search` (third party): finished rows emit
`pad_token_id`, a row finishes on `eos_token_id`, the loop stops when all rows are finished or `max_new_tokens` tokens were added.
"""
import torch

from . import ops
from .trainable import _Direct

BF16 = torch.bfloat16


class KVCache:
    """K / V of every decoder layer, [layers][N, Tmax, H] bf16 each; `len` = tokens stored."""

    def __init__(self, layers, N, Tmax, H, device):
        self.k = torch.empty((layers, N, Tmax, H), device=device, dtype=BF16)
        self.v = torch.empty((layers, N, Tmax, H), device=device, dtype=BF16)
        self.N, self.Tmax, self.H, self.len = N, Tmax, H, 0


class GenerateMixin:
    def _decode_step(self, x, cache, logits_out=True):
        """One token per sequence through the decoder stack.  x bf16 [N, H] (token embeddings) at position `cache.len`.
        -> (final-norm hidden [N, H], logits [N, V] | None); the cache grows by one."""
        c = self.config.llama
        F = _Direct
        N, H = x.shape
        pos = cache.len
        assert pos < cache.Tmax
        cos, sin, _ = self._rope(cache.Tmax)
        s = c.lora_alpha / c.lora_r if c.lora_r > 0 else 0.0
        hd, heads = c.head_dim, c.heads
        att = torch.empty((N, H), device=x.device, dtype=BF16)
        for i in range(c.layers):
            p = f"model.layers.{i}."
            h = F.norm(x, self._w(p + "input_layernorm.weight", F), None, c.eps, True)
            if c.lora_r > 0:
                lp = p + "self_attn."
                qkv = F.lora_qkv(h, self._w(p + "qkv", F), self._w(lp + "q_proj.lora_A.default.weight", F),
                                 self._w(lp + "q_proj.lora_B.default.weight", F), self._w(lp + "v_proj.lora_A.default.weight", F),
                                 self._w(lp + "v_proj.lora_B.default.weight", F), s)
            else:
                qkv = ops.gemm(h, self._wcat(p + "qkv", [p + f"self_attn.{n}_proj.weight" for n in "qkv"], F))
            ld = qkv.stride(0)
            ops.rope_(qkv, cos[pos:pos + 1], sin[pos:pos + 1], N, 1, 2 * heads, hd, ld)       # q and k of every row at position `pos`
            kc, vc = cache.k[i], cache.v[i]
            kc[:, pos].copy_(qkv[:, H:2 * H])
            vc[:, pos].copy_(qkv[:, 2 * H:3 * H])
            ops.attention(qkv, kc, vc, att, batch=N, heads=heads, Nq=1, Nk=pos + 1, head_dim=hd, q_strides=(ld, hd, ld),
                          k_strides=(cache.Tmax * H, hd, H), v_strides=(cache.Tmax * H, hd, H), o_strides=(H, hd, H))
            x = ops.gemm(att, self._w(p + "self_attn.o_proj.weight", F), residual=x)
            h = F.norm(x, self._w(p + "post_attention_layernorm.weight", F), None, c.eps, True)
            gu = ops.gemm(h, self._wcat(p + "gate_up", [p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight"], F))
            x = ops.gemm(ops.swiglu(gu, c.inter), self._w(p + "mlp.down_proj.weight", F), residual=x)
        cache.len = pos + 1
        hidden = F.norm(x, self._w("model.norm.weight", F), None, c.eps, True)
        return hidden, (ops.gemm(hidden, self._w("lm_head.weight", F)) if logits_out else None)

    @torch.no_grad()
    def generate(self, images_clip, input_ids, max_new_tokens=32, eos_token_id=2, pad_token_id=0):
        """Greedy generation.  images_clip bf16 [N, 3, 224, 224] (one image per sequence), input_ids int64 [N, L] holding exactly one
        IMAGE_TOKEN_INDEX each, no padding (evaluate() passes no attention mask).
        -> (sequences int64 [N, L + n_new], hidden bf16 [N, T + n_new - 1, H]: final-norm hidden state of every token but the last)."""
        self.prepare()
        c = self.config
        cl = c.llama
        dev = self.device_
        N, L = input_ids.shape
        Pn, H = c.n_img_tokens, cl.hidden
        T = L - 1 + Pn
        assert max_new_tokens >= 1
        plan = self.make_plan(input_ids, None, torch.ones((N, L), dtype=torch.bool), list(range(N + 1)), None, inference=False)
        F = _Direct
        proj = self.encode_images(images_clip.to(dev, BF16))
        embeds = F.embed_splice(input_ids.to(dev).contiguous(), self._w("model.embed_tokens.weight", F), proj[1:], Pn, (Pn + 1) * H, plan.tok_index)
        cache = KVCache(cl.layers, N, T + max_new_tokens, H, dev)

        def keep_kv(i, qkv):                                   # qkv [N*T, 3H] after the in-place RoPE of q and k
            cache.k[i, :, :T].copy_(qkv[:, H:2 * H].view(N, T, H))
            cache.v[i, :, :T].copy_(qkv[:, 2 * H:3 * H].view(N, T, H))
        hidden_p = self._llama(embeds, plan.key_mask, F, kv_out=keep_kv)          # [N, T, H]
        cache.len = T
        hidden = torch.empty((N, T + max_new_tokens - 1, H), device=dev, dtype=BF16)
        hidden[:, :T] = hidden_p
        emb_w = self._w("model.embed_tokens.weight", F)
        logits = ops.gemm(hidden_p[:, -1].contiguous(), self._w("lm_head.weight", F))     # only the last position's logits are needed
        seqs = [input_ids.to(dev)]
        unfinished = torch.ones((N,), dtype=torch.int64, device=dev)
        n_new = 0
        while True:
            nxt = logits.float().argmax(-1)
            if eos_token_id is not None:
                nxt = nxt * unfinished + pad_token_id * (1 - unfinished)
            seqs.append(nxt[:, None])
            n_new += 1
            if eos_token_id is not None:
                unfinished = unfinished * (nxt != eos_token_id).long()
                if int(unfinished.max()) == 0:                 # the one host synchronisation of a step (HF's loop has the same)
                    break
            if n_new == max_new_tokens:
                break
            h1, logits = self._decode_step(ops.gather_rows(emb_w, nxt), cache)
            hidden[:, T + n_new - 1] = h1
        return torch.cat(seqs, 1), hidden[:, :T + n_new - 1]

    @torch.no_grad()
    def seg_embeddings(self, output_ids, hidden):
        """LISA.py:497-521: rows of `hidden` whose NEXT token is [SEG] (255 leading positions = the image-token expansion) through
        `text_hidden_fcs`; -> list (per sequence) of bf16 [n_seg, out_dim].  Gather first, then the two small GEMMs on those rows only."""
        F = _Direct
        N = output_ids.shape[0]
        Pn = self.config.n_img_tokens
        m = output_ids[:, 1:] == self.seg_token_idx
        m = torch.cat([torch.zeros((N, Pn - 1), dtype=torch.bool, device=m.device), m], 1)
        assert m.shape[1] == hidden.shape[1], (m.shape, hidden.shape)
        idx = m.reshape(-1).nonzero().flatten()
        cnt = [0] + m.sum(1).cumsum(0).tolist()
        rows = ops.gather_rows(hidden.reshape(-1, hidden.shape[-1]), idx)
        if rows.shape[0]:
            h = ops.gemm(rows, self._w("model.text_hidden_fcs.0.0.weight", F), bias=self._w("model.text_hidden_fcs.0.0.bias", F), act=ops.ACT_RELU)
            rows = ops.gemm(h, self._w("model.text_hidden_fcs.0.2.weight", F), bias=self._w("model.text_hidden_fcs.0.2.bias", F))
        else:
            rows = torch.empty((0, self.config.out_dim), device=hidden.device, dtype=BF16)
        return [rows[cnt[i]:cnt[i + 1]] for i in range(N)]
