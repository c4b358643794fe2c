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


class DecodeState:
    """Everything a decode step touches, at fixed addresses (so the step can be captured once and replayed): the KV cache
    [layers][N, capacity, H] bf16 x 2, the device-side position {pos, pos + 1}, the step's input embeddings and its outputs."""

    def __init__(self, layers, N, cap, H, V, device):
        self.k = torch.empty((layers, N, cap, H), device=device, dtype=BF16)
        self.v = torch.empty((layers, N, cap, H), device=device, dtype=BF16)
        self.pos = torch.zeros((2,), device=device, dtype=torch.int32)       # [0] = position of the token being fed, [1] = keys present after it
        self.x = torch.empty((N, H), device=device, dtype=BF16)
        self.hidden = torch.empty((N, H), device=device, dtype=BF16)
        self.logits = torch.empty((N, V), device=device, dtype=BF16)
        self.N, self.cap, self.H = N, cap, H
        self.graph = None
        self.fused = False           # decode step on merged-LoRA weights (+ norm / SwiGLU on the GEMM's A load for a single sequence)
        self.qkv_w = None            # per-layer q|k|v weights with the LoRA deltas merged in (fused steps)


class GenerateMixin:
    def _merge_lora(self):
        """q|k|v weights with the LoRA deltas folded in (W + (alpha / r) B A on the q and v blocks, peft `merge_and_unload` -- what the
        reference's released checkpoints are, merge_lora_weights_and_save_hf_model.py), rebuilt from the CURRENT LoRA matrices at every
        `generate()` call into persistent buffers (3.2 GB at Llama-7B; ~2 ms): a decode step then needs no LoRA launches.  The rounding of
        W + delta to bf16 differs from the training forward's x W^T + (x A^T) B^T by one bf16 ulp of W."""
        c = self.config.llama
        F = _Direct
        H = c.hidden
        bufs = self.__dict__.setdefault("_merged_qkv", [None] * c.layers)
        s = c.lora_alpha / c.lora_r
        for i in range(c.layers):
            p = f"model.layers.{i}."
            w = self._wcat(p + "qkv", [p + f"self_attn.{n}_proj.weight" for n in "qkv"], F)
            if bufs[i] is None:
                bufs[i] = torch.empty_like(w)
            bufs[i].copy_(w)
            lp = p + "self_attn."
            for blk, n in ((0, "q_proj"), (2, "v_proj")):
                a_, b_ = self._w(lp + n + ".lora_A.default.weight", F), self._w(lp + n + ".lora_B.default.weight", F)
                r = a_.shape[0]
                if r % 8:                                     # the GEMM contracts in 16-byte chunks: pad the rank with zeros
                    rp = (r + 7) // 8 * 8
                    a_ = torch.cat([a_, torch.zeros((rp - r, H), device=a_.device, dtype=a_.dtype)], 0)
                    b_ = torch.cat([b_, torch.zeros((H, rp - r), device=b_.device, dtype=b_.dtype)], 1).contiguous()
                view = bufs[i][blk * H:(blk + 1) * H]
                ops.gemm(b_, a_, trans_w=True, alpha=s, residual=view, out=view)           # [H, r] @ [r, H] + W block
        return bufs

    def _decode_body(self, st):
        """One token per sequence through the decoder stack: st.x (embeddings of the token at position st.pos[0]) -> st.hidden, st.logits;
        K / V of the token are appended to the cache and the position advances.  No host-visible state: the step is a pure kernel
        sequence over fixed buffers, replayable from a hipGraph.  Six launches per layer for a single sequence (LoRA
        merged): q|k|v GEMM with the input RMSNorm on its A load, RoPE + KV append, attention, o_proj (+ residual), gate|up GEMM with the
        post-attention RMSNorm on its A load, down_proj (+ residual) with SwiGLU on its A load."""
        c = self.config.llama
        F = _Direct
        N, H = st.x.shape
        if st.fused:
            onload = N == 1      # RMSNorm / SwiGLU on the skinny GEMM's A load: every wave redoes the transform, a win for one row only (two rows: 4.02 vs
            #                      3.90 ms per token with the norm on load, 5.97 vs 4.34 with both): more rows keep the separate launches
            cos, sin, _ = self._rope(st.cap)
            hd, heads = c.head_dim, c.heads
            x = st.x
            att = torch.empty((N, H), device=x.device, dtype=BF16)
            scratch = ops.decode_attn_scratch(N, heads, x.device) if hd == 128 else None
            for i in range(c.layers):
                p = f"model.layers.{i}."
                wq = st.qkv_w[i] if st.qkv_w is not None else self._wcat(p + "qkv", [p + f"self_attn.{n}_proj.weight" for n in "qkv"], F)
                if onload:
                    qkv = ops.gemm(x, wq, a_norm_w=self._w(p + "input_layernorm.weight", F), a_norm_eps=c.eps)
                else:
                    qkv = ops.gemm(F.norm(x, self._w(p + "input_layernorm.weight", F), None, c.eps, True), wq)
                if hd == 128:                                  # RoPE + KV append + attention over the cache: one launch (+ the split's merge)
                    ops.decode_attn(qkv, cos, sin, st.k[i], st.v[i], st.pos, heads, hd, out=att, scratch=scratch)
                else:
                    ld = qkv.stride(0)
                    ops.rope_kv_append_(qkv, cos, sin, st.k[i], st.v[i], st.pos, heads, hd)
                    ops.attention(qkv, st.k[i], st.v[i], att, batch=N, heads=heads, Nq=1, Nk=st.cap, head_dim=hd, q_strides=(ld, hd, ld),
                                  k_strides=(st.cap * H, hd, H), v_strides=(st.cap * H, hd, H), o_strides=(H, hd, H), nk_dev=st.pos[1:])
                x = ops.gemm(att, self._w(p + "self_attn.o_proj.weight", F), residual=x)
                wgu = self._wcat(p + "gate_up", [p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight"], F)
                if onload:
                    gu = ops.gemm(x, wgu, a_norm_w=self._w(p + "post_attention_layernorm.weight", F), a_norm_eps=c.eps)
                else:
                    gu = ops.gemm(F.norm(x, self._w(p + "post_attention_layernorm.weight", F), None, c.eps, True), wgu)
                if onload:
                    x = ops.gemm(gu, self._w(p + "mlp.down_proj.weight", F), residual=x, a_swiglu=True)
                else:
                    x = ops.gemm(ops.swiglu(gu, c.inter), self._w(p + "mlp.down_proj.weight", F), residual=x)
            ops.norm(x, self._w("model.norm.weight", F), None, eps=c.eps, rms=True, out=st.hidden)
            ops.gemm(st.hidden, self._w("lm_head.weight", F), out=st.logits)
            st.pos.add_(1)
            return
        cos, sin, _ = self._rope(st.cap)
        s = c.lora_alpha / c.lora_r if c.lora_r > 0 else 0.0
        hd, heads = c.head_dim, c.heads
        x = st.x
        att = torch.empty((N, H), device=x.device, dtype=BF16)
        for i in range(c.layers):
            p = f"model.layers.{i}."
            h = F.norm(x, self._w(p + "input_layernorm.weight", F), None, c.eps, True)
            qkv = ops.gemm(h, self._wcat(p + "qkv", [p + f"self_attn.{n}_proj.weight" for n in "qkv"], F))      # skinny GEMM: a weight stream
            if c.lora_r > 0:
                lp = p + "self_attn."
                aq, bq = self._w(lp + "q_proj.lora_A.default.weight", F), self._w(lp + "q_proj.lora_B.default.weight", F)
                av, bv = self._w(lp + "v_proj.lora_A.default.weight", F), self._w(lp + "v_proj.lora_B.default.weight", F)
                if c.lora_r == 8:                              # rank-8 kernels: [h Aq^T | h Av^T] in one pass, then the two rank-8 updates
                    xa = ops.lora_down(h, aq, x2=h, w2=av)
                    ops.lora_apply_(qkv[:, :H], xa, bq, alpha=s)
                    ops.lora_apply_(qkv[:, 2 * H:], xa[:, 8:], bv, alpha=s)
                else:
                    ops.gemm(ops.gemm(h, aq), bq, residual=qkv[:, :H], out=qkv[:, :H], alpha=s)
                    ops.gemm(ops.gemm(h, av), bv, residual=qkv[:, 2 * H:], out=qkv[:, 2 * H:], alpha=s)
            ld = qkv.stride(0)
            ops.rope_kv_append_(qkv, cos, sin, st.k[i], st.v[i], st.pos, heads, hd)       # q, k rotated at the device-side position; k, v -> cache
            ops.attention(qkv, st.k[i], st.v[i], att, batch=N, heads=heads, Nq=1, Nk=st.cap, head_dim=hd, q_strides=(ld, hd, ld),
                          k_strides=(st.cap * H, hd, H), v_strides=(st.cap * H, hd, H), o_strides=(H, hd, H), nk_dev=st.pos[1:])
            x = ops.gemm(att, self._w(p + "self_attn.o_proj.weight", F), residual=x)
            h = F.norm(x, self._w(p + "post_attention_layernorm.weight", F), None, c.eps, True)
            gu = ops.gemm(h, self._wcat(p + "gate_up", [p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight"], F))
            x = ops.gemm(ops.swiglu(gu, c.inter), self._w(p + "mlp.down_proj.weight", F), residual=x)
        ops.norm(x, self._w("model.norm.weight", F), None, eps=c.eps, rms=True, out=st.hidden)
        ops.gemm(st.hidden, self._w("lm_head.weight", F), out=st.logits)
        st.pos.add_(1)

    def _decode_step(self, st, use_graph=True):
        """Run one decode step; from the second step of a state on, replay it from a hipGraph (captured once per (N, capacity): a step is
        ~420 launches of 5-20 us kernels, the host cannot issue them as fast as the GPU finishes them)."""
        if not use_graph:
            return self._decode_body(st)
        if st.graph is None:
            if not getattr(st, "warm", False):
                st.warm = True
                return self._decode_body(st)                    # first step of this state: eager (also warms every allocation)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):      # (another thread -- an RCCL watchdog -- may query events meanwhile: train.py::_capture)
                self._decode_body(st)
            st.graph = g                                        # capture records, it does not execute: fall through to the replay
        st.graph.replay()

    @torch.no_grad()
    def generate(self, images_clip, input_ids, max_new_tokens=32, eos_token_id=2, pad_token_id=0, use_graph=True, fuse_decode=True):
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
        cap = (T + max_new_tokens + 63) // 64 * 64
        states = self.__dict__.setdefault("_decode_states", {})
        st = states.get((N, cap))
        if st is None:
            st = states[(N, cap)] = DecodeState(cl.layers, N, cap, H, cl.vocab, dev)

        def keep_kv(i, qkv):                                   # qkv [N*T, 3H] after the in-place RoPE of q and k
            st.k[i, :, :T].copy_(qkv[:, H:2 * H].view(N, T, H))
            st.v[i, :, :T].copy_(qkv[:, 2 * H:3 * H].view(N, T, H))
        hidden_p = self._llama(embeds, plan.key_mask, F, kv_out=keep_kv)          # [N, T, H]
        fused = bool(fuse_decode)
        if fused != st.fused:
            st.fused, st.graph, st.warm = fused, None, False                      # a different kernel sequence: capture again
        st.qkv_w = self._merge_lora() if (fused and cl.lora_r > 0) else None
        st.pos.copy_(torch.tensor([T, T + 1], dtype=torch.int32))
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
            st.x.copy_(ops.gather_rows(emb_w, nxt))
            self._decode_step(st, use_graph)
            hidden[:, T + n_new - 1] = st.hidden
            logits = st.logits
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
