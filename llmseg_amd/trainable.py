"""The trainable half of the path (mixed into `LISAForCausalLM`): LLaVA splice + Llama-7B stack (+LoRA) + lm_head/CE,
`text_hidden_fcs`, mask pooling, the mask-selection transformer and the losses -- reference `model/LISA.py:254-474`,
`model/llava/model/language_model/llava_llama.py:55-135`, `model/transformer.py:215-341`, `model/loss.py:50-94`.

Every op goes through a small namespace `F`: `_Direct` calls the kernels straight (inference / no-grad), `_Auto` routes the
same calls through the autograd Functions of `autograd.py` so that `loss.backward()` runs the HIP backward kernels.
What is trainable follows the reference (`training.py:183-241`): LoRA A/B on every q_proj/v_proj, `embed_tokens`,
`lm_head`, `text_hidden_fcs`, every `lisa_*` module; CLIP, mm_projector, SAM, DINOv2 and the Llama base weights are frozen.

Host-side index plumbing (where the `<image>` token sits, which hidden rows are `[SEG]`, how conversations map to images, the
spliced labels) is computed ONCE per batch into a `BatchPlan` (`make_plan`); `model_forward(..., plan=plan)` then issues kernels only --
no device->host synchronisation, so a whole fwd+bwd micro-step can be captured in a hipGraph (`llmseg_amd/train.py`).
"""
import torch

from . import autograd as ag
from . import ops

IMAGE_TOKEN_INDEX = -200
IGNORE_INDEX = -100
BF16 = torch.bfloat16


class _Direct:
    grad = False
    linear = staticmethod(lambda x, w, b=None, act=ops.ACT_NONE, residual=None, wt=None: ops.gemm(x, w, bias=b, act=act, residual=residual))
    norm = staticmethod(lambda x, w, b=None, eps=1e-5, rms=False: ops.norm(x, w, b, eps=eps, rms=rms))
    norm_pass = staticmethod(lambda x, w, b=None, eps=1e-5, rms=False, pre=None: ((ops.norm(x, w, b, eps=eps, rms=rms) if pre is None else pre), x))

    @staticmethod
    def linear_norm(x, w, residual, wt, norm_w, eps):
        """(x @ w^T + residual, RMSNorm of that * norm_w) in one GEMM call (`llmseg_gemm_args.norm_out`)."""
        h = torch.empty((x.shape[0], w.shape[0]), device=x.device, dtype=BF16)
        return ops.gemm(x, w, residual=residual, norm_w=norm_w, norm_eps=eps, norm_out=h), h
    attn_packed = staticmethod(lambda qkv, batch, n, heads, hd, causal=False, key_mask=None:
                               ops.attention_packed(qkv, batch, n, heads, hd, causal=causal, key_mask=key_mask))
    swiglu = staticmethod(lambda gu, inter: ops.swiglu(gu, inter))
    embed_splice = staticmethod(lambda ids, emb, feats, P, fs, tok=None: ops.embed_splice(ids, emb, feats, P, feats_stride_n=fs))
    gather_rows = staticmethod(lambda x, idx: ops.gather_rows(x, idx))
    maskpool = staticmethod(lambda feat, segs, g, S: ops.upsample_maskpool(feat, segs, g, S))

    @staticmethod
    def rope_attn(qkv, rope, batch, n, heads, hd, causal, key_mask, pre_rotated=False):
        if not pre_rotated:
            ops.rope_(qkv, rope[0], rope[1], batch * n, n, 2 * heads, hd, qkv.stride(0))
        return ops.attention_packed(qkv, batch, n, heads, hd, causal=causal, key_mask=key_mask)

    @staticmethod
    def mlp(x, wgu, wgu_t, wd, wd_t, residual, norm_w, eps):
        M, inter = x.shape[0], wd.shape[1]
        gu = torch.empty((M, 2 * inter), device=x.device, dtype=BF16)
        h = torch.empty((M, inter), device=x.device, dtype=BF16)
        ops.gemm(x, wgu, out=gu, swiglu_out=h)
        if norm_w is None:
            return ops.gemm(h, wd, residual=residual), None
        pre = torch.empty((M, wd.shape[0]), device=x.device, dtype=BF16)
        return ops.gemm(h, wd, residual=residual, norm_w=norm_w, norm_eps=eps, norm_out=pre), pre

    @staticmethod
    def lora_qkv(x, wqkv, aq, bq, av, bv, s, wqkv_t=None, drop=None, rope=None):
        H = wqkv.shape[1]
        if aq.shape[0] == 8:
            return ag.lora_qkv_fused(x, wqkv, aq, bq, av, bv, s, None, rope=rope)[0]
        assert rope is None
        qkv = ops.gemm(x, wqkv)
        ops.gemm(ops.gemm(x, aq), bq, residual=qkv[:, :H], out=qkv[:, :H], alpha=s)
        ops.gemm(ops.gemm(x, av), bv, residual=qkv[:, 2 * H:], out=qkv[:, 2 * H:], alpha=s)
        return qkv

    @staticmethod
    def ce(logits, labels):
        acc = ops.ce_loss(logits, labels)
        return acc[0] / acc[1]

    @staticmethod
    def align_reg(e, t, pred, gt_iou, gt_iop):
        return ops.align_reg_loss(e, t, gt_iou, pred, gt_iop)

    @staticmethod
    def bcast_add(s, add, Cn, K):
        for ci in range(Cn):
            blk = s[ci * K:(ci + 1) * K]
            ops.add_rows(blk, add[ci:ci + 1].contiguous(), out=blk)
        return s

    @staticmethod
    def cross_attn_1q(q, kv, Cn, K, heads, hd):
        D = heads * hd
        o = torch.empty((Cn, D), device=q.device, dtype=BF16)
        ops.attention(q, kv, kv[:, D:], o, batch=Cn, heads=heads, Nq=1, Nk=K, head_dim=hd, q_strides=(D, hd, D),
                      k_strides=(K * 2 * D, hd, 2 * D), v_strides=(K * 2 * D, hd, 2 * D), o_strides=(D, hd, D))
        return o


class _Auto:
    grad = True
    linear = staticmethod(ag.linear)
    norm = staticmethod(ag.norm)
    norm_pass = staticmethod(ag.norm_pass)
    linear_norm = staticmethod(ag.linear_norm)
    attn_packed = staticmethod(lambda qkv, batch, n, heads, hd, causal=False, key_mask=None:
                               ag.PackedAttnFn.apply(qkv, batch, n, heads, hd, causal, key_mask, None))
    rope_attn = staticmethod(ag.rope_attention)
    swiglu = staticmethod(lambda gu, inter: ag.SwigluFn.apply(gu, inter))
    embed_splice = staticmethod(lambda ids, emb, feats, P, fs, tok=None: ag.EmbedSpliceFn.apply(ids, emb, feats, P, fs, tok))
    gather_rows = staticmethod(lambda x, idx: ag.GatherRowsFn.apply(x, idx))
    maskpool = staticmethod(lambda feat, segs, g, S: ag.MaskPoolFn.apply(feat, segs, g, S))
    lora_qkv = staticmethod(lambda x, wqkv, aq, bq, av, bv, s, wqkv_t=None, drop=None, rope=None:
                            ag.LoraQKVFn.apply(x, wqkv, aq, bq, av, bv, s, wqkv_t, drop, rope))
    mlp = staticmethod(ag.mlp)
    norm_lora_qkv = staticmethod(ag.norm_lora_qkv)
    norm_mlp = staticmethod(ag.norm_mlp)
    attn_oproj = staticmethod(ag.attn_oproj)
    ce = staticmethod(lambda logits, labels: ag.CELossFn.apply(logits, labels))
    align_reg = staticmethod(lambda e, t, pred, gt_iou, gt_iop: ag.AlignRegFn.apply(e, t, pred, gt_iou, gt_iop))
    bcast_add = staticmethod(lambda s, add, Cn, K: ag.BcastAddFn.apply(s, add, Cn, K))
    cross_attn_1q = staticmethod(lambda q, kv, Cn, K, heads, hd: ag.CrossAttn1QFn.apply(q, kv, Cn, K, heads, hd))


class BatchPlan:
    """Index plumbing of one batch, computed on the host once (`TrainableMixin.make_plan`): python structure (`sig`: everything that
    shapes the kernel sequence) + small device index tensors (`tensors`: their VALUES may change from batch to batch while `sig`
    stays the same, which is what lets a captured hipGraph be replayed on new data after `copy_tensors_from`)."""

    def __init__(self):
        self.tensors = {}
        self.sig = None

    def __getattr__(self, k):
        t = self.__dict__.get("tensors", {})
        if k in t:
            return t[k]
        raise AttributeError(k)

    def clone(self):
        """A plan with its own device tensors (what a captured hipGraph keeps: the caller's plan is never written to)."""
        p = BatchPlan()
        p.__dict__.update({k: v for k, v in self.__dict__.items() if k != "tensors"})
        p.tensors = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.tensors.items()}
        return p

    def copy_tensors_from(self, other):
        assert self.sig == other.sig, "batch structure changed: capture a new graph"
        for k, v in other.tensors.items():
            if v is not None:
                self.tensors[k].copy_(v, non_blocking=True)


def rank_dropout_seed(seed, rank):
    """Philox key of data-parallel rank r's LoRA-dropout stream; rank 0 keeps `seed`.  The reference never seeds (training.py:369-381
    hands the model to DeepSpeed with torch's default generator, whose seed is the same on every rank), so its ranks draw the same mask
    sequence on different data; here every rank draws its own stream.  Either way the masks are builder-defined (peft absent: unpinned)."""
    return (int(seed) + int(rank) * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF


class TrainableMixin:
    # ------------------------------------------------------------------------------------------------ plumbing
    def set_trainable(self):
        """requires_grad as `training.py:183-241` leaves it, minus the parameters whose gradient is identically zero or
        absent on this path -- freezing them is equivalent under AdamW(wd=0) and keeps DDP's reducer free of unused
        parameters: the q/k projections of attentions over a single key (softmax over one key is constant:
        `cross_attn_token_to_image`, `lisa_final_attn`), and `lisa_dino_conv` when the SAM backbone feeds the head."""
        dead = ("cross_attn_token_to_image.q_proj", "cross_attn_token_to_image.k_proj", "lisa_final_attn.q_proj", "lisa_final_attn.k_proj")
        for name, prm in self.params.named_parameters():
            on = any(k in name for k in ("lm_head", "embed_tokens", "text_hidden_fcs", "lisa_", "lora_"))
            if any(d in name for d in dead) or (self.config.backbone == "sam" and "lisa_dino_conv" in name):
                on = False
            prm.requires_grad_(on)
        return self

    def trainable_parameters(self):
        return [p for p in self.params.parameters() if p.requires_grad]

    def _F(self):
        return _Auto if torch.is_grad_enabled() else _Direct

    def _w(self, name, F):
        """Weight tensor for `name`: the Parameter itself under autograd, its storage otherwise."""
        t = self.params.flat[name]
        if F.grad and isinstance(t, torch.nn.Parameter):
            return t
        return t.data if isinstance(t, torch.nn.Parameter) else t

    def _wcat(self, fused, members, F):
        """Fused [q|k|v]-style operand: the single backing tensor (frozen members, or arena mode where its gradient block is fused
        too), else a cat of the member Parameters so that plain autograd reaches each of them."""
        buf = self.params.flat[fused]
        if F.grad and ag.g32_of(buf) is None and any(self.params.flat[m].requires_grad for m in members):
            return torch.cat([self.params.flat[m] for m in members], 0)
        return buf

    def _wT(self, name, F):
        """Cached [K, N] transposed copy of a FROZEN [N, K] weight (one-time re-layout): lets dX = dY W use the fast TN kernel."""
        if not F.grad:
            return None
        t = self.params.flat[name]
        if isinstance(t, torch.nn.Parameter) and t.requires_grad:
            return None
        cache = self.__dict__.setdefault("_wt_cache", {})
        if name not in cache:
            with torch.no_grad():
                cache[name] = (t.data if isinstance(t, torch.nn.Parameter) else t).t().contiguous()
        return cache[name]

    def _frozen(self, members):
        return not any(self.params.flat[m].requires_grad for m in members)

    def _qkv_wb(self, p, F):
        names_w = [p + f"{n}_proj.weight" for n in "qkv"]
        names_b = [p + f"{n}_proj.bias" for n in "qkv"]
        return self._wcat(p + "qkv.weight", names_w, F), self._wcat(p + "qkv.bias", names_b, F)

    def _kv_wb(self, pc, F, D):
        """k|v projection of a head attention as ONE operand (rows D.. of the fused q|k|v tensor)."""
        views = self.__dict__.get("_arena_views", {})
        if F.grad and (pc + "kv.weight") in views:                                   # arena mode: slice views carrying their arena block
            return views[pc + "kv.weight"], views[pc + "kv.bias"]
        if F.grad and self.params.flat[pc + "k_proj.weight"].requires_grad:
            return (torch.cat([self.params.flat[pc + "k_proj.weight"], self.params.flat[pc + "v_proj.weight"]], 0),
                    torch.cat([self.params.flat[pc + "k_proj.bias"], self.params.flat[pc + "v_proj.bias"]], 0))
        return self.params[pc + "qkv.weight"][D:], self.params[pc + "qkv.bias"][D:]

    # LoRA dropout state (peft lora_dropout, training.py:91): device {seed, offset}; the trainer advances `offset` every micro-step
    def dropout_state(self):
        """Device int64 [2] = {Philox key of THIS rank's LoRA-dropout stream, offset}: what the kernels read (`llmseg_dropout.rng_state`)."""
        st = self.__dict__.get("_rng_state")
        if st is None:
            st = self.__dict__["_rng_state"] = torch.tensor([0x5EED, 0], device=self.device_, dtype=torch.int64)
            self.__dict__.setdefault("_dropout_base", 0x5EED)
        return st

    def dropout_base_seed(self):
        """The seed as the caller set it (rank 0's key); what a checkpoint stores."""
        self.dropout_state()
        return int(self.__dict__["_dropout_base"])

    def set_dropout_seed(self, seed, offset=0):
        """`seed` is the BASE seed; the key in device memory is derived from it and the module's dropout rank (`set_dropout_rank`),
        so calling this (or constructing a second Trainer) any number of times never derives a seed from a derived seed."""
        self.__dict__["_dropout_base"] = int(seed)
        key = rank_dropout_seed(seed, self.__dict__.get("_dropout_rank", 0))
        self.dropout_state().copy_(torch.tensor([key, int(offset)], dtype=torch.int64))

    def set_dropout_rank(self, rank):
        """Data-parallel rank of this replica: re-derives the key from the base seed, keeps the offset."""
        off = int(self.dropout_state()[1])
        self.__dict__["_dropout_rank"] = int(rank)
        self.set_dropout_seed(self.__dict__["_dropout_base"], off)

    def advance_dropout(self, n=1):
        if n:
            self.dropout_state()[1:].add_(int(n))

    # ------------------------------------------------------------------------------------------------ batch plan
    def make_plan(self, input_ids, labels, attention_masks, offset, sam_segs_list=None, inference=False, micro_batches=1, **_):
        """Host-side index plumbing of a batch -> BatchPlan (the only place that synchronises with the device).
        micro_batches = k > 1: the batch is the concatenation of the k micro-batches of ONE gradient-accumulation window
        (`merge_micro_batches`; reference: `gradient_accumulation_steps` micro-steps per optimizer step, training.py:79-82,532-547) run as a single
        pass so that every GEMM sees k x the rows.  The loss of the pass is the SUM of the k micro-batch losses as the reference forms each of them
        -- CE averaged over the micro-batch's OWN labelled tokens, align / IoP losses averaged over its own images -- so the gradient equals the
        one k micro-steps accumulate; the LoRA dropout of micro-batch j draws the mask its own step would draw (`llmseg_dropout.seg_rows`)."""
        c = self.config
        dev = self.device_
        ids = input_ids.detach().cpu()
        N, L = ids.shape
        Pn = c.n_img_tokens
        T = L - 1 + Pn
        off = [int(v) for v in (offset.tolist() if torch.is_tensor(offset) else offset)]
        B = len(off) - 1
        is_img = ids == IMAGE_TOKEN_INDEX
        assert bool((is_img.sum(1) == 1).all()), "exactly one <image> per sequence (the reference's seg_token_mask assumes it too)"
        pos = is_img.int().argmax(1)
        am = attention_masks.detach().cpu().bool()
        key_mask = torch.cat([torch.ones((N, T - L), dtype=torch.bool), am], 1).to(torch.uint8)
        if inference:
            clip_index = torch.zeros((N,), dtype=torch.int64)                               # LISA.py:271-276: one image, expanded
        else:
            clip_index = torch.tensor([i for i in range(B) for _ in range(off[i + 1] - off[i])], dtype=torch.int64)   # LISA.py:293-303
        new_labels = None
        if labels is not None:                                                          # label splice (llava_arch.py:242-251)
            lab = labels.detach().cpu()
            ar = torch.arange(T)[None]
            src = torch.where(ar < pos[:, None], ar, (ar - Pn + 1).clamp(min=0))
            new_labels = torch.gather(lab, 1, src.clamp(max=L - 1))
            new_labels = torch.where((ar >= pos[:, None]) & (ar < pos[:, None] + Pn), torch.full_like(new_labels, IGNORE_INDEX), new_labels)
        # CE rows: only hidden rows whose NEXT token carries a label reach the loss (shifted CE, llava_llama.py:108-118), so lm_head and the
        # CE run on those rows only -- same loss, same gradients (the other rows' dlogits are zero), ~10 x fewer lm_head rows in training
        k = int(micro_batches)
        assert k >= 1 and B % k == 0 and N % k == 0 and not (inference and k > 1), (k, B, N)
        Bm, Nm = B // k, N // k
        if k > 1:
            assert all(off[(j + 1) * Bm] - off[j * Bm] == Nm for j in range(k)), "fused accumulation: every micro-batch must hold the same number of sequences"
        ce_rows = ce_labels = None
        ce_segs = []
        if new_labels is not None:
            valid = new_labels[:, 1:] != IGNORE_INDEX                                    # [N, T-1]: row t predicts label t + 1
            grid = torch.arange(N)[:, None] * T + torch.arange(T - 1)[None, :]
            rows_l, labs_l, pos = [], [], 0
            for j in range(k):                                                           # one CE segment per micro-batch (its own mean)
                sl = slice(j * Nm, (j + 1) * Nm)
                flat = grid[sl][valid[sl]]
                if flat.numel() > 0:
                    rows_l.append(torch.cat([flat, flat[:1]]))                           # + one trailing row: the shifted CE never uses the last position as a predictor
                    labs_l.append(torch.cat([torch.tensor([IGNORE_INDEX]), new_labels[sl, 1:][valid[sl]]]))
                    ce_segs.append((pos, pos + int(flat.numel()) + 1))
                    pos += int(flat.numel()) + 1
                else:
                    assert k == 1, "fused accumulation: a micro-batch without a labelled token has no CE term (its own step would produce nan)"
            if rows_l:
                ce_rows = torch.cat(rows_l)
                ce_labels = torch.cat(labs_l)[None, :]                                   # [1, sum (R_j + 1)]
        tok = ag.embed_token_index(ids, Pn)
        # [SEG] rows: mask shifted by one and by the P-1 extra image tokens (LISA.py:254-266)
        segm = torch.zeros((N, T), dtype=torch.bool)
        segm[:, Pn - 1:Pn - 1 + L - 1] = ids[:, 1:] == self.seg_token_idx
        seg_idx = segm.view(-1).nonzero().flatten()
        cnt = [0] + segm.sum(1).cumsum(0).tolist()
        seg_off = [int(cnt[o]) for o in off]
        rounds = [seg_off[b + 1] - seg_off[b] for b in range(B)]
        plan = BatchPlan()
        segs_shapes = tuple(tuple(s.shape) for s in sam_segs_list) if sam_segs_list is not None else None
        plan.sig = (N, L, T, B, tuple(off), tuple(seg_off), bool(inference), labels is not None, segs_shapes, None if ce_rows is None else int(ce_rows.numel()),
                    k, tuple(ce_segs) if k > 1 else None)
        plan.N, plan.L, plan.T, plan.B, plan.off, plan.seg_off, plan.rounds = N, L, T, B, off, seg_off, rounds
        plan.micro, plan.ce_segs = k, ce_segs
        plan.clip_identity = N == (1 if inference else B) and clip_index.tolist() == list(range(N))     # sequence i reads image i: no expansion copy
        plan.drop_seg_rows = Nm * T if k > 1 else 0                                       # rows of one micro-batch in the Llama activations
        plan.loss_div = float(B) if k == 1 else 1.0                                      # k > 1: 1 / (images per micro-batch) is folded into the item weights
        # groups of images with the same proposal count (the head runs once per group), and the loss weights 1 / (R + 1e-8) of
        # every (image, round) item in group order (LISA.py:452-455)
        plan.groups, loss_w = {}, {}
        if sam_segs_list is not None:
            for b in range(B):
                if rounds[b] > 0:
                    plan.groups.setdefault(int(sam_segs_list[b].shape[0]), []).append(b)
            plan.groups = {K: plan.groups[K] for K in sorted(plan.groups)}
            for K, members in plan.groups.items():
                loss_w[K] = torch.tensor([1.0 / (rounds[b] + 1e-8) / (Bm if k > 1 else 1.0) for b in members for _ in range(rounds[b])], dtype=torch.float32)
        # pinned staging + non-blocking copies: a pageable host->device copy makes the host wait until the stream has drained, i.e. until the
        # previous micro-step's graph has finished -- with a loader in the loop that idles the GPU for the whole host side of a step
        up = lambda t: None if t is None else t.contiguous().pin_memory().to(dev, non_blocking=True)
        plan.tensors = dict(key_mask=up(key_mask), clip_index=up(clip_index), tok_index=up(tok.reshape(-1)), seg_idx=up(seg_idx),
                            new_labels=up(new_labels), ce_rows=up(ce_rows), ce_labels=up(ce_labels))
        for K, w in loss_w.items():
            plan.tensors[f"loss_w{K}"] = up(w)
        return plan

    # ------------------------------------------------------------------------------------------------ language
    def _llama(self, embeds, key_mask_u8, F, kv_out=None, drop_seg_rows=0):
        """32 x [RMSNorm -> q|k|v GEMM (+LoRA) -> RoPE -> causal attention -> o_proj(+res) -> RMSNorm -> gate|up GEMM ->
        SwiGLU -> down(+res)], final RMSNorm (HF LlamaModel, transformers 4.29; call site llava_llama.py:93-102).
        Activations of every layer are kept for the backward pass (288 GB HBM: no recompute, unlike the reference's
        gradient checkpointing, training.py:165-166)."""
        c = self.config.llama
        N, T, H = embeds.shape
        x = embeds.reshape(N * T, H)
        rope = self._rope(T)
        s = c.lora_alpha / c.lora_r if c.lora_r > 0 else 0.0
        p_drop = c.lora_dropout if (F.grad and self.training) else 0.0
        rng = self.dropout_state() if p_drop > 0 else None
        pre = None           # RMSNorm of x under the NEXT pre-norm's weight, when the GEMM that produced x wrote it as its second output
        fuse = self.fuse_residual_norm
        # RoPE inside the q|k|v GEMM's store (round 6): the rank-8 LoRA route at head_dim 128; kv_out (generation prefill) reads the rotated buffer either way
        rope_in_gemm = ag.FUSE_ROPE_FWD and c.lora_r == 8 and c.head_dim == 128
        for i in range(c.layers):
            p = f"model.layers.{i}."
            drop_i = (((rng, i, p_drop, drop_seg_rows) if drop_seg_rows else (rng, i, p_drop)) if p_drop > 0 else None)
            # round 6: pre-norm + projection as one node where the whole backward tail (dX product, LoRA dX, norm backward, residual gradient) is one GEMM call
            norm_frozen = F.grad and not self.params.flat[p + "input_layernorm.weight"].requires_grad and not self.params.flat[p + "post_attention_layernorm.weight"].requires_grad
            merged = F.grad and ag.FUSE_NORM_BWD and norm_frozen and c.hidden % 8 == 0
            if merged and c.lora_r == 8 and self._wT(p + "qkv", F) is not None:
                lp = p + "self_attn."
                qkv, x = F.norm_lora_qkv(x, self._w(p + "input_layernorm.weight", F), c.eps, pre, self._w(p + "qkv", F), self._w(lp + "q_proj.lora_A.default.weight", F),
                                         self._w(lp + "q_proj.lora_B.default.weight", F), self._w(lp + "v_proj.lora_A.default.weight", F),
                                         self._w(lp + "v_proj.lora_B.default.weight", F), s, self._wT(p + "qkv", F), drop_i,
                                         (rope[0], rope[1], T) if rope_in_gemm else None)
                h = None
            else:
                h, x = F.norm_pass(x, self._w(p + "input_layernorm.weight", F), None, c.eps, True, pre)      # x: the residual branch of the same node
            if h is None:
                pass
            elif c.lora_r > 0:
                lp = p + "self_attn."
                qkv = F.lora_qkv(h, self._w(p + "qkv", F), self._w(lp + "q_proj.lora_A.default.weight", F),
                                 self._w(lp + "q_proj.lora_B.default.weight", F), self._w(lp + "v_proj.lora_A.default.weight", F),
                                 self._w(lp + "v_proj.lora_B.default.weight", F), s, self._wT(p + "qkv", F), drop_i,
                                 (rope[0], rope[1], T) if rope_in_gemm else None)
            else:
                mem = [p + f"self_attn.{n}_proj.weight" for n in "qkv"]
                qkv = F.linear(h, self._wcat(p + "qkv", mem, F), None, ops.ACT_NONE, None, self._wT(p + "qkv", F) if self._frozen(mem) else None)
            if F.grad and ag.FUSE_DELTA and c.head_dim == 128 and self._wT(p + "self_attn.o_proj.weight", F) is not None:
                # attention + o_proj as one node: dX(o_proj)'s reduce launch also writes the attention backward's delta
                x, pre = F.attn_oproj(qkv, rope, N, T, c.heads, c.head_dim, True, key_mask_u8, rope_in_gemm, self._w(p + "self_attn.o_proj.weight", F),
                                      self._wT(p + "self_attn.o_proj.weight", F), x, self._w(p + "post_attention_layernorm.weight", F) if fuse else None, c.eps)
                a = None
            else:
                a = F.rope_attn(qkv, rope, N, T, c.heads, c.head_dim, True, key_mask_u8, rope_in_gemm)
            if kv_out is not None:                     # generation prefill (no-grad path): qkv now holds the rotated K and V
                kv_out(i, qkv)
            if a is not None:
                pre = None
            if a is None:
                pass
            elif fuse:         # o_proj + residual and the post-attention norm of the sum: one GEMM call (its K-slice reduce launch writes both)
                x, pre = F.linear_norm(a, self._w(p + "self_attn.o_proj.weight", F), x, self._wT(p + "self_attn.o_proj.weight", F),
                                       self._w(p + "post_attention_layernorm.weight", F), c.eps)
            else:
                x = F.linear(a, self._w(p + "self_attn.o_proj.weight", F), None, ops.ACT_NONE, x, self._wT(p + "self_attn.o_proj.weight", F))
            mem = [p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight"]
            if merged and ag.FUSE_MLP and self._frozen(mem + [p + "mlp.down_proj.weight"]) and c.inter % 8 == 0:
                nxt = self._w(f"model.layers.{i + 1}.input_layernorm.weight", F) if (fuse and i + 1 < c.layers) else None
                x, pre = F.norm_mlp(x, self._w(p + "post_attention_layernorm.weight", F), c.eps, pre, self._wcat(p + "gate_up", mem, F), self._wT(p + "gate_up", F),
                                    self._w(p + "mlp.down_proj.weight", F), self._wT(p + "mlp.down_proj.weight", F), nxt)
                continue
            h, x = F.norm_pass(x, self._w(p + "post_attention_layernorm.weight", F), None, c.eps, True, pre)
            if ag.FUSE_MLP and self._frozen(mem + [p + "mlp.down_proj.weight"]) and c.inter % 8 == 0:
                # frozen MLP (the reference's LoRA targets are q_proj / v_proj only): one node, swiglu and its backward inside the GEMMs' stores
                nxt = self._w(f"model.layers.{i + 1}.input_layernorm.weight", F) if (fuse and i + 1 < c.layers) else None
                x, pre = F.mlp(h, self._wcat(p + "gate_up", mem, F), self._wT(p + "gate_up", F), self._w(p + "mlp.down_proj.weight", F),
                               self._wT(p + "mlp.down_proj.weight", F), x, nxt, c.eps)
                continue
            gu = F.linear(h, self._wcat(p + "gate_up", mem, F), None, ops.ACT_NONE, None, self._wT(p + "gate_up", F) if self._frozen(mem) else None)
            pre = None
            if fuse and i + 1 < c.layers:      # down_proj + residual and the NEXT layer's input norm
                x, pre = F.linear_norm(F.swiglu(gu, c.inter), self._w(p + "mlp.down_proj.weight", F), x, self._wT(p + "mlp.down_proj.weight", F),
                                       self._w(f"model.layers.{i + 1}.input_layernorm.weight", F), c.eps)
            else:
                x = F.linear(F.swiglu(gu, c.inter), self._w(p + "mlp.down_proj.weight", F), None, ops.ACT_NONE, x, self._wT(p + "mlp.down_proj.weight", F))
        return F.norm(x, self._w("model.norm.weight", F), None, c.eps, True).view(N, T, H)

    def llava_forward(self, images_clip, input_ids, plan, want_logits=True, clip_proj=None):
        """LlavaLlamaForCausalLM.forward (llava_llama.py:55-135): splice, decoder stack, lm_head, shifted CE.
        `images_clip`: one CLIP input per IMAGE of the batch (the reference expands it to one per conversation before the tower, LISA.py:293-303:
        the same rows C times; here the frozen tower runs once per image and its projected tokens are expanded through `plan.clip_index`).
        `clip_proj`: the projected tokens [B*(P+1), H] when they were computed elsewhere (`encode_towers`).
        -> (ce_loss | None, logits | None, final-norm hidden [N,T,H])."""
        c = self.config
        F = self._F()
        N, T = plan.N, plan.T
        Pn = c.n_img_tokens
        H = c.llama.hidden
        with torch.no_grad():                                              # CLIP + mm_projector are frozen
            proj = self.encode_images(images_clip) if clip_proj is None else clip_proj      # [B*(P+1), H]
            if not plan.__dict__.get("clip_identity", False):              # one block of P+1 rows per SEQUENCE
                proj = proj.view(-1, (Pn + 1) * H).index_select(0, plan.clip_index).view(N * (Pn + 1), H)
        embeds = F.embed_splice(input_ids.contiguous(), self._w("model.embed_tokens.weight", F), proj[1:], Pn, (Pn + 1) * H, plan.tok_index)
        hidden = self._llama(embeds, plan.key_mask, F, drop_seg_rows=plan.__dict__.get("drop_seg_rows", 0))
        if F.grad and self.__dict__.get("_split_backward"):
            # `Trainer(overlap_exchange=True)`: cut the autograd graph at the Llama output.  backward(loss) then stops at `leaf` with everything
            # downstream of the decoder stack (lm_head, text_hidden_fcs, the mask-selection head: the tail of the gradient arena) final, and
            # `root.backward(leaf.grad)` runs the decoder stack's backward -- the trainer issues the tail's all-reduce between the two.
            root, hidden = hidden, hidden.detach().requires_grad_(True)
            self.__dict__["_split_pair"] = (root, hidden)
        logits, loss = None, None
        gathered = plan.new_labels is not None and not want_logits and plan.ce_rows is not None and self.ce_gather_first
        if want_logits or (plan.new_labels is not None and not gathered):
            logits = F.linear(hidden.view(N * T, H), self._w("lm_head.weight", F)).view(N, T, -1)
        if gathered:                                                       # lm_head + CE on the label-carrying rows only (see make_plan)
            rows = F.gather_rows(hidden.view(N * T, H), plan.ce_rows)
            lg = F.linear(rows, self._w("lm_head.weight", F)).view(1, rows.shape[0], -1)
            if plan.__dict__.get("micro", 1) > 1:                          # fused accumulation window: each micro-batch's own mean, summed
                loss = None
                for a, b in plan.ce_segs:
                    part = F.ce(lg[:, a:b], plan.ce_labels[:, a:b])
                    loss = part if loss is None else loss + part
            else:
                loss = F.ce(lg, plan.ce_labels)
        elif plan.new_labels is not None:
            assert plan.__dict__.get("micro", 1) == 1, "fused accumulation needs ce_gather_first (per-micro-batch CE segments)"
            loss = F.ce(logits, plan.new_labels)
        return loss, logits, hidden

    # ------------------------------------------------------------------------------------------------------ head
    def _head_attn_1key(self, p, text, F):
        """Attention whose key/value is a single token: softmax over one key == 1, so out = out_proj(v_proj(text)) for every
        query (transformer.py:319-341 with Nk = 1); q_proj/k_proj get exactly-zero gradients, as in the reference."""
        v = F.linear(text, self._w(p + "v_proj.weight", F), self._w(p + "v_proj.bias", F))
        return F.linear(v, self._w(p + "out_proj.weight", F), self._w(p + "out_proj.bias", F))

    def _ln(self, x, name, F):
        return F.norm(x, self._w(name + ".weight", F), self._w(name + ".bias", F), 1e-5, False)

    def _mask_head(self, pooled, text, F, stacked=False):
        """LISA.py:363-391: pooled [K, D] bf16 of one image, text [C, D] bf16 -> (pred_iou [C*K] bf16, emb [C*K, D] bf16).
        stacked: `pooled` is already the [C*K, D] row matrix (row = c*K + k) of C conversations from several images."""
        Cn = text.shape[0]
        D = pooled.shape[1]
        K = pooled.shape[0] // Cn if stacked else pooled.shape[0]
        nh, hd = 8, D // 8
        if stacked:
            s = pooled
        else:
            s = pooled.repeat(Cn, 1) if Cn > 1 else (pooled.clone() if not F.grad else pooled)  # row = c*K + k (LISA.py:372)
        t = text.contiguous()
        trace = self.__dict__.get("_relu_trace")            # tests only: the ReLU outputs by feeding Linear (gate patterns, see tests/backward_checks.py)

        def lin(x, p, act=ops.ACT_NONE, res=None):
            y = F.linear(x, self._w(p + ".weight", F), self._w(p + ".bias", F), act, res)
            if trace is not None and act == ops.ACT_RELU:
                trace.setdefault(p, []).append(y.detach())
            return y
        for i in range(2):
            p = f"model.lisa_attention_layers.{i}."
            w, b = self._qkv_wb(p + "self_attn.", F)
            a = F.attn_packed(F.linear(s, w, b), Cn, K, nh, hd)
            s = self._ln(lin(a, p + "self_attn.out_proj", res=s), p + "norm1", F)
            s = self._ln(F.bcast_add(s, self._head_attn_1key(p + "cross_attn_token_to_image.", t, F), Cn, K), p + "norm2", F)
            s = self._ln(lin(lin(s, p + "mlp.lin1", ops.ACT_RELU), p + "mlp.lin2", res=s), p + "norm3", F)
            # image -> token: q = text (1 query per conversation), k = v = mask features
            pc = p + "cross_attn_image_to_token."
            q = lin(t, pc + "q_proj")
            kvw, kvb = self._kv_wb(pc, F, D)
            kv = F.linear(s, kvw, kvb)                                                      # [C*K, 2D] = k | v
            o = F.cross_attn_1q(q, kv, Cn, K, nh, hd)
            t = self._ln(lin(o, pc + "out_proj", res=t), p + "norm4", F)
        s = self._ln(F.bcast_add(s, self._head_attn_1key("model.lisa_final_attn.", t, F), Cn, K), "model.lisa_norm_final_attn", F)
        iou = lin(lin(s, "model.lisa_iou_head.0", ops.ACT_RELU), "model.lisa_iou_head.2", ops.ACT_SIGMOID)
        emb = lin(lin(s, "model.lisa_embedding_head.0", ops.ACT_RELU), "model.lisa_embedding_head.2")
        return iou.view(Cn * K), emb

    def _mask_head_f32(self, pooled, text):
        """`_mask_head` with fp32 ACTIVATIONS (inference only; LISA.py:363-391, transformer.py:215-341): pooled fp32 [K, D] of one image, text fp32 [C, D]
        -> (pred_iou fp32 [C*K], emb fp32 [C*K, D]).  Same weights (the model's bf16 tensors), same wiring; every intermediate stays fp32."""
        P = self.params
        Cn, (K, D) = text.shape[0], pooled.shape
        nh, hd = 8, D // 8
        s = pooled.repeat(Cn, 1) if Cn > 1 else pooled                                      # row = c*K + k (LISA.py:372)
        t = text.contiguous()
        lin = lambda x, p, act=ops.ACT_NONE, res=None: ops.linear_f32(x, P[p + ".weight"], P[p + ".bias"], act, res)
        ln = lambda x, p: ops.layernorm_f32(x, P[p + ".weight"], P[p + ".bias"], 1e-5)
        one_key = lambda p, tt: lin(lin(tt, p + "v_proj"), p + "out_proj")                  # softmax over a single key == 1 (see `_head_attn_1key`)
        bcast = lambda ss, add: (ss.view(Cn, K, D) + add[:, None, :]).view(Cn * K, D)       # (glue: a broadcast add of [C, D] rows)
        for i in range(2):
            p = f"model.lisa_attention_layers.{i}."
            qkv = ops.linear_f32(s, P[p + "self_attn.qkv.weight"], P[p + "self_attn.qkv.bias"])            # [C*K, 3D] = q | k | v
            a = torch.empty((Cn * K, D), device=s.device, dtype=torch.float32)
            st = (K * 3 * D, hd, 3 * D)
            ops.attention_f32(qkv, qkv[:, D:], qkv[:, 2 * D:], a, Cn, nh, K, K, hd, st, st, st, (K * D, hd, D))
            s = ln(lin(a, p + "self_attn.out_proj", res=s), p + "norm1")
            s = ln(bcast(s, one_key(p + "cross_attn_token_to_image.", t)), p + "norm2")
            s = ln(lin(lin(s, p + "mlp.lin1", ops.ACT_RELU), p + "mlp.lin2", res=s), p + "norm3")
            pc = p + "cross_attn_image_to_token."
            q = lin(t, pc + "q_proj")
            kv = ops.linear_f32(s, P[pc + "qkv.weight"][D:], P[pc + "qkv.bias"][D:])                       # [C*K, 2D] = k | v
            o = torch.empty((Cn, D), device=s.device, dtype=torch.float32)
            ops.attention_f32(q, kv, kv[:, D:], o, Cn, nh, 1, K, hd, (D, hd, D), (K * 2 * D, hd, 2 * D), (K * 2 * D, hd, 2 * D), (D, hd, D))
            t = ln(lin(o, pc + "out_proj", res=t), p + "norm4")
        s = ln(bcast(s, one_key("model.lisa_final_attn.", t)), "model.lisa_norm_final_attn")
        iou = lin(lin(s, "model.lisa_iou_head.0", ops.ACT_RELU), "model.lisa_iou_head.2", ops.ACT_SIGMOID)
        emb = lin(lin(s, "model.lisa_embedding_head.0", ops.ACT_RELU), "model.lisa_embedding_head.2")
        return iou.view(Cn * K), emb

    def _inference_scores_f32(self, feat, rows_per_img, row0, g, hidden, plan, sam_segs_list, masks_list, return_aux, logits):
        """The inference tail of `model_forward` (LISA.py:318-408) with fp32 activations from the two bf16 trunk outputs on: [SEG] rows of the
        final-norm hidden states -> text_hidden_fcs; proposals x upsampled features -> mask pooling as (segs . U) . feat / (sum segs + 1e-8) with the
        pulled-back masks kept in fp32; the head; cosine scores.  ~1 GFLOP per image: plain fp32 kernels (csrc/head_f32.hip)."""
        P, c = self.params, self.config
        N, T, H = hidden.shape
        B = plan.B
        if plan.seg_idx.numel():
            hs = ops.gather_rows(hidden.view(N * T, H), plan.seg_idx).float()
            hs = ops.linear_f32(hs, P["model.text_hidden_fcs.0.0.weight"], P["model.text_hidden_fcs.0.0.bias"], ops.ACT_RELU)
            pred = ops.linear_f32(hs, P["model.text_hidden_fcs.0.2.weight"], P["model.text_hidden_fcs.0.2.bias"])
        else:
            pred = torch.empty((0, c.out_dim), device=hidden.device, dtype=torch.float32)
        pred_embeddings = [pred[plan.seg_off[b]:plan.seg_off[b + 1]] for b in range(B)]
        sims, ious = [], []
        for b in range(B):
            if plan.rounds[b] == 0:
                sims.append(None); ious.append(None)
                continue
            segs = sam_segs_list[b].to(BF16).contiguous()
            K, S = segs.shape[0], segs.shape[1]
            fb = feat[b * rows_per_img + row0: b * rows_per_img + row0 + g * g]
            pb, ws = ops.mask_pullback_f32(segs, g, S)                                      # pb fp32 [K, g*g] = segs . U, ws fp32 [K] = sum of segs
            pooled = ops.linear_f32(pb, fb, w_kn=True) / (ws[:, None] + 1e-8)
            iou, emb = self._mask_head_f32(pooled, pred_embeddings[b])
            Cn = plan.rounds[b]
            e0 = emb.view(Cn, K, -1)[0].contiguous()
            sims.append(torch.stack([ops.cosine_f32(pred_embeddings[b][ci].contiguous(), e0) for ci in range(Cn)]))
            ious.append(iou.view(Cn, K)[:1])
        out = {"pred_similarity": sims, "gt_masks": masks_list, "pred_iou": ious}
        if return_aux:
            out.update(logits=logits, hidden=hidden, feats=feat, pred_embeddings=pred_embeddings)
        return out

    # ------------------------------------------------------------------------------------------------ model_forward
    fp32_head = True               # inference: mask pooling, text_hidden_fcs, the head and the cosine scores with fp32 activations (class default; False = the bf16 MFMA route the training pass uses)
    fuse_residual_norm = True      # o_proj / down_proj + residual + the following RMSNorm as one GEMM call (class default; False = the separate norm launch)
    ce_gather_first = True         # lm_head + CE on the label-carrying rows only (class default; False = all N*T rows, as the reference computes them)
    overlap_towers = True          # issue the frozen segmentation backbone on its own HIP stream (class default; set False to serialise)

    def _tower_stream(self):
        st = self.__dict__.get("_side_stream")
        if st is None:
            st = self.__dict__["_side_stream"] = torch.cuda.Stream(device=self.device_)
        return st

    def visual_features_cl(self, images, F, tower_visual=None, n_images=None):
        """-> (channels-last feature rows bf16, rows per image, row offset of the first patch, grid).
        tower_visual: the frozen backbone's rows for these images when they were computed elsewhere (`encode_towers`)."""
        c = self.config
        with torch.no_grad():                                              # both backbones are frozen
            if c.backbone == "sam":
                return (self._sam_encoder_cl(images) if tower_visual is None else tower_visual), c.sam.grid ** 2, 0, c.sam.grid
            if tower_visual is None:
                x, n_tok, (gh, gw) = self._dinov2_tokens(images)
            else:
                x = tower_visual
                n_tok = x.shape[0] // n_images
                gh = gw = int(round((n_tok - 1) ** 0.5))
                assert gh * gw + 1 == n_tok, "tower_visual of the DINOv2 backbone: square token grids only"
        assert gh == gw
        d = self.prepare()
        views = self.__dict__.get("_arena_views", {})
        if F.grad and "model.lisa_dino_conv.weight2d" in views:
            w = views["model.lisa_dino_conv.weight2d"]
        else:
            w = self._w("model.lisa_dino_conv.weight", F)
            w = w.reshape(c.out_dim, c.dino.dim) if F.grad else d["dino.conv_w"]
        y = F.linear(x, w, self._w("model.lisa_dino_conv.bias", F))                          # 1x1 conv (LISA.py:245)
        return y, n_tok, 1, gh

    def forward(self, **kwargs):
        return self.model_forward(**kwargs)

    def model_forward(self, images, images_clip, input_ids, labels, attention_masks, offset, masks_list=None, label_list=None,
                      resize_list=None, sam_segs_list=None, sam_ious_list=None, sam_iops_list=None, inference=False,
                      return_aux=False, plan=None, tower_visual=None, tower_clip=None, **kwargs):
        """Same arguments and return keys as the reference (model/LISA.py:225-474).  `plan` (optional): the BatchPlan of this batch
        (`make_plan`); without it the plan is built here, which synchronises with the device once.
        `tower_visual` / `tower_clip` (optional, not in the reference): the outputs of the two FROZEN towers for this batch's images, computed
        elsewhere by `encode_towers` (e.g. once per accumulation window over all of its images, `Trainer.window_step`); `images` /
        `images_clip` are then not read and may be None."""
        if plan is None:
            plan = self.make_plan(input_ids, None if inference else labels, attention_masks, offset, sam_segs_list, inference)
        towers = None if tower_visual is None else (tower_visual, tower_clip)
        if inference:
            with torch.no_grad():
                return self._model_forward(images, images_clip, input_ids, masks_list, sam_segs_list, sam_ious_list, sam_iops_list, True,
                                           return_aux, plan, towers)
        return self._model_forward(images, images_clip, input_ids, masks_list, sam_segs_list, sam_ious_list, sam_iops_list, False, return_aux, plan, towers)

    @torch.no_grad()
    def encode_towers(self, images, images_clip):
        """The two frozen towers of the path on a batch of images: -> (tower_visual, tower_clip) for `model_forward(..., tower_visual=, tower_clip=)`.
        tower_visual: the segmentation backbone's token rows [B * rows_per_img, C] -- SAM ViT-H's neck output (LISA.py:173-184) or DINOv2's
        x_norm tokens BEFORE the trainable 1x1 conv (LISA.py:186-199,242-245); tower_clip: CLIP-L -> mm_projector rows [B * (P+1), H]
        (clip_encoder.py:41-60, llava_arch.py:93-96).  Neither has a gradient, so they are INPUTS of the trainable part: B may be the images of a
        whole accumulation window (every GEMM of the towers then runs at accum x the rows), and slices of the results feed the micro-steps."""
        images, images_clip = images.to(BF16).contiguous(), images_clip.to(BF16).contiguous()
        c = self.config
        if c.backbone == "sam":
            vis = self._sam_encoder_cl(images)
        else:
            vis = self._dinov2_tokens(images)[0]
        return vis, self.encode_images(images_clip)

    def tower_rows_per_image(self, img_hw=None):
        """(rows of tower_visual, rows of tower_clip) per image."""
        c = self.config
        if c.backbone == "sam":
            return c.sam.grid ** 2, c.n_img_tokens + 1
        hh, ww = img_hw
        return (hh // c.dino.patch) * (ww // c.dino.patch) + 1, c.n_img_tokens + 1

    def _model_forward(self, images, images_clip, input_ids, masks_list, sam_segs_list, sam_ious_list, sam_iops_list, inference, return_aux, plan, towers=None):
        c = self.config
        F = self._F()
        B = plan.B
        if towers is None:
            images, images_clip = images.to(BF16).contiguous(), images_clip.to(BF16).contiguous()
            assert B == images.shape[0]
        if inference and towers is None:
            assert images_clip.shape[0] == 1                                             # LISA.py:271
        # The frozen segmentation backbone (SAM ViT-H / DINOv2) and the CLIP -> Llama chain do not depend on each other until the mask
        # pooling: they are issued on two HIP streams, so the short-matrix Llama GEMMs (M = N_seq x 319 rows leaves CUs idle) and the
        # backbone's kernels share the chip.  Inside a captured micro-step the two streams become parallel branches of the hipGraph.
        side = self._tower_stream() if (towers is None and self.overlap_towers and (c.backbone == "sam" or not F.grad)) else None   # (DINOv2's 1x1 conv trains: keep autograd on one stream)
        if side is not None:
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                feat, rows_per_img, row0, g = self.visual_features_cl(images, F)
            images.record_stream(side)
        else:
            feat, rows_per_img, row0, g = self.visual_features_cl(images, F, None if towers is None else towers[0], B)
        ce, logits, hidden = self.llava_forward(images_clip, input_ids, plan, want_logits=return_aux or (not inference and plan.new_labels is None),
                                                clip_proj=None if towers is None else towers[1])
        if side is not None:
            cur.wait_stream(side)
            feat.record_stream(cur)

        if inference and self.fp32_head:
            return self._inference_scores_f32(feat, rows_per_img, row0, g, hidden, plan, sam_segs_list, masks_list, return_aux, logits)

        # [SEG] rows: gather first, then the MLP (identical to the reference's MLP-on-everything + boolean gather, LISA.py:318-323)
        N, T, H = hidden.shape
        seg_off = plan.seg_off
        if plan.seg_idx.numel():
            hs = F.gather_rows(hidden.view(N * T, H), plan.seg_idx)
            hs = F.linear(hs, self._w("model.text_hidden_fcs.0.0.weight", F), self._w("model.text_hidden_fcs.0.0.bias", F), ops.ACT_RELU)
            if self.__dict__.get("_relu_trace") is not None:
                self.__dict__["_relu_trace"].setdefault("model.text_hidden_fcs.0.0", []).append(hs.detach())
            pred = F.linear(hs, self._w("model.text_hidden_fcs.0.2.weight", F), self._w("model.text_hidden_fcs.0.2.bias", F))
        else:
            pred = torch.empty((0, c.out_dim), device=hidden.device, dtype=BF16)
        pred_embeddings = [pred[seg_off[b]:seg_off[b + 1]] for b in range(B)]

        # mask pooling per image, then the mask-selection head ONCE per group of images with the same proposal count: the head's
        # weights are shared, so its rows (image, conversation, proposal) are stacked into one matrix (the reference loops over
        # images, LISA.py:355-391; per-image launches of M = K-row GEMMs leave 250 of the 256 CUs idle)
        ious, embs = [None] * B, [None] * B
        pooled_of, pool_groups = {}, {}
        segs_of = {}
        for b in range(B):
            if plan.rounds[b] == 0:
                if not inference:
                    raise ValueError("number of rounds = 0")                          # LISA.py:435-437
                continue
            segs_of[b] = sam_segs_list[b].to(BF16).contiguous()
            pool_groups.setdefault(tuple(segs_of[b].shape), []).append(b)
        for (K, S, _), members in pool_groups.items():
            if len(members) > 1 and not (F.grad and feat.requires_grad) and members == list(range(members[0], members[0] + len(members))):
                # frozen feature map (SAM backbone): all images of the group pooled by one strided-batched GEMM
                pooled = ops.maskpool_batched(feat[members[0] * rows_per_img:], rows_per_img, row0, [segs_of[b] for b in members], g, S)
                for i, b in enumerate(members):
                    pooled_of[b] = pooled[i]
            else:
                for b in members:
                    fb = feat[b * rows_per_img + row0: b * rows_per_img + row0 + g * g]
                    pooled_of[b] = F.maskpool(fb, segs_of[b], g, S)
        group_out = {}
        for K, members in plan.groups.items():
            if len(members) == 1:
                b = members[0]
                iou, emb = self._mask_head(pooled_of[b], pred_embeddings[b], F)
            else:
                s0 = torch.cat([pooled_of[b].repeat(plan.rounds[b], 1) if plan.rounds[b] > 1 else pooled_of[b] for b in members], 0)
                iou, emb = self._mask_head(s0, torch.cat([pred_embeddings[b] for b in members], 0), F, stacked=True)
            group_out[K] = (members, iou, emb)
            r0 = 0
            for b in members:
                Cn = plan.rounds[b]
                ious[b] = iou[r0 * K:(r0 + Cn) * K].view(Cn, K)
                embs[b] = emb[r0 * K:(r0 + Cn) * K].view(Cn, K, -1)
                r0 += Cn

        if inference:
            # every conversation's [SEG] embedding against the FIRST conversation's mask features (LISA.py:394-403: [C, K]; C = 1 in the reference's loops)
            sims = [torch.stack([ops.cosine_scores(pred_embeddings[b][c], embs[b][0]) for c in range(plan.rounds[b])]) for b in range(B)]
            out = {"pred_similarity": sims, "gt_masks": masks_list, "pred_iou": [ious[b][:1].float() for b in range(B)]}
            if return_aux:
                out.update(logits=logits, hidden=hidden, feats=feat, pred_embeddings=pred_embeddings)
            return out

        # losses (LISA.py:416-466): every (image, round) item of a group in ONE launch; an image's rounds are averaged (1/(R+1e-8)),
        # then the images
        align = torch.zeros((), device=hidden.device, dtype=torch.float32)
        reg = torch.zeros((), device=hidden.device, dtype=torch.float32)
        for K, (members, iou_s, emb_s) in group_out.items():
            t_s = torch.cat([pred_embeddings[b] for b in members], 0).contiguous() if len(members) > 1 else pred_embeddings[members[0]].contiguous()
            Rg = t_s.shape[0]
            gi = torch.cat([sam_ious_list[b].float().reshape(-1, K) for b in members], 0).contiguous()
            gp = torch.cat([sam_iops_list[b].float().reshape(-1, K) for b in members], 0).contiguous()
            o = F.align_reg(emb_s.reshape(Rg, K, -1), t_s, iou_s.reshape(Rg, K), gi, gp)                    # [Rg, 2]
            w = plan.tensors[f"loss_w{K}"]
            align = align + (o[:, 0] * w).sum()
            reg = reg + (o[:, 1] * w).sum()
        div = plan.__dict__.get("loss_div", float(B))
        align, reg = align / div, reg / div
        ce = ce * c.ce_loss_weight
        align = align * c.align_loss_weight
        reg = reg * c.regression_loss_weight
        out = {"loss": ce + align + reg, "ce_loss": ce, "align_loss": align, "regression_loss": reg}
        if return_aux:
            out.update(logits=logits, hidden=hidden, feats=feat)
        return out
