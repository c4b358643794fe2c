"""torch.autograd.Function wrappers: forward AND backward of every op on the trainable part of the path run in
libllmseg_hip.so (the autograd engine is only the graph plumbing).  Frozen towers (CLIP, SAM, DINOv2) never come here.

Backward GEMMs use the transposed-operand modes of `llmseg_gemm_bf16` (dX = dY W, dW = dY^T X); attention backward is the fused
recompute kernel (`llmseg_attn_bwd`; the single-query cross attentions of the head keep a small materialised chain).

Parameter gradients have two destinations:
  * arena mode (training, `llmseg_amd.train.GradArena`): a parameter tensor carries `_g32`, an fp32 view into the trainer's flat
    gradient arena; the backward kernels ACCUMULATE into it (`C += ...` GEMM epilogue, fixed-order partial sums) and the Function returns None
    for that input, so gradients are summed over micro-steps in fp32 (the reference's DeepSpeed engine accumulates
    `gradient_accumulation_steps` micro-batches, training.py:79-82,292-332) and no bf16 `.grad` round trip exists;
  * plain autograd: without `_g32` the Function returns a bf16 gradient and autograd fills `.grad` as usual (parity tests).
"""
import math
import os

import torch
from torch.autograd import Function

from . import ops

BF16 = torch.bfloat16


BIG_LINEAR = 1 << 34          # M*N*K above which LinearFn.backward transposes/pads its operands for the LDS-DMA GEMM kernels
BIG_WEIGHT = 1 << 26          # N * K of a Linear whose dX / dW always take the transposed-operand route below (lm_head: 131 M)
# A/B switches of the round-6 Llama-layer fusions (tests compare each fused route with the unfused one bit for bit)
FUSE_ROPE_BWD = os.environ.get("LLMSEG_NO_FUSE_ROPE_BWD") is None      # inverse RoPE inside the attention backward's dq / dk store
FUSE_ROPE_FWD = os.environ.get("LLMSEG_NO_FUSE_ROPE_FWD") is None      # RoPE inside the q|k|v GEMM's store (rank-8 LoRA route, head_dim 128)
FUSE_NORM_BWD = os.environ.get("LLMSEG_NO_FUSE_NORM_BWD") is None      # pre-norm backward (+ LoRA dX + residual gradient) inside the dX product's K-slice reduce launch
FUSE_LORA_PARTS = os.environ.get("LLMSEG_NO_FUSE_LORA_PARTS") is None  # the backward's rank-8 down projection is finished inside the dX(q|k|v) product's tail
FUSE_DELTA = os.environ.get("LLMSEG_NO_FUSE_DELTA") is None            # the attention backward's delta inside the dX(o_proj) product's reduce launch
FUSE_MLP = os.environ.get("LLMSEG_NO_FUSE_MLP") is None                # swiglu / swiglu_bwd inside the gate|up and dX(down) GEMMs' stores (frozen MLP weights)


class Leaves:
    """Weight-gradient kernels that accumulate into the trainer's fp32 arena are LEAVES of the backward pass: nothing later in the pass
    reads what they write.  With `Trainer(leaf_stream=True)` (`Leaves.on`) they are issued on ONE side stream that waits for the
    producing kernels, so the dependent chain of the backward pass (dX GEMMs, norm / attention backward) does not queue behind them; inside
    a captured hipGraph the two streams become parallel branches.  MEASURED SLOWER (42.95 -> 43.81 ms at two images per step): a replayed
    graph has no launch gaps to hide and the branch takes CUs from the GEMM chain -- off by default, kept as an A/B switch.  One side stream keeps the leaves in issue order (a parameter used twice
    accumulates in the same order every run: results stay bit-reproducible).  Operands are kept alive until `join()` -- the caching
    allocator would otherwise hand a freed operand's block to the next main-stream allocation while the side kernel has not run."""
    on = False
    _streams = {}
    _keep = []
    _used = False

    @classmethod
    def run(cls, fn, *operands):
        if not cls.on:
            return fn()
        cur = torch.cuda.current_stream()
        key = cur.device.index
        side = cls._streams.get(key)
        if side is None:
            side = cls._streams[key] = torch.cuda.Stream(device=cur.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            out = fn()
        cls._keep.extend(operands)
        cls._used = True
        return out

    @classmethod
    def join(cls):
        """The current stream waits for every leaf issued so far (end of a backward pass; inside a capture this closes the side branch)."""
        if cls._used:
            cur = torch.cuda.current_stream()
            side = cls._streams.get(cur.device.index)
            if side is not None:
                cur.wait_stream(side)
        cls._keep.clear()
        cls._used = False


def g32_of(t):
    """fp32 gradient-arena view attached to a parameter tensor (None outside arena mode)."""
    return getattr(t, "_g32", None)


def _pad8_cols(t):
    """[M, N] -> [M, roundup8(N)] zero-padded copy (tiny tensors only: N < 8 heads such as the 1-wide IoP head)."""
    M, N = t.shape
    out = torch.zeros((M, (N + 7) // 8 * 8), device=t.device, dtype=t.dtype)
    out[:, :N] = t
    return out


class LinearFn(Function):
    """y = act(x @ w^T + b) + residual  (act in {none, relu, sigmoid}; act and residual are not combined on this path)."""

    @staticmethod
    def forward(ctx, x, w, b, act, residual, wt=None):
        """wt: optional pre-transposed copy of a FROZEN w ([K, N]) so that dX runs on the fast K-contiguous DMA kernel."""
        y = ops.gemm(x, w, bias=b, act=act, residual=residual)
        ctx.wt = wt
        ctx.act = act
        ctx.has_res = residual is not None
        ctx.gw, ctx.gb = g32_of(w), (g32_of(b) if b is not None else None)
        ctx.b_needs = b is not None and b.requires_grad
        assert not (act != ops.ACT_NONE and residual is not None)
        assert act in (ops.ACT_NONE, ops.ACT_RELU, ops.ACT_SIGMOID), "only relu/sigmoid epilogues are differentiated"
        ctx.save_for_backward(x, w, y if act != ops.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        dpre = ops.act_bwd(dy, y, ctx.act) if ctx.act != ops.ACT_NONE else dy
        N = w.shape[0]
        M, Kin = x.shape
        dx = dw = db = None
        gw = ctx.gw
        need_w = gw is not None or ctx.needs_input_grad[1]
        big = M * N * Kin >= BIG_LINEAR or (N * Kin >= BIG_WEIGHT and M >= 16)      # the weight itself is big: also with few rows (lm_head on the label rows only)
        if big and Kin % 8 == 0 and (need_w or ctx.wt is None or N % 8 != 0):
            # wide trainable Linear (lm_head, [32004, 4096]): pad the contraction dims to multiples of 64 and transpose the
            # operands once, so that both gradient GEMMs run on the K-contiguous LDS-DMA kernels instead of the
            # transposed-operand register-staged one (3-4x slower at this size)
            Np, Mp = (N + 63) // 64 * 64, (M + 63) // 64 * 64
            if Np != N:
                dpp = torch.zeros((M, Np), device=dpre.device, dtype=BF16)
                dpp[:, :N] = dpre
            else:
                dpp = dpre
            if ctx.needs_input_grad[0]:
                dx = ops.gemm(dpp, ops.transpose_pad(w, Np))                             # [M, Np] @ [Np, Kin]   (W^T: [Kin, Np])
            if need_w:
                if gw is not None:
                    def wgrad():
                        dT, xT = ops.transpose_pad(dpp, Mp), ops.transpose_pad(x, Mp)    # [Np, Mp], [Kin, Mp]
                        ops.gemm(dT[:N], xT, out=gw, accumulate=True)                     # fp32 arena += dY^T X
                    Leaves.run(wgrad, dpp, x)
                else:
                    dT, xT = ops.transpose_pad(dpp, Mp), ops.transpose_pad(x, Mp)
                    dw = ops.gemm(dT, xT)[:N]
        else:
            if N % 8 != 0:                                   # tiny head (N = 1): pad the contraction/leading dims
                dpre_p = _pad8_cols(dpre)
                w_p = torch.zeros((dpre_p.shape[1], w.shape[1]), device=w.device, dtype=w.dtype)
                w_p[:N] = w
            else:
                dpre_p, w_p = dpre, w
            if ctx.needs_input_grad[0]:
                dx = ops.gemm(dpre, ctx.wt) if (ctx.wt is not None and N % 8 == 0) else ops.gemm(dpre_p, w_p, trans_w=True)   # [M,N] @ [N,K]
            if gw is not None:
                Leaves.run(lambda: ops.gemm(dpre_p[:, :N], x, out=gw, trans_a=True, trans_w=True, accumulate=True), dpre_p, x)   # [N,M] @ [M,K] -> fp32 arena
            elif need_w:
                dw = ops.gemm(dpre_p, x, trans_a=True, trans_w=True)[:N]      # [N,M] @ [M,K]
                if dw.shape[0] != N or not dw.is_contiguous():
                    dw = dw.contiguous()
        if ctx.gb is not None:
            Leaves.run(lambda: ops.colsum(dpre, out=ctx.gb), dpre)
        elif ctx.b_needs and ctx.needs_input_grad[2]:
            db = ops.colsum(dpre).to(BF16)
        dres = dy if (ctx.has_res and ctx.needs_input_grad[4]) else None
        return dx, dw, db, None, dres, None


def linear(x, w, b=None, act=ops.ACT_NONE, residual=None, wt=None):
    return LinearFn.apply(x, w, b, act, residual, wt)


class LinearNormFn(Function):
    """(y, h) = (x @ w^T + residual, RMSNorm(y) * norm_w): a residual-stream projection together with the stream's NEXT pre-norm
    (`llmseg_gemm_args.norm_out`: when the product runs as K-slices its reduce launch writes both).  h is a forward-only by-product: it is handed
    to `norm_pass(..., pre=h)`, whose node owns the norm's backward; this node's backward is `LinearFn`'s."""

    @staticmethod
    def forward(ctx, x, w, b, act, residual, wt, norm_w, eps):
        assert b is None and act == ops.ACT_NONE and residual is not None
        h = torch.empty((x.shape[0], w.shape[0]), device=x.device, dtype=BF16)
        y = ops.gemm(x, w, residual=residual, norm_w=norm_w, norm_eps=eps, norm_out=h)
        ctx.wt, ctx.act, ctx.has_res = wt, act, True
        ctx.gw, ctx.gb, ctx.b_needs = g32_of(w), None, False
        ctx.save_for_backward(x, w, None)
        ctx.mark_non_differentiable(h)
        ctx.set_materialize_grads(False)        # no zero-fill for h's absent gradient
        return y, h

    @staticmethod
    def backward(ctx, dy, _dh):
        return LinearFn.backward(ctx, dy) + (None, None)


def linear_norm(x, w, residual, wt, norm_w, eps):
    return LinearNormFn.apply(x, w, None, ops.ACT_NONE, residual, wt, norm_w, eps)


def lora_qkv_fused(x, wqkv, aq, bq, av, bv, s, drop=None, want_bt=False, rope=None):
    """qkv = x Wqkv^T + s (drop_q(x) Aq^T) Bq^T on the q block + s (drop_v(x) Av^T) Bv^T on the v block, rank 8.  The two updates
    ride in the qkv GEMM as one extra 64-wide K-tile: A2 = [x Aq^T | x Av^T | 0], W2 rows of the q block = [s Bq | 0], rows of the
    v block = [0 | s Bv | 0] (no read-modify-write pass over q and v).  Both operands are rebuilt from the current LoRA matrices
    on every call (two small kernels): nothing is cached, so an in-place optimizer update or a load_state_dict cannot leave a
    stale operand behind.  drop = (rng_state, layer, p[, seg_rows]) or None.  rope = (cos, sin, T): the q | k heads (width 128) leave the GEMM
    already rotated (`llmseg_gemm_args.fx`, round 6: `rope_` used to be a launch of its own).  -> (qkv, A2, B^T | None)"""
    H = wqkv.shape[1]
    M = x.shape[0]
    a2 = torch.empty((M, 64), device=x.device, dtype=BF16)
    w2b = torch.empty((3 * H, 64), device=x.device, dtype=BF16)
    bt = torch.empty((16, H), device=x.device, dtype=BF16) if want_bt else None      # Bq^T | Bv^T for the backward's down projection
    ops.lora_down(x, aq, out=a2, zero_cols=48, drop=_drops(drop)[0], x2=x, w2=av,      # [x Aq^T | x Av^T | 0], x read once; the pack rides in its finish launch
                  pack=(aq, bq, av, bv, s, w2b, None, bt))
    return ops.gemm(x, wqkv, a2=a2, w2=w2b, rope=None if rope is None else (rope[0], rope[1], rope[2], 2 * H)), a2, bt


def _drops(drop):
    if drop is None or drop[2] <= 0.0:
        return None, None
    rng, layer, p = drop[:3]
    seg = drop[3:]                                                   # optional rows-per-segment of a fused accumulation window
    return (rng, 2 * layer, p) + tuple(seg), (rng, 2 * layer + 1, p) + tuple(seg)


class LoraQKVFn(Function):
    """qkv = x Wqkv^T with the LoRA deltas of q_proj and v_proj added in place:
    q += s (drop(x) Aq^T) Bq^T, v += s (drop(x) Av^T) Bv^T, s = alpha / r, independent dropout masks on the two LoRA branch inputs
    (peft 0.4.0 Linear: `lora_B(lora_A(lora_dropout(x))) * scaling`; base weight frozen).  PARITY UNPINNED (peft absent)."""

    @staticmethod
    def forward(ctx, x, wqkv, aq, bq, av, bv, s, wqkv_t=None, drop=None, rope=None):
        """rope = (cos, sin, T) (rank-8 route only): the output's q | k heads are ROTATED inside the GEMM.  This node's backward still expects the
        gradient w.r.t. the UNROTATED q|k|v -- which is what `PackedAttnFn(..., pre_rotated=True)` returns (its backward applies the inverse rotation
        in the attention kernel's store): the two nodes are a pair, the rotated tensor must have no other consumer."""
        ctx.wqkv_t = wqkv_t
        H = wqkv.shape[1]
        ctx.fast = aq.shape[0] == 8                                      # rank-8 skinny kernels
        ctx.drop = drop if (drop is not None and drop[2] > 0.0) else None
        ctx.g = tuple(g32_of(t) for t in (aq, bq, av, bv))
        if ctx.fast:
            qkv, a2, ctx.bt = lora_qkv_fused(x, wqkv, aq, bq, av, bv, s, ctx.drop, want_bt=True, rope=rope)
            xaq, xav = a2[:, :8], a2[:, 8:16]
        else:
            assert ctx.drop is None, "LoRA dropout is implemented for rank 8"
            assert rope is None, "the fused rotation rides on the rank-8 route"
            qkv = ops.gemm(x, wqkv)
            xaq, xav = ops.gemm(x, aq), ops.gemm(x, av)                   # [M, r]
            ops.gemm(xaq, bq, residual=qkv[:, :H], out=qkv[:, :H], alpha=s)
            ops.gemm(xav, bv, residual=qkv[:, 2 * H:], out=qkv[:, 2 * H:], alpha=s)
        ctx.s = s
        if ctx.fast:
            ctx.save_for_backward(x, wqkv, aq, bq, av, bv, a2, None)      # the [M, 64] extension operand itself (xaq / xav are its first 16 columns)
        else:
            ctx.save_for_backward(x, wqkv, aq, bq, av, bv, xaq, xav)
        return qkv

    @staticmethod
    def backward(ctx, d):
        x, wqkv, aq, bq, av, bv, xaq, xav = ctx.saved_tensors
        if ctx.fast:
            a2 = xaq
            xaq, xav = a2[:, :8], a2[:, 8:16]
        s, H = ctx.s, wqkv.shape[1]
        d = d.contiguous()
        M = d.shape[0]
        dq, dv = d[:, :H], d[:, 2 * H:]
        gaq, gbq, gav, gbv = ctx.g
        if ctx.fast:
            drq = _drops(ctx.drop)[0]                                        # stream of the q branch; the v branch is stream + 1
            t2 = torch.empty((M, 64), device=d.device, dtype=BF16)       # [s dq Bq | s dv Bv | 0]
            ops.lora_down(dq, ctx.bt[:8], alpha=s, out=t2, zero_cols=48, x2=dv, w2=ctx.bt[8:])      # B^T packed in the forward: K-contiguous rows
            tq, tv = t2[:, :8], t2[:, 8:16]
            if ctx.wqkv_t is not None and ctx.drop is None:               # dx = d Wqkv + tq Aq + tv Av in ONE GEMM
                w2a = torch.empty((H, 64), device=d.device, dtype=BF16)
                ops.lora_pack(aq, bq, av, bv, s, w2a=w2a)
                dx = ops.gemm(d, ctx.wqkv_t, a2=t2, w2=w2a)
            else:                                                         # dropout masks the LoRA branches' dx element-wise
                dx = ops.gemm(d, ctx.wqkv_t) if ctx.wqkv_t is not None else ops.gemm(d, wqkv, trans_w=True)
                ops.lora_apply_(dx, t2, aq, w_rn=True, drop=drq, w2=av)
            def wgrads():
                b_ = ops.lora_outer(dq, xaq, alpha=s, out=gbq, a2=dv, b2=xav, out2=gbv)               # [H, 8] = s d^T (drop(x) A^T)
                a_ = ops.lora_outer(x, tq, out_rn=True, out=gaq, drop=drq, a2=x, b2=tv, out2=gav)      # [8, H] = t^T drop(x)
                return b_, a_
            if all(g is not None for g in ctx.g):                            # arena mode: the four gradients in ONE launch (+ one fold), off the main chain
                Leaves.run(lambda: ops.lora_wgrads(d, H, x, a2, t2, gbq, gbv, gaq, gav, s, drop=drq), d, x, t2, a2)
                dbq = dbv = daq = dav = None
            else:
                (dbq, dbv), (daq, dav) = wgrads()
            outs = [None if (g is not None or t is None) else t.to(BF16) for g, t in ((gaq, daq), (gbq, dbq), (gav, dav), (gbv, dbv))]
            return dx, None, outs[0], outs[1], outs[2], outs[3], None, None, None, None
        tq = ops.gemm(dq, bq, trans_w=True, alpha=s)                      # [M, r] = s dq Bq
        tv = ops.gemm(dv, bv, trans_w=True, alpha=s)
        dx = ops.gemm(d, ctx.wqkv_t) if ctx.wqkv_t is not None else ops.gemm(d, wqkv, trans_w=True)
        ops.gemm(tq, aq, trans_w=True, residual=dx, out=dx)
        ops.gemm(tv, av, trans_w=True, residual=dx, out=dx)
        grads = []
        for g, a_, b_, al in ((gaq, tq, x, 1.0), (gbq, dq, xaq, s), (gav, tv, x, 1.0), (gbv, dv, xav, s)):   # dA = t^T x, dB = s d^T (xA)
            if g is not None:
                ops.gemm(a_, b_, out=g, trans_a=True, trans_w=True, alpha=al, accumulate=True)
                grads.append(None)
            else:
                grads.append(ops.gemm(a_, b_, trans_a=True, trans_w=True, alpha=al))
        return dx, None, grads[0], grads[1], grads[2], grads[3], None, None, None, None


class NormLoraQKVFn(Function):
    """RMSNorm pre-norm + the LoRA'd q|k|v projection of it as ONE node (round 6): (qkv, x) = (lora_qkv(norm(x) * w), x) -- `NormPassFn` followed by
    `LoraQKVFn` (rank 8, frozen base weight, frozen norm weight).  What the merge buys is the backward: the dX product of q|k|v, the LoRA branches' dX
    (`lora_apply_`), the norm backward and the residual branch's gradient are ONE GEMM call (`llmseg_gemm_args.nb_x`: all of it in the K-sliced product's
    reduce launch) instead of a reduce, a lora_apply and a norm_bwd launch with two round trips of the [M, H] gradient.  Same bits as the two nodes."""

    @staticmethod
    def forward(ctx, x, norm_w, eps, pre, wqkv, aq, bq, av, bv, s, wqkv_t, drop, rope):
        x = x.contiguous()
        h = ops.norm(x, norm_w, None, eps=eps, rms=True) if pre is None else pre
        ctx.drop = drop if (drop is not None and drop[2] > 0.0) else None
        ctx.g = tuple(g32_of(t) for t in (aq, bq, av, bv))
        qkv, a2, ctx.bt = lora_qkv_fused(h, wqkv, aq, bq, av, bv, s, ctx.drop, want_bt=True, rope=rope)
        ctx.s, ctx.eps, ctx.wqkv_t = s, eps, wqkv_t
        ctx.save_for_backward(x, norm_w, h, wqkv, aq, bq, av, bv, a2)
        ctx.set_materialize_grads(False)
        return qkv, x.view_as(x)

    @staticmethod
    def backward(ctx, d, dpass):
        x, norm_w, h, wqkv, aq, bq, av, bv, a2 = ctx.saved_tensors
        if d is None:
            return (dpass,) + (None,) * 12
        s, H = ctx.s, wqkv.shape[1]
        d = d.contiguous()
        M = d.shape[0]
        gaq, gbq, gav, gbv = ctx.g
        drq = _drops(ctx.drop)[0]
        t2 = torch.empty((M, 64), device=d.device, dtype=BF16)       # [s dq Bq | s dv Bv | 0]
        # the rank-8 down projection leaves its K-slice partials: the dX product's tail finishes them (and writes t2 for the weight gradients below)
        part, pS, pscale = None, 0, 0.0
        if FUSE_LORA_PARTS:
            _, part, pS, pscale = ops.lora_down(d[:, :H], ctx.bt[:8], alpha=s, out=t2, zero_cols=48, x2=d[:, 2 * H:], w2=ctx.bt[8:], parts=True)
        else:
            ops.lora_down(d[:, :H], ctx.bt[:8], alpha=s, out=t2, zero_cols=48, x2=d[:, 2 * H:], w2=ctx.bt[8:])
        dres = None if dpass is None else dpass.contiguous()
        dx = ops.gemm(d, ctx.wqkv_t, normbwd=(x, norm_w, ctx.eps, True, dres), nb_lora=(t2, aq, av, 1.0, drq, part, pS, pscale, 48))
        if all(g is not None for g in ctx.g):                            # arena mode: the four gradients in ONE launch (+ one fold), off the main chain
            Leaves.run(lambda: ops.lora_wgrads(d, H, h, a2, t2, gbq, gbv, gaq, gav, s, drop=drq), d, h, t2, a2)
            outs = [None] * 4
        else:
            dbq, dbv = ops.lora_outer(d[:, :H], a2[:, :8], alpha=s, out=gbq, a2=d[:, 2 * H:], b2=a2[:, 8:16], out2=gbv)
            daq, dav = ops.lora_outer(h, t2[:, :8], out_rn=True, out=gaq, drop=drq, a2=h, b2=t2[:, 8:16], out2=gav)
            outs = [None if (g is not None or t is None) else t.to(BF16) for g, t in ((gaq, daq), (gbq, dbq), (gav, dav), (gbv, dbv))]
        return dx, None, None, None, None, outs[0], outs[1], outs[2], outs[3], None, None, None, None


def norm_lora_qkv(x, norm_w, eps, pre, wqkv, aq, bq, av, bv, s, wqkv_t, drop, rope):
    return NormLoraQKVFn.apply(x, norm_w, eps, pre, wqkv, aq, bq, av, bv, s, wqkv_t, drop, rope)


class NormMlpFn(Function):
    """A whole pre-norm MLP block of HF LlamaDecoderLayer with frozen weights as ONE node (round 6):
        y = x + down(silu(gate) * up),  gate|up = (RMSNorm(x) * norm_w) Wgu^T      (, pre = RMSNorm(y) * next_norm_w)
    = `NormPassFn` + `MlpFn`.  Backward: d(gate|up) inside the dX(down) GEMM's store, then dX(gate|up) + the norm backward + the residual gradient (= dy) in
    ONE GEMM call (`llmseg_gemm_args.nb_x`).  Same bits as the separate nodes.  llava_llama.py:93-102."""

    @staticmethod
    def forward(ctx, x, norm_w, eps, pre, wgu, wgu_t, wd, wd_t, next_norm_w):
        x = x.contiguous()
        h = ops.norm(x, norm_w, None, eps=eps, rms=True) if pre is None else pre
        M, inter = x.shape[0], wd.shape[1]
        gu = torch.empty((M, 2 * inter), device=x.device, dtype=BF16)
        act = torch.empty((M, inter), device=x.device, dtype=BF16)
        ops.gemm(h, wgu, out=gu, swiglu_out=act)
        ctx.wgu_t, ctx.wd_t, ctx.eps = wgu_t, wd_t, eps
        ctx.save_for_backward(x, norm_w, gu)
        ctx.set_materialize_grads(False)
        if next_norm_w is None:
            return ops.gemm(act, wd, residual=x), None
        nxt = torch.empty_like(x)
        y = ops.gemm(act, wd, residual=x, norm_w=next_norm_w, norm_eps=eps, norm_out=nxt)
        ctx.mark_non_differentiable(nxt)
        return y, nxt

    @staticmethod
    def backward(ctx, dy, _dnxt):
        x, norm_w, gu = ctx.saved_tensors
        dy = dy.contiguous()
        dgu = ops.gemm(dy, ctx.wd_t, swiglu_bwd_of=gu)
        dx = ops.gemm(dgu, ctx.wgu_t, normbwd=(x, norm_w, ctx.eps, True, dy))
        return dx, None, None, None, None, None, None, None, None


def norm_mlp(x, norm_w, eps, pre, wgu, wgu_t, wd, wd_t, next_norm_w):
    return NormMlpFn.apply(x, norm_w, eps, pre, wgu, wgu_t, wd, wd_t, next_norm_w)


class NormFn(Function):
    @staticmethod
    def forward(ctx, x, w, b, eps, rms):
        x = x.contiguous()
        ctx.eps, ctx.rms, ctx.has_b = eps, rms, b is not None
        ctx.gw, ctx.gb = g32_of(w), (g32_of(b) if b is not None else None)
        ctx.save_for_backward(x, w)
        return ops.norm(x, w, b, eps=eps, rms=rms)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        if ctx.gw is not None:                                   # arena mode: the kernel adds into the fp32 arena views
            dx = ops.norm_bwd(dy.contiguous(), x, w, ctx.eps, ctx.rms, ctx.gw, ctx.gb)
            return dx, None, None, None, None
        need_w = ctx.needs_input_grad[1]
        dw = torch.zeros(w.shape, device=w.device, dtype=torch.float32) if need_w else None
        db = torch.zeros(w.shape, device=w.device, dtype=torch.float32) if (need_w and ctx.has_b) else None
        dx = ops.norm_bwd(dy.contiguous(), x, w, ctx.eps, ctx.rms, dw, db)
        return dx, (dw.to(BF16) if need_w else None), (db.to(BF16) if db is not None else None), None, None


def norm(x, w, b=None, eps=1e-5, rms=False):
    return NormFn.apply(x, w, b, eps, rms)


class NormPassFn(Function):
    """(norm(x), x): the pre-norm block's input with its pass-through to the residual connection as ONE node, so that the two gradients
    of x (through the norm, through the residual) are summed inside the norm-backward kernel instead of by an autograd add pass
    (64 of them per Llama micro-step)."""

    @staticmethod
    def forward(ctx, x, w, b, eps, rms, pre=None):
        """pre: norm(x) already computed by the producer of x (`LinearNormFn`): used as the output instead of a launch."""
        x = x.contiguous()
        ctx.eps, ctx.rms, ctx.has_b = eps, rms, b is not None
        ctx.gw, ctx.gb = g32_of(w), (g32_of(b) if b is not None else None)
        ctx.save_for_backward(x, w)
        ctx.set_materialize_grads(False)
        return (ops.norm(x, w, b, eps=eps, rms=rms) if pre is None else pre.view_as(pre)), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dpass):
        x, w = ctx.saved_tensors
        if dy is None:
            return dpass, None, None, None, None, None
        dres = None if dpass is None else dpass.contiguous()
        if ctx.gw is not None:
            return ops.norm_bwd(dy.contiguous(), x, w, ctx.eps, ctx.rms, ctx.gw, ctx.gb, dres=dres), None, None, None, None, None
        need_w = ctx.needs_input_grad[1]
        dw = torch.zeros(w.shape, device=w.device, dtype=torch.float32) if need_w else None
        db = torch.zeros(w.shape, device=w.device, dtype=torch.float32) if (need_w and ctx.has_b) else None
        dx = ops.norm_bwd(dy.contiguous(), x, w, ctx.eps, ctx.rms, dw, db, dres=dres)
        return dx, (dw.to(BF16) if need_w else None), (db.to(BF16) if db is not None else None), None, None, None


def norm_pass(x, w, b=None, eps=1e-5, rms=False, pre=None):
    return NormPassFn.apply(x, w, b, eps, rms, pre)


def attention_backward(q, k, v, do, dq, dk, dv, *, batch, heads, Nq, Nk, hd, qs, ks, vs, dos, dqs, dks, dvs, scale, causal=False,
                       key_mask=None):
    """Materialised attention backward.  q/k/v/do and the outputs dq/dk/dv are tensors whose data_ptr is the (b=0,h=0,row=0)
    element; *s = (batch stride, head stride, row stride) in elements.  Writes dq [Nq,hd], dk/dv [Nk,hd] per (b,h)."""
    dev = q.device
    Tp = (Nk + 7) // 8 * 8
    BH = batch * heads
    # S = Q K^T (fp32)
    S = torch.empty((batch, heads, Nq, Tp), device=dev, dtype=torch.float32)
    ops.gemm_batched(q, k, S, M=Nq, N=Nk, K=hd, lda=qs[2], ldw=ks[2], ldc=Tp, batch=heads, sA=qs[1], sW=ks[1], sC=Nq * Tp,
                     batch2=batch, sA2=qs[0], sW2=ks[0], sC2=heads * Nq * Tp, out_f32=True)
    assert Nq == Nk or not causal
    P = ops.softmax_rows(S, BH, Nq, Nk, Tp, scale, causal=causal, key_mask=key_mask, heads=heads)
    # dV[k][d] = sum_q P[q][k] dO[q][d]
    ops.gemm_batched(P, do, dv, M=Nk, N=hd, K=Nq, lda=Tp, ldw=dos[2], ldc=dvs[2], batch=heads, sA=Nq * Tp, sW=dos[1], sC=dvs[1],
                     batch2=batch, sA2=heads * Nq * Tp, sW2=dos[0], sC2=dvs[0], out_f32=False, trans_a=True, trans_w=True)
    # dP = dO V^T (fp32), dS = scale * P * (dP - rowsum(P dP))
    dP = S
    ops.gemm_batched(do, v, dP, M=Nq, N=Nk, K=hd, lda=dos[2], ldw=vs[2], ldc=Tp, batch=heads, sA=dos[1], sW=vs[1], sC=Nq * Tp,
                     batch2=batch, sA2=dos[0], sW2=vs[0], sC2=heads * Nq * Tp, out_f32=True)
    dS = ops.attn_ds(P.view(BH, Nq, Tp), dP, Nk, Tp, scale)
    # dQ[q][d] = sum_k dS[q][k] K[k][d]   (dS rows are zero-padded to Tp)
    ops.gemm_batched(dS, k, dq, M=Nq, N=hd, K=Nk, lda=Tp, ldw=ks[2], ldc=dqs[2], batch=heads, sA=Nq * Tp, sW=ks[1], sC=dqs[1],
                     batch2=batch, sA2=heads * Nq * Tp, sW2=ks[0], sC2=dqs[0], out_f32=False, trans_w=True)
    # dK[k][d] = sum_q dS[q][k] Q[q][d]
    ops.gemm_batched(dS, q, dk, M=Nk, N=hd, K=Nq, lda=Tp, ldw=qs[2], ldc=dks[2], batch=heads, sA=Nq * Tp, sW=qs[1], sC=dks[1],
                     batch2=batch, sA2=heads * Nq * Tp, sW2=qs[0], sC2=dks[0], out_f32=False, trans_a=True, trans_w=True)


class PackedAttnFn(Function):
    """Self attention on a packed qkv [batch*n, 3*heads*hd] buffer (Llama causal+key-mask; head self-attention), optionally with
    the rotate-half RoPE of the q|k part applied first (in place on qkv; positions = row % n).  Backward is the fused recompute
    kernel (`llmseg_attn_bwd`) for head_dim 32/64/128, the materialised GEMM chain otherwise, followed by the inverse rotation on
    the gradient buffer this node itself allocated (RoPE and attention are ONE autograd node, so no other node ever sees the
    buffer that is rotated in place)."""

    @staticmethod
    def forward(ctx, qkv, batch, n, heads, hd, causal, key_mask, rope=None, pre_rotated=False):
        """rope = (cos, sin, -sin) fp32 [n, hd/2] tables or None.  pre_rotated: the q | k part of qkv was rotated by its producer (`LoraQKVFn(rope=)`:
        inside the GEMM); the backward still applies the inverse rotation, i.e. returns the gradient w.r.t. the UNROTATED q|k|v that producer expects."""
        D = heads * hd
        if rope is not None and not pre_rotated:
            ops.rope_(qkv, rope[0], rope[1], batch * n, n, 2 * heads, hd, qkv.stride(0))
            ctx.mark_dirty(qkv)
            ctx.set_materialize_grads(False)          # the rotated qkv output has no consumer: its gradient stays None
        fused = hd in (32, 64, 128)
        lse = torch.empty((batch, heads, n), device=qkv.device, dtype=torch.float32) if fused else None
        out = ops.attention_packed(qkv, batch, n, heads, hd, causal=causal, key_mask=key_mask, lse=lse)
        ctx.args = (batch, n, heads, hd, causal)
        ctx.rope = rope
        ctx.save_for_backward(qkv, key_mask, out if fused else None, lse)
        return (qkv, out) if (rope is not None and not pre_rotated) else out

    @staticmethod
    def backward(ctx, *grads):
        do = grads[-1]
        qkv, key_mask, out, lse = ctx.saved_tensors
        batch, n, heads, hd, causal = ctx.args
        D = heads * hd
        do = do.contiguous()
        dqkv = torch.empty_like(qkv)
        ld = qkv.stride(0)
        st = (n * ld, hd, ld)
        dst = (n * D, hd, D)
        rope_in_bwd = ctx.rope is not None and lse is not None and hd in (64, 128) and FUSE_ROPE_BWD
        if lse is not None:
            ops.attention_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], out, do, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], lse, batch=batch, heads=heads,
                              Nq=n, Nk=n, head_dim=hd, q_strides=st, k_strides=st, v_strides=st, o_strides=(n * out.stride(0), hd, out.stride(0)),
                              do_strides=dst, dq_strides=st, dk_strides=st, dv_strides=st, causal=causal, key_mask=key_mask,
                              rope=(ctx.rope[0], ctx.rope[2]) if rope_in_bwd else None)      # the inverse rotation rides in the dq / dk store
        else:
            attention_backward(qkv, qkv[:, D:], qkv[:, 2 * D:], do, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], batch=batch, heads=heads, Nq=n, Nk=n,
                               hd=hd, qs=st, ks=st, vs=st, dos=dst, dqs=st, dks=st, dvs=st, scale=1.0 / math.sqrt(hd), causal=causal,
                               key_mask=key_mask)
        if ctx.rope is not None:
            assert len(grads) == 1 or grads[0] is None, "the rotated qkv buffer must not be consumed outside this node"
            if not rope_in_bwd:
                ops.rope_(dqkv, ctx.rope[0], ctx.rope[2], batch * n, n, 2 * heads, hd, ld)
        return dqkv, None, None, None, None, None, None, None, None


class AttnOProjFn(Function):
    """(RoPE +) causal attention on a packed q|k|v buffer and the o_proj + residual (+ the residual stream's next RMSNorm) behind it as ONE node (round 6; frozen
    o_proj weight): `PackedAttnFn` followed by `LinearNormFn` / `LinearFn`.  What the merge buys is the backward: the K-sliced dX(o_proj) product's reduce launch also
    writes the attention backward's row statistic delta = rowsum(dO * O) (`llmseg_gemm_args.dl_o`), so `llmseg_attn_bwd` skips its delta launch.  Same bits."""

    @staticmethod
    def forward(ctx, qkv, batch, n, heads, hd, causal, key_mask, rope, pre_rotated, wo, wo_t, residual, norm_w, eps):
        if rope is not None and not pre_rotated:
            ops.rope_(qkv, rope[0], rope[1], batch * n, n, 2 * heads, hd, qkv.stride(0))
            ctx.mark_dirty(qkv)
        lse = torch.empty((batch, heads, n), device=qkv.device, dtype=torch.float32)
        out = ops.attention_packed(qkv, batch, n, heads, hd, causal=causal, key_mask=key_mask, lse=lse)
        ctx.args = (batch, n, heads, hd, causal)
        ctx.rope, ctx.wo_t = rope, wo_t
        ctx.save_for_backward(qkv, key_mask, out, lse)
        ctx.set_materialize_grads(False)
        rot = rope is not None and not pre_rotated
        if norm_w is None:
            y = ops.gemm(out, wo, residual=residual)
            return (qkv, y, None) if rot else (y, None)
        pre = torch.empty((out.shape[0], wo.shape[0]), device=qkv.device, dtype=BF16)
        y = ops.gemm(out, wo, residual=residual, norm_w=norm_w, norm_eps=eps, norm_out=pre)
        ctx.mark_non_differentiable(pre)
        return (qkv, y, pre) if rot else (y, pre)

    @staticmethod
    def backward(ctx, *grads):
        dy = grads[-2]
        qkv, key_mask, out, lse = ctx.saved_tensors
        batch, n, heads, hd, causal = ctx.args
        D = heads * hd
        dy = dy.contiguous()
        do = torch.empty_like(out)
        delta = torch.empty_like(lse) if hd == 128 else None
        ops.gemm(dy, ctx.wo_t, out=do, delta_of=None if delta is None else (out, delta, heads, n))
        dqkv = torch.empty_like(qkv)
        ld = qkv.stride(0)
        st, dst = (n * ld, hd, ld), (n * D, hd, D)
        fuse_rope = ctx.rope is not None and hd in (64, 128) and FUSE_ROPE_BWD
        ops.attention_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], out, do, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], lse, batch=batch, heads=heads, Nq=n, Nk=n, head_dim=hd,
                          q_strides=st, k_strides=st, v_strides=st, o_strides=(n * out.stride(0), hd, out.stride(0)), do_strides=dst, dq_strides=st,
                          dk_strides=st, dv_strides=st, causal=causal, key_mask=key_mask, rope=(ctx.rope[0], ctx.rope[2]) if fuse_rope else None, delta=delta)
        if ctx.rope is not None and not fuse_rope:
            ops.rope_(dqkv, ctx.rope[0], ctx.rope[2], batch * n, n, 2 * heads, hd, ld)
        return (dqkv,) + (None,) * 10 + (dy if ctx.needs_input_grad[11] else None, None, None)


def attn_oproj(qkv, rope, batch, n, heads, hd, causal, key_mask, pre_rotated, wo, wo_t, residual, norm_w, eps):
    """-> (o_proj(attention) + residual, RMSNorm of that * norm_w | None)"""
    r = AttnOProjFn.apply(qkv, batch, n, heads, hd, causal, key_mask, rope, pre_rotated, wo, wo_t, residual, norm_w, eps)
    return r[-2], r[-1]


def rope_attention(qkv, rope, batch, n, heads, hd, causal, key_mask, pre_rotated=False):
    """RoPE (in place on the q|k part of qkv, unless its producer already rotated it) + attention as one autograd node -> attention output."""
    r = PackedAttnFn.apply(qkv, batch, n, heads, hd, causal, key_mask, rope, pre_rotated)
    return r[1] if isinstance(r, tuple) else r


class CrossAttn1QFn(Function):
    """Head's image->token attention: one query per conversation (q [C, D]) over K keys (kv [C*K, 2D] = k | v)."""

    @staticmethod
    def forward(ctx, q, kv, Cn, K, heads, hd):
        D = heads * hd
        o = torch.empty((Cn, D), device=q.device, dtype=BF16)
        ops.attention(q, kv, kv[:, D:], o, batch=Cn, heads=heads, Nq=1, Nk=K, head_dim=hd, q_strides=(D, hd, D),
                      k_strides=(K * 2 * D, hd, 2 * D), v_strides=(K * 2 * D, hd, 2 * D), o_strides=(D, hd, D))
        ctx.args = (Cn, K, heads, hd)
        ctx.save_for_backward(q, kv)
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv = ctx.saved_tensors
        Cn, K, heads, hd = ctx.args
        D = heads * hd
        do = do.contiguous()
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        kvs = (K * 2 * D, hd, 2 * D)
        attention_backward(q, kv, kv[:, D:], do, dq, dkv, dkv[:, D:], batch=Cn, heads=heads, Nq=1, Nk=K, hd=hd, qs=(D, hd, D), ks=kvs,
                           vs=kvs, dos=(D, hd, D), dqs=(D, hd, D), dks=kvs, dvs=kvs, scale=1.0 / math.sqrt(hd))
        return dq, dkv, None, None, None, None


class SwigluFn(Function):
    @staticmethod
    def forward(ctx, gu, inter):
        ctx.inter = inter
        ctx.save_for_backward(gu)
        return ops.swiglu(gu, inter)

    @staticmethod
    def backward(ctx, d):
        (gu,) = ctx.saved_tensors
        return ops.swiglu_bwd(gu, d.contiguous(), ctx.inter), None


class MlpFn(Function):
    """HF LlamaMLP with FROZEN weights + the residual add (+ optionally the residual stream's next RMSNorm) as ONE node (round 6):
        gate|up = x Wgu^T,  h = silu(gate) * up (inside that GEMM's store),  y = h Wd^T + residual (, pre = RMSNorm(y) * norm_w);
    backward:  d(gate|up) = swiglu'(dy Wd) inside the dX GEMM's store,  dx = d(gate|up) Wgu,  d(residual) = dy.
    `swiglu` and `swiglu_bwd` used to be launches of their own (8.8 + 13.4 us per layer at 2 images, each re-reading what a GEMM had just written);
    h is not kept for the backward (no weight gradient needs it).  Same bits as linear -> SwigluFn -> linear_norm.  llava_llama.py:93-102 (HF LlamaMLP)."""

    @staticmethod
    def forward(ctx, x, wgu, wgu_t, wd, wd_t, residual, norm_w, eps):
        M, inter = x.shape[0], wd.shape[1]
        gu = torch.empty((M, 2 * inter), device=x.device, dtype=BF16)
        h = torch.empty((M, inter), device=x.device, dtype=BF16)
        ops.gemm(x, wgu, out=gu, swiglu_out=h)
        ctx.wgu_t, ctx.wd_t = wgu_t, wd_t
        ctx.save_for_backward(gu)
        ctx.set_materialize_grads(False)
        if norm_w is None:
            return ops.gemm(h, wd, residual=residual), None
        pre = torch.empty((M, wd.shape[0]), device=x.device, dtype=BF16)
        y = ops.gemm(h, wd, residual=residual, norm_w=norm_w, norm_eps=eps, norm_out=pre)
        ctx.mark_non_differentiable(pre)
        return y, pre

    @staticmethod
    def backward(ctx, dy, _dpre):
        (gu,) = ctx.saved_tensors
        dy = dy.contiguous()
        dgu = ops.gemm(dy, ctx.wd_t, swiglu_bwd_of=gu)          # [M, 2I] = swiglu_bwd(gu, dy Wd)
        dx = ops.gemm(dgu, ctx.wgu_t) if ctx.needs_input_grad[0] else None
        return dx, None, None, None, None, (dy if ctx.needs_input_grad[5] else None), None, None


def mlp(x, wgu, wgu_t, wd, wd_t, residual, norm_w, eps):
    return MlpFn.apply(x, wgu, wgu_t, wd, wd_t, residual, norm_w, eps)


def embed_token_index(ids, P):
    """Embedding row that receives the gradient of every spliced position ([N, T] int64): text positions -> token id, the image
    span -> -1 (CLIP + mm_projector are frozen).  Index plumbing of the splice (llava_arch.py:185-208)."""
    N, L = ids.shape
    T = L - 1 + P
    pos = (ids == -200).int().argmax(1)
    ar = torch.arange(T, device=ids.device)[None]
    src = torch.where(ar < pos[:, None], ar, (ar - P + 1).clamp(min=0)).clamp(max=L - 1)
    tok = torch.gather(ids, 1, src)
    return torch.where((ar >= pos[:, None]) & (ar < pos[:, None] + P), torch.full_like(tok, -1), tok).contiguous()


class EmbedSpliceFn(Function):
    """LLaVA splice; gradient flows to the embedding table only (CLIP + mm_projector are frozen, training.py:173-176)."""

    @staticmethod
    def forward(ctx, ids, embed, img_feats, P, fstride, tok_index=None):
        out = ops.embed_splice(ids, embed, img_feats, P, feats_stride_n=fstride)
        ctx.P = P
        ctx.vocab = embed.shape
        ctx.g = g32_of(embed)
        ctx.save_for_backward(ids, tok_index)
        return out

    @staticmethod
    def backward(ctx, d):
        ids, tok = ctx.saved_tensors
        N, L = ids.shape
        T = L - 1 + ctx.P
        if tok is None:
            tok = embed_token_index(ids, ctx.P)
        src = d.contiguous().view(N * T, -1)
        if ctx.g is not None:
            ops.scatter_add_rows(src, tok.reshape(-1), ctx.g)
            return None, None, None, None, None, None
        g32 = torch.zeros(ctx.vocab, device=d.device, dtype=torch.float32)
        ops.scatter_add_rows(src, tok.reshape(-1).contiguous(), g32)
        return None, g32.to(BF16), None, None, None, None


class GatherRowsFn(Function):
    @staticmethod
    def forward(ctx, x, idx):
        ctx.shape = x.shape
        ctx.save_for_backward(idx)
        return ops.gather_rows(x, idx)

    @staticmethod
    def backward(ctx, d):
        (idx,) = ctx.saved_tensors
        g32 = torch.zeros(ctx.shape, device=d.device, dtype=torch.float32)
        if idx.numel():
            ops.scatter_add_rows(d.contiguous(), idx, g32)
        return g32.to(BF16), None


class CELossFn(Function):
    @staticmethod
    def forward(ctx, logits, labels):
        acc = ops.ce_loss(logits, labels)
        ctx.save_for_backward(logits, labels, acc)
        return acc[0] / acc[1]

    @staticmethod
    def backward(ctx, g):
        logits, labels, acc = ctx.saved_tensors
        coef = (g.float() / acc[1]).reshape(1).contiguous()
        return ops.ce_bwd(logits, labels, coef), None


class AlignRegFn(Function):
    """softmax_align_loss + iou_regression_loss for one (image, round) -> fp32[2], or for R stacked items (e [R,K,D], t [R,D], pred /
    gt [R,K]) -> fp32 [R, 2] in one launch; grads come from the same kernel."""

    @staticmethod
    def forward(ctx, e, t, pred, gt_iou, gt_iop):
        out, d_e, d_t, d_p = ops.align_reg_loss(e, t, gt_iou, pred, gt_iop, want_grads=True)
        ctx.save_for_backward(d_e, d_t, d_p)
        return out

    @staticmethod
    def backward(ctx, g):
        d_e, d_t, d_p = ctx.saved_tensors
        g = g.float()
        if g.dim() == 2:                                   # batched: g [R, 2]
            return ((d_e * g[:, 0, None, None]).to(BF16), (d_t * g[:, 0, None]).to(BF16), (d_p * g[:, 1, None]).to(BF16), None, None)
        return (d_e * g[0]).to(BF16), (d_t * g[0]).to(BF16), (d_p * g[1]).to(BF16), None, None


class DiceBceFn(Function):
    """dice_loss + sigmoid_ce_loss on mask logits (reference model/loss.py:4-47; named by the north_star, no caller in the reference):
    logits / targets fp32 [M, H, W] -> fp32[2] = (dice, bce), both already divided by (num_masks + 1e-8)."""

    @staticmethod
    def forward(ctx, logits, targets, num_masks):
        logits, targets = logits.contiguous(), targets.contiguous()
        ctx.num = float(num_masks)
        ctx.save_for_backward(logits, targets)
        return ops.dice_bce(logits, targets, num_masks)

    @staticmethod
    def backward(ctx, g):
        logits, targets = ctx.saved_tensors
        return ops.dice_bce_bwd(logits, targets, g.float().contiguous(), ctx.num), None, None


class MaskPoolFn(Function):
    @staticmethod
    def forward(ctx, feat_cl, segs, g, S):
        if feat_cl.requires_grad:
            out, pb, ws = ops.upsample_maskpool(feat_cl, segs, g, S, want_aux=True)
            ctx.save_for_backward(pb, ws)
        else:
            out = ops.upsample_maskpool(feat_cl, segs, g, S)
        return out

    @staticmethod
    def backward(ctx, d):
        pb, ws = ctx.saved_tensors
        # d_feat[s][c] = sum_k pb[k][s] * d[k][c] / (wsum_k + 1e-8)
        dn = (d.float() / (ws[:, None] + 1e-8)).to(BF16).contiguous()
        return ops.gemm(pb.to(BF16).contiguous(), dn, trans_a=True, trans_w=True), None, None, None


class BcastAddFn(Function):
    """s[c*K + k] += add[c]  (the head's single-key cross attentions broadcast one vector per conversation)."""

    @staticmethod
    def forward(ctx, s, add, Cn, K):
        out = torch.empty_like(s)
        for ci in range(Cn):
            ops.add_rows(s[ci * K:(ci + 1) * K], add[ci:ci + 1].contiguous(), out=out[ci * K:(ci + 1) * K])
        ctx.args = (Cn, K)
        return out

    @staticmethod
    def backward(ctx, d):
        Cn, K = ctx.args
        d = d.contiguous()
        dadd = torch.zeros((Cn, d.shape[1]), device=d.device, dtype=torch.float32)
        for ci in range(Cn):
            ops.colsum(d[ci * K:(ci + 1) * K], out=dadd[ci])
        return d, dadd.to(BF16), None, None
