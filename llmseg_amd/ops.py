"""Tensor-level wrappers over the C ABI (include/llmseg_hip.h).

PyTorch is plumbing here: it owns device memory (`torch.empty`) and the current HIP stream; every
arithmetic op below is a kernel of libllmseg_hip.so.  Inputs must be CUDA(HIP) tensors; nothing in
this module computes on the CPU.
"""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_NONE, ACT_QUICKGELU, ACT_RELU, ACT_SIGMOID, ACT_SILU, AttnArgs, AttnBwdArgs, GemmArgs)

BF16 = torch.bfloat16


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _req(t, dtype=BF16):
    assert t.is_cuda and t.dtype == dtype, f"expected cuda {dtype}, got {t.device} {t.dtype}"
    return t


WS_BYTES = 192 << 20
_WS = {}


def _workspace(dev, stream):
    """Per-(device, stream) scratch of the split-K GEMM path (fp32 partial tiles), allocated on first use and kept."""
    key = (dev.index, stream)
    ws = _WS.get(key)
    if ws is None:
        ws = _WS[key] = torch.zeros(WS_BYTES, device=dev, dtype=torch.uint8)
    return ws


REDUCE_WS_BYTES = 16 << 20          # == LLMSEG_REDUCE_WS_BYTES of include/llmseg_hip.h
_RWS = {}


def _reduce_ws(dev):
    """(pointer, bytes) of the per-(device, stream) scratch that the fixed-order reductions write their per-workgroup partials to."""
    key = (dev.index, torch.cuda.current_stream().cuda_stream)
    ws = _RWS.get(key)
    if ws is None:
        ws = _RWS[key] = torch.zeros(REDUCE_WS_BYTES, device=dev, dtype=torch.uint8)
    return C.c_void_p(ws.data_ptr()), ws.numel()


def gemm(a, w, bias=None, act=ACT_NONE, residual=None, gamma=None, out=None, alpha=1.0, out_f32=False, trans_a=False, trans_w=False,
         a2=None, w2=None, accumulate=False, a_norm_w=None, a_norm_eps=1e-6, a_swiglu=False, norm_w=None, norm_eps=1e-6, norm_out=None,
         rope=None, swiglu_out=None, swiglu_bwd_of=None, normbwd=None, nb_lora=None, delta_of=None):
    """out[M,N] = residual + gamma * act(alpha * A @ W^T + bias) with A = a [M,K] (or a^T when trans_a: a stored [K,M]) and
    W = w [N,K] (or w^T when trans_w: w stored [K,N]).  2-D bf16 operands, last dim contiguous.  a2 [M,64] / w2 [N,64]: optional
    extension of the contraction (A @ W^T + a2 @ w2^T), e.g. zero-padded low-rank updates.  accumulate (fp32 `out` only):
    out += result (gradient accumulation).  M <= 8 only (decode steps): a_norm_w -> A := RMSNorm(A) * a_norm_w on load; a_swiglu ->
    a holds [gate | up] rows of width 2K and A := silu(gate) * up on load.
    Fused Llama-layer epilogues (llmseg_gemm_args.fx; the same bits as the pointwise launch each replaces):
      rope = (cos, sin, T, cols): the heads (width 128) of the first `cols` output columns are rotated, position = row % T (`rope_` after the product);
      swiglu_out = h [M, N/2]: out = gate|up as usual and h = silu(gate) * up (`swiglu` after the product);
      swiglu_bwd_of = gu [M, 2N]: the product is d(h) [M, N]; out [M, 2N] = d(gate|up) (`swiglu_bwd` of the stored product).
    Norm-backward tail (llmseg_gemm_args.nb_x): normbwd = (x, w, eps, rms, dres | None): the product is the gradient of a pre-norm's output and
    out = norm_bwd(product, x, w) + dres; nb_lora = (t [M, >=16], w0 [8, N], w1 [8, N] | None, alpha, drop | None) adds the LoRA branches' dX to the
    product first (`lora_apply_` with w_rn).  One pass in the K-sliced route, the three launches otherwise; same bits.
    Delta tail (llmseg_gemm_args.dl_o): delta_of = (o [M, N], delta fp32 [batch, heads, T], heads, T): the product is dO of an attention with output o and
    the call also writes delta = rowsum_d(dO * o) per head (width 128) -- pass it to `attention_bwd(..., delta=)`."""
    _req(a); _req(w)
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    if a_swiglu:
        K //= 2
    N, Kw = (w.shape[1], w.shape[0]) if trans_w else w.shape
    assert K == Kw, (a.shape, w.shape, trans_a, trans_w)
    if out is None:
        assert not accumulate
        out = torch.empty((M, 2 * N if swiglu_bwd_of is not None else N), device=a.device, dtype=torch.float32 if out_f32 else BF16)
    else:
        out_f32 = out.dtype == torch.float32
    assert out.shape == (M, 2 * N if swiglu_bwd_of is not None else N) and out.stride(1) == 1 and (out_f32 or not accumulate)
    stream = torch.cuda.current_stream().cuda_stream
    g = GemmArgs(A=a.data_ptr(), W=w.data_ptr(), C=out.data_ptr(),
                 bias=None if bias is None else _req(bias).data_ptr(),
                 gamma=None if gamma is None else _req(gamma).data_ptr(),
                 residual=None if residual is None else _req(residual).data_ptr(),
                 M=M, N=N, K=K, lda=a.stride(0), ldw=w.stride(0), ldc=out.stride(0),
                 ldr=0 if residual is None else residual.stride(0),
                 batch=1, strideA=0, strideW=0, strideC=0, alpha=alpha, act=act, out_f32=1 if out_f32 else 0,
                 trans_a=1 if trans_a else 0, trans_w=1 if trans_w else 0, accumulate=1 if accumulate else 0)
    if norm_out is not None:                                            # second output: RMSNorm(out) * norm_w (llmseg_gemm_args.norm_out, ABI 6)
        assert norm_w is not None and not out_f32 and norm_out.shape == out.shape and norm_out.stride(1) == 1 and norm_out.dtype == BF16
        g.norm_w, g.norm_eps, g.norm_out, g.ldn = _req(norm_w).data_ptr(), norm_eps, _req(norm_out).data_ptr(), norm_out.stride(0)
    if rope is not None:
        cos, sin, T, cols = rope
        assert cos.dtype == torch.float32 and sin.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous() and cos.shape == sin.shape == (T, 64)
        g.fx, g.fx_T, g.fx_cols, g.fx_cos, g.fx_sin = _lib.FX_ROPE, T, cols, cos.data_ptr(), sin.data_ptr()
    elif swiglu_out is not None:
        assert _req(swiglu_out).shape == (M, N // 2) and swiglu_out.stride(1) == 1
        g.fx, g.fx_out, g.fx_ld = _lib.FX_SWIGLU, swiglu_out.data_ptr(), swiglu_out.stride(0)
    elif swiglu_bwd_of is not None:
        assert _req(swiglu_bwd_of).shape == (M, 2 * N) and swiglu_bwd_of.stride(1) == 1
        g.fx, g.fx_in, g.fx_ld = _lib.FX_SWIGLU_BWD, swiglu_bwd_of.data_ptr(), swiglu_bwd_of.stride(0)
    keep = None
    if normbwd is not None:
        nx, nw, neps, nrms, ndres = normbwd
        assert _req(nx).shape == (M, N) and nx.is_contiguous() and _req(nw).numel() == N and out.is_contiguous() and not out_f32
        g.nb_x, g.nb_w, g.nb_eps, g.nb_rms = nx.data_ptr(), nw.data_ptr(), neps, 1 if nrms else 0
        if ndres is not None:
            assert _req(ndres).shape == (M, N) and ndres.is_contiguous()
            g.nb_dres = ndres.data_ptr()
        if nb_lora is not None:
            lt, lw0, lw1, lalpha, ldrop = nb_lora[:5]
            if len(nb_lora) > 5 and nb_lora[5] is not None:                       # t still as lora_down's K-slice partials: (..., partials, S, scale, zero_cols)
                lpart, lS, lscale, lzero = nb_lora[5:9]
                assert lpart.dtype == torch.float32 and lpart.numel() >= lS * M * 16 and lt.shape[1] >= 16 + lzero
                g.nb_lora_part, g.nb_lora_S, g.nb_lora_scale, g.nb_lora_zero = lpart.data_ptr(), lS, lscale, lzero
            assert _req(lt).shape[0] == M and lt.stride(1) == 1 and _req(lw0).shape == (8, N) and lw0.is_contiguous() and (lw1 is None or (_req(lw1).shape == (8, N) and lw1.is_contiguous()))
            g.nb_lora_t, g.nb_lora_ldt, g.nb_lora_w0, g.nb_lora_alpha = lt.data_ptr(), lt.stride(0), lw0.data_ptr(), lalpha
            g.nb_lora_w1 = None if lw1 is None else lw1.data_ptr()
            if ldrop is not None and ldrop[2] > 0.0:
                rng, st, pd = ldrop[:3]
                keep = _lib.Dropout(rng_state=rng.data_ptr(), stream=int(st), drop_thr=int(round(pd * 65536)), seg_rows=int(ldrop[3]) if len(ldrop) > 3 else 0, reserved0=0)
                g.nb_lora_drop = C.addressof(keep)
    else:
        assert nb_lora is None
    if delta_of is not None:
        do_, dl, dh, dT = delta_of
        assert _req(do_).shape == (M, N) and do_.stride(1) == 1 and N == dh * 128 and M % dT == 0 and dl.dtype == torch.float32 and dl.is_contiguous() and dl.numel() == (M // dT) * dh * dT
        g.dl_o, g.dl_ldo, g.dl_out, g.dl_heads, g.dl_T = do_.data_ptr(), do_.stride(0), dl.data_ptr(), dh, dT
    if a_norm_w is not None:
        g.a_norm_w, g.a_norm_eps = _req(a_norm_w).data_ptr(), a_norm_eps
    if a_swiglu:
        g.a_swiglu = 1
    if a2 is not None:
        _req(a2); _req(w2)
        assert a2.shape == (M, 64) and w2.shape == (N, 64) and a2.stride(1) == 1 and w2.stride(1) == 1
        g.A2, g.W2, g.lda2, g.ldw2 = a2.data_ptr(), w2.data_ptr(), a2.stride(0), w2.stride(0)
    if K >= 256 or normbwd is not None:                                 # split-K candidates (and the two-launch route of the norm-backward tail): hand the scratch over
        ws = _workspace(a.device, stream)
        g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel()
    if residual is not None:
        assert residual.shape == (M, N) and residual.stride(1) == 1
    _lib.check(_lib.load().llmseg_gemm_bf16(C.byref(g), C.c_void_p(stream)), "gemm")
    return out


def gemm_batched(a, w, out, M, N, K, lda, ldw, ldc, batch, sA, sW, sC, out_f32=True, alpha=1.0, trans_a=False, trans_w=False,
                 batch2=1, sA2=0, sW2=0, sC2=0):
    """Strided-batched GEMM on raw strides (elements); used for the per-head q.R^T relative-position products."""
    g = GemmArgs(A=a.data_ptr(), W=w.data_ptr(), C=out.data_ptr(), bias=None, gamma=None, residual=None,
                 M=M, N=N, K=K, lda=lda, ldw=ldw, ldc=ldc, ldr=0, batch=batch, strideA=sA, strideW=sW, strideC=sC,
                 alpha=alpha, act=ACT_NONE, out_f32=1 if out_f32 else 0, trans_a=1 if trans_a else 0, trans_w=1 if trans_w else 0,
                 batch2=batch2, strideA2=sA2, strideW2=sW2, strideC2=sC2)
    if K >= 256 and batch2 <= 1:
        ws = _workspace(a.device, torch.cuda.current_stream().cuda_stream)
        g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel()
    _lib.check(_lib.load().llmseg_gemm_bf16(C.byref(g), _stream()), "gemm_batched")
    return out


def attention(q, k, v, out, *, batch, heads, Nq, Nk, head_dim, q_strides, k_strides, v_strides, o_strides, scale=None,
              causal=False, key_mask=None, rel_h=None, rel_w=None, rel_ld=0, grid_hw=(0, 0), o_row_map=None, rel_tab_h=None, rel_tab_w=None, lse=None,
              nk_dev=None, win_gather=None):
    """Strided fused attention.  *_strides = (batch, head, row) in elements relative to the given tensors' data_ptr.
    win_gather = (grid, pad_q, pad_k, pad_v): SAM 14x14 windows gathered from / scattered to UNPARTITIONED token rows inside the kernel (see llmseg_attn_args.win_grid)."""
    a = AttnArgs(Q=q.data_ptr(), K=k.data_ptr(), V=v.data_ptr(), O=out.data_ptr(),
                 q_stride_b=q_strides[0], q_stride_h=q_strides[1], q_stride_row=q_strides[2],
                 k_stride_b=k_strides[0], k_stride_h=k_strides[1], k_stride_row=k_strides[2],
                 v_stride_b=v_strides[0], v_stride_h=v_strides[1], v_stride_row=v_strides[2],
                 o_stride_b=o_strides[0], o_stride_h=o_strides[1], o_stride_row=o_strides[2],
                 batch=batch, heads=heads, Nq=Nq, Nk=Nk, head_dim=head_dim,
                 scale=(1.0 / math.sqrt(head_dim)) if scale is None else scale, causal=1 if causal else 0,
                 key_mask=None if key_mask is None else key_mask.data_ptr(),
                 rel_h=None if rel_h is None else rel_h.data_ptr(), rel_w=None if rel_w is None else rel_w.data_ptr(),
                 rel_ld=rel_ld, grid_h=grid_hw[0], grid_w=grid_hw[1],
                 o_row_map=None if o_row_map is None else o_row_map.data_ptr(),
                 rel_tab_h=None if rel_tab_h is None else rel_tab_h.data_ptr(), rel_tab_w=None if rel_tab_w is None else rel_tab_w.data_ptr(),
                 lse=None if lse is None else lse.data_ptr(), nk_dev=None if nk_dev is None else nk_dev.data_ptr())
    if win_gather is not None:
        g, pq, pk, pv = win_gather
        a.win_grid, a.win_nw, a.pad_q, a.pad_k, a.pad_v = g, (g + 13) // 14, pq.data_ptr(), pk.data_ptr(), pv.data_ptr()
    _lib.check(_lib.load().llmseg_attn_fwd(C.byref(a), _stream()), "attn_fwd")
    return out


def attention_packed(qkv, batch, n_tok, heads, head_dim, out=None, win_pad=None, **kw):
    """qkv [batch*n_tok, 3*heads*head_dim] (q|k|v, heads-major inside each) -> out [batch*n_tok, heads*head_dim].
    win_pad = (grid, pad [3*heads*head_dim] bf16): qkv / out hold the rows of the unpartitioned tokens of images on a grid x grid grid, `batch` counts the 14x14 windows
    (window gather, see `attention`); pad = the q|k|v row of a padded token (the projection's bias)."""
    D = heads * head_dim
    if win_pad is not None:
        g, pad = win_pad
        nw = (g + 13) // 14
        assert qkv.shape == (batch // (nw * nw) * g * g, 3 * D) and qkv.stride(1) == 1 and pad.shape == (3 * D,) and pad.dtype == BF16 and pad.is_contiguous()
        kw["win_gather"] = (g, pad, pad[D:], pad[2 * D:])
    else:
        assert qkv.shape == (batch * n_tok, 3 * D) and qkv.stride(1) == 1
    ld = qkv.stride(0)
    if out is None:
        out = torch.empty((qkv.shape[0], D), device=qkv.device, dtype=BF16)
    st = (n_tok * ld, head_dim, ld)
    return attention(qkv, qkv[:, D:], qkv[:, 2 * D:], out, batch=batch, heads=heads, Nq=n_tok, Nk=n_tok, head_dim=head_dim,
                     q_strides=st, k_strides=st, v_strides=st, o_strides=(n_tok * out.stride(0), head_dim, out.stride(0)), **kw)


def attention_bwd(q, k, v, o, do, dq, dk, dv, lse, *, batch, heads, Nq, Nk, head_dim, q_strides, k_strides, v_strides, o_strides, do_strides,
                  dq_strides, dk_strides, dv_strides, scale=None, causal=False, key_mask=None, rope=None, delta=None):
    """Fused attention backward (dq, dk, dv are written).  lse = the fp32 [batch, heads, Nq] row statistics the forward wrote.
    rope = (cos, sin) fp32 [Nq, head_dim/2]: every dq / dk row is rotated at its position before the store (pass -sin for the inverse).
    delta: fp32 [batch, heads, Nq] = rowsum(dO * O) already computed (`gemm(..., delta_of=)`): the kernel's own delta launch is skipped."""
    assert lse.dtype == torch.float32 and lse.is_contiguous() and lse.numel() == batch * heads * Nq
    ready = delta is not None
    if ready:
        assert delta.dtype == torch.float32 and delta.is_contiguous() and delta.numel() == lse.numel()
    else:
        delta = torch.empty_like(lse)
    kw = {}
    for name, st in (("q", q_strides), ("k", k_strides), ("v", v_strides), ("o", o_strides), ("do", do_strides), ("dq", dq_strides),
                     ("dk", dk_strides), ("dv", dv_strides)):
        kw[f"{name}_stride_b"], kw[f"{name}_stride_h"], kw[f"{name}_stride_row"] = st
    a = AttnBwdArgs(Q=q.data_ptr(), K=k.data_ptr(), V=v.data_ptr(), O=o.data_ptr(), dO=do.data_ptr(), dQ=dq.data_ptr(), dK=dk.data_ptr(),
                    dV=dv.data_ptr(), batch=batch, heads=heads, Nq=Nq, Nk=Nk, head_dim=head_dim,
                    scale=(1.0 / math.sqrt(head_dim)) if scale is None else scale, causal=1 if causal else 0,
                    key_mask=None if key_mask is None else key_mask.data_ptr(), lse=lse.data_ptr(), delta=delta.data_ptr(), **kw)
    a.delta_ready = 1 if ready else 0
    if rope is not None:
        assert all(t.dtype == torch.float32 and t.is_contiguous() and t.shape == (Nq, head_dim // 2) for t in rope)
        a.rope_cos, a.rope_sin = rope[0].data_ptr(), rope[1].data_ptr()
    _lib.check(_lib.load().llmseg_attn_bwd(C.byref(a), _stream()), "attn_bwd")


def norm(x, w, b=None, eps=1e-5, rms=False, out=None, row_map=None, out_rows=None):
    _req(x); _req(w)
    assert x.dim() == 2 and x.stride(1) == 1
    rows, cols = x.shape
    if out is None:
        out = torch.empty((rows if out_rows is None else out_rows, cols), device=x.device, dtype=BF16)
    _lib.check(_lib.load().llmseg_norm(_ptr(x), _ptr(w), _ptr(b), _ptr(out), rows, cols, x.stride(0), out.stride(0), eps,
                                       1 if rms else 0, _ptr(row_map), _stream()), "norm")
    return out


def rope_(x, cos, sin, rows, T, heads, head_dim, ld):
    _lib.check(_lib.load().llmseg_rope(_ptr(x), _ptr(cos), _ptr(sin), rows, T, heads, head_dim, ld, _stream()), "rope")
    return x


def rope_kv_append_(qkv, cos, sin, kcache, vcache, pos_dev, heads, head_dim):
    """Decode step: rotate q (in place) and k at position *pos_dev, write k / v into the caches [N, capacity, heads*head_dim]."""
    N = qkv.shape[0]
    assert kcache.shape == vcache.shape and kcache.shape[0] == N and kcache.stride(1) == heads * head_dim and pos_dev.dtype == torch.int32
    _lib.check(_lib.load().llmseg_rope_kv_append(_ptr(qkv), qkv.stride(0), _ptr(cos), _ptr(sin), _ptr(kcache), _ptr(vcache), kcache.stride(0), _ptr(pos_dev),
                                                 N, heads, head_dim, _stream()), "rope_kv_append")
    return qkv


def decode_attn(qkv, cos, sin, kcache, vcache, pos_dev, heads, head_dim, out=None, scale=None, scratch=None):
    """Decode step in one launch (+ merge): k rotated / v copied into the caches at position *pos_dev, out[n] = attention of the rotated q
    over the pos + 1 cached keys.  head_dim 128.  scratch: fp32 buffer from decode_attn_scratch (None = one workgroup per head)."""
    N = qkv.shape[0]
    assert kcache.shape == vcache.shape and kcache.shape[0] == N and kcache.stride(1) == heads * head_dim and pos_dev.dtype == torch.int32
    if out is None:
        out = torch.empty((N, heads * head_dim), device=qkv.device, dtype=torch.bfloat16)
    _lib.check(_lib.load().llmseg_decode_attn(_ptr(qkv), qkv.stride(0), _ptr(cos), _ptr(sin), _ptr(kcache), _ptr(vcache), kcache.stride(0), _ptr(pos_dev),
                                              N, heads, head_dim, float(scale if scale is not None else head_dim ** -0.5), _ptr(out), out.stride(0),
                                              _ptr(scratch) if scratch is not None else None, scratch.numel() * 4 if scratch is not None else 0,
                                              _stream()), "decode_attn")
    return out


def decode_attn_scratch(N, heads, device):
    """fp32 scratch of decode_attn's key split: [N][heads][splits][130]."""
    splits = min(16, max(1, 256 // (N * heads)))
    return torch.empty((N * heads * splits * 130,), device=device, dtype=torch.float32) if splits > 1 else None


def act_(x, act):
    """In-place elementwise activation (bf16, contiguous)."""
    assert x.is_contiguous() and x.numel() % 8 == 0
    _lib.check(_lib.load().llmseg_act(_ptr(x), _ptr(x), x.numel(), act, _stream()), "act")
    return x


def sam_postprocess(low, input_size, original_size, img_size=1024, nested=True):
    """low fp32 [n, 65536] mask logits -> fp32 [n, H, W] (Sam.postprocess_masks)."""
    assert low.dtype == torch.float32 and low.is_contiguous() and low.shape[1] == 65536
    n = low.shape[0]
    out = torch.empty((n, int(original_size[0]), int(original_size[1])), device=low.device, dtype=torch.float32)
    if n:
        _lib.check(_lib.load().llmseg_sam_postprocess(_ptr(low), _ptr(out), n, img_size, int(input_size[0]), int(input_size[1]), int(original_size[0]),
                                                      int(original_size[1]), 1 if nested else 0, _stream()), "sam_postprocess")
    return out


def sam_mask_stats(low, iou, iou_thresh, input_size, original_size, img_size=1024, mask_threshold=0.0, offset=1.0, nested=True):
    """low fp32 [n, 65536], iou fp32 [n] -> int32 [n, 7] = {|m > thr+off|, |m > thr-off|, |m > thr|, min x, min y, max x, max y} of every candidate
    whose predicted IoU passes (rows of the others keep their initial value)."""
    n = low.shape[0]
    assert low.dtype == torch.float32 and low.is_contiguous() and low.shape[1] == 65536 and iou.dtype == torch.float32 and iou.numel() == n
    st = torch.tensor([0, 0, 0, 2 ** 31 - 1, 2 ** 31 - 1, -1, -1], device=low.device, dtype=torch.int32).repeat(n, 1)
    _lib.check(_lib.load().llmseg_sam_mask_stats(_ptr(low), _ptr(iou), float(iou_thresh), _ptr(st), n, img_size, int(input_size[0]), int(input_size[1]),
                                                 int(original_size[0]), int(original_size[1]), 1 if nested else 0, float(mask_threshold), float(offset),
                                                 _stream()), "sam_mask_stats")
    return st


def sam_binarize(low, sel, input_size, original_size, img_size=1024, mask_threshold=0.0, nested=True):
    """uint8 [len(sel), H, W] binary masks of the selected candidates (rows `sel` of low)."""
    k = sel.shape[0]
    out = torch.empty((k, int(original_size[0]), int(original_size[1])), device=low.device, dtype=torch.uint8)
    if k:
        _lib.check(_lib.load().llmseg_sam_binarize(_ptr(low), _ptr(sel.to(torch.int32).contiguous()), _ptr(out), k, img_size, int(input_size[0]), int(input_size[1]),
                                                   int(original_size[0]), int(original_size[1]), 1 if nested else 0, float(mask_threshold), _stream()), "sam_binarize")
    return out


def image_resize_u8(img, out_h, out_w, crop_box=None):
    """img uint8 [H, W, C] on the device (contiguous) -> uint8 [out_h, out_w, C]: Pillow's BILINEAR `Image.resize`, bit-identical
    (`ResizeLongestSide.apply_image`).  crop_box = (x0, y0, x1, y1) resizes that window of img without copying it."""
    assert img.dtype == torch.uint8 and img.is_contiguous() and img.dim() == 3 and img.is_cuda
    H, W, ch = img.shape
    x0, y0, x1, y1 = (0, 0, W, H) if crop_box is None else [int(v) for v in crop_box]
    assert 0 <= x0 < x1 <= W and 0 <= y0 < y1 <= H
    lib = _lib.load()
    nb = lib.llmseg_image_resize_workspace(y1 - y0, x1 - x0, int(out_h), int(out_w), ch)
    ws = torch.empty((nb,), device=img.device, dtype=torch.uint8)
    out = torch.empty((int(out_h), int(out_w), ch), device=img.device, dtype=torch.uint8)
    _lib.check(lib.llmseg_image_resize_u8(img.data_ptr() + (y0 * W + x0) * ch, W * ch, _ptr(out), y1 - y0, x1 - x0, int(out_h), int(out_w), ch, _ptr(ws), nb,
                                          _stream()), "image_resize_u8")
    return out


def sam_preprocess(img, img_size, mean, std):
    """img uint8 [h, w, 3] -> bf16 [1, 3, img_size, img_size]: (x - mean) / std, zero padding (`Sam.preprocess`)."""
    import ctypes
    assert img.dtype == torch.uint8 and img.is_contiguous() and img.dim() == 3 and img.shape[2] == 3
    out = torch.empty((1, 3, img_size, img_size), device=img.device, dtype=torch.bfloat16)
    m3, s3 = (ctypes.c_float * 3)(*[float(v) for v in mean]), (ctypes.c_float * 3)(*[float(v) for v in std])
    _lib.check(_lib.load().llmseg_sam_preprocess(_ptr(img), _ptr(out), img.shape[0], img.shape[1], int(img_size), ctypes.cast(m3, ctypes.c_void_p),
                                                 ctypes.cast(s3, ctypes.c_void_p), _stream()), "sam_preprocess")
    return out


SMALL_REGIONS_WS_BYTES = 1 << 30


def mask_small_regions_(masks, min_area):
    """masks uint8 [K, H, W] IN PLACE: holes below min_area filled, then islands below min_area removed (`remove_small_regions` twice, as
    `postprocess_small_regions` applies it) -> uint8 [K] changed flags."""
    assert masks.dtype == torch.uint8 and masks.is_contiguous() and masks.dim() == 3
    K, H, W = masks.shape
    changed = torch.zeros((K,), device=masks.device, dtype=torch.uint8)
    if K:
        lib = _lib.load()
        # masks in chunks that keep the label workspace (8 bytes per pixel) under a fixed budget: 300 masks of a 1500 x 2250 photograph
        # would otherwise ask for 8 GB at once (the reference cleans one mask at a time, amg.py:267-291)
        per = max(1, int(lib.llmseg_mask_small_regions_workspace(1, H, W)))
        kc = max(1, min(K, SMALL_REGIONS_WS_BYTES // per))
        nb = lib.llmseg_mask_small_regions_workspace(kc, H, W)
        ws = torch.empty((nb,), device=masks.device, dtype=torch.uint8)
        for k0 in range(0, K, kc):
            k1 = min(K, k0 + kc)
            _lib.check(lib.llmseg_mask_small_regions(_ptr(masks[k0:k1]), k1 - k0, H, W, int(min_area), _ptr(changed[k0:k1]), _ptr(ws), nb, _stream()),
                       "mask_small_regions")
    return changed


def mask_boxes(masks):
    """masks uint8 [K, H, W] -> (boxes int32 [K, 4] XYXY inclusive, zeros when empty; areas int32 [K]) (`batched_mask_to_box`)."""
    assert masks.dtype == torch.uint8 and masks.is_contiguous() and masks.dim() == 3
    K, H, W = masks.shape
    boxes = torch.zeros((K, 4), device=masks.device, dtype=torch.int32)
    areas = torch.zeros((K,), device=masks.device, dtype=torch.int32)
    if K:
        ws = torch.empty((K * 5,), device=masks.device, dtype=torch.int32)
        _lib.check(_lib.load().llmseg_mask_boxes(_ptr(masks), K, H, W, _ptr(boxes), _ptr(areas), _ptr(ws), K * 20, _stream()), "mask_boxes")
    return boxes, areas


def nms(boxes, order, iou_threshold):
    """boxes fp32 [*, 4] XYXY, order int32 [n] (candidate indices by decreasing score) -> uint8 [n] keep flags (torchvision.ops.nms semantics)."""
    n = order.shape[0]
    keep = torch.empty((n,), device=boxes.device, dtype=torch.uint8)
    if n:
        assert boxes.dtype == torch.float32 and boxes.is_contiguous() and order.dtype == torch.int32 and order.is_contiguous()
        _lib.check(_lib.load().llmseg_nms(_ptr(boxes), _ptr(order), n, float(iou_threshold), _ptr(keep), _stream()), "nms")
    return keep


def swiglu(gu, inter, out=None):
    rows = gu.shape[0]
    if out is None:
        out = torch.empty((rows, inter), device=gu.device, dtype=BF16)
    _lib.check(_lib.load().llmseg_swiglu(_ptr(gu), _ptr(out), rows, inter, gu.stride(0), out.stride(0), _stream()), "swiglu")
    return out


def add_rows(x, add, out=None):
    """x [rows, cols] + add[(row % add.shape[0])]."""
    assert x.is_contiguous() and add.is_contiguous() and x.shape[1] == add.shape[1]
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.load().llmseg_add_rows(_ptr(x), _ptr(add), _ptr(out), x.shape[0], x.shape[1], add.shape[0], _stream()), "add_rows")
    return out


def patchify(img, p, ldo, rows_per_img=None, row_off=0, out=None):
    _req(img)
    B, Cc, H, W = img.shape
    assert Cc == 3 and img.is_contiguous()
    n = (H // p) * (W // p)
    rows_per_img = n if rows_per_img is None else rows_per_img
    if out is None:
        out = torch.zeros((B * rows_per_img, ldo), device=img.device, dtype=BF16) if row_off else \
            torch.empty((B * rows_per_img, ldo), device=img.device, dtype=BF16)
    _lib.check(_lib.load().llmseg_patchify(_ptr(img), _ptr(out), B, H, W, p, ldo, rows_per_img, row_off, _stream()), "patchify")
    return out


def im2col3x3(x, B, H, W, Cc):
    out = torch.empty((B * H * W, 9 * Cc), device=x.device, dtype=BF16)
    _lib.check(_lib.load().llmseg_im2col3x3(_ptr(x), _ptr(out), B, H, W, Cc, _stream()), "im2col3x3")
    return out


def embed_splice(ids, embed, img_feats, P, feats_stride_n=None):
    N, L = ids.shape
    H = embed.shape[1]
    out = torch.empty((N, L - 1 + P, H), device=embed.device, dtype=BF16)
    _lib.check(_lib.load().llmseg_embed_splice(_ptr(ids), _ptr(embed), _ptr(img_feats), _ptr(out), N, L, P, H, embed.shape[0],
                                               P * H if feats_stride_n is None else feats_stride_n, _stream()), "embed_splice")
    return out


def gather_rows(x, idx):
    out = torch.empty((idx.shape[0], x.shape[1]), device=x.device, dtype=BF16)
    if idx.shape[0] == 0:
        return out
    _lib.check(_lib.load().llmseg_gather_rows(_ptr(x), _ptr(idx), _ptr(out), idx.shape[0], x.shape[1], x.stride(0), _stream()),
               "gather_rows")
    return out


def upsample_maskpool(feat_cl, segs, g, S, want_aux=False):
    """feat_cl bf16 [g*g, C] channels-last, segs bf16 [K, S, S] -> pooled bf16 [K, C] (+ (pulled_back fp32 [K,g*g], wsum fp32 [K]))."""
    _req(feat_cl); _req(segs)
    assert feat_cl.is_contiguous() and segs.is_contiguous()
    K, Cc = segs.shape[0], feat_cl.shape[1]
    out = torch.empty((K, Cc), device=segs.device, dtype=BF16)
    pb = torch.empty((K, g * g), device=segs.device, dtype=torch.float32) if want_aux else None
    ws = torch.empty((K,), device=segs.device, dtype=torch.float32) if want_aux else None
    wn = torch.empty((K, g * g), device=segs.device, dtype=BF16)
    _lib.check(_lib.load().llmseg_upsample_maskpool(_ptr(feat_cl), _ptr(segs), _ptr(out), _ptr(wn), _ptr(pb), _ptr(ws), K, Cc, g, S, _stream()),
               "upsample_maskpool")
    return (out, pb, ws) if want_aux else out


def maskpool_batched(feat_cl, rows_per_img, row0, segs_list, g, S):
    """Mask pooling of several images with equal proposal count: feat_cl bf16 [B*rows_per_img, C] (the g*g patch rows of image b
    start at row b*rows_per_img + row0), segs_list[b] bf16 [K, S, S] -> pooled bf16 [B, K, C].  One pull-back launch per image,
    ONE strided-batched GEMM for all of them."""
    B, K, Cc = len(segs_list), segs_list[0].shape[0], feat_cl.shape[1]
    assert feat_cl.is_contiguous()
    wn = torch.empty((B, K, g * g), device=feat_cl.device, dtype=BF16)
    lib = _lib.load()
    for b, segs in enumerate(segs_list):
        _req(segs)
        assert segs.is_contiguous() and segs.shape == (K, S, S)
        _lib.check(lib.llmseg_mask_pullback(_ptr(segs), _ptr(wn[b]), None, None, K, g, S, _stream()), "mask_pullback")
    out = torch.empty((B, K, Cc), device=feat_cl.device, dtype=BF16)
    gemm_batched(wn, feat_cl[row0:], out, M=K, N=Cc, K=g * g, lda=g * g, ldw=Cc, ldc=Cc, batch=B, sA=K * g * g, sW=rows_per_img * Cc,
                 sC=K * Cc, out_f32=False, trans_w=True)
    return out


def cosine_scores(t, e):
    K, D = e.shape
    out = torch.empty((K,), device=e.device, dtype=torch.float32)
    _lib.check(_lib.load().llmseg_cosine_scores(_ptr(t), _ptr(e), _ptr(out), K, D, _stream()), "cosine_scores")
    return out


def align_reg_loss(e, t, gt_iou, pred_iou, gt_iop, tau=0.05, want_grads=False):
    """One item: e bf16 [K, D], t [D], gt_iou / gt_iop fp32 [K], pred_iou bf16 [K] -> out fp32[2] = (align KL, IoP regression) and
    optionally (d_e [K,D], d_t [D], d_pred [K]) fp32.  Batched: e [R, K, D], t [R, D], the K-vectors [R, K] -> out [R, 2] etc.,
    one launch for all R items."""
    batched = e.dim() == 3
    R = e.shape[0] if batched else 1
    K, D = e.shape[-2:]
    for x in (e, t, gt_iou, pred_iou, gt_iop):
        assert x.is_contiguous()
    assert t.numel() == R * D and gt_iou.numel() == R * K and pred_iou.numel() == R * K and gt_iop.numel() == R * K
    shp = (R,) if batched else ()
    out = torch.empty(shp + (2,), device=e.device, dtype=torch.float32)
    d_e = torch.empty(shp + (K, D), device=e.device, dtype=torch.float32) if want_grads else None
    d_t = torch.empty(shp + (D,), device=e.device, dtype=torch.float32) if want_grads else None
    d_p = torch.empty(shp + (K,), device=e.device, dtype=torch.float32) if want_grads else None
    _lib.check(_lib.load().llmseg_align_reg_loss(_ptr(e), _ptr(t), _ptr(gt_iou), _ptr(pred_iou), _ptr(gt_iop), _ptr(out), _ptr(d_e),
                                                 _ptr(d_t), _ptr(d_p), K, D, tau, R, _stream()), "align_reg_loss")
    return (out, d_e, d_t, d_p) if want_grads else out


def dice_bce(logits, targets, num_masks):
    M = logits.shape[0]
    HW = logits[0].numel()
    out = torch.zeros((2,), device=logits.device, dtype=torch.float32)
    _lib.check(_lib.load().llmseg_dice_bce(_ptr(logits), _ptr(targets), _ptr(out), M, HW, float(num_masks), *_reduce_ws(logits.device), _stream()), "dice_bce")
    return out


def dice_bce_bwd(logits, targets, g, num_masks):
    """g fp32[2] = upstream gradients of (dice, bce) -> d logits (fp32, same shape)."""
    M = logits.shape[0]
    HW = logits[0].numel()
    dx = torch.empty_like(logits)
    _lib.check(_lib.load().llmseg_dice_bce_bwd(_ptr(logits), _ptr(targets), _ptr(g), _ptr(dx), M, HW, float(num_masks), _stream()), "dice_bce_bwd")
    return dx


def ce_loss(logits, labels):
    """logits bf16 [N,T,V(ld)], labels int64 [N,T] (spliced) -> fp32[2] = (sum nll, count)."""
    N, T, V = logits.shape
    acc = torch.zeros((2,), device=logits.device, dtype=torch.float32)
    _lib.check(_lib.load().llmseg_ce_loss(_ptr(logits), _ptr(labels), _ptr(acc), N, T, V, logits.stride(1), *_reduce_ws(logits.device), _stream()), "ce_loss")
    return acc


def intersection_union(pred_u8, target_u8, ignore_index=255, out=None):
    """-> int64[6] = {I0, I1, U0, U1, T0, T1} accumulated into `out` (validate_threshold's per-image I/U, training.py:764)."""
    assert pred_u8.dtype == torch.uint8 and target_u8.dtype == torch.uint8 and pred_u8.numel() == target_u8.numel()
    if out is None:
        out = torch.zeros((6,), device=pred_u8.device, dtype=torch.int64)
    _lib.check(_lib.load().llmseg_intersection_union(_ptr(pred_u8.contiguous()), _ptr(target_u8.contiguous()), pred_u8.numel(), ignore_index,
                                                     _ptr(out), _stream()), "intersection_union")
    return out


def union_resize_iou(segs_hwk_u8, select_u8, gt_u8, out_size=1024, ignore_index=255, out=None):
    """Union of selected proposals (segs [H,W,K] uint8) + nearest resize of it and of gt [Hg,Wg] to out_size (an int: square; a (h, w)
    pair; or None = the ground truth's own size) + I/U -> int64[6]."""
    H, W, K = segs_hwk_u8.shape
    Hg, Wg = gt_u8.shape
    oh, ow = (Hg, Wg) if out_size is None else ((out_size, out_size) if isinstance(out_size, int) else out_size)
    if out is None:
        out = torch.zeros((6,), device=segs_hwk_u8.device, dtype=torch.int64)
    _lib.check(_lib.load().llmseg_union_resize_iou(_ptr(segs_hwk_u8.contiguous()), _ptr(select_u8.contiguous()), _ptr(gt_u8.contiguous()), H, W, K,
                                                   Hg, Wg, oh, ow, ignore_index, _ptr(out), _stream()), "union_resize_iou")
    return out


# ---- backward / optimizer -------------------------------------------------------------------------------------------
def colsum(x, out=None):
    M, N = x.shape
    if out is None:
        out = torch.zeros((N,), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().llmseg_colsum(_ptr(x), _ptr(out), M, N, x.stride(0), *_reduce_ws(x.device), _stream()), "colsum")
    return out


def norm_bwd(dy, x, w, eps, rms, dw=None, db=None, dres=None):
    """dres: gradient reaching x through a residual connection, added to dx in the same pass."""
    rows, cols = x.shape
    assert dy.is_contiguous() and x.is_contiguous() and (dres is None or (dres.is_contiguous() and dres.shape == x.shape))
    dx = torch.empty_like(x)
    _lib.check(_lib.load().llmseg_norm_bwd_add(_ptr(dy), _ptr(x), _ptr(w), _ptr(dres), _ptr(dx), _ptr(dw), _ptr(db), rows, cols, eps, 1 if rms else 0,
                                               *_reduce_ws(x.device), _stream()), "norm_bwd")
    return dx


def swiglu_bwd(gu, dout, inter):
    assert gu.is_contiguous() and dout.is_contiguous()
    dgu = torch.empty_like(gu)
    _lib.check(_lib.load().llmseg_swiglu_bwd(_ptr(gu), _ptr(dout), _ptr(dgu), gu.shape[0], inter, _stream()), "swiglu_bwd")
    return dgu


def act_bwd(dy, y, act):
    assert dy.is_contiguous() and y.is_contiguous()
    out = torch.empty_like(dy)
    _lib.check(_lib.load().llmseg_act_bwd(_ptr(dy), _ptr(y), _ptr(out), dy.numel(), act, _stream()), "act_bwd")
    return out


def softmax_rows(S, BH, Tq, Tk, ld, scale, causal=False, key_mask=None, heads=1):
    P = torch.empty((BH, Tq, ld), device=S.device, dtype=BF16)
    _lib.check(_lib.load().llmseg_softmax_rows(_ptr(S), _ptr(P), BH, Tq, Tk, ld, scale, 1 if causal else 0, _ptr(key_mask), heads, _stream()),
               "softmax_rows")
    return P


def attn_ds(P, dP, T, ld, scale):
    dS = torch.empty_like(P)
    _lib.check(_lib.load().llmseg_attn_ds(_ptr(P), _ptr(dP), _ptr(dS), P.shape[0] * P.shape[1], T, ld, scale, _stream()), "attn_ds")
    return dS


def ce_bwd(logits, labels, coef):
    N, T, V = logits.shape
    dl = torch.empty_like(logits)
    _lib.check(_lib.load().llmseg_ce_bwd(_ptr(logits), _ptr(labels), _ptr(coef), _ptr(dl), N, T, V, logits.stride(1), _stream()), "ce_bwd")
    return dl


def scatter_add_rows(src, idx, dst):
    _lib.check(_lib.load().llmseg_scatter_add_rows(_ptr(src), _ptr(idx), _ptr(dst), src.shape[0], src.shape[1], _stream()), "scatter_add_rows")
    return dst


def _drop(drop):
    """drop = (rng_state int64[2] device tensor, stream id, p[, seg_rows]) or None -> ctypes pointer (or None).  seg_rows: rows per
    segment of a fused accumulation window (`llmseg_dropout.seg_rows`; 0 / absent = one segment)."""
    if drop is None or drop[2] <= 0.0:
        return None
    rng, stream, p = drop[:3]
    seg = int(drop[3]) if len(drop) > 3 else 0
    assert rng.is_cuda and rng.dtype == torch.int64 and rng.numel() == 2
    return C.byref(_lib.Dropout(rng_state=rng.data_ptr(), stream=int(stream), drop_thr=int(round(p * 65536)), seg_rows=seg, reserved0=0))


def lora_down(x, w, w_kr=False, alpha=1.0, out=None, zero_cols=0, drop=None, x2=None, w2=None, pack=None, parts=False):
    """y [M, 8] = alpha * drop(x) [M, K] @ W^T; W stored [8, K] (or [K, 8] when w_kr).  x may be a column view (row stride = ld).
    With (x2, w2) the second branch lands in columns 8..15 of the same rows (its dropout stream is drop's + 1; x2 may be x).  `out`
    may be the leading columns of a wider row; `zero_cols` further columns of every row are zero-filled.
    pack = (aq, bq, av, bv, s, w2b, w2a, bt): `lora_pack` of the same layer in the same call (`llmseg_lora_down_pack`: it rides in the K-slice finish launch).
    parts=True (`llmseg_lora_down_parts`): -> (y, partials | None, S, scale): where the product runs as K slices it is left UNFINISHED (y not written; hand
    (partials, S, scale, zero_cols) to `gemm(..., nb_lora=)`, whose tail finishes it and writes y); S = 0: y is complete."""
    M, K = x.shape
    nb = 1 if w2 is None else 2
    y = torch.empty((M, 8 * nb), device=x.device, dtype=BF16) if out is None else out
    assert y.shape[0] == M and y.stride(1) == 1 and (x2 is None or (x2.shape == x.shape and x2.stride(0) == x.stride(0)))
    scratch = None
    if not w_kr and (M + 15) // 16 < 256 and K % 256 == 0:       # short activations: K-sliced over several workgroups per row tile (fp32 partials)
        scratch = torch.empty((32 * 2 * M * 16,), device=x.device, dtype=torch.float32)
    args = (_ptr(x), _ptr(x2 if w2 is not None and x2 is not None else (x if w2 is not None else None)), x.stride(0),
            _ptr(w), _ptr(w2), _ptr(y), y.stride(0), M, K, 1 if w_kr else 0, alpha, zero_cols, _drop(drop),
            _ptr(scratch), 0 if scratch is None else scratch.numel() * 4)
    if parts:
        assert pack is None
        S_out, sc_out = C.c_int32(0), C.c_float(0.0)
        _lib.check(_lib.load().llmseg_lora_down_parts(*args, C.byref(S_out), C.byref(sc_out), _stream()), "lora_down_parts")
        return y, (scratch if S_out.value > 0 else None), S_out.value, sc_out.value
    if pack is None:
        _lib.check(_lib.load().llmseg_lora_down_ws(*args, _stream()), "lora_down")
    else:
        aq, bq, av, bv, ps, w2b, w2a, bt = pack
        for t in (aq, bq, av, bv):
            assert t.is_contiguous()
        _lib.check(_lib.load().llmseg_lora_down_pack(*args, _ptr(aq), _ptr(bq), _ptr(av), _ptr(bv), _ptr(w2b), _ptr(w2a), _ptr(bt), aq.shape[1], ps, _stream()),
                   "lora_down_pack")
    return y


def lora_outer(a, b, out_rn=False, alpha=1.0, out=None, drop=None, a2=None, b2=None, out2=None):
    """out [N, 8] (or [8, N] when out_rn) += alpha * drop(a)[M, N]^T @ b[M, 8], fp32 (a fresh zero buffer unless `out` is given).
    With (a2, b2) a second, independent product of the same shapes runs in the same launch (dropout stream + 1) -> (out, out2)."""
    M, N = a.shape
    shape = (8, N) if out_rn else (N, 8)
    if out is None:
        out = torch.zeros(shape, device=a.device, dtype=torch.float32)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == 8 * N and b.stride(1) == 1
    if b2 is not None:
        a2 = a if a2 is None else a2
        if out2 is None:
            out2 = torch.zeros(shape, device=a.device, dtype=torch.float32)
        assert a2.shape == a.shape and a2.stride(0) == a.stride(0) and b2.stride(0) == b.stride(0) and out2.is_contiguous() and out2.numel() == 8 * N
    _lib.check(_lib.load().llmseg_lora_outer(_ptr(a), _ptr(a2 if b2 is not None else None), a.stride(0), _ptr(b), _ptr(b2), b.stride(0), _ptr(out),
                                             _ptr(out2 if b2 is not None else None), M, N, 1 if out_rn else 0, alpha, _drop(drop), *_reduce_ws(a.device),
                                             _stream()), "lora_outer")
    return out if b2 is None else (out, out2)


def lora_wgrads(d, H, x, a2, t2, gbq, gbv, gaq, gav, s, drop=None):
    """The four LoRA weight gradients of a q|k|v projection in one launch: d = dqkv [M, 3H] (q block = columns 0..H, v block = 2H..3H),
    x [M, H] the projection's input, a2 [M, >= 16] = [drop_q(x) Aq^T | drop_v(x) Av^T], t2 [M, >= 16] = [s dq Bq | s dv Bv];
    gbq / gbv fp32 [H, 8] and gaq / gav fp32 [8, H] are accumulated into (arena views)."""
    M = d.shape[0]
    for g in (gbq, gbv, gaq, gav):
        assert g.dtype == torch.float32 and g.is_contiguous() and g.numel() == 8 * H
    assert d.stride(1) == 1 and x.stride(1) == 1 and a2.stride(1) == 1 and t2.stride(1) == 1 and x.shape == (M, H)
    dq, dv = d[:, :H], d[:, 2 * H:]
    _lib.check(_lib.load().llmseg_lora_wgrads(_ptr(dq), _ptr(dv), d.stride(0), _ptr(x), x.stride(0), _ptr(a2), a2.stride(0), _ptr(t2), t2.stride(0),
                                              _ptr(gbq), _ptr(gbv), _ptr(gaq), _ptr(gav), M, H, s, _drop(drop), *_reduce_ws(d.device), _stream()), "lora_wgrads")


def lora_apply_(y, xa, w, w_rn=False, alpha=1.0, drop=None, w2=None):
    """y [M, N] += alpha * mask * (xa [M, 8] @ W^T) in place; W stored [N, 8] (or [8, N] when w_rn).  y may be a column view.
    With w2: + alpha * mask2 * (xa[:, 8:16] @ W2^T) in the same pass (dropout stream + 1)."""
    M, N = y.shape
    _lib.check(_lib.load().llmseg_lora_apply(_ptr(y), y.stride(0), _ptr(xa), xa.stride(0), _ptr(w), _ptr(w2), M, N, 1 if w_rn else 0, alpha, _drop(drop),
                                             _stream()), "lora_apply")
    return y


def lora_pack(aq, bq, av, bv, s, w2b=None, w2a=None, bt=None):
    """Extension operands of the LoRA'd qkv GEMMs from the current LoRA matrices (aq/av [8, H], bq/bv [H, 8]):
    w2b [3H, 64] (forward) and/or w2a [H, 64] (backward dX) and/or bt [16, H] = Bq^T | Bv^T; written in place."""
    H = aq.shape[1]
    for t in (aq, bq, av, bv):
        assert t.is_contiguous()
    _lib.check(_lib.load().llmseg_lora_pack(_ptr(aq), _ptr(bq), _ptr(av), _ptr(bv), _ptr(w2b), _ptr(w2a), _ptr(bt), H, s, _stream()), "lora_pack")


def sumsq(x, out):
    _lib.check(_lib.load().llmseg_sumsq(_ptr(x), x.numel(), 1 if x.dtype == torch.float32 else 0, _ptr(out), *_reduce_ws(x.device), _stream()), "sumsq")
    return out


def adamw_(p, master, grad, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=None):
    _lib.check(_lib.load().llmseg_adamw(_ptr(p), _ptr(master), _ptr(grad), 1 if grad.dtype == torch.float32 else 0, _ptr(m), _ptr(v), p.numel(),
                                        lr, beta1, beta2, eps, wd, step, _ptr(grad_scale), _stream()), "adamw")


def transpose_pad(x, rows_pad=None, out=None):
    """x bf16 [R, C] (row stride any multiple of 8) -> out [C, rows_pad]: out[c][r] = x[r][c], zeros for R <= r < rows_pad.
    `out` may be wider than rows_pad (its remaining columns are left untouched)."""
    _req(x)
    R, Cc = x.shape
    rows_pad = (R + 7) // 8 * 8 if rows_pad is None else rows_pad
    assert x.stride(1) == 1 and rows_pad >= R and rows_pad % 8 == 0
    if out is None:
        out = torch.empty((Cc, rows_pad), device=x.device, dtype=BF16)
    assert out.shape[0] == Cc and out.shape[1] >= rows_pad and out.stride(1) == 1
    _lib.check(_lib.load().llmseg_transpose_pad(_ptr(x), _ptr(out), R, Cc, x.stride(0), out.stride(0), rows_pad, _stream()), "transpose_pad")
    return out


def prof_enable(on):
    _lib.load().llmseg_prof_enable(1 if on else 0)


def prof_collect():
    """-> dict(all=(ms, flops, launches), dominant=(ms, flops, launches), dominant_kernel=name) of the GEMM launches since
    prof_enable(True); dominant = the GEMM kernel class with the largest total time."""
    ms, fl, n, dms, dfl, dn = C.c_double(), C.c_double(), C.c_int64(), C.c_double(), C.c_double(), C.c_int64()
    _lib.load().llmseg_prof_collect(C.byref(ms), C.byref(fl), C.byref(n), C.byref(dms), C.byref(dfl), C.byref(dn))
    name = _lib.load().llmseg_prof_dominant_kernel()
    info = (C.c_int64 * 4)()
    _lib.load().llmseg_prof_dominant_info(info)
    return {"all": (ms.value, fl.value, n.value), "dominant": (dms.value, dfl.value, dn.value), "dominant_kernel": (name or b"").decode(),
            "dominant_alg_bytes": _lib.load().llmseg_prof_dominant_bytes(),
            "dominant_info": {"calls": int(info[0]), "calls_as_k_slices": int(info[1]), "k_slices": int(info[2]), "kernel_launches": int(info[3])}}


# ---- fp32-activation head (inference scores): csrc/head_f32.hip ------------------------------------------------------------------------------
F32 = torch.float32


def linear_f32(x, w, bias=None, act=ACT_NONE, residual=None, w_kn=False, alpha=1.0):
    """act(alpha * x @ W^T + bias) + residual with fp32 activations: x fp32 [M, K] (last dim contiguous), w bf16 [N, K] (or [K, N] when w_kn),
    bias bf16 [N], residual fp32 [M, N] -> fp32 [M, N]."""
    _req(x, F32); _req(w)
    assert x.dim() == 2 and w.dim() == 2 and x.stride(1) == 1 and w.stride(1) == 1
    M, K = x.shape
    N = w.shape[1] if w_kn else w.shape[0]
    assert (w.shape[0] if w_kn else w.shape[1]) == K, (x.shape, w.shape, w_kn)
    if bias is not None:
        _req(bias); assert bias.numel() == N and bias.is_contiguous()
    if residual is not None:
        _req(residual, F32); assert residual.shape == (M, N) and residual.stride(1) == 1
    y = torch.empty((M, N), device=x.device, dtype=F32)
    _lib.check(_lib.load().llmseg_linear_f32(_ptr(x), x.stride(0), _ptr(w), w.stride(0), 1 if w_kn else 0, _ptr(bias), _ptr(residual),
                                             residual.stride(0) if residual is not None else 0, _ptr(y), N, M, N, K, act, float(alpha), _stream()), "linear_f32")
    return y


def layernorm_f32(x, w, b=None, eps=1e-5):
    _req(x, F32); _req(w)
    assert x.is_contiguous() and x.dim() == 2 and w.numel() == x.shape[1]
    y = torch.empty_like(x)
    _lib.check(_lib.load().llmseg_layernorm_f32(_ptr(x), _ptr(w), _ptr(b), _ptr(y), x.shape[0], x.shape[1], float(eps), _stream()), "layernorm_f32")
    return y


def attention_f32(q, k, v, o, batch, heads, Nq, Nk, head_dim, q_strides, k_strides, v_strides, o_strides, scale=None):
    """softmax(q k^T * scale) v, all operands fp32; *_strides = (batch, head, row) in elements."""
    for t in (q, k, v, o):
        _req(t, F32)
    st = (C.c_int64 * 12)(*[int(s) for tup in (q_strides, k_strides, v_strides, o_strides) for s in tup])
    _lib.check(_lib.load().llmseg_attn_f32(_ptr(q), _ptr(k), _ptr(v), _ptr(o), st, batch, heads, Nq, Nk, head_dim,
                                           float(scale if scale is not None else 1.0 / math.sqrt(head_dim)), _stream()), "attn_f32")
    return o


def cosine_f32(t, e):
    _req(t, F32); _req(e, F32)
    assert t.is_contiguous() and e.is_contiguous() and e.dim() == 2 and t.numel() == e.shape[1]
    out = torch.empty((e.shape[0],), device=e.device, dtype=F32)
    _lib.check(_lib.load().llmseg_cosine_f32(_ptr(t), _ptr(e), _ptr(out), e.shape[0], e.shape[1], _stream()), "cosine_f32")
    return out


def mask_pullback_f32(segs, g, S):
    """segs bf16 [K, S, S] -> (pulled_back fp32 [K, g*g] = segs . U, the proposals pulled back through the adjoint of the bilinear upsampling;
    wsum fp32 [K] = sum of each proposal's pixels): stage 1 of the mask pooling with its fp32 outputs, no GEMM."""
    _req(segs)
    assert segs.is_contiguous()
    K = segs.shape[0]
    pb = torch.empty((K, g * g), device=segs.device, dtype=F32)
    ws = torch.empty((K,), device=segs.device, dtype=F32)
    wn = torch.empty((K, g * g), device=segs.device, dtype=BF16)
    _lib.check(_lib.load().llmseg_mask_pullback(_ptr(segs), _ptr(wn), _ptr(pb), _ptr(ws), K, g, S, _stream()), "mask_pullback")
    return pb, ws
