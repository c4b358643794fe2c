"""Backward-pass parity: HIP backward kernels / autograd Functions vs torch autograd on fp32 CPU copies, and the whole
model's parameter gradients vs the oracle (and the reference-generated gradient fixture)."""
import math

import torch
import torch.nn.functional as F

from llmseg_amd import autograd as ag
from llmseg_amd import ops
from tests.kernel_checks import BF, DEV, err, rnd


def rel_tol(ref, k=2.0 ** -6):
    return k * max(1e-6, ref.float().abs().max().item())


def check_gemm_layouts():
    out = []
    M, N, K = 304, 264, 200
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / 14)
    ref = a.float() @ w.float().t()
    ad, wd = a.to(DEV), w.to(DEV)
    at, wt = a.t().contiguous().to(DEV), w.t().contiguous().to(DEV)      # stored [K, M] / [K, N]
    out.append(("gemm trans_w", err(ops.gemm(ad, wt, trans_w=True), ref), rel_tol(ref)))
    out.append(("gemm trans_a", err(ops.gemm(at, wd, trans_a=True), ref), rel_tol(ref)))
    out.append(("gemm trans_a+trans_w", err(ops.gemm(at, wt, trans_a=True, trans_w=True), ref), rel_tol(ref)))
    # odd contraction length with both operands transposed (attention backward: K' = 319)
    M, N, K = 319, 128, 319
    a, w = rnd(K, 320, seed=3), rnd(K, N, seed=4, scale=1 / 16)
    ref = a[:, :M].float().t() @ w.float()
    got = torch.empty(M, N, device=DEV, dtype=BF)
    ops.gemm_batched(a.to(DEV), w.to(DEV), got, M=M, N=N, K=K, lda=320, ldw=N, ldc=N, batch=1, sA=0, sW=0, sC=0, out_f32=False,
                     trans_a=True, trans_w=True)
    out.append(("gemm trans both K=319", err(got, ref), rel_tol(ref)))
    # K = 319 with a zero-padded K-contiguous A (dQ = dS K)
    a = rnd(64, 320, seed=5)
    a[:, 319] = 0
    w = rnd(319, 128, seed=6, scale=1 / 16)
    ref = a[:, :319].float() @ w.float()
    got = torch.empty(64, 128, device=DEV, dtype=BF)
    ops.gemm_batched(a.to(DEV), w.to(DEV), got, M=64, N=128, K=319, lda=320, ldw=128, ldc=128, batch=1, sA=0, sW=0, sC=0, out_f32=False,
                     trans_w=True)
    out.append(("gemm K=319 padded A, trans_w", err(got, ref), rel_tol(ref)))
    # 2-D batch
    b1, b2, M, N, K = 3, 2, 70, 40, 64
    a, w = rnd(b2, b1, M, K, seed=7), rnd(b2, b1, N, K, seed=8, scale=1 / 8)
    ref = a.float() @ w.float().transpose(-1, -2)
    got = torch.empty(b2, b1, M, N, device=DEV, dtype=torch.float32)
    ops.gemm_batched(a.to(DEV), w.to(DEV), got, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, batch=b1, sA=M * K, sW=N * K, sC=M * N, batch2=b2,
                     sA2=b1 * M * K, sW2=b1 * N * K, sC2=b1 * M * N, out_f32=True)
    out.append(("gemm 2-D batch", err(got, ref), 1e-3))
    return out


def _grads(fn_ref, fn_hip, inputs, names):
    """inputs: list of bf16 CPU tensors.  Returns [(name, err, tol)] for the output and every input gradient."""
    res = []
    xs = [t.float().clone().requires_grad_(True) for t in inputs]
    y = fn_ref(*xs)
    gen = torch.Generator().manual_seed(99)
    gy = torch.randn(y.shape, generator=gen).to(BF)
    y.backward(gy.float())
    xd = [t.to(DEV).clone().requires_grad_(True) for t in inputs]
    yd = fn_hip(*xd)
    yd.backward(gy.to(DEV))
    res.append((names[0] + " fwd", err(yd, y), rel_tol(y)))
    for n, a, b in zip(names[1:], xd, xs):
        if b.grad is not None:
            res.append((f"{names[0]} d{n}", err(a.grad, b.grad), rel_tol(b.grad, 2.0 ** -5)))
    return res


def check_autograd_ops():
    out = []
    x, w, b, r = rnd(200, 256, seed=1), rnd(264, 256, seed=2, scale=1 / 16), rnd(264, seed=3), rnd(200, 264, seed=4)
    out += _grads(lambda x, w, b, r: F.linear(x, w, b) + r, lambda x, w, b, r: ag.linear(x, w, b, ops.ACT_NONE, r), [x, w, b, r],
                  ["linear+res", "x", "w", "b", "res"])
    out += _grads(lambda x, w, b: F.relu(F.linear(x, w, b)), lambda x, w, b: ag.linear(x, w, b, ops.ACT_RELU), [x, w, b], ["linear relu", "x", "w", "b"])
    # the wide-Linear path (lm_head): operands transposed + padded to multiples of 64, forced here at a small size; N = 260 is
    # neither a multiple of 8 nor of 64, M = 200 pads to 256
    ag.BIG_LINEAR, keep = 0, ag.BIG_LINEAR
    try:
        wl = rnd(260, 256, seed=31, scale=1 / 16)
        out += _grads(lambda x, w: F.linear(x, w), lambda x, w: ag.linear(x, w), [x, wl], ["linear wide-path", "x", "w"])
    finally:
        ag.BIG_LINEAR = keep
    w1, b1 = rnd(1, 256, seed=5, scale=1 / 16), rnd(1, seed=6)
    out += _grads(lambda x, w, b: torch.sigmoid(F.linear(x, w, b)), lambda x, w, b: ag.linear(x, w, b, ops.ACT_SIGMOID), [x, w1, b1],
                  ["linear N=1 sigmoid", "x", "w", "b"])
    g, bb = rnd(256, seed=7), rnd(256, seed=8)
    out += _grads(lambda x, g, b: F.layer_norm(x, (256,), g, b, 1e-5), lambda x, g, b: ag.norm(x, g, b, 1e-5, False), [x, g, bb], ["layernorm", "x", "w", "b"])
    out += _grads(lambda x, g: g * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)), lambda x, g: ag.norm(x, g, None, 1e-6, True), [x, g],
                  ["rmsnorm", "x", "w"])
    gu = rnd(100, 2 * 128, seed=9)
    out += _grads(lambda t: F.silu(t[:, :128]) * t[:, 128:], lambda t: ag.SwigluFn.apply(t, 128), [gu], ["swiglu", "gu"])
    # packed causal attention with a key mask (Llama form), hd = 128, T = 37 (odd: exercises the padded P buffers)
    Nn, T, nh, hd = 2, 37, 2, 128
    qkv = rnd(Nn * T, 3 * nh * hd, seed=10, scale=0.5)
    km = torch.ones(Nn, T, dtype=torch.uint8)
    km[1, 30:] = 0

    def attn_ref(t):
        x = t.view(Nn, T, 3, nh, hd).permute(2, 0, 3, 1, 4)
        s = (x[0] @ x[1].transpose(-1, -2)) / math.sqrt(hd)
        m = torch.ones(T, T, dtype=torch.bool).tril()[None, None] & (km != 0)[:, None, None, :]
        p = torch.softmax(s.masked_fill(~m, -1e30), -1)
        return (p @ x[2]).transpose(1, 2).reshape(Nn * T, nh * hd)
    out += _grads(attn_ref, lambda t: ag.PackedAttnFn.apply(t, Nn, T, nh, hd, True, km.to(DEV)), [qkv], ["attn causal", "qkv"])
    # head shapes: self attention hd 32 and the 1-query cross attention
    Cn, K, nh, hd = 2, 48, 8, 32
    D = nh * hd
    qkv = rnd(Cn * K, 3 * D, seed=11, scale=0.5)

    def sa_ref(t):
        x = t.view(Cn, K, 3, nh, hd).permute(2, 0, 3, 1, 4)
        p = torch.softmax((x[0] @ x[1].transpose(-1, -2)) / math.sqrt(hd), -1)
        return (p @ x[2]).transpose(1, 2).reshape(Cn * K, D)
    out += _grads(sa_ref, lambda t: ag.PackedAttnFn.apply(t, Cn, K, nh, hd, False, None), [qkv], ["attn head self", "qkv"])
    # the bench shape of the fused backward: T = 319 (three 128-row owner blocks, five 64-row tiles, causal tile skipping) with
    # right padding, and a plain hd = 64 case with ragged 200 rows
    for (fN, fT, fh, fd, causal, pad) in ((2, 319, 2, 128, True, 300), (1, 200, 2, 64, False, None), (2, 130, 1, 32, True, None)):
        qkv = rnd(fN * fT, 3 * fh * fd, seed=20 + fd, scale=0.5)
        km = torch.ones(fN, fT, dtype=torch.uint8)
        if pad:
            km[1, pad:] = 0

        def ref(t, Nn=fN, T=fT, nh=fh, hd=fd, causal=causal, km=km):
            x = t.view(Nn, T, 3, nh, hd).permute(2, 0, 3, 1, 4)
            s = (x[0] @ x[1].transpose(-1, -2)) / math.sqrt(hd)
            m = (km != 0)[:, None, None, :].expand(Nn, 1, T, T)
            if causal:
                m = m & torch.ones(T, T, dtype=torch.bool).tril()[None, None]
            p = torch.softmax(s.masked_fill(~m, -1e30), -1)
            return (p @ x[2]).transpose(1, 2).reshape(Nn * T, nh * hd)
        kmd = km.to(DEV) if pad else None
        out += _grads(ref, lambda t, a=(fN, fT, fh, fd, causal, kmd): ag.PackedAttnFn.apply(t, *a), [qkv],
                      [f"attn fused T={fT} hd={fd}", "qkv"])
    q, kv = rnd(Cn, D, seed=12), rnd(Cn * K, 2 * D, seed=13, scale=0.5)

    def ca_ref(q, kv):
        qq = q.view(Cn, 1, nh, hd).transpose(1, 2)
        kk = kv[:, :D].view(Cn, K, nh, hd).transpose(1, 2)
        vv = kv[:, D:].view(Cn, K, nh, hd).transpose(1, 2)
        p = torch.softmax((qq @ kk.transpose(-1, -2)) / math.sqrt(hd), -1)
        return (p @ vv).transpose(1, 2).reshape(Cn, D)
    out += _grads(ca_ref, lambda q, kv: ag.CrossAttn1QFn.apply(q, kv, Cn, K, nh, hd), [q, kv], ["attn head 1q", "q", "kv"])
    # RoPE + causal attention as ONE autograd node (the inverse rotation runs on the node's own gradient buffer)
    T, nh, hd, N = 19, 2, 128, 2
    qkv = rnd(N * T, 3 * nh * hd, seed=14, scale=0.5)
    inv = 1.0 / (10000 ** (torch.arange(0, hd, 2).float() / hd))
    ang = torch.outer(torch.arange(T).float(), inv)
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()

    def rope_attn_ref(t):
        x = t.view(N, T, 3, nh, hd)
        c2, s2 = torch.cat([cos, cos], -1)[None, :, None], torch.cat([sin, sin], -1)[None, :, None]
        rot = lambda u: torch.cat([-u[..., hd // 2:], u[..., :hd // 2]], -1)
        q, k, v = (x[:, :, 0] * c2 + rot(x[:, :, 0]) * s2).transpose(1, 2), (x[:, :, 1] * c2 + rot(x[:, :, 1]) * s2).transpose(1, 2), x[:, :, 2].transpose(1, 2)
        sc = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
        p = torch.softmax(sc.masked_fill(~torch.ones(T, T, dtype=torch.bool).tril()[None, None], -1e30), -1)
        return (p @ v).transpose(1, 2).reshape(N * T, nh * hd)
    rope_tabs = (cos.to(DEV), sin.to(DEV), (-sin).to(DEV))
    out += _grads(rope_attn_ref, lambda t: ag.rope_attention(t.clone(), rope_tabs, N, T, nh, hd, True, None), [qkv], ["rope+attn", "qkv"])
    # round 6: the inverse rotation rides in the attention backward's dq / dk store == a separate llmseg_rope launch over the stored
    # gradient, BIT FOR BIT (the benchmark's T = 319 with right padding; head_dim 128 and 64)
    for hd2 in (128, 64):
        T2, N2, nh2 = 319, 2, 2
        qkv2 = rnd(N2 * T2, 3 * nh2 * hd2, seed=33 + hd2, scale=0.5).to(DEV)
        go = rnd(N2 * T2, nh2 * hd2, seed=34, scale=0.1).to(DEV)
        km2 = torch.ones(N2, T2, dtype=torch.uint8)
        km2[1, 300:] = 0
        ang2 = torch.outer(torch.arange(T2).float(), 1.0 / (10000 ** (torch.arange(0, hd2, 2).float() / hd2)))
        tabs2 = (ang2.cos().contiguous().to(DEV), ang2.sin().contiguous().to(DEV), (-ang2.sin()).contiguous().to(DEV))
        keep = ag.FUSE_ROPE_BWD
        gs = []
        try:
            for fuse in (True, False):
                ag.FUSE_ROPE_BWD = fuse
                t = qkv2.clone().requires_grad_(True)
                ag.rope_attention(t.clone(), tabs2, N2, T2, nh2, hd2, True, km2.to(DEV)).backward(go)
                gs.append(t.grad.float().cpu())
        finally:
            ag.FUSE_ROPE_BWD = keep
        out.append((f"rope inside attn-bwd store == rope launch, hd={hd2} (bits)", float((gs[0] - gs[1]).abs().max()), 0.0))
    # CE
    Nn, T, V = 2, 7, 1000
    logits = rnd(Nn, T, V, seed=15, scale=2.0)
    labels = torch.randint(0, V, (Nn, T), generator=torch.Generator().manual_seed(1))
    labels[:, :2] = -100
    out += _grads(lambda l: F.cross_entropy(l[:, :-1].reshape(-1, V), labels[:, 1:].reshape(-1), ignore_index=-100),
                  lambda l: ag.CELossFn.apply(l, labels.to(DEV)), [logits], ["shifted CE", "logits"])
    # LoRA-fused qkv
    H, r, M = 256, 8, 50
    x, wq = rnd(M, H, seed=16), rnd(3 * H, H, seed=17, scale=1 / 16)
    aq, bq, av, bv = rnd(r, H, seed=18, scale=1 / 16), rnd(H, r, seed=19, scale=0.3), rnd(r, H, seed=20, scale=1 / 16), rnd(H, r, seed=21, scale=0.3)

    def lora_ref(x, aq, bq, av, bv):
        y = F.linear(x, wq.float())
        dq, dv = 2.0 * F.linear(F.linear(x, aq), bq), 2.0 * F.linear(F.linear(x, av), bv)
        return torch.cat([y[:, :H] + dq, y[:, H:2 * H], y[:, 2 * H:] + dv], 1)
    out += _grads(lora_ref, lambda x, aq, bq, av, bv: ag.LoraQKVFn.apply(x, wq.to(DEV), aq, bq, av, bv, 2.0), [x, aq, bq, av, bv],
                  ["lora qkv", "x", "Aq", "Bq", "Av", "Bv"])
    # embedding splice + gather
    Nn, L, P, Hd, V = 2, 9, 4, 64, 50
    ids = torch.randint(0, V, (Nn, L), generator=torch.Generator().manual_seed(2))
    ids[:, 2] = -200
    ids[0, 5] = ids[0, 6]                                                   # a repeated token: gradient accumulation
    emb, feats = rnd(V, Hd, seed=22), rnd(Nn, P, Hd, seed=23)

    def splice_ref(e):
        return torch.stack([torch.cat([e[ids[n, :2]], feats[n].float(), e[ids[n, 3:]]]) for n in range(Nn)])
    out += _grads(splice_ref, lambda e: ag.EmbedSpliceFn.apply(ids.to(DEV), e, feats.to(DEV), P, P * Hd), [emb], ["embed_splice", "embed"])
    idx = torch.tensor([3, 17, 0], dtype=torch.int64)
    xx = rnd(20, 64, seed=24)
    out += _grads(lambda t: t[idx], lambda t: ag.GatherRowsFn.apply(t, idx.to(DEV)), [xx], ["gather_rows", "x"])
    # mask pooling backward
    K, Cc, g, S = 8, 64, 16, 64
    feat = rnd(g * g, Cc, seed=25)
    segs = (torch.rand(K, S, S, generator=torch.Generator().manual_seed(3)) > 0.6).to(BF)

    def pool_ref(f):
        up = F.interpolate(f.t().reshape(1, Cc, g, g), size=(S, S), mode="bilinear", align_corners=False)[0]
        wv = segs.float().flatten(1)
        return (wv @ up.flatten(1).t()) / (wv.sum(-1, keepdim=True) + 1e-8)
    out += _grads(pool_ref, lambda f: ag.MaskPoolFn.apply(f, segs.to(DEV), g, S), [feat], ["maskpool", "feat"])
    # dice + BCE (model/loss.py:4-47) forward and backward, fp32 logits
    from oracle import losses as olosses
    xl = torch.randn(3, 40, 52, generator=torch.Generator().manual_seed(28)) * 2
    yl = (torch.rand(3, 40, 52, generator=torch.Generator().manual_seed(29)) > 0.6).float()
    xr = xl.clone().requires_grad_(True)
    ref = 0.7 * olosses.dice(xr, yl, 3) + 1.3 * olosses.sigmoid_ce(xr, yl, 3)
    ref.backward()
    xd = xl.to(DEV).requires_grad_(True)
    o = ag.DiceBceFn.apply(xd, yl.to(DEV), 3.0)
    (0.7 * o[0] + 1.3 * o[1]).backward()
    out.append(("dice+bce fwd", abs(float(0.7 * o[0] + 1.3 * o[1]) - float(ref)), 1e-4 * max(1.0, abs(float(ref)))))
    out.append(("dice+bce dlogits", err(xd.grad, xr.grad), 1e-4 * xr.grad.abs().max().item()))
    s, add = rnd(2 * 10, 64, seed=26), rnd(2, 64, seed=27)
    out += _grads(lambda s, a: s + a.repeat_interleave(10, 0), lambda s, a: ag.BcastAddFn.apply(s, a, 2, 10), [s, add], ["bcast_add", "s", "add"])
    return out


def _oracle_drop(x, stream, st, p):
    from oracle import dropout as odrop
    return odrop.apply(x, st[0], st[1], stream, p)


def check_lora_paths():
    """The LoRA'd qkv projection on the paths the training step really takes: rank-8 kernels + extension K-tile with the
    pre-transposed frozen weight (dX in ONE GEMM), the same under dropout (masked dX through lora_apply), gradients accumulated
    into fp32 arena views over two backward passes, at a shape that dispatches to the ping-pong GEMM."""
    out = []
    H, r, M, s_ = 512, 8, 700, 2.0
    x, wq = rnd(M, H, seed=16), rnd(3 * H, H, seed=17, scale=1 / 22)
    aq, bq, av, bv = rnd(r, H, seed=18, scale=1 / 22), rnd(H, r, seed=19, scale=0.3), rnd(r, H, seed=20, scale=1 / 22), rnd(H, r, seed=21, scale=0.3)
    wqd = wq.to(DEV)
    wqt = wqd.t().contiguous()
    st = (0xABCDEF12345, 3)
    rng = torch.tensor(list(st), device=DEV, dtype=torch.int64)
    for p_drop in (0.0, 0.25):
        def lora_ref(x, aq, bq, av, bv, p_drop=p_drop):
            y = F.linear(x, wq.float())
            xq = _oracle_drop(x, 2 * 5, st, p_drop) if p_drop > 0 else x
            xv = _oracle_drop(x, 2 * 5 + 1, st, p_drop) if p_drop > 0 else x
            dq, dv = s_ * F.linear(F.linear(xq, aq), bq), s_ * F.linear(F.linear(xv, av), bv)
            return torch.cat([y[:, :H] + dq, y[:, H:2 * H], y[:, 2 * H:] + dv], 1)
        drop = (rng, 5, p_drop) if p_drop > 0 else None
        tag = f"lora qkv ext-tile{' + dropout' if p_drop else ''}"
        out += _grads(lora_ref, lambda x, aq, bq, av, bv, drop=drop: ag.LoraQKVFn.apply(x, wqd, aq, bq, av, bv, s_, wqt, drop), [x, aq, bq, av, bv],
                      [tag, "x", "Aq", "Bq", "Av", "Bv"])
        # arena mode: two backward passes accumulate 2 x the gradient in fp32, nothing lands in .grad
        xs = [t.float().clone().requires_grad_(True) for t in (x, aq, bq, av, bv)]
        y = lora_ref(*xs)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).to(BF)
        y.backward(gy.float())
        prm = [t.to(DEV).clone().requires_grad_(True) for t in (aq, bq, av, bv)]
        for t in prm:
            t._g32 = torch.zeros(t.shape, device=DEV, dtype=torch.float32)
        for _ in range(2):
            xd = x.to(DEV).clone().requires_grad_(True)
            ag.LoraQKVFn.apply(xd, wqd, *prm, s_, wqt, drop).backward(gy.to(DEV))
        for n, t, ref in zip(("Aq", "Bq", "Av", "Bv"), prm, xs[1:]):
            assert t.grad is None
            out.append((f"{tag} arena d{n} (2 passes)", err(t._g32 / 2, ref.grad), rel_tol(ref.grad, 2.0 ** -6)))
    # the bit-exact mask: lora_down with all-ones weights row 0 counts the kept elements of every row
    Mm, Hh, p_drop = 37, 256, 0.05
    from oracle import dropout as odrop
    ones = torch.ones(Mm, Hh).to(BF)
    wsel = torch.zeros(8, Hh).to(BF)
    wsel[0] = 1
    got = ops.lora_down(ones.to(DEV), wsel.to(DEV), drop=(rng, 9, p_drop))[:, 0].float().cpu()
    keep = odrop.keep_mask(Mm, Hh, st[0], st[1], 9, p_drop)
    ref = keep.float().sum(1) * odrop.drop_scale(p_drop)
    out.append(("dropout mask: kept count per row vs the Philox oracle", (got - ref).abs().max().item(), 2.0 ** -8 * ref.max().item()))
    out.append(("dropout keep rate", abs(keep.float().mean().item() - (1 - p_drop)), 0.02))
    # exact positions: lora_apply adds mask * scale * 1 onto zeros
    yz = torch.zeros(Mm, Hh, device=DEV, dtype=BF)
    xa1, wn1 = torch.zeros(Mm, 8).to(BF), torch.zeros(Hh, 8).to(BF)
    xa1[:, 0] = 1
    wn1[:, 0] = 1
    ops.lora_apply_(yz, xa1.to(DEV), wn1.to(DEV), drop=(rng, 9, p_drop))
    out.append(("dropout mask positions (lora_apply) vs the Philox oracle: mismatching elements", float(((yz.cpu() != 0) != keep).sum()), 0.0))
    a_ones = torch.ones(Mm, Hh).to(BF)
    b_sel = torch.zeros(Mm, 8).to(BF)
    b_sel[:, 0] = 1
    colcnt = ops.lora_outer(a_ones.to(DEV), b_sel.to(DEV), drop=(rng, 9, p_drop))[:, 0].cpu()
    out.append(("dropout mask column counts (lora_outer)", (colcnt - keep.float().sum(0) * odrop.drop_scale(p_drop)).abs().max().item(), 1e-3))
    # segments (llmseg_dropout.seg_rows, ABI 5): rows [j seg, (j + 1) seg) draw the mask of their own pass at offset + j -- positions through
    # lora_apply, per-row counts through both lora_down kernels (per-row and MFMA), column counts through lora_outer, on a ragged last segment
    seg = 16
    keep_s = odrop.keep_mask(Mm, Hh, st[0], st[1], 9, p_drop, seg_rows=seg)
    ref_rows = torch.cat([odrop.keep_mask(min(seg, Mm - r0), Hh, st[0], st[1] + j, 9, p_drop) for j, r0 in enumerate(range(0, Mm, seg))], 0)
    assert torch.equal(keep_s, ref_rows) and not torch.equal(keep_s, keep)
    yz = torch.zeros(Mm, Hh, device=DEV, dtype=BF)
    ops.lora_apply_(yz, xa1.to(DEV), wn1.to(DEV), drop=(rng, 9, p_drop, seg))
    out.append(("segmented dropout mask positions (lora_apply) vs the oracle: mismatching elements", float(((yz.cpu() != 0) != keep_s).sum()), 0.0))
    ref_s = keep_s.float().sum(1) * odrop.drop_scale(p_drop)
    got = ops.lora_down(ones.to(DEV), wsel.to(DEV), drop=(rng, 9, p_drop, seg))[:, 0].float().cpu()
    out.append(("segmented dropout: kept count per row (lora_down)", (got - ref_s).abs().max().item(), 2.0 ** -8 * ref_s.max().item()))
    colcnt = ops.lora_outer(a_ones.to(DEV), b_sel.to(DEV), drop=(rng, 9, p_drop, seg))[:, 0].cpu()
    out.append(("segmented dropout: column counts (lora_outer)", (colcnt - keep_s.float().sum(0) * odrop.drop_scale(p_drop)).abs().max().item(), 1e-3))
    # a tall activation takes the other lora_down route (RW = 4 rows per wave / MFMA tile without K-slices): same rule
    Mt = 2100
    ones_t = torch.ones(Mt, Hh).to(BF)
    got = ops.lora_down(ones_t.to(DEV), wsel.to(DEV), drop=(rng, 9, p_drop, 700))[:, 0].float().cpu()
    ref_t = odrop.keep_mask(Mt, Hh, st[0], st[1], 9, p_drop, seg_rows=700).float().sum(1) * odrop.drop_scale(p_drop)
    out.append(("segmented dropout: kept count per row, tall activation (lora_down)", (got - ref_t).abs().max().item(), 2.0 ** -8 * ref_t.max().item()))
    return out


def check_arena_ops():
    """Parameter gradients accumulated by the kernels into fp32 `_g32` views (two passes = 2 x grad), per autograd Function."""
    out = []

    def run(name, fn_ref, fn_hip, acts, prms):
        """acts / prms: bf16 CPU tensors; fn_*(acts..., prms...)."""
        xs = [t.float().clone().requires_grad_(True) for t in acts + prms]
        y = fn_ref(*xs)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7)).to(BF)
        y.backward(gy.float())
        pd = [t.to(DEV).clone().requires_grad_(True) for t in prms]
        for t in pd:
            t._g32 = torch.full(t.shape, 0.0, device=DEV, dtype=torch.float32)
        for _ in range(2):
            ad = [t.to(DEV).clone().requires_grad_(True) for t in acts]
            fn_hip(*ad, *pd).backward(gy.to(DEV))
        for i, (t, ref) in enumerate(zip(pd, xs[len(acts):])):
            assert t.grad is None, name
            out.append((f"arena {name} dparam{i}", err(t._g32 / 2, ref.grad), rel_tol(ref.grad, 2.0 ** -6)))
        out.append((f"arena {name} dx", err(ad[0].grad, xs[0].grad), rel_tol(xs[0].grad, 2.0 ** -5)))

    x, w, b = rnd(200, 256, seed=1), rnd(264, 256, seed=2, scale=1 / 16), rnd(264, seed=3)
    run("linear", lambda x, w, b: F.linear(x, w, b), lambda x, w, b: ag.linear(x, w, b), [x], [w, b])
    run("linear relu", lambda x, w, b: F.relu(F.linear(x, w, b)), lambda x, w, b: ag.linear(x, w, b, ops.ACT_RELU), [x], [w, b])
    w1, b1 = rnd(1, 256, seed=5, scale=1 / 16), rnd(1, seed=6)
    run("linear N=1 sigmoid", lambda x, w, b: torch.sigmoid(F.linear(x, w, b)), lambda x, w, b: ag.linear(x, w, b, ops.ACT_SIGMOID), [x], [w1, b1])
    ag.BIG_LINEAR, keep = 0, ag.BIG_LINEAR
    try:
        wl = rnd(260, 256, seed=31, scale=1 / 16)
        run("linear wide-path", lambda x, w: F.linear(x, w), lambda x, w: ag.linear(x, w), [x], [wl])
    finally:
        ag.BIG_LINEAR = keep
    g, bb = rnd(256, seed=7), rnd(256, seed=8)
    run("layernorm", lambda x, g, b: F.layer_norm(x, (256,), g, b, 1e-5), lambda x, g, b: ag.norm(x, g, b, 1e-5, False), [x], [g, bb])
    Nn, L, P, Hd, V = 2, 9, 4, 64, 50
    ids = torch.randint(0, V, (Nn, L), generator=torch.Generator().manual_seed(2))
    ids[:, 2] = -200
    ids[0, 5] = ids[0, 6]
    emb, feats = rnd(V, Hd, seed=22), rnd(Nn, P, Hd, seed=23)
    e32 = emb.float().clone().requires_grad_(True)
    y = torch.stack([torch.cat([e32[ids[n, :2]], feats[n].float(), e32[ids[n, 3:]]]) for n in range(Nn)])
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(8)).to(BF)
    y.backward(gy.float())
    ed = emb.to(DEV).clone().requires_grad_(True)
    ed._g32 = torch.zeros(ed.shape, device=DEV, dtype=torch.float32)
    tok = ag.embed_token_index(ids, P).reshape(-1).to(DEV)
    for _ in range(2):
        ag.EmbedSpliceFn.apply(ids.to(DEV), ed, feats.to(DEV), P, P * Hd, tok).backward(gy.to(DEV))
    out.append(("arena embed_splice (precomputed token index)", err(ed._g32 / 2, e32.grad), rel_tol(e32.grad, 2.0 ** -6)))
    return out


def check_adamw():
    n = 1000
    gen = torch.Generator().manual_seed(0)
    w0 = torch.randn(n, generator=gen)
    p = w0.to(BF).to(DEV)
    master = p.float().clone()
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    ref_w = p.float().cpu().clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref_w], lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01)
    for step in range(1, 4):
        g = torch.randn(n, generator=gen).to(BF)
        ref_w.grad = g.float()
        opt.step()
        ops.adamw_(p, master, g.to(DEV), m, v, 1e-2, 0.9, 0.95, 1e-8, 0.01, step)
    acc = torch.zeros(1, device=DEV)
    ops.sumsq(p, acc)
    return [("adamw master", err(master, ref_w.detach()), 1e-5), ("adamw bf16 copy", err(p, ref_w.detach()), 2.0 ** -7 * 4),
            ("sumsq", abs(acc.item() - p.float().pow(2).sum().item()), 1e-2)]


def check_model_grads(golden_loader=None):
    """Parameter gradients of the tiny end-to-end model: HIP backward vs autograd through the fp32 oracle (same bf16-rounded
    weights), plus the fixture recorded from the imported reference."""
    from oracle import cases, lisa as olisa
    from tests import model_checks as mc
    cfg = cases.tiny_lisa_cfg("dinov2")
    m, sd = mc.build_pair(cfg)
    names = ["model.text_hidden_fcs.0.2.weight", "model.text_hidden_fcs.0.0.bias", "lm_head.weight", "model.lisa_iou_head.0.weight",
             "model.lisa_embedding_head.2.weight", "model.layers.1.self_attn.q_proj.weight", "model.layers.0.mlp.down_proj.weight",
             "model.embed_tokens.weight", "model.lisa_attention_layers.0.self_attn.q_proj.weight",
             "model.lisa_attention_layers.1.cross_attn_image_to_token.k_proj.weight", "model.lisa_attention_layers.0.norm1.weight",
             "model.lisa_dino_conv.weight", "model.lisa_final_attn.v_proj.weight", "model.layers.1.input_layernorm.weight"]
    for n in names:
        sd[n].requires_grad_(True)
    batch = mc._round_batch(cases.tiny_lisa_batch(img_size=896))
    gt = GateTrace()
    gt.oracle("ref", lambda: olisa.model_forward(sd, cfg, **batch, inference=False)["loss"].backward())
    # the same gradients from the bf16 CPU oracle: the error scale of bf16 arithmetic itself on this (ill-conditioned) tiny case
    sd_lo = {k: v.detach().to(BF) for k, v in sd.items()}
    for n in names:
        sd_lo[n].requires_grad_(True)
    gt.oracle("lo", lambda: olisa.model_forward(sd_lo, cfg, **mc._bf16_batch(batch), inference=False)["loss"].backward())
    for n, p in m.params.named_parameters():
        p.requires_grad_(n in names)
    gt.hip_begin(m)
    out = m.model_forward(**mc._dev(batch), inference=False)
    out["loss"].backward()
    gt.hip_collect(m)
    gt.hip_end(m)
    flips = gt.flipped_rows()
    all_stats = []
    res = [(f"ReLU gates recorded on both sides for {len(flips)} Linears ({sum(int(f.sum()) for f in flips.values())} units flipped)", 0.0 if len(flips) == len(GateTrace.NAMES) else 1.0, 0.5)]
    prm = dict(m.params.named_parameters())
    for n in names:
        ref = sd[n].grad
        got = prm[n].grad
        assert got is not None, n
        ratio, desc = grad_err(got, ref, sd_lo[n].grad.float(), floor=3e-4, skip_rows=GateTrace.rows_for(n, flips), stats=all_stats)   # floor: tensors whose true gradient vanishes hold rounding noise only
        res.append((f"grad {n}: {desc}; shown as err / tol", ratio, 1.0))
    res += ratio_summary(all_stats, "model grads")
    if golden_loader is not None:
        # gradients recorded from the imported reference (fp32, un-rounded weights): same policy, the bf16-CPU error as the scale
        g = golden_loader("lisa_tiny.pt")["grads"]
        for key, n in (("text_fc2_w", "model.text_hidden_fcs.0.2.weight"), ("iou_head0_w", "model.lisa_iou_head.0.weight")):
            ref = g[key]
            ratio, desc = grad_err(prm[n].grad, ref, sd_lo[n].grad.float(), floor=3e-4, skip_rows=GateTrace.rows_for(n, flips))
            res.append((f"grad {key} vs reference fixture: {desc}; shown as err / tol", ratio, 1.0))
    return res


# Multipliers on the bf16-CPU oracle's own error for ONE tensor (no trimming; round 3: 2.5 / 4.0 after a blanket 2 % trim).  HIP and the bf16-CPU
# oracle are two independent draws of bf16 rounding noise; measured over 6 dropout seeds x 92 tensors with the flipped-gate rows excluded
# (profiles/r04b_spread_grads_K2.md): the ratio HIP error / bf16-CPU error has median 1.11, 90 % 1.40 (RMS) / 1.59 (max), 99 % 2.6, and a tail
# to 3.5 on narrow tensors (a 128-entry bias) -- with ~1000 (tensor, step) comparisons per suite run a bound of 2 x fails 3 % of them by chance.
# What the per-tensor bound cannot see -- a SYSTEMATIC loss of precision -- is bounded over all tensors of a test: median RMS ratio <= Q50_RMS,
# 90 % quantile <= Q90_RMS (`ratio_summary`).
K_RMS, K_MAX = 4.0, 4.0
Q50_RMS, Q90_RMS = 1.5, 2.5


def ratio_summary(stats, tag):
    """stats: grad_err's `stats` tuples of every tensor of a test -> two result lines bounding the DISTRIBUTION of RMS err HIP / RMS err bf16-CPU
    over the tensors that carry a gradient (|ref|max above 10 x the 3e-4 noise floor): no systematic loss against the reference's dtype."""
    rr = sorted(st[0] / max(st[1], 1e-12) for st in stats if st[5] > 3e-3 and st[1] > 0)
    if len(rr) < 30:                                      # quantiles of a dozen ratios are noise themselves
        return []
    q = lambda p: rr[min(len(rr) - 1, int(p * len(rr)))]
    return [(f"{tag}: median over {len(rr)} tensors of RMS err HIP / RMS err bf16-CPU oracle", q(0.5), Q50_RMS),
            (f"{tag}: 90 % quantile of the same ratio", q(0.9), Q90_RMS)]


class GateTrace:
    """ReLU gate patterns of one forward pass on each side.  The five ReLUs of the path (text_hidden_fcs.0.0, the two
    MLP blocks' lin1, lisa_iou_head.0, lisa_embedding_head.0) follow a Linear; a unit whose pre-activation is within bf16 rounding noise
    of zero is ON in one evaluation and OFF in another, and that moves ROW `unit` of the Linear's weight gradient (and element `unit` of
    its bias gradient) by that sample's whole contribution -- measured in round 3: 0.14 at |ref|max 0.36, every other row < 0.006.
    Instead of trimming the largest 2 % of every tensor's errors (round 3), the tests record the gates (output > 0) of the HIP forward and of
    the fp32 / bf16 oracle forwards on the same inputs and exclude exactly the rows whose gate provably differs on some sample."""

    NAMES = ("model.text_hidden_fcs.0.0", "model.lisa_attention_layers.0.mlp.lin1", "model.lisa_attention_layers.1.mlp.lin1",
             "model.lisa_iou_head.0", "model.lisa_embedding_head.0")

    def __init__(self):
        self.gates = {"hip": {}, "ref": {}, "lo": {}}

    def oracle(self, side, fn):
        """Run fn() with the oracle's trace hook on; its ReLU outputs (all calls, in order) become side's gates of this pass."""
        from oracle import mask_head
        mask_head.TRACE = {}
        try:
            out = fn()
            for n, ys in mask_head.TRACE.items():
                self.gates[side].setdefault(n, []).append(torch.cat([y.float() for y in ys], 0) > 0)
        finally:
            mask_head.TRACE = None
        return out

    def hip_begin(self, model):
        model.__dict__["_relu_trace"] = {}

    def hip_collect(self, model, calls_per_pass=None):
        """After a forward (or a hipGraph replay): the LAST pass's ReLU outputs.  calls_per_pass: head invocations per forward (groups of
        images with equal proposal count; 1 for the uniform test batches) -- a replayed graph refreshes the tensors recorded at capture."""
        tr = model.__dict__["_relu_trace"]
        for n, ys in tr.items():
            k = calls_per_pass or 1
            self.gates["hip"].setdefault(n, []).append(torch.cat([y.float().cpu() for y in ys[-k:]], 0) > 0)

    def hip_end(self, model):
        model.__dict__.pop("_relu_trace", None)

    def flipped_rows(self):
        """name of the Linear -> bool [units]: the gate of that unit differs between fp32 oracle and HIP, or fp32 and bf16 oracle, on some
        sample of some pass (the same rows are excluded from the HIP error and from the bf16-CPU yardstick)."""
        out = {}
        for n, ref in self.gates["ref"].items():
            fl = torch.zeros(ref[0].shape[1], dtype=torch.bool)
            for side in ("hip", "lo"):
                got = self.gates[side].get(n, [])
                assert len(got) == len(ref), (n, side, len(got), len(ref))
                for a, b in zip(got, ref):
                    assert a.shape == b.shape, (n, side, a.shape, b.shape)
                    fl |= (a != b).any(0)
            out[n] = fl
        return out

    @staticmethod
    def rows_for(param_name, flips):
        for n, fl in (flips or {}).items():
            if param_name in (n + ".weight", n + ".bias"):
                return fl
        return None


def grad_err(gh, gr, gl, floor=3e-4, skip_rows=None, stats=None):
    """Gradient tolerance policy (round 4).  gh: HIP, gr: fp32 oracle, gl: the same oracle run in bf16 on the CPU (the
    reference's own arithmetic).  Over ALL entries of the tensor except the rows of a Linear-before-ReLU whose gate provably flipped
    (`skip_rows`, bool over dim 0, from `GateTrace.flipped_rows`):
        RMS error  <= max(3 % of rms(ref), K_RMS x RMS error of the bf16-CPU oracle)
        max error  <= max(3 % of |ref|max, K_MAX x max error of the bf16-CPU oracle)
    and the excluded rows must stay finite and within 10 x the tensor's largest gradient.  No trimming: a kernel bug that corrupts one tile-edge row or one head of a weight gradient fails the max bound.
    -> (worst ratio err / tol, description); `stats` (a list) receives (rms err, rms bf16-CPU err, max err, max bf16-CPU err, rms ref, max ref, skipped)."""
    shp = gr.shape
    gh, gr, gl = gh.reshape(shp).double().cpu(), gr.double(), gl.double().reshape(shp)
    finite = bool(torch.isfinite(gh).all())
    n_skip = 0
    d_skip = dl_skip = 0.0
    if skip_rows is not None and bool(skip_rows.any()):
        keep = ~skip_rows
        n_skip = int(skip_rows.sum())
        d_skip = float((gh[skip_rows] - gr[skip_rows]).abs().max())
        dl_skip = float((gl[skip_rows] - gr[skip_rows]).abs().max())
        ref_max_all = float(gr.abs().max())
        gh, gr, gl = gh[keep], gr[keep], gl[keep]
    else:
        ref_max_all = float(gr.abs().max()) if gr.numel() else 0.0
    gh, gr, gl = gh.flatten(), gr.flatten(), gl.flatten()
    rms = lambda x: float(x.pow(2).mean().sqrt()) if x.numel() else 0.0
    d, dl = (gh - gr).abs(), (gl - gr).abs()
    small = gh.numel() < 50                                   # a handful of entries (a bias of 1, a [1, 8] head): no statistics, one looser bound
    # round 5 (full depth, K = 256): a tensor whose TRUE gradient vanishes (|ref|max below the noise floor: the head's q / k projections when the
    # pooled mask features of all proposals nearly coincide -- softmax over identical keys) holds nothing but each pipeline's rounding noise;
    # the comparison is noise against ONE draw of noise (measured ratio 4.1-4.5 at full depth): the same doubled multiplier as the small tensors
    noise_only = ref_max_all < floor
    loose = small or noise_only
    d_max, dl_max = (float(d.max()), float(dl.max())) if d.numel() else (0.0, 0.0)
    # small tensors: ONE bf16-CPU draw of one or a few numbers is no yardstick (the ratio of two such draws exceeds 9 in 7 % of cases): 10 % relative
    t_rms = max((0.10 if small else 0.03) * rms(gr), (2 * K_RMS if loose else K_RMS) * rms(dl), floor / 4)
    t_max = max((0.10 if small else 0.03) * ref_max_all, (2 * K_MAX if loose else K_MAX) * dl_max, floor)
    # the excluded rows: a flipped gate moves a row by one sample's whole contribution, which for a bias (a sum of cancelling terms) exceeds the
    # row's own gradient (measured 6.9e-3 at |ref|max 1.8e-3; at full depth 1.2e-2 at |ref|max 1.4e-4, where the kept rows cancel almost completely) --
    # they have to stay finite and within an order of magnitude of the tensor OR of what the bf16-CPU oracle's own flipped rows moved by (round 5)
    t_skip = max(10.0 * ref_max_all, K_MAX * dl_skip, floor)
    if stats is not None:
        stats.append((rms(d), rms(dl), d_max, dl_max, rms(gr), ref_max_all, n_skip))
    r = max(rms(d) / t_rms, d_max / t_max, d_skip / t_skip, 0.0 if finite else 1e9)
    return r, (f"rms err {rms(d):.2e} (tol {t_rms:.2e}, bf16-CPU {rms(dl):.2e}), max err {d_max:.2e} (tol {t_max:.2e}, bf16-CPU {dl_max:.2e}) "
               f"over all {gh.numel()} entries" + (f" outside the {n_skip} rows with a flipped ReLU gate (largest error there {d_skip:.2e})" if n_skip else "") +
               f", |ref| {ref_max_all:.2e}")


def _lora_case(backbone="sam", p_drop=0.05, K=16):
    from oracle import cases
    from tests import model_checks as mc
    cfg = cases.tiny_lisa_cfg(backbone, lora_r=8)
    cfg.llama.lora_dropout = p_drop
    m, sd = mc.build_pair(cfg)
    m.set_trainable()
    img = 896 if backbone == "dinov2" else cfg.sam.img
    batch = mc._round_batch(cases.tiny_lisa_batch(img_size=img, K=K))
    return cfg, m, sd, batch


def check_model_grads_lora(backbone="sam", dropout=None, all_tensors=False, stats=None):
    """The configuration the benchmark times, at tiny size: LoRA r = 8 (random B) with dropout 0.05 on q/v, trainable embed / lm_head /
    text_hidden_fcs / lisa_*, gradients accumulated by the kernels into the fp32 arena over TWO micro-steps -- against autograd through the
    fp32 oracle with the same dropout masks.  Covers the extension K-tile GEMMs, the rank-8 kernels, the fused weight-gradient blocks."""
    from llmseg_amd.train import GradArena
    from oracle import lisa as olisa
    from tests import model_checks as mc
    cfg, m, sd, batch = _lora_case(backbone)
    names = [n for n, p in m.params.named_parameters() if p.requires_grad]
    for n in names:
        sd[n].requires_grad_(True)
    seed, off = (0x1234ABCD, 7) if dropout is None else dropout
    gt = GateTrace()

    def run_ref():
        o = olisa.model_forward(sd, cfg, **batch, inference=False, dropout_state=(seed, off))
        o["loss"].backward()
        return o
    ref = gt.oracle("ref", run_ref)
    sd_lo = {k: v.detach().to(BF) for k, v in sd.items()}
    for n in names:
        sd_lo[n].requires_grad_(True)

    def run_lo():
        o = olisa.model_forward(sd_lo, cfg, **mc._bf16_batch(batch), inference=False, dropout_state=(seed, off))
        o["loss"].backward()
        return o
    lo = gt.oracle("lo", run_lo)
    arena = GradArena(m)
    m.set_dropout_seed(seed, off)
    db = mc._dev(batch)
    plan = m.make_plan(**db)
    gt.hip_begin(m)
    for _ in range(2):                                   # the same micro-step twice (same masks): the arena holds 2 x the gradient
        out = m.model_forward(**db, inference=False, plan=plan)
        out["loss"].backward()
    gt.hip_collect(m)
    gt.hip_end(m)
    flips = gt.flipped_rows()
    res = []
    for k in ("ce_loss", "align_loss", "regression_loss", "loss"):
        r = float(ref[k])
        res.append((f"lora+dropout train {k} (ref {r:.4f})", abs(float(out[k]) - r), max(5e-3 * max(1.0, abs(r)), 1.5 * abs(float(lo[k]) - r))))
    prm = dict(m.params.named_parameters())
    pick = [n for n in names if any(t in n for t in ("layers.0.self_attn.q_proj.lora_A", "layers.1.self_attn.q_proj.lora_B", "layers.1.self_attn.v_proj.lora_A",
                                                     "layers.0.self_attn.v_proj.lora_B", "embed_tokens", "lm_head", "text_hidden_fcs.0.0.weight",
                                                     "text_hidden_fcs.0.2.bias", "lisa_attention_layers.0.self_attn.k_proj.weight",
                                                     "lisa_attention_layers.1.self_attn.v_proj.bias", "cross_attn_image_to_token.v_proj.weight",
                                                     "cross_attn_image_to_token.q_proj.weight", "lisa_final_attn.v_proj.weight", "lisa_attention_layers.0.norm2.weight",
                                                     "lisa_iou_head.2.weight", "lisa_embedding_head.0.bias"))]
    assert len(pick) >= 16, pick
    all_stats = []
    for n in names:                                      # every trainable tensor feeds the distribution bound; the picked ones are listed one by one
        if not all_tensors and n not in pick:
            grad_err(prm[n]._g32 / 2, sd[n].grad, sd_lo[n].grad.float(), floor=3e-4, skip_rows=GateTrace.rows_for(n, flips), stats=all_stats)
    if all_tensors:
        pick = names
    for n in pick:
        assert prm[n].grad is None, n
        got, r = prm[n]._g32 / 2, sd[n].grad
        # floor 3e-4 (other gradients here are 1e-2 .. 2): a tensor whose true gradient vanishes (k-projections: softmax is
        # shift-invariant, |ref| ~ 1e-6) holds rounding noise only, and that noise moves with every change of summation order upstream
        st = [] if stats is not None else None
        ratio, desc = grad_err(got, r, sd_lo[n].grad.float(), floor=3e-4, skip_rows=GateTrace.rows_for(n, flips), stats=st)
        if stats is not None:
            stats.append((n, ratio) + st[0])
            all_stats.append(st[0])
        else:
            grad_err(got, r, sd_lo[n].grad.float(), floor=3e-4, skip_rows=GateTrace.rows_for(n, flips), stats=all_stats)
        res.append((f"arena grad {n}: {desc}; shown as err / tol", ratio, 1.0))
    res += ratio_summary(all_stats, "lora arena grads")
    arena.detach()
    return res


def check_trainer(use_graph=False, opt_steps=3, accum=2, K=16):
    """`Trainer` on the HIP model (arena, HipAdamW, clip, WarmupDecayLR, LoRA + dropout; optionally the hipGraph micro-step) against the
    fp32 oracle, TEACHER-FORCED: at every optimizer step the oracle is evaluated at the weights the HIP model holds at that moment, so a
    difference is this step's arithmetic and not the divergence of two trajectories (AdamW turns the rounding noise of near-zero gradients
    into +-lr steps, which made the round-2 form of this test a test of that noise).  Per optimizer step:
      * losses of every micro-step (same weights, same dropout masks);
      * the accumulated gradient the optimizer consumes (fp32 arena, observed through `Trainer.grad_hook`) against autograd through the
        oracle under `grad_err`'s policy (RMS <= 2.5 x, max <= 4 x the bf16-CPU oracle's over the 98 % best entries, 3 % floors), direction 1 - cos <= max(0.05,
        1.5 x (1 - cos) of the bf16-CPU oracle);
      * the update: fp32 master weights after the step against a float64 AdamW (clip 1.0, WarmupDecayLR, bias correction) applied to
        the arena's own gradient -- the optimizer / clipping / schedule arithmetic, to rounding.
    K = 512, accum = 8: the workload shape of BASELINE configs[4] (512 candidate masks per image, grad-accum 8; training.py:79-82)."""
    from llmseg_amd.train import Trainer, warmup_decay_lr
    from oracle import lisa as olisa
    from tests import model_checks as mc
    cfg, m, sd, batch = _lora_case("sam", K=K)
    names = [n for n, p in m.params.named_parameters() if p.requires_grad]
    prm = dict(m.params.named_parameters())
    lr, clip, betas, eps, seed = 2e-3, 1.0, (0.9, 0.95), 1e-8, 99
    tr = Trainer(m, lr=lr, betas=betas, clip=clip, grad_accum=accum, warmup=1, total_steps=10, use_graph=use_graph, graph_warmup=1)
    m.set_dropout_seed(seed, 0)
    db = mc._dev(batch)
    plan = m.make_plan(**db)
    seen = {}
    tr.grad_hook = lambda t, ss: seen.update(g={n: prm[n]._g32.detach().float().cpu().clone() for n in names}, ss=float(ss))
    tag = ("graph" if use_graph else "eager") + (f" K={K} accum={accum}" if K != 16 else "")
    cpu = lambda ts: [t.detach().double().cpu().clone() for t in ts]
    res, hip_losses, offset = [], [], 0
    first_last = []
    for step in range(opt_steps):
        w0, m0, v0 = cpu(tr.opt.master), cpu(tr.opt.m), cpu(tr.opt.v)
        held = {n: prm[n].detach().float().cpu().clone() for n in names}          # the bf16 copies the forward reads
        gt = GateTrace()
        if "_relu_trace" not in m.__dict__:
            gt.hip_begin(m)                                                       # kept on across steps: a hipGraph refreshes the tensors recorded at capture
        losses = []
        for _ in range(accum):
            losses.append(float(tr.micro_step(db, plan)["loss"].detach()))
            gt.hip_collect(m)
        hip_losses += losses
        assert tr.opt_steps == step + 1 and "g" in seen
        w = {n: held[n].clone().requires_grad_(True) for n in names}
        wl = {n: held[n].to(BF).requires_grad_(True) for n in names}
        sdw, sdl = {**sd, **w}, {**{k: v.to(BF) for k, v in sd.items()}, **wl}
        ref_losses = []
        for a in range(accum):
            offset += 1
            def run_ref():
                o = olisa.model_forward(sdw, cfg, **batch, inference=False, dropout_state=(seed, offset))
                o["loss"].backward()
                return o
            ref_losses.append(float(gt.oracle("ref", run_ref)["loss"]))
            gt.oracle("lo", lambda: olisa.model_forward(sdl, cfg, **mc._bf16_batch(batch), inference=False, dropout_state=(seed, offset))["loss"].backward())
        flips = gt.flipped_rows()
        first_last += [ref_losses[0], ref_losses[-1]]
        res.append((f"trainer[{tag}] step {step}: micro-step losses vs the oracle at the same weights (ref {ref_losses[0]:.4f} ..)",
                    max(abs(h - r) for h, r in zip(losses, ref_losses)), 5e-3 * max(1.0, abs(ref_losses[0]))))
        gmax = max(w[n].grad.abs().max().item() for n in names)
        worst_e, worst_c = (0.0, ""), (0.0, "")
        step_stats = []
        for n in names:
            gr, gl, gh = w[n].grad, wl[n].grad.float(), seen["g"][n].reshape(w[n].shape)
            ratio, desc = grad_err(gh, gr, gl, floor=3e-4 * accum, skip_rows=GateTrace.rows_for(n, flips), stats=step_stats)
            worst_e = max(worst_e, (ratio, f"{n}: {desc}"))
            if gr.abs().max().item() > 1e-3 * gmax:                                  # direction, for tensors that carry a gradient at all
                cos = lambda x, y: float((x.flatten().double() @ y.flatten().double()) / (x.double().norm() * y.double().norm() + 1e-30))
                c_h, c_l = 1.0 - cos(gh, gr), 1.0 - cos(gl, gr)
                tol_c = max(0.05, 1.5 * c_l)
                worst_c = max(worst_c, (c_h / tol_c, f"{n}: 1-cos {c_h:.2e} tol {tol_c:.2e} (bf16-CPU {c_l:.2e})"))
        res.append((f"trainer[{tag}] step {step}: accumulated gradient, worst of {len(names)} tensors = {worst_e[1]}; shown as err / tol", worst_e[0], 1.0))
        res.append((f"trainer[{tag}] step {step}: gradient direction, worst tensor = {worst_c[1]}; shown as (1-cos) / tol", worst_c[0], 1.0))
        res += ratio_summary(step_stats, f"trainer[{tag}] step {step}")
        # the update, from the arena's own gradient (float64 restatement of the recipe)
        g = [seen["g"][n].double().reshape(-1) / accum for n in names]
        norm = torch.sqrt(sum((x * x).sum() for x in g))
        res.append((f"trainer[{tag}] step {step}: squared gradient norm (one reduction over the arena) vs float64", abs(seen["ss"] - float(norm ** 2) * accum * accum) / float(norm ** 2 * accum * accum + 1e-30), 1e-4))
        coef = min(1.0, clip / (float(norm) + 1e-6))
        lr_t, t = warmup_decay_lr(step, lr, 1, 10), step + 1
        worst_u = 0.0
        for i, n in enumerate(names):
            gn = g[i] * coef
            m1 = betas[0] * m0[i].reshape(-1) + (1 - betas[0]) * gn
            v1 = betas[1] * v0[i].reshape(-1) + (1 - betas[1]) * gn * gn
            w1 = w0[i].reshape(-1) - lr_t * ((m1 / (1 - betas[0] ** t)) / ((v1 / (1 - betas[1] ** t)).sqrt() + eps))
            worst_u = max(worst_u, (tr.opt.master[i].detach().double().cpu().reshape(-1) - w1).abs().max().item())
            wb = prm[n].detach().float().cpu().reshape(-1)
            assert torch.equal(wb, tr.opt.master[i].detach().to(BF).float().cpu().reshape(-1)), f"bf16 copy of {n} is not the rounded master"
        res.append((f"trainer[{tag}] step {step}: fp32 masters after AdamW vs float64 on the same gradient (lr {lr_t:.1e}, clip coef {coef:.3f})", worst_u, 2e-3 * max(lr_t, 1e-5)))
    moved = max(first_last) - min(first_last)
    res.append((f"trainer[{tag}] the loss moved over the optimizer steps (|range| = {moved:.3f})", 0.0 if moved > 0.01 else 1.0, 0.5))
    if use_graph:
        assert tr.graph_error is None, tr.graph_error
        assert any(e["graph"] is not None for e in tr._graphs.values()), "the hipGraph path was never taken"
    m.__dict__.pop("_relu_trace", None)
    tr.close()
    return res, hip_losses


def check_trainer_graph_vs_eager():
    r_e, l_e = check_trainer(False)
    r_g, l_g = check_trainer(True)
    res = r_e + r_g
    res.append((f"trainer graph vs eager: the losses of every micro-step are identical ({l_e} vs {l_g})", 0.0 if l_e == l_g else 1.0, 0.5))
    return res


def check_determinism():
    """The same micro-step from the same state, run twice (eager, then again as a replayed hipGraph): losses and the fp32 gradient arena
    must agree BIT FOR BIT.  Holds because no kernel adds floats with atomics (include/llmseg_hip.h "Determinism"); a data race in a
    kernel (LDS-DMA staging, split-K slabs, the two streams of the forward) would show up here as a differing element."""
    from llmseg_amd.train import Trainer
    from tests import model_checks as mc
    res = []
    for K in (16, 96):
        cfg, m, sd, batch = _lora_case("sam", K=K)
        db = mc._dev(batch)
        plan = m.make_plan(**db)
        runs = []
        for use_graph in (False, False, True, True):
            tr = Trainer(m, lr=1e-3, grad_accum=1000, use_graph=use_graph, graph_warmup=0)
            arenas, losses = [], []
            for rep in range(3):
                m.set_dropout_seed(77, 0)
                tr.arena.zero_()
                out = tr.micro_step(db, plan)
                torch.cuda.synchronize()
                losses.append(tuple(float(out[k]) for k in ("loss", "ce_loss", "align_loss", "regression_loss")))
                arenas.append(tr.arena.flat.clone())
            if use_graph:
                assert tr.graph_error is None, tr.graph_error
                assert any(e["graph"] is not None for e in tr._graphs.values()), "the hipGraph path was never taken"
            runs.append((arenas, losses))
            tr.close()
        ref_a, ref_l = runs[0][0][0], runs[0][1][0]
        nz = float((ref_a != 0).float().mean())
        res.append((f"determinism K={K}: the arena holds gradients (non-zero fraction {nz:.3f})", 0.0 if nz > 0.2 else 1.0, 0.5))
        for ri, (arenas, losses) in enumerate(runs):
            tag = ("eager", "eager (second trainer)", "hipGraph", "hipGraph (second capture)")[ri]
            for rep, (a, l) in enumerate(zip(arenas, losses)):
                res.append((f"determinism K={K}: {tag} repeat {rep}: arena elements that differ from the first run", float((a != ref_a).sum()), 0.0))
                res.append((f"determinism K={K}: {tag} repeat {rep}: losses identical ({l} vs {ref_l})", 0.0 if l == ref_l else 1.0, 0.5))
    return res


def check_graph_rotating_batches():
    """A loader that rotates several resident batches (and their plans) through a hipGraph trainer: every micro-step must see ITS batch.
    The graph owns its input buffers and its plan (the round-3 trainer aliased the tensors present at capture and skipped the
    copy on a pointer match, so the set that was present at capture was replayed stale once another set had been copied over it).
    Checked against the eager trainer on the same sequence: identical losses and master weights, bit for bit; the caller's tensors and
    plans are unchanged afterwards."""
    from llmseg_amd.train import Trainer
    from tests import model_checks as mc

    def run(use_graph):
        cfg, m, sd, batch = _lora_case("sam")
        m.set_dropout_seed(31, 0)
        base = mc._dev(batch)
        sets = []
        for i in range(3):
            b = dict(base)
            b["images"] = (base["images"].float() * (1.0 - 0.3 * i) + 0.1 * i).to(base["images"].dtype)
            b["images_clip"] = base["images_clip"].roll(i, -1).contiguous()
            b["sam_ious_list"] = [(t * (1.0 - 0.2 * i)).contiguous() for t in base["sam_ious_list"]]
            ids = base["input_ids"].clone()
            ids[:, 10 + i] = 7 + i                                              # a different prompt token: the plan's tensors differ too
            b["input_ids"] = ids
            sets.append((b, m.make_plan(**b)))
        keep = [(b["images"].clone(), b["input_ids"].clone(), {k: v.clone() for k, v in p.tensors.items() if torch.is_tensor(v)}) for b, p in sets]
        tr = Trainer(m, lr=1e-3, grad_accum=3, warmup=0, total_steps=20, use_graph=use_graph, graph_warmup=1)
        losses = [float(tr.micro_step(*sets[i % 3])["loss"]) for i in range(9)]
        if use_graph:
            assert tr.graph_error is None, tr.graph_error
            assert any(e["graph"] is not None for e in tr._graphs.values()), "the hipGraph path was never taken"
        same = all(torch.equal(b["images"], k[0]) and torch.equal(b["input_ids"], k[1]) and all(torch.equal(p.tensors[n], t) for n, t in k[2].items())
                   for (b, p), k in zip(sets, keep))
        w = torch.cat([x.flatten().cpu() for x in tr.opt.master])
        tr.close()
        return losses, w, same
    le, we, se = run(False)
    lg, wg, sg = run(True)
    return [(f"rotating batches: the three sets give different losses ({le[:3]})", 0.0 if len(set(le[:3])) == 3 else 1.0, 0.5),
            (f"rotating batches: hipGraph losses identical to eager ({lg} vs {le})", 0.0 if le == lg else 1.0, 0.5),
            ("rotating batches: master weights after 3 optimizer steps, elements that differ graph vs eager", float((we != wg).sum()), 0.0),
            ("rotating batches: the caller's batches and plans are untouched", 0.0 if (se and sg) else 1.0, 0.5)]


def check_checkpoint_resume(tmp_dir):
    """Save after the first optimizer step, resume in a FRESH model + trainer, take the second step: same parameters and losses as the
    uninterrupted run; and a reference-layout file (PEFT prefix, rotary buffers, SAM decoder keys) loads with those extras ignored."""
    import os
    from llmseg_amd import checkpoint as ck
    from llmseg_amd.train import Trainer
    from tests import model_checks as mc
    kw = dict(lr=2e-3, grad_accum=2, warmup=0, total_steps=20)

    def fresh():
        cfg, m, sd, batch = _lora_case("sam")
        m.set_dropout_seed(11, 0)
        return m, Trainer(m, **kw), mc._dev(batch)
    m, tr, db = fresh()
    la = [float(tr.micro_step(db)["loss"]) for _ in range(4)]
    pa = torch.cat([w.flatten().cpu() for w in tr.opt.master])
    mom_a = torch.cat([w.flatten().cpu() for w in tr.opt.m] + [w.flatten().cpu() for w in tr.opt.v])
    tr.close()
    m, tr, db = fresh()
    lb = [float(tr.micro_step(db)["loss"]) for _ in range(2)]
    ck.save_checkpoint(tmp_dir, m, tr, global_step=tr.opt_steps)
    tr.close()
    m2, tr2, db = fresh()
    with torch.no_grad():                                   # scramble the fresh model: everything must come from the checkpoint
        for p in m2.trainable_parameters():
            p.add_(0.05)
    info = ck.load_checkpoint(tmp_dir, m2, tr2, steps_per_epoch=1)
    lb += [float(tr2.micro_step(db)["loss"]) for _ in range(2)]
    pb = torch.cat([w.flatten().cpu() for w in tr2.opt.master])
    # Round 4: the library has no floating-point atomics left (fixed-order reductions), so a resumed run must reproduce the uninterrupted
    # one BIT FOR BIT -- losses, fp32 masters, moments.  Anything else is a state that was not restored (e.g. the dropout offset).
    ma = torch.cat([w.flatten().cpu() for w in tr2.opt.m] + [w.flatten().cpu() for w in tr2.opt.v])
    res = [("resume: optimizer state restored, tag / epoch parsed", 0.0 if (info["optimizer_restored"] and info["global_steps"] == 1 and info["start_epoch"] == 1) else 1.0, 0.5),
           (f"resume: losses of the 4 micro-steps identical to the uninterrupted run ({la} vs {lb})", 0.0 if la == lb else 1.0, 0.5),
           ("resume: fp32 master weights after step 2 bit-identical to the uninterrupted run: number of differing elements", float((pa != pb).sum()), 0.0),
           ("resume: largest master difference", (pa - pb).abs().max().item(), 0.0),
           ("resume: Adam moments bit-identical to the uninterrupted run: number of differing elements", float((ma != mom_a).sum()), 0.0)]
    # a reference-shaped file: PEFT prefix + buffers / decoder tensors that are not on the path
    sdm = {ck.PEFT_PREFIX + k: v.cpu() for k, v in m2.state_dict().items()}
    sdm[ck.PEFT_PREFIX + "model.layers.0.self_attn.rotary_emb.inv_freq"] = torch.ones(64)
    sdm[ck.PEFT_PREFIX + "model.visual_model.mask_decoder.iou_token.weight"] = torch.ones(1, 256)
    rdir = os.path.join(tmp_dir, "ref", "global_step5000")
    os.makedirs(rdir)
    torch.save({"module": sdm, "global_steps": 5000, "ds_version": "0.10.0"}, os.path.join(rdir, "mp_rank_00_model_states.pt"))
    open(os.path.join(tmp_dir, "ref", "latest"), "w").write("global_step5000")
    m3, tr3, _ = fresh()
    info = ck.load_checkpoint(os.path.join(tmp_dir, "ref"), m3, tr3)
    same = all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(m3.state_dict().values(), m2.state_dict().values()))
    res.append(("reference-layout checkpoint: weights identical, extras ignored, epoch 10 parsed",
                0.0 if (same and len(info["ignored"]) == 2 and not info["missing"] and info["start_epoch"] == 10 and not info["optimizer_restored"]) else 1.0, 0.5))
    res.append(("reference-layout checkpoint: fp32 masters re-read from the loaded weights",
                max((w.cpu() - p.detach().float().cpu()).abs().max().item() for w, p in zip(tr3.opt.master, tr3.opt.params)), 0.0))
    tr2.close(); tr3.close()
    # a model built WITH the SAM prompt encoder / mask decoder (evaluate() reads them): save -> load must restore them too
    import dataclasses
    from llmseg_amd import lisa as hip_lisa
    dcfg = dataclasses.replace(m2.config, sam_decoder=True)
    md = hip_lisa.LISAForCausalLM(dcfg, device=mc.DEV).init_random(seed=21)
    ddir = os.path.join(tmp_dir, "dec")
    ck.save_checkpoint(ddir, md, None, global_step=7)
    me = hip_lisa.LISAForCausalLM(dataclasses.replace(dcfg), device=mc.DEV).init_random(seed=22)
    info = ck.load_reference_checkpoint(me, ddir, strict=True)
    dec = [k for k in md.state_dict() if ".prompt_encoder." in k or ".mask_decoder." in k]
    sde, sdd = me.state_dict(), md.state_dict()
    ok = len(dec) > 40 and not info["missing"] and not info["ignored"] and all(torch.equal(sde[k], sdd[k]) for k in sdd)
    res.append((f"sam_decoder=True checkpoint round trip: {len(dec)} prompt-encoder / mask-decoder tensors restored", 0.0 if ok else 1.0, 0.5))
    # ... and the same file into a model WITHOUT the decoder: those keys are reported as ignored, nothing is missing
    info = ck.load_reference_checkpoint(m3, ddir)
    res.append(("decoder checkpoint into a decoder-less model: decoder keys ignored", 0.0 if (len(info["ignored"]) == len(dec) and not info["missing"]) else 1.0, 0.5))
    return res


ALL = [check_gemm_layouts, check_autograd_ops, check_lora_paths, check_arena_ops, check_adamw]


def _variant_batches(batch, k):
    """k micro-batches of the structure of `batch` with different data (text tokens, images, proposals rotated; specials and labels kept in place)."""
    out = []
    for j in range(k):
        b = {key: ([t.clone() for t in v] if isinstance(v, list) else (v.clone() if torch.is_tensor(v) else v)) for key, v in batch.items()}
        ids = b["input_ids"]
        text = (ids >= 3) & (ids < 31999)
        ids[text] = (ids[text] - 3 + 977 * j) % 31996 + 3
        lab = b["labels"]
        keep = lab >= 0
        lab[keep] = ids[keep]
        b["images"] = torch.roll(b["images"], shifts=37 * j, dims=-1)
        b["images_clip"] = torch.roll(b["images_clip"], shifts=11 * j, dims=-2)
        b["sam_segs_list"] = [torch.roll(t, shifts=j, dims=0) for t in b["sam_segs_list"]]
        b["sam_ious_list"] = [torch.roll(t, shifts=3 * j, dims=-1) for t in b["sam_ious_list"]]
        out.append(b)
    return out


def check_fused_accum(k=3):
    """the k micro-batches of one accumulation window run as ONE pass (`Trainer(fused_accum=k)`, `make_plan(micro_batches=k)`,
    `llmseg_dropout.seg_rows`) must leave in the fp32 arena the gradient of  sum_j loss_j  with every loss_j formed as the reference forms it for
    micro-batch j alone -- CE averaged over ITS labelled tokens, align / IoP losses over ITS images, LoRA dropout mask of ITS step (offset + j).
    (a) against the ORACLE: autograd through `oracle.lisa.model_forward` on each of the k DIFFERENT micro-batches with dropout state (seed, j + 1),
        summed -- `grad_err` policy with the bf16-CPU oracle as the yardstick, gates traced on both sides;
    (b) against k sequential micro-steps of the HIP trainer on the same weights (lr = 0 keeps them): losses, the squared norm, `lm_head` (no ReLU /
        softmax-temperature amplification downstream of it: 1 %), the whole gradient's direction.  Tensor by tensor the two HIP runs are two draws of
        bf16 rounding noise (another M picks other split-K slice counts: activations differ in isolated ulps, measured 6.6e-3 relative on the head's
        embeddings), which the tiny head's 1 / tau = 20 softmax amplifies to ~10 % of a head tensor's RMS -- the same spread HIP and the bf16-CPU
        oracle show against fp32 here (tools/probes/fused_diag*.py), hence the oracle-based bound in (a) and the aggregate bounds in (b);
    (c) the fused pass replayed from a hipGraph leaves the same BITS as the eager fused pass; the dropout offset ends where k micro-steps leave it."""
    from llmseg_amd.train import Trainer, merge_micro_batches
    from oracle import lisa as olisa
    from tests import model_checks as mc
    cfg, m, sd, batch = _lora_case("sam")
    names = [n for n, p in m.params.named_parameters() if p.requires_grad]
    prm = dict(m.params.named_parameters())
    cpu_batches = _variant_batches(batch, k)
    batches = [mc._dev(b) for b in cpu_batches]
    seed = 4242
    grab = lambda store: (lambda t, ss: store.update(g={n: prm[n]._g32.detach().clone() for n in names}, ss=float(ss)))
    scal = lambda out: {kk: float(v.detach()) for kk, v in out.items() if torch.is_tensor(v) and v.numel() == 1}
    # (b) sequential: k micro-steps, one optimizer step (lr 0: the weights stay)
    seq = {}
    tr = Trainer(m, lr=0.0, grad_accum=k, warmup=1, total_steps=10)
    tr.grad_hook = grab(seq)
    m.set_dropout_seed(seed, 0)
    seq_losses = [scal(tr.micro_step(b, m.make_plan(**b))) for b in batches]
    assert tr.opt_steps == 1 and "g" in seq
    end_offset = int(m.dropout_state()[1])
    tr.close()
    merged = merge_micro_batches(batches)
    plan = m.make_plan(**merged, micro_batches=k)
    assert merged["offset"].tolist() == [0] + [int(batches[0]["offset"][-1]) * j + int(o) for j in range(k) for o in batches[0]["offset"][1:]]
    res, outs = [], {}
    gt = GateTrace()
    for mode, use_graph in (("eager", False), ("hipGraph", True)):
        fus = {}
        tr = Trainer(m, lr=0.0, grad_accum=1, warmup=1, total_steps=10, fused_accum=k, use_graph=use_graph, graph_warmup=1)
        tr.grad_hook = grab(fus)
        if not use_graph:
            gt.hip_begin(m)
        for _ in range(3 if use_graph else 1):             # graph: eager warm-up, capture + replay, replay
            m.set_dropout_seed(seed, 0)
            out = tr.micro_step(merged, plan)
        torch.cuda.synchronize()
        if use_graph:
            assert tr.graph_error is None, tr.graph_error
            assert any(e["graph"] is not None for e in tr._graphs.values()), "the hipGraph path was never taken"
        else:
            gt.hip_collect(m)
            gt.hip_end(m)
        assert int(m.dropout_state()[1]) == end_offset, "the fused pass must leave the dropout offset where k micro-steps leave it"
        outs[mode] = (scal(out), fus["g"], fus["ss"])
        tr.close()
    # (a) the oracle: sum over the micro-batches of the reference's own per-micro-batch loss, each under its own dropout state
    w = {n: sd[n].detach().clone().requires_grad_(True) for n in names}
    wl = {n: sd[n].detach().to(BF).requires_grad_(True) for n in names}
    sdw, sdl = {**sd, **w}, {**{kk: v.to(BF) for kk, v in sd.items()}, **wl}
    ref_losses = []
    for j, b in enumerate(cpu_batches):
        def run_ref(b=b, j=j):
            o = olisa.model_forward(sdw, cfg, **b, inference=False, dropout_state=(seed, j + 1))
            o["loss"].backward()
            return o
        ref_losses.append(scal(gt.oracle("ref", run_ref)))
        gt.oracle("lo", lambda b=b, j=j: olisa.model_forward(sdl, cfg, **mc._bf16_batch(b), inference=False, dropout_state=(seed, j + 1))["loss"].backward())
    for side in ("ref", "lo"):                               # the fused pass is ONE forward whose head rows are the micro-batches' rows in order
        gt.gates[side] = {n: [torch.cat(v, 0)] for n, v in gt.gates[side].items()}
    flips = gt.flipped_rows()
    for kk in ("loss", "ce_loss", "align_loss", "regression_loss"):
        r = sum(l[kk] for l in ref_losses)
        res.append((f"fused accum k={k}: {kk} of the pass vs the oracle's sum over the {k} micro-batches ({r:.4f})", abs(outs["eager"][0][kk] - r), 5e-3 * max(1.0, abs(r))))
        sq = sum(l[kk] for l in seq_losses)
        res.append((f"fused accum k={k}: {kk} of the pass vs the sum of the {k} HIP micro-step values ({sq:.4f})", abs(outs["eager"][0][kk] - sq), 2e-3 * max(1.0, abs(sq))))
    worst, stats = (0.0, ""), []
    for n in names:
        ratio, desc = grad_err(outs["eager"][1][n], w[n].grad, wl[n].grad.float(), floor=3e-4 * k, skip_rows=GateTrace.rows_for(n, flips), stats=stats)
        worst = max(worst, (ratio, f"{n}: {desc}"))
    res.append((f"fused accum k={k}: arena after ONE fused pass vs autograd through the oracle on the {k} micro-batches, worst of {len(names)} tensors = {worst[1]}; "
                "shown as err / tol", worst[0], 1.0))
    res += ratio_summary(stats, f"fused accum k={k}")
    flat = lambda g: torch.cat([g[n].double().flatten() for n in names])
    a, b_ = flat(seq["g"]), flat(outs["eager"][1])
    res.append((f"fused accum k={k}: direction of the whole gradient, fused pass vs {k} micro-steps; shown as 1 - cos", 1.0 - float((a @ b_) / (a.norm() * b_.norm())), 5e-3))
    res.append((f"fused accum k={k}: squared gradient norm, fused vs sequential", abs(outs["eager"][2] - seq["ss"]) / max(seq["ss"], 1e-30), 5e-2))
    la, lb = seq["g"]["lm_head.weight"].double().flatten(), outs["eager"][1]["lm_head.weight"].double().flatten()
    res.append((f"fused accum k={k}: lm_head gradient (per-micro-batch CE means), relative RMS difference fused vs sequential", float((la - lb).norm() / la.norm()), 1e-2))
    same = all(torch.equal(outs["eager"][1][n], outs["hipGraph"][1][n]) for n in names) and outs["eager"][0] == outs["hipGraph"][0]
    res.append((f"fused accum k={k}: the replayed hipGraph of the fused pass leaves the same bits as the eager pass", 0.0 if same else 1.0, 0.5))
    return res


def _mix_batch(c, j, img, K=16, L=24):
    """One image carrying c conversations (BASELINE configs[3]: batch_size = 1 per GPU; 1-3 conversations per sampled image depending on the
    source the 9:3:1 draw picked, utils/dataset.py:499-502), every tensor seeded by the micro-step index j; one right-padded sequence when c > 1."""
    from oracle import cases, seeded
    ids = cases.prompt_ids(c, L, 500 + 7 * j)
    labels = ids.clone()
    labels[:, :10] = -100
    am = torch.ones(c, L, dtype=torch.bool)
    if c > 1:
        am[c - 1, L - 2:] = False
    return dict(images=seeded.uniform((1, 3, img, img), 600 + j, -2, 2), images_clip=seeded.uniform((1, 3, 224, 224), 700 + j, -2, 2),
                input_ids=ids, labels=labels, attention_masks=am, offset=torch.tensor([0, c]),
                sam_segs_list=[(seeded.uniform((K, 256, 256), 800 + j) > 0.4).float()],
                sam_ious_list=[seeded.uniform((c, K), 900 + j, 0, 1).double()], sam_iops_list=[seeded.uniform((c, K), 1000 + j, 0, 1).double()])


def check_mix_window(accum=10, sampler_seed=1):
    """BASELINE configs[3]'s per-GPU workload through the Trainer: batch_size = 1, the source of every sample drawn 9:3:1 by
    `synthetic.HybridSampler` (reference `HybridDataset.__getitem__`, utils/dataset.py:499-502; `--sample_rates 9,3,1`, training.py:66-70), hence a
    window of `grad_accumulation_steps` = 10 micro-batches (training.py:532-547) whose STRUCTURE changes from micro-step to micro-step: 1, 2 or 3
    conversations on the image = N = 1..3 sequences through CLIP + Llama, `offset` = [0, c], the head over c x K rows -- three kernel sequences,
    three hipGraphs.  One accumulation window:
      (a) eager Trainer: every micro-step's losses and the fp32 arena the optimizer consumes, against autograd through `oracle.lisa.model_forward`
          over the same ten batches, each under its own dropout state (seed, j + 1) -- `grad_err` policy, gates traced per pass;
      (b) hipGraph Trainer (one graph per structure, captured on first reuse): the SECOND pass over the window -- mostly replays, the plan tensors
          re-uploaded whenever the structure repeats with other data -- leaves the same BITS in the arena and the same losses as (a)."""
    from llmseg_amd import synthetic
    from llmseg_amd.train import Trainer
    from oracle import lisa as olisa
    from tests import model_checks as mc
    cfg, m, sd, _ = _lora_case("sam")
    names = [n for n, p in m.params.named_parameters() if p.requires_grad]
    prm = dict(m.params.named_parameters())
    sampler = synthetic.HybridSampler((9, 3, 1), seed=sampler_seed)
    draws = [sampler.draw() for _ in range(accum)]
    convs = [c for _, c in draws]
    assert set(convs) == {1, 2, 3}, f"the window must hold all three batch structures, drew {draws}"
    cpu_batches = [mc._round_batch(_mix_batch(c, j, cfg.sam.img)) for j, c in enumerate(convs)]
    batches = [mc._dev(b) for b in cpu_batches]
    plans = [m.make_plan(**b) for b in batches]
    assert len({p.sig for p in plans}) == 3
    seed = 777
    scal = lambda out: {kk: float(v.detach()) for kk, v in out.items() if torch.is_tensor(v) and v.numel() == 1}
    grab = lambda store: (lambda t, ss: store.update(g={n: prm[n]._g32.detach().clone() for n in names}, ss=float(ss)))

    def window(tr, gt=None):
        m.set_dropout_seed(seed, 0)
        out = []
        for b, p in zip(batches, plans):
            out.append(scal(tr.micro_step(b, p)))
            if gt is not None:
                gt.hip_collect(m)
        return out
    # (a) eager
    gt = GateTrace()
    eag = {}
    tr = Trainer(m, lr=0.0, grad_accum=accum, warmup=1, total_steps=10)
    tr.grad_hook = grab(eag)
    gt.hip_begin(m)
    eager_losses = window(tr, gt)
    gt.hip_end(m)
    assert tr.opt_steps == 1 and "g" in eag
    tr.close()
    # (b) hipGraph: window 1 = eager warm-up of each structure + captures, window 2 = replays
    gra = {}
    tr = Trainer(m, lr=0.0, grad_accum=accum, warmup=1, total_steps=10, use_graph=True, graph_warmup=1)
    tr.grad_hook = grab(gra)
    window(tr)
    graph_losses = window(tr)
    torch.cuda.synchronize()
    assert tr.graph_error is None, tr.graph_error
    n_graphs = sum(1 for e in tr._graphs.values() if e["graph"] is not None)
    tr.close()
    # the oracle over the same ten batches
    w = {n: sd[n].detach().clone().requires_grad_(True) for n in names}
    wl = {n: sd[n].detach().to(BF).requires_grad_(True) for n in names}
    sdw, sdl = {**sd, **w}, {**{kk: v.to(BF) for kk, v in sd.items()}, **wl}
    ref_losses = []
    for j, b in enumerate(cpu_batches):
        def run_ref(b=b, j=j):
            o = olisa.model_forward(sdw, cfg, **b, inference=False, dropout_state=(seed, j + 1))
            o["loss"].backward()
            return o
        ref_losses.append(scal(gt.oracle("ref", run_ref)))
        gt.oracle("lo", lambda b=b, j=j: olisa.model_forward(sdl, cfg, **mc._bf16_batch(b), inference=False, dropout_state=(seed, j + 1))["loss"].backward())
    flips = gt.flipped_rows()
    res = [(f"mix 9:3:1 window: conversations per micro-step {convs} (sources {[s for s, _ in draws]}): three structures", 0.0, 0.5),
           (f"mix 9:3:1 window: one hipGraph per batch structure ({n_graphs} captured)", 0.0 if n_graphs == 3 else 1.0, 0.5)]
    for kk in ("loss", "ce_loss", "align_loss", "regression_loss"):
        worst = max(abs(h[kk] - r[kk]) / (5e-3 * max(1.0, abs(r[kk]))) for h, r in zip(eager_losses, ref_losses))
        res.append((f"mix 9:3:1 window: {kk} of every micro-step vs the oracle on the same batch (ref {[round(r[kk], 4) for r in ref_losses]}); shown as err / tol", worst, 1.0))
    worst, stats = (0.0, ""), []
    for n in names:
        ratio, desc = grad_err(eag["g"][n], w[n].grad, wl[n].grad.float(), floor=3e-4 * accum, skip_rows=GateTrace.rows_for(n, flips), stats=stats)
        worst = max(worst, (ratio, f"{n}: {desc}"))
    res.append((f"mix 9:3:1 window: arena after the {accum} micro-steps vs autograd through the oracle over the same batches, worst of {len(names)} tensors = {worst[1]}; "
                "shown as err / tol", worst[0], 1.0))
    res += ratio_summary(stats, "mix 9:3:1 window")
    g64 = [w[n].grad.double().reshape(-1) for n in names]
    ss_ref = float(sum((x * x).sum() for x in g64))
    res.append(("mix 9:3:1 window: squared norm of the accumulated gradient vs the oracle's", abs(eag["ss"] - ss_ref) / max(ss_ref, 1e-30), 5e-2))
    same = all(torch.equal(eag["g"][n], gra["g"][n]) for n in names)
    res.append(("mix 9:3:1 window: the hipGraph trainer's second pass leaves the same bits in the arena as the eager trainer", 0.0 if same else 1.0, 0.5))
    res.append((f"mix 9:3:1 window: losses of every micro-step identical, graph vs eager", 0.0 if graph_losses == eager_losses else 1.0, 0.5))
    return res


def check_window_towers(k=3):
    """the frozen towers (SAM ViT-H, CLIP-L + mm_projector) batched over the accumulation window -- `Trainer.window_step` /
    `encode_window` / `model_forward(tower_visual=, tower_clip=)` (reference: the towers carry no gradient, LISA.py:173-184,242-245;
    clip_encoder.py:41-60 runs under no_grad).
      (1) the plumbing is EXACT: micro-steps fed the tower outputs computed per micro-batch (`encode_towers` at the micro-batch's own row count)
          leave the same BITS in the arena and the same losses as micro-steps that run the towers themselves -- features are inputs;
      (2) one `window_step` over the k micro-batches (towers at k x the rows): the tower outputs against the per-micro-batch ones (bit-equal
          wherever the kernels keep their summation order across row counts -- reported; bounded by 2 bf16 ulps of the feature scale otherwise),
          the arena against (1) in aggregate (direction, norm, lm_head) and bit for bit when the features are;
      (3) `window_step` replayed from hipGraphs with the next window's towers PREFETCHED on the side stream == the eager window, bit for bit;
      (4) at the real width (SAM ViT-H dim 1280, one windowed + one global block + neck; 1024 x 1024): images encoded 3 at a time == encoded
          one by one, bit for bit -- the shapes the benchmark's window pass runs."""
    from llmseg_amd.train import Trainer
    from tests import model_checks as mc
    cfg, m, sd, batch = _lora_case("sam")
    names = [n for n, p in m.params.named_parameters() if p.requires_grad]
    prm = dict(m.params.named_parameters())
    batches = [mc._dev(b) for b in _variant_batches(batch, k)]
    plans = [m.make_plan(**b) for b in batches]
    seed = 515
    scal = lambda out: {kk: float(v.detach()) for kk, v in out.items() if torch.is_tensor(v) and v.numel() == 1}
    grab = lambda store: (lambda t, ss: store.update(g={n: prm[n]._g32.detach().clone() for n in names}, ss=float(ss)))
    flat = lambda g: torch.cat([g[n].double().flatten() for n in names])

    def run(mode, use_graph=False):
        st = {}
        tr = Trainer(m, lr=0.0, grad_accum=k, warmup=1, total_steps=10, use_graph=use_graph, graph_warmup=1)
        tr.grad_hook = grab(st)
        losses = None
        for rep in range(3 if use_graph else 1):
            m.set_dropout_seed(seed, 0)
            if mode == "own":
                losses = [scal(tr.micro_step(b, p)) for b, p in zip(batches, plans)]
            elif mode == "per_micro_batch":
                losses = []
                for b, p in zip(batches, plans):
                    tv, tc = m.encode_towers(b["images"], b["images_clip"])
                    mb = {kk: v for kk, v in b.items() if kk not in ("images", "images_clip")}
                    losses.append(scal(tr.micro_step(dict(mb, images=None, images_clip=None, tower_visual=tv, tower_clip=tc), p)))
            elif mode == "window":
                losses = [scal(o) for o in tr.window_step(batches, plans)]
            else:                                         # the next window's towers issued on the side stream before this window's micro-steps
                tw = st.pop("next", None) or tr.encode_window(batches, prefetch=True)
                st["next"] = tr.encode_window(batches, prefetch=True)
                losses = [scal(o) for o in tr.window_step(batches, plans, towers=tw)]
        torch.cuda.synchronize()
        if use_graph:
            assert tr.graph_error is None, tr.graph_error
            assert any(e["graph"] is not None for e in tr._graphs.values()), "the hipGraph path was never taken"
        tr.close()
        return losses, st["g"], st["ss"]
    l_own, g_own, ss_own = run("own")
    l_pmb, g_pmb, _ = run("per_micro_batch")
    l_win, g_win, ss_win = run("window")
    l_gra, g_gra, _ = run("window_prefetch", use_graph=True)
    res = [("window towers (1): tower outputs as INPUTS, computed per micro-batch: arena elements that differ from the self-computing micro-steps",
            float(sum((g_own[n] != g_pmb[n]).sum() for n in names)), 0.0),
           (f"window towers (1): losses identical ({l_own[0]['loss']:.5f} ..)", 0.0 if l_own == l_pmb else 1.0, 0.5)]
    with torch.no_grad():
        imgs = torch.cat([b["images"] for b in batches]); clips = torch.cat([b["images_clip"] for b in batches])
        vw, cw = m.encode_towers(imgs, clips)
        per = [m.encode_towers(b["images"], b["images_clip"]) for b in batches]
        vp, cp = torch.cat([p[0] for p in per]), torch.cat([p[1] for p in per])
    ulp = lambda t: 2.0 ** -7 * float(t.float().abs().max())
    feq = bool(torch.equal(vw, vp) and torch.equal(cw, cp))
    res.append((f"window towers (2): SAM rows at {k} x the batch vs per micro-batch (bit-equal: {bool(torch.equal(vw, vp))}); max diff", float((vw.float() - vp.float()).abs().max()), 2 * ulp(vp)))
    res.append((f"window towers (2): CLIP -> projector rows (bit-equal: {bool(torch.equal(cw, cp))}); max diff", float((cw.float() - cp.float()).abs().max()), 2 * ulp(cp)))
    if feq:
        res.append(("window towers (2): same tower bits => same arena bits: differing elements", float(sum((g_own[n] != g_win[n]).sum() for n in names)), 0.0))
    a, b_ = flat(g_own), flat(g_win)
    res.append(("window towers (2): direction of the whole gradient, window pass vs self-computing micro-steps; shown as 1 - cos", 1.0 - float((a @ b_) / (a.norm() * b_.norm())), 5e-3))
    res.append(("window towers (2): squared gradient norm", abs(ss_win - ss_own) / max(ss_own, 1e-30), 5e-2))
    la, lb = g_own["lm_head.weight"].double().flatten(), g_win["lm_head.weight"].double().flatten()
    res.append(("window towers (2): lm_head gradient, relative RMS difference", float((la - lb).norm() / la.norm()), 1e-2))
    res.append(("window towers (2): losses of the micro-steps", max(abs(x["loss"] - y["loss"]) for x, y in zip(l_own, l_win)), 2e-3 * max(1.0, abs(l_own[0]["loss"]))))
    res.append(("window towers (3): hipGraph micro-steps + prefetched towers vs the eager window: differing arena elements", float(sum((g_win[n] != g_gra[n]).sum() for n in names)), 0.0))
    res.append(("window towers (3): losses identical", 0.0 if l_win == l_gra else 1.0, 0.5))
    del m
    # (4) full width
    from oracle import cases, sam_encoder as osam, seeded
    scfg = osam.SamCfg(depth=2, global_idx=(1,))
    fcfg = cases.tiny_lisa_cfg("sam")
    fcfg.sam = scfg
    from llmseg_amd import lisa as hip_lisa
    mf = hip_lisa.LISAForCausalLM(mc.to_hip_cfg(fcfg), device=mc.DEV)
    mf.load_state_dict({kk: v.to(BF).float() for kk, v in seeded.fill_state_dict(seeded.lisa_shapes(fcfg), 5).items()}, strict=False)
    img = seeded.uniform((6, 3, 1024, 1024), 19, -2, 2).to(BF).to(mc.DEV)
    clip = seeded.uniform((6, 3, 224, 224), 20, -2, 2).to(BF).to(mc.DEV)
    with torch.no_grad():
        v6, c6 = mf.encode_towers(img, clip)
        two = [mf.encode_towers(img[i:i + 2], clip[i:i + 2]) for i in range(0, 6, 2)]
        one = [mf.encode_towers(img[i:i + 1], clip[i:i + 1]) for i in range(6)]
    v2, v1 = torch.cat([o[0] for o in two]), torch.cat([o[0] for o in one])
    # a K-sliced GEMM plan (short matrices only) sums in another order than the unsliced one: a different row count may pick another plan, and isolated
    # outputs then differ by one bf16 rounding.  Measured on MI355X (round 6): one image at a time (M = 4096 / 4900 rows) vs 3 at once: 896 of 3.1 M
    # elements differ by one ulp.  The bound is 2 ulps either way; the counts are in the line's text.
    # Measured (profiles/r06b_window_towers_lines.txt): two at a time vs six at once -- the benchmark's micro-batch against its window pass -- 0 of 6.3 M
    # elements differ, so the headline's `window_towers` line computes the SAME gradient bits as its micro-steps; one at a time: 1780 by one ulp.
    res.append((f"window towers (4): full-width SAM-H blocks + neck, 6 images at once vs two at a time (the benchmark's micro-batch, M = 8192 / 9800): "
                f"differing elements of {v6.numel()}", float((v6 != v2).sum()), 0.0))
    nd = int((v6 != v1).sum())
    res.append((f"window towers (4): ... vs one at a time (configs[3]'s micro-batch, M = 4096 / 4900: K-sliced plans): {nd} elements differ; max difference",
                float((v6.float() - v1.float()).abs().max()), 2 * ulp(v6)))
    return res


def check_overlap_exchange():
    """`Trainer(overlap_exchange=True)` (the reference's DeepSpeed `overlap_comm`, training.py:321-329): every micro-step's backward
    cut at the Llama output -- half A (lm_head / CE, text_hidden_fcs, the mask-selection head), [the arena tail's all-reduce leaves here on the last
    micro-step of a window], half B (decoder stack, LoRA, embedding rows).  Without a process group nothing is exchanged, so this checks the cut itself:
    the same kernels in the same order on the same data -> the SAME BITS in the arena, the same losses and master weights as the uncut trainer, eagerly and
    as the two-graph replay; and that the arena's tail is exactly what half A finishes (the head of the arena is still zero between the halves)."""
    from llmseg_amd.train import Trainer
    from tests import model_checks as mc
    res = []
    runs = {}
    for mode, kw in (("plain eager", dict()), ("cut eager", dict(overlap_exchange=True)), ("plain graph", dict(use_graph=True, graph_warmup=1)),
                     ("cut graph", dict(use_graph=True, graph_warmup=1, overlap_exchange=True))):
        cfg, m, sd, batch = _lora_case("sam")
        m.set_dropout_seed(21, 0)
        tr = Trainer(m, lr=2e-3, grad_accum=2, warmup=0, total_steps=20, **kw)
        seen = []
        tr.grad_hook = lambda t, ss: seen.append((t.arena.flat.detach().clone(), float(ss)))
        db = mc._dev(batch)
        plan = m.make_plan(**db)
        losses = [float(tr.micro_step(db, plan)["loss"]) for _ in range(6)]
        torch.cuda.synchronize()
        if kw.get("use_graph"):
            assert tr.graph_error is None, tr.graph_error
            ents = [e for e in tr._graphs.values() if e["graph"] is not None]
            assert ents and all((e.get("graph_b") is not None) == bool(kw.get("overlap_exchange")) for e in ents)
        if kw.get("overlap_exchange"):
            # the tail boundary: between the halves of an eager micro-step the arena's head (embedding, LoRA) must still be untouched
            tr.arena.zero_()
            m.__dict__["_split_backward"] = True
            out = m.model_forward(**db, plan=plan)
            root, leaf = m.__dict__.pop("_split_pair"); m.__dict__.pop("_split_backward")
            out["loss"].backward()
            head_nz = float((tr.arena.flat[: tr._tail_start] != 0).sum())
            tail_nz = float((tr.arena.flat[tr._tail_start:] != 0).float().mean())
            root.backward(leaf.grad)
            head_after = float((tr.arena.flat[: tr._tail_start] != 0).sum())
            res.append((f"overlap exchange [{mode}]: after half A the arena's head (embedding + LoRA blocks, {tr._tail_start} elements) is still zero; non-zero elements", head_nz, 0.0))
            res.append((f"overlap exchange [{mode}]: half A filled the tail (non-zero fraction {tail_nz:.3f}) and half B the head ({int(head_after)} elements)",
                        0.0 if (tail_nz > 0.3 and head_after > 0) else 1.0, 0.5))
        runs[mode] = (losses, [a for a, _ in seen], [s_ for _, s_ in seen], torch.cat([w.flatten() for w in tr.opt.master]).clone())
        tr.close()
    ref = runs["plain eager"]
    for mode in ("cut eager", "plain graph", "cut graph"):
        l, arenas, sss, w = runs[mode]
        res.append((f"overlap exchange: [{mode}] losses identical to the uncut eager trainer", 0.0 if l == ref[0] else 1.0, 0.5))
        res.append((f"overlap exchange: [{mode}] arena elements that differ over 3 optimizer steps", float(sum((a != b).sum() for a, b in zip(arenas, ref[1]))), 0.0))
        res.append((f"overlap exchange: [{mode}] master weights that differ after 3 optimizer steps", float((w != ref[3]).sum()), 0.0))
    return res
