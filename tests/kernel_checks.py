"""Per-kernel parity checks: HIP kernel (through the C ABI) vs a plain fp32 PyTorch/oracle computation on the CPU.

Each check returns a list of (name, max_abs_err, tolerance).  Used by tests/test_kernels_gpu.py (asserts) and
tools/gpu_diag.py (prints the whole table without stopping at the first failure).
"""
import math

import torch
import torch.nn.functional as F

from llmseg_amd import ops
from oracle import losses as olosses
from oracle import mask_head as ohead

DEV = "cuda"
BF = torch.bfloat16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def err(got, ref):
    return (got.detach().float().cpu() - ref.float()).abs().max().item()


def tol_bf16(ref, k=1.0):
    return k * (2.0 ** -7) * max(1.0, ref.float().abs().max().item())


# ------------------------------------------------------------------------------------------------------------------ GEMM
def check_gemm():
    out = []
    for i, (M, N, K) in enumerate([(256, 256, 256), (319, 4096, 4096), (130, 200, 72), (1000, 1280, 1280), (64, 1, 128),
                                   (4100, 384, 592)]):
        a, w = rnd(M, K, seed=10 + i), rnd(N, K, seed=20 + i, scale=1 / math.sqrt(K))
        ref = a.float() @ w.float().t()
        got = ops.gemm(a.to(DEV), w.to(DEV))
        out.append((f"gemm {M}x{N}x{K}", err(got, ref), tol_bf16(ref)))
    # full epilogue: bias + gelu + LayerScale + residual
    M, N, K = 300, 520, 256
    a, w, b, g, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / 16), rnd(N, seed=3), rnd(N, seed=4), rnd(M, N, seed=5)
    for act, fn in ((ops.ACT_GELU, F.gelu), (ops.ACT_RELU, F.relu), (ops.ACT_SILU, F.silu), (ops.ACT_SIGMOID, torch.sigmoid),
                    (ops.ACT_QUICKGELU, lambda x: x * torch.sigmoid(1.702 * x))):
        ref = r.float() + g.float() * fn(a.float() @ w.float().t() + b.float())
        got = ops.gemm(a.to(DEV), w.to(DEV), bias=b.to(DEV), act=act, residual=r.to(DEV), gamma=g.to(DEV))
        out.append((f"gemm epilogue act={act}", err(got, ref), tol_bf16(ref, 1.5)))
    # skinny GEMM (M <= 8: the decode step of generation, single-row head GEMMs): every row count, ragged N, K below / across /
    # far above one 512-element K-step, the full epilogue, fp32 output with accumulate, strided A
    for i, (M, N, K) in enumerate([(1, 4096, 4096), (2, 12288, 4096), (3, 37, 24), (4, 4096, 11008), (5, 250, 520), (7, 16, 1024), (8, 1000, 1544)]):
        a, w = rnd(M, K, seed=110 + i), rnd(N, K, seed=120 + i, scale=1 / math.sqrt(K))
        ref = a.float() @ w.float().t()
        out.append((f"gemm skinny {M}x{N}x{K}", err(ops.gemm(a.to(DEV), w.to(DEV)), ref), tol_bf16(ref)))
    M, N, K = 2, 530, 1032
    a, w, b, g, r = rnd(M, K, seed=131), rnd(N, K, seed=132, scale=1 / 32), rnd(N, seed=133), rnd(N, seed=134), rnd(M, N, seed=135)
    ref = r.float() + g.float() * F.silu(0.5 * (a.float() @ w.float().t()) + b.float())
    got = ops.gemm(a.to(DEV), w.to(DEV), bias=b.to(DEV), act=ops.ACT_SILU, residual=r.to(DEV), gamma=g.to(DEV), alpha=0.5)
    out.append(("gemm skinny epilogue (alpha, bias, silu, gamma, residual)", err(got, ref), tol_bf16(ref, 1.5)))
    big = rnd(3, 3 * 256, seed=136)
    w = rnd(45, 256, seed=137)
    acc0 = torch.randn(3, 45, generator=torch.Generator().manual_seed(138))
    o = acc0.clone().to(DEV)
    ops.gemm(big.to(DEV)[:, 256:512], w.to(DEV), out=o, out_f32=True, accumulate=True)
    out.append(("gemm skinny f32-out accumulate strided-A", err(o, acc0 + big[:, 256:512].float() @ w.float().t()), 1e-3))
    # decode-step fusions of the skinny route: RMSNorm / SwiGLU applied to the A rows on load == norm / swiglu launch followed by the GEMM
    for M, N, K in ((1, 4096, 4096), (4, 530, 1032), (7, 256, 24)):
        x, nw, w, r = rnd(M, K, seed=151, scale=2.0), rnd(K, seed=152), rnd(N, K, seed=153, scale=1 / math.sqrt(K)), rnd(M, N, seed=154)
        xd, nwd, wd, rd = x.to(DEV), nw.to(DEV), w.to(DEV), r.to(DEV)
        two = ops.gemm(ops.norm(xd, nwd, None, eps=1e-6, rms=True), wd, residual=rd)
        one = ops.gemm(xd, wd, residual=rd, a_norm_w=nwd, a_norm_eps=1e-6)
        out.append((f"gemm skinny + RMSNorm on load {M}x{N}x{K} == norm then gemm", 0.0 if torch.equal(one, two) else 1.0, 0.0))
        gu = rnd(M, 2 * K, seed=155, scale=1.5).to(DEV)
        two = ops.gemm(ops.swiglu(gu, K), wd, residual=rd)
        one = ops.gemm(gu, wd, residual=rd, a_swiglu=True)
        out.append((f"gemm skinny + SwiGLU on load {M}x{N}x{K} == swiglu then gemm", 0.0 if torch.equal(one, two) else 1.0, 0.0))
    # fp32 output, odd ldc, alpha, strided A (a view into a wider buffer)
    M, N, K = 200, 27, 80
    big = rnd(M, 3 * K, seed=6)
    w = rnd(N, K, seed=7)
    ref = 0.5 * (big[:, K:2 * K].float() @ w.float().t())
    bigd = big.to(DEV)
    got = ops.gemm(bigd[:, K:2 * K], w.to(DEV), alpha=0.5, out_f32=True)
    out.append(("gemm f32-out strided-A N=27", err(got, ref), 1e-3))
    # strided-batched (per-head q . R^T as used for SAM rel-pos)
    heads, hd, rows, R = 2, 80, 196, 27
    q = rnd(rows, 3 * heads * hd, seed=8)
    tab = rnd(32, hd, seed=9)
    qd, tabd = q.to(DEV), tab.to(DEV)
    o = torch.zeros(heads, rows, 32, device=DEV, dtype=torch.float32)
    ops.gemm_batched(qd, tabd, o, M=rows, N=R, K=hd, lda=3 * heads * hd, ldw=hd, ldc=32, batch=heads, sA=hd, sW=0, sC=rows * 32)
    ref = torch.stack([q[:, h * hd:(h + 1) * hd].float() @ tab[:R].float().t() for h in range(heads)])
    out.append(("gemm batched rel-pos", err(o[:, :, :R], ref), 1e-3))
    # the 256 x 256 ping-pong kernel: picked automatically for long-K / many-tile shapes, and forced (variant 8) on ragged
    # edge tiles, the K = 128 minimum, the fused epilogue, fp32 output and a strided batch
    from llmseg_amd import _lib
    M, N, K = 4096, 4096, 2048
    a, w = rnd(M, K, seed=31), rnd(N, K, seed=32, scale=1 / math.sqrt(K))
    ref = a.float() @ w.float().t()
    out.append((f"gemm auto->ping-pong {M}x{N}x{K}", err(ops.gemm(a.to(DEV), w.to(DEV)), ref), tol_bf16(ref)))
    lib = _lib.load()
    lib.llmseg_gemm_set_variant(8)
    try:
        for i, (M, N, K) in enumerate([(1000, 520, 128), (257, 300, 192), (513, 256, 1280)]):
            a, w = rnd(M, K, seed=40 + i), rnd(N, K, seed=50 + i, scale=1 / math.sqrt(K))
            ref = a.float() @ w.float().t()
            out.append((f"gemm ping-pong {M}x{N}x{K}", err(ops.gemm(a.to(DEV), w.to(DEV)), ref), tol_bf16(ref)))
        M, N, K = 300, 520, 256
        a, w, b, g, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / 16), rnd(N, seed=3), rnd(N, seed=4), rnd(M, N, seed=5)
        ref = r.float() + g.float() * F.gelu(a.float() @ w.float().t() + b.float())
        got = ops.gemm(a.to(DEV), w.to(DEV), bias=b.to(DEV), act=ops.ACT_GELU, residual=r.to(DEV), gamma=g.to(DEV))
        out.append(("gemm ping-pong epilogue gelu", err(got, ref), tol_bf16(ref, 1.5)))
        ref = 0.5 * (a.float() @ w.float().t())
        out.append(("gemm ping-pong f32-out", err(ops.gemm(a.to(DEV), w.to(DEV), alpha=0.5, out_f32=True), ref), 1e-3))
        # extension K-tile (fused low-rank update): C = A W^T + A2 W2^T + bias, incl. K = 64 (the extension is the second tile)
        for M, N, K in ((513, 300, 192), (300, 520, 64)):
            a, w, b = rnd(M, K, seed=71), rnd(N, K, seed=72, scale=1 / math.sqrt(K)), rnd(N, seed=73)
            a2, w2 = rnd(M, 64, seed=74), rnd(N, 64, seed=75, scale=1 / 8)
            a2[:, 16:] = 0
            ref = a.float() @ w.float().t() + a2.float() @ w2.float().t() + b.float()
            got = ops.gemm(a.to(DEV), w.to(DEV), bias=b.to(DEV), a2=a2.to(DEV), w2=w2.to(DEV))
            out.append((f"gemm ping-pong + extension tile {M}x{N}x{K}", err(got, ref), tol_bf16(ref)))
        nb, M, N, K = 3, 260, 300, 128
        ab, wb = rnd(nb * M, K, seed=61), rnd(nb * N, K, seed=62, scale=1 / 8)
        ob = torch.zeros(nb, M, N, device=DEV, dtype=torch.float32)
        ops.gemm_batched(ab.to(DEV), wb.to(DEV), ob, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, batch=nb, sA=M * K, sW=N * K, sC=M * N)
        ref = torch.stack([ab[i * M:(i + 1) * M].float() @ wb[i * N:(i + 1) * N].float().t() for i in range(nb)])
        out.append(("gemm ping-pong batched", err(ob, ref), 1e-3))
    finally:
        lib.llmseg_gemm_set_variant(5)
    # the 128 x 256 ping-pong kernel (variant 9): same cases as the 256 x 256 one (its half-tile DMA ring holds MI + 4 = 6 instructions)
    lib.llmseg_gemm_set_variant(9)
    try:
        for i, (M, N, K) in enumerate([(1000, 520, 128), (257, 300, 192), (638, 512, 1280), (129, 256, 4096)]):
            a, w = rnd(M, K, seed=140 + i), rnd(N, K, seed=150 + i, scale=1 / math.sqrt(K))
            ref = a.float() @ w.float().t()
            out.append((f"gemm ping-pong 128x256 {M}x{N}x{K}", err(ops.gemm(a.to(DEV), w.to(DEV)), ref), tol_bf16(ref)))
        M, N, K = 300, 520, 256
        a, w, b, g, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / 16), rnd(N, seed=3), rnd(N, seed=4), rnd(M, N, seed=5)
        ref = r.float() + g.float() * F.gelu(a.float() @ w.float().t() + b.float())
        got = ops.gemm(a.to(DEV), w.to(DEV), bias=b.to(DEV), act=ops.ACT_GELU, residual=r.to(DEV), gamma=g.to(DEV))
        out.append(("gemm ping-pong 128x256 epilogue gelu", err(got, ref), tol_bf16(ref, 1.5)))
        for M, N, K in ((513, 300, 192), (300, 520, 64)):
            a, w, b = rnd(M, K, seed=71), rnd(N, K, seed=72, scale=1 / math.sqrt(K)), rnd(N, seed=73)
            a2, w2 = rnd(M, 64, seed=74), rnd(N, 64, seed=75, scale=1 / 8)
            a2[:, 16:] = 0
            ref = a.float() @ w.float().t() + a2.float() @ w2.float().t() + b.float()
            got = ops.gemm(a.to(DEV), w.to(DEV), bias=b.to(DEV), a2=a2.to(DEV), w2=w2.to(DEV))
            out.append((f"gemm ping-pong 128x256 + extension tile {M}x{N}x{K}", err(got, ref), tol_bf16(ref)))
    finally:
        lib.llmseg_gemm_set_variant(5)
    # split-K (K-slices write fp32 slabs into the workspace, one reduce launch applies the epilogue): forced slice counts incl. an
    # uneven last slice (K = 1344 = 21 tiles in 4 slices of 6, 6, 6, 3), both tile heights, bf16 / fp32 / accumulating output
    M, N, K = 638, 520, 1344
    a, w, b, r = rnd(M, K, seed=91), rnd(N, K, seed=92, scale=1 / math.sqrt(K)), rnd(N, seed=93), rnd(M, N, seed=94)
    ref = r.float() + F.relu(a.float() @ w.float().t() + b.float())
    for v, S in ((8, 4), (9, 4), (9, 7), (8, 2)):
        lib.llmseg_gemm_set_variant(v | (S << 8))
        try:
            got = ops.gemm(a.to(DEV), w.to(DEV), bias=b.to(DEV), act=ops.ACT_RELU, residual=r.to(DEV))
            out.append((f"gemm split-K variant {v} x{S} bias+relu+residual", err(got, ref), tol_bf16(ref, 1.5)))
            c32 = torch.full((M, N), 2.0, device=DEV, dtype=torch.float32)
            ops.gemm(a.to(DEV), w.to(DEV), out=c32, accumulate=True, alpha=0.5)
            out.append((f"gemm split-K variant {v} x{S} fp32 +=", err(c32, 2.0 + 0.5 * (a.float() @ w.float().t())), 2e-3))
        finally:
            lib.llmseg_gemm_set_variant(5)
    # split-K + extension K-tile (the LoRA dX GEMM at M = 2 x 319: the rank-8 products are one more slab)
    M, N, K = 638, 512, 4096
    a, w = rnd(M, K, seed=95), rnd(N, K, seed=96, scale=1 / math.sqrt(K))
    a2, w2 = rnd(M, 64, seed=97), rnd(N, 64, seed=98, scale=1 / 8)
    a2[:, 16:] = 0
    ref = a.float() @ w.float().t() + a2.float() @ w2.float().t()
    for v, S in ((9, 5), (8, 3), (5, 0)):
        lib.llmseg_gemm_set_variant(v | (S << 8))
        try:
            out.append((f"gemm split-K variant {v} x{S} + extension tile", err(ops.gemm(a.to(DEV), w.to(DEV), a2=a2.to(DEV), w2=w2.to(DEV)), ref), tol_bf16(ref, 1.5)))
        finally:
            lib.llmseg_gemm_set_variant(5)
    # automatic dispatch on the short-and-long shapes of BASELINE configs[2] (M = 2 x 319): whatever the cost model picks must agree
    for i, (M, N, K) in enumerate([(638, 4096, 4096), (638, 1024, 11008), (514, 1024, 4096), (638, 12288, 1024)]):
        a, w = rnd(M, K, seed=160 + i), rnd(N, K, seed=170 + i, scale=1 / math.sqrt(K))
        ref = a.float() @ w.float().t()
        out.append((f"gemm auto {M}x{N}x{K}", err(ops.gemm(a.to(DEV), w.to(DEV)), ref), tol_bf16(ref)))
    # fp32 accumulate (gradient arena): every kernel family, incl. the transposed-operand layouts of dW = dY^T X
    M, N, K = 200, 264, 256
    a, w = rnd(M, K, seed=181), rnd(N, K, seed=182, scale=1 / 16)
    for v in (0, 2, 8, 9):
        lib.llmseg_gemm_set_variant(v)
        try:
            c32 = torch.full((M, N), -1.0, device=DEV, dtype=torch.float32)
            ops.gemm(a.to(DEV), w.to(DEV), out=c32, accumulate=True)
            ops.gemm(a.to(DEV), w.to(DEV), out=c32, accumulate=True)
            out.append((f"gemm fp32 += (variant {v}, twice)", err(c32, -1.0 + 2.0 * (a.float() @ w.float().t())), 2e-3))
        finally:
            lib.llmseg_gemm_set_variant(5)
    dy, x = rnd(300, 136, seed=183), rnd(300, 72, seed=184)               # dW [136, 72] += dY^T X
    c32 = torch.full((136, 72), 0.5, device=DEV, dtype=torch.float32)
    ops.gemm(dy.to(DEV), x.to(DEV), out=c32, trans_a=True, trans_w=True, accumulate=True)
    out.append(("gemm fp32 += transposed operands", err(c32, 0.5 + dy.float().t() @ x.float()), 2e-3))
    # the same through the automatic dispatch on a small shape (128 x 128 kernel + accumulate launch)
    M, N, K = 200, 264, 256
    a, w = rnd(M, K, seed=81), rnd(N, K, seed=82, scale=1 / 16)
    a2, w2 = rnd(M, 64, seed=83), rnd(N, 64, seed=84, scale=1 / 8)
    ref = a.float() @ w.float().t() + a2.float() @ w2.float().t()
    out.append(("gemm extension tile, fallback path", err(ops.gemm(a.to(DEV), w.to(DEV), a2=a2.to(DEV), w2=w2.to(DEV)), ref), tol_bf16(ref, 1.5)))
    return out


def check_gemm_hot_shapes():
    """The GEMMs the benchmark spends its time in, at their real sizes (24 and 2 images per step), checked on a strided sample of output
    rows / columns against an fp32 reference computed on the device in fp64-free chunks (A sample rows x full W)."""
    out = []
    shapes = [(98304, 5120, 1280, ops.ACT_GELU, "sam lin1 B=24"), (7656, 22016, 4096, ops.ACT_NONE, "llama gate_up B=24"),
              (638, 4096, 11008, ops.ACT_NONE, "llama down B=2 (split-K)"), (638, 12288, 4096, ops.ACT_NONE, "llama qkv B=2 (128x256 tiles)")]
    g = torch.Generator(device=DEV).manual_seed(0)
    for M, N, K, act, tag in shapes:
        a = (torch.rand((M, K), device=DEV, generator=g) * 2 - 1).to(BF)
        w = ((torch.rand((N, K), device=DEV, generator=g) * 2 - 1) * (3.0 / K) ** 0.5).to(BF)
        b = torch.randn((N,), device=DEV, generator=g).to(BF)
        got = ops.gemm(a, w, bias=b, act=act)
        rows = torch.arange(0, M, max(1, M // 97), device=DEV)
        rows = torch.cat([rows, torch.tensor([M - 1, max(0, M - 130)], device=DEV)])        # incl. the ragged last tile
        # the reference is computed on the HOST in fp32 (independent of the device's vendor GEMM): ~100 sampled rows x all of W
        ref = a[rows].float().cpu() @ w.float().cpu().t() + b.float().cpu()
        if act == ops.ACT_GELU:
            ref = F.gelu(ref)
        e = (got[rows].float().cpu() - ref).abs().max().item()
        out.append((f"gemm hot shape {tag} {M}x{N}x{K} (sampled rows, host fp32 reference)", e, tol_bf16(ref, 1.5)))
    return out


def check_gemm_norm_out():
    """`llmseg_gemm_args.norm_out` (the residual-stream projection with the stream's next RMSNorm as a second output) against the two-call
    spelling gemm -> norm: same bits in both outputs, on the shapes whose K-slice reduce launch writes the norm (Llama o_proj / down_proj at
    2 images per step) and on shapes that take the library's own gemm + norm route (one K slice, M past the workgroup-norm range, narrow N)."""
    out = []
    g = torch.Generator(device=DEV).manual_seed(3)
    for M, N, K, tag in [(638, 4096, 4096, "o_proj B=2"), (638, 4096, 11008, "down_proj B=2"), (319, 4096, 4096, "one image"),
                         (77, 4096, 4096, "ragged M"), (7656, 4096, 4096, "B=24 (gemm + norm)"), (638, 1024, 4096, "narrow N (gemm + norm)"),
                         (638, 4096, 128, "short K (gemm + norm)")]:
        a = (torch.rand((M, K), device=DEV, generator=g) * 2 - 1).to(BF)
        w = ((torch.rand((N, K), device=DEV, generator=g) * 2 - 1) * (3.0 / K) ** 0.5).to(BF)
        r = torch.randn((M, N), device=DEV, generator=g).to(BF)
        nw = (1.0 + 0.1 * torch.randn((N,), device=DEV, generator=g)).to(BF)
        y0 = ops.gemm(a, w, residual=r)
        h0 = ops.norm(y0, nw, None, eps=1e-6, rms=True)
        h1 = torch.full((M, N), float("nan"), device=DEV, dtype=BF)
        y1 = ops.gemm(a, w, residual=r, norm_w=nw, norm_eps=1e-6, norm_out=h1)
        ref = (a.float() @ w.float().t() + r.float()).cpu()
        out.append((f"gemm norm_out {tag}: C vs fp32", err(y1, ref), tol_bf16(ref, 1.5)))
        out.append((f"gemm norm_out {tag}: C bits == gemm", float((y1.view(torch.int16) != y0.view(torch.int16)).sum().item()), 0.0))
        out.append((f"gemm norm_out {tag}: norm bits == gemm -> norm", float((h1.view(torch.int16) != h0.view(torch.int16)).sum().item()), 0.0))
    return out


# ------------------------------------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, scale, bias=None):
    s = (q.float() @ k.float().transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias
    return torch.softmax(s, -1) @ v.float()


def check_gemm_fx():
    """Round 6: the fused Llama-layer epilogues of llmseg_gemm_bf16 (fx = rope / swiglu / swiglu_bwd) against the product followed by the
    pointwise launch each replaces -- BIT FOR BIT -- at the benchmark's shapes (the fused 128 x 256 kernel: one library launch) and at small
    shapes (the library's own two-launch route), and against an fp32 reference."""
    from llmseg_amd import _lib
    lib = _lib.load()
    out = []

    def launches(fn):
        torch.cuda.synchronize()
        n0 = lib.llmseg_launch_count()
        r = fn()
        return r, lib.llmseg_launch_count() - n0

    bits = lambda a, b: float((a.float() - b.float()).abs().max())
    for (M, H, T, tag) in ((638, 4096, 319, "bench shape"), (2552, 4096, 319, "8 sequences: the 256 x 256 tile's fused forms"), (50, 256, 25, "small")):
        D = H
        x, w = rnd(M, H, seed=1, scale=1.0).to(DEV), rnd(3 * D, H, seed=2, scale=H ** -0.5).to(DEV)
        a2, w2 = rnd(M, 64, seed=3, scale=0.3).to(DEV), rnd(3 * D, 64, seed=4, scale=0.1).to(DEV)
        ang = torch.outer(torch.arange(T).float(), 1.0 / (10000 ** (torch.arange(0, 128, 2).float() / 128)))
        cos, sin = ang.cos().contiguous().to(DEV), ang.sin().contiguous().to(DEV)
        ref = ops.gemm(x, w, a2=a2, w2=w2)
        ops.rope_(ref, cos, sin, M, T, 2 * D // 128, 128, 3 * D)
        got, n = launches(lambda: ops.gemm(x, w, a2=a2, w2=w2, rope=(cos, sin, T, 2 * D)))
        out.append((f"gemm fx rope == gemm + rope ({tag}, {n} launch{'es' if n > 1 else ''}) (bits)", bits(got, ref), 0.0))
        if M == 638:
            out.append(("gemm fx rope ran as ONE launch at the bench shape", float(n), 1.0))
            y = (x.float() @ w.float().t() + a2.float() @ w2.float().t()).cpu()
            q = y.view(M, 3 * D // 128, 128).clone()
            pos = torch.arange(M) % T
            c, s_ = ang.cos()[pos][:, None, :], ang.sin()[pos][:, None, :]
            nh = 2 * D // 128
            a_, b_ = q[:, :nh, :64].clone(), q[:, :nh, 64:].clone()
            q[:, :nh, :64], q[:, :nh, 64:] = a_ * c - b_ * s_, b_ * c + a_ * s_
            out.append(("gemm fx rope vs fp32", err(got, q.reshape(M, 3 * D)), tol_bf16(q, 2.0)))
        # gate|up + swiglu, and the backward of it on the dX product of down_proj
        I = 11008 if M >= 638 else 192
        wg = rnd(2 * I, H, seed=5, scale=H ** -0.5).to(DEV)
        gu_ref = ops.gemm(x, wg)
        h_ref = ops.swiglu(gu_ref, I)
        h = torch.empty(M, I, device=DEV, dtype=BF)
        gu, n = launches(lambda: ops.gemm(x, wg, swiglu_out=h))
        out.append((f"gemm fx swiglu: gate|up == gemm ({tag}, {n} launch{'es' if n > 1 else ''}) (bits)", bits(gu, gu_ref), 0.0))
        out.append((f"gemm fx swiglu: h == swiglu(gemm) ({tag}) (bits)", bits(h, h_ref), 0.0))
        if M == 638:
            out.append(("gemm fx swiglu ran as ONE launch at the bench shape", float(n), 1.0))
            yf = (x.float() @ wg.float().t()).cpu()
            out.append(("gemm fx swiglu vs fp32", err(h, F.silu(yf[:, :I]) * yf[:, I:]), tol_bf16(yf, 2.0)))
        dy, wd_t = rnd(M, H, seed=6, scale=0.5).to(DEV), rnd(I, H, seed=7, scale=H ** -0.5).to(DEV)      # dX = dY @ Wd with Wd^T stored [I, H]
        dh_ref = ops.gemm(dy, wd_t)
        dgu_ref = ops.swiglu_bwd(gu_ref, dh_ref, I)
        dgu, n = launches(lambda: ops.gemm(dy, wd_t, swiglu_bwd_of=gu_ref))
        out.append((f"gemm fx swiglu_bwd == swiglu_bwd(gemm) ({tag}, {n} launch{'es' if n > 1 else ''}) (bits)", bits(dgu, dgu_ref), 0.0))
        if M == 638:
            out.append(("gemm fx swiglu_bwd ran as ONE launch at the bench shape", float(n), 1.0))
    return out


def check_gemm_normbwd_tail():
    """Round 6: llmseg_gemm_args.nb_x -- dX product + LoRA dX + pre-norm backward + residual gradient as ONE call -- against the product,
    `lora_apply_` and `norm_bwd` launches it replaces, BIT FOR BIT: the K-sliced route at the benchmark's shapes (dX of q|k|v with the LoRA term and
    dropout, dX of gate|up without), and a small shape on the library's own three-launch route."""
    from llmseg_amd import _lib
    lib = _lib.load()
    out = []
    bits = lambda a, b: float((a.float() - b.float()).abs().max())
    rng = torch.tensor([1234, 7], device=DEV, dtype=torch.int64)
    for (M, N, K, lora, tag) in ((638, 4096, 12288, True, "dX(q|k|v) + LoRA, dropout"), (638, 4096, 22016, False, "dX(gate|up)"), (50, 256, 512, True, "small")):
        d, wt = rnd(M, K, seed=1, scale=0.3).to(DEV), rnd(N, K, seed=2, scale=K ** -0.5).to(DEV)
        x, w, dres = rnd(M, N, seed=3).to(DEV), rnd(N, seed=4).to(DEV), rnd(M, N, seed=5, scale=0.2).to(DEV)
        t2, a0, a1 = rnd(M, 64, seed=6, scale=0.3).to(DEV), rnd(8, N, seed=7, scale=0.1).to(DEV), rnd(8, N, seed=8, scale=0.1).to(DEV)
        drop = (rng, 6, 0.05)
        ref = ops.gemm(d, wt)
        if lora:
            ops.lora_apply_(ref, t2, a0, w_rn=True, drop=drop, w2=a1)
        ref = ops.norm_bwd(ref, x, w, 1e-6, True, dres=dres)
        torch.cuda.synchronize()
        n0 = lib.llmseg_launch_count()
        got = ops.gemm(d, wt, normbwd=(x, w, 1e-6, True, dres), nb_lora=(t2, a0, a1, 1.0, drop) if lora else None)
        n = lib.llmseg_launch_count() - n0
        out.append((f"gemm nb tail == gemm + lora_apply + norm_bwd: {tag} ({n} launches) (bits)", bits(got, ref), 0.0))
        if lora:
            # the LoRA operand as lora_down's UNFINISHED K-slice partials: the tail finishes them, uses them and writes t (for the weight gradients)
            dq, bt = rnd(M, 2 * N, seed=9, scale=0.3).to(DEV), rnd(16, N, seed=10, scale=0.1).to(DEV)
            t_ref = torch.empty(M, 64, device=DEV, dtype=BF)
            ops.lora_down(dq[:, :N], bt[:8], alpha=2.0, out=t_ref, zero_cols=48, x2=dq[:, N:], w2=bt[8:])
            ref2 = ops.gemm(d, wt)
            ops.lora_apply_(ref2, t_ref, a0, w_rn=True, drop=drop, w2=a1)
            ref2 = ops.norm_bwd(ref2, x, w, 1e-6, True, dres=dres)
            t_out = torch.full((M, 64), 7.0, device=DEV, dtype=BF)
            _, part, pS, pscale = ops.lora_down(dq[:, :N], bt[:8], alpha=2.0, out=t_out, zero_cols=48, x2=dq[:, N:], w2=bt[8:], parts=True)
            got2 = ops.gemm(d, wt, normbwd=(x, w, 1e-6, True, dres), nb_lora=(t_out, a0, a1, 1.0, drop, part, pS, pscale, 48))
            out.append((f"gemm nb tail fed lora_down's K-slice partials (S = {pS}): dx == finished route, {tag} (bits)", bits(got2, ref2), 0.0))
            out.append((f"gemm nb tail fed partials: the t operand it writes == lora_down's finish, {tag} (bits)", bits(t_out, t_ref), 0.0))
        if M == 638:
            out.append((f"gemm nb tail: {tag} ran as the K-sliced kernel + ONE tail launch", float(n), 2.0))
    return out


def check_gemm_delta_tail():
    """Round 6: llmseg_gemm_args.dl_o -- dX(o_proj) with the attention backward's delta = rowsum(dO * O) written by the K-sliced product's reduce launch --
    against the product followed by llmseg_attn_bwd's own delta launch: dO, delta and therefore dq | dk | dv BIT FOR BIT (2 x 319 rows, 32 heads x 128)."""
    from llmseg_amd import _lib
    lib = _lib.load()
    out = []
    bits = lambda a, b: float((a.float() - b.float()).abs().max())
    N, T, H, hd = 2, 319, 32, 128
    D, M = H * hd, N * T
    dy, wt = rnd(M, D, seed=1, scale=0.3).to(DEV), rnd(D, D, seed=2, scale=D ** -0.5).to(DEV)
    qkv = rnd(M, 3 * D, seed=3, scale=0.5).to(DEV)
    km = torch.ones(N, T, dtype=torch.uint8)
    km[1, 300:] = 0
    km = km.to(DEV)
    lse = torch.empty(N, H, T, device=DEV)
    o = ops.attention_packed(qkv, N, T, H, hd, causal=True, key_mask=km, lse=lse)
    ld, st, dst = 3 * D, (T * 3 * D, hd, 3 * D), (T * D, hd, D)

    def bwd(do, delta):
        dqkv = torch.empty_like(qkv)
        ops.attention_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], o, do, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], lse, batch=N, heads=H, Nq=T, Nk=T, head_dim=hd, q_strides=st,
                          k_strides=st, v_strides=st, o_strides=dst, do_strides=dst, dq_strides=st, dk_strides=st, dv_strides=st, causal=True, key_mask=km, delta=delta)
        return dqkv
    do_ref = ops.gemm(dy, wt)
    g_ref = bwd(do_ref, None)
    delta = torch.empty(N, H, T, device=DEV)
    torch.cuda.synchronize()
    n0 = lib.llmseg_launch_count()
    do = ops.gemm(dy, wt, delta_of=(o, delta, H, T))
    n = lib.llmseg_launch_count() - n0
    g = bwd(do, delta)
    dref = (do_ref.float().view(N, T, H, hd) * o.float().view(N, T, H, hd)).sum(-1).permute(0, 2, 1)
    out.append((f"gemm delta tail: dO == gemm ({n} launches) (bits)", bits(do, do_ref), 0.0))
    out.append(("gemm delta tail: delta vs fp32 rowsum(dO * O)", float((delta - dref).abs().max()), 1e-4 * max(1.0, float(dref.abs().max()))))
    out.append(("gemm delta tail: attention backward with the precomputed delta == with its own delta launch (bits)", bits(g, g_ref), 0.0))
    out.append(("gemm delta tail ran as the K-sliced kernel + ONE tail launch", float(n), 2.0))
    return out


def check_attention():
    out = []
    for hd, B, H, N in [(32, 2, 8, 256), (64, 1, 4, 257), (128, 2, 2, 319), (80, 3, 2, 196), (64, 1, 2, 1100)]:
        D = H * hd
        qkv = rnd(B * N, 3 * D, seed=hd + N, scale=1.0)
        o = ops.attention_packed(qkv.to(DEV), B, N, H, hd)
        x = qkv.view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
        ref = _attn_ref(x[0], x[1], x[2], hd ** -0.5).transpose(1, 2).reshape(B * N, D)
        out.append((f"attn plain hd={hd} N={N}", err(o, ref), tol_bf16(ref, 2.0)))
    # causal + key padding (Llama form)
    hd, B, H, N = 128, 2, 2, 319
    D = H * hd
    qkv = rnd(B * N, 3 * D, seed=77)
    km = torch.ones(B, N, dtype=torch.uint8)
    km[1, 300:] = 0
    o = ops.attention_packed(qkv.to(DEV), B, N, H, hd, causal=True, key_mask=km.to(DEV))
    x = qkv.view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    bias = torch.zeros(B, 1, N, N)
    bias.masked_fill_(torch.ones(N, N, dtype=torch.bool).triu(1)[None, None], -1e30)
    bias.masked_fill_((km == 0)[:, None, None, :], -1e30)
    ref = _attn_ref(x[0], x[1], x[2], hd ** -0.5, bias).transpose(1, 2).reshape(B * N, D)
    out.append(("attn causal+keymask hd=128", err(o, ref), tol_bf16(ref, 2.0)))
    # decomposed rel-pos: generic grid (14x14 window) and the tile-aligned 64-wide grid
    for (gh, gw, B, H) in [(14, 14, 3, 2), (30, 30, 1, 2), (64, 64, 1, 1)]:
        hd, N = 80, gh * gw
        D = H * hd
        qkv = rnd(B * N, 3 * D, seed=gh, scale=0.7)
        R = 2 * max(gh, gw) - 1
        ld = (R + 3) // 4 * 4
        relh = (torch.randn(B, H, N, ld, generator=torch.Generator().manual_seed(gh + 1)) * 0.5)
        relw = (torch.randn(B, H, N, ld, generator=torch.Generator().manual_seed(gh + 2)) * 0.5)
        # kernel layout is [heads][batch*Nq][ld] (what the strided-batched q.R^T GEMM writes)
        o = ops.attention_packed(qkv.to(DEV), B, N, H, hd, rel_h=relh.transpose(0, 1).contiguous().to(DEV),
                                 rel_w=relw.transpose(0, 1).contiguous().to(DEV), rel_ld=ld, grid_hw=(gh, gw))
        x = qkv.view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
        qi = torch.arange(N)
        qh_, qw_ = qi // gw, qi % gw
        ih = (qh_[:, None] - qh_[None, :] + gh - 1)       # [Nq, Nk] index into rel_h rows
        iw = (qw_[:, None] - qw_[None, :] + gw - 1)
        bias = torch.gather(relh, 3, ih[None, None].expand(B, H, N, N)) + torch.gather(relw, 3, iw[None, None].expand(B, H, N, N))
        ref = _attn_ref(x[0], x[1], x[2], hd ** -0.5, bias).transpose(1, 2).reshape(B * N, D)
        out.append((f"attn rel-pos grid {gh}x{gw}", err(o, ref), tol_bf16(ref, 2.0)))
    # fused window rel-pos: pass the tables, q.R^T is computed inside the kernel (SAM 14x14, hd 80)
    gh = gw = 14
    hd, N, B, H = 80, 196, 3, 2
    D = H * hd
    qkv = rnd(B * N, 3 * D, seed=41, scale=0.7)
    th, tw = torch.zeros(32, hd, dtype=BF), torch.zeros(32, hd, dtype=BF)
    th[:27], tw[:27] = rnd(27, hd, seed=42, scale=0.3), rnd(27, hd, seed=43, scale=0.3)
    o = ops.attention_packed(qkv.to(DEV), B, N, H, hd, rel_tab_h=th.to(DEV), rel_tab_w=tw.to(DEV), grid_hw=(gh, gw))
    x = qkv.view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    qi = torch.arange(N)
    qh_, qw_ = qi // gw, qi % gw
    Gh, Gw = x[0].float() @ th[:27].float().t(), x[0].float() @ tw[:27].float().t()          # [B,H,N,27]
    ih = (qh_[:, None] - qh_[None, :] + gh - 1)[None, None].expand(B, H, N, N)
    iw = (qw_[:, None] - qw_[None, :] + gw - 1)[None, None].expand(B, H, N, N)
    bias = torch.gather(Gh, 3, ih) + torch.gather(Gw, 3, iw)
    ref = _attn_ref(x[0], x[1], x[2], hd ** -0.5, bias).transpose(1, 2).reshape(B * N, D)
    out.append(("attn fused window rel-pos (tables)", err(o, ref), tol_bf16(ref, 2.0)))
    # o_row_map: scatter rows (window un-partition): reverse order, skip every 5th query
    hd, B, H, N = 80, 2, 2, 196
    D = H * hd
    qkv = rnd(B * N, 3 * D, seed=5)
    rm = torch.arange(B * N - 1, -1, -1, dtype=torch.int32)
    rm[::5] = -1
    od = torch.zeros(B * N, D, device=DEV, dtype=BF)
    ops.attention_packed(qkv.to(DEV), B, N, H, hd, out=od, o_row_map=rm.to(DEV))
    x = qkv.view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    full = _attn_ref(x[0], x[1], x[2], hd ** -0.5).transpose(1, 2).reshape(B * N, D)
    ref = torch.zeros(B * N, D)
    keep = rm >= 0
    ref[rm[keep].long()] = full[keep]
    out.append(("attn o_row_map", err(od, ref), tol_bf16(ref, 2.0)))
    # window gather (llmseg_attn_args.win_grid): partition + zero padding + un-partition inside the kernel == the row-map route over padded windows, bit for bit,
    # and against the fp32 computation of the padded windows (image_encoder.py:178-183, 263-318); grid 30 -> 3 x 3 windows (12 padded rows / columns), grid 64 -> SAM's 5 x 5
    for g, Bi, H in [(30, 2, 2), (64, 1, 3), (14, 3, 1)]:
        hd, ws = 80, 14
        D, nw = H * hd, (g + 13) // 14
        tok = rnd(Bi * g * g, 3 * D, seed=900 + g, scale=0.7)
        pad = rnd(3 * D, seed=901 + g, scale=0.7)
        th, tw = torch.zeros(32, hd, dtype=BF), torch.zeros(32, hd, dtype=BF)
        th[:27], tw[:27] = rnd(27, hd, seed=902 + g, scale=0.3), rnd(27, hd, seed=903 + g, scale=0.3)
        yy, xx = torch.meshgrid(torch.arange(g), torch.arange(g), indexing="ij")
        part1 = (((yy // ws) * nw + xx // ws) * ws * ws + (yy % ws) * ws + xx % ws).reshape(-1)
        per_img = nw * nw * ws * ws
        part = torch.cat([part1 + b * per_img for b in range(Bi)])
        unpart = torch.full((Bi * per_img,), -1, dtype=torch.int32)
        unpart[part] = torch.arange(Bi * g * g, dtype=torch.int32)
        win = pad[None].repeat(Bi * per_img, 1)
        win[part] = tok
        kw = dict(rel_tab_h=th.to(DEV), rel_tab_w=tw.to(DEV), grid_hw=(14, 14))
        o_map = torch.zeros(Bi * g * g, D, device=DEV, dtype=BF)
        ops.attention_packed(win.to(DEV), Bi * nw * nw, ws * ws, H, hd, out=o_map, o_row_map=unpart.to(DEV), **kw)
        o_gat = torch.zeros(Bi * g * g, D, device=DEV, dtype=BF)
        ops.attention_packed(tok.to(DEV), Bi * nw * nw, ws * ws, H, hd, out=o_gat, win_pad=(g, pad.to(DEV)), **kw)
        out.append((f"attn window gather grid {g}: differing elements vs the row-map route", float((o_gat != o_map).sum()), 0.0))
        Bw, N = Bi * nw * nw, ws * ws
        x = win.view(Bw, N, 3, H, hd).permute(2, 0, 3, 1, 4)
        qi = torch.arange(N)
        qh_, qw_ = qi // ws, qi % ws
        Gh, Gw = x[0].float() @ th[:27].float().t(), x[0].float() @ tw[:27].float().t()
        ih = (qh_[:, None] - qh_[None, :] + ws - 1)[None, None].expand(Bw, H, N, N)
        iw = (qw_[:, None] - qw_[None, :] + ws - 1)[None, None].expand(Bw, H, N, N)
        full = _attn_ref(x[0], x[1], x[2], hd ** -0.5, torch.gather(Gh, 3, ih) + torch.gather(Gw, 3, iw)).transpose(1, 2).reshape(Bw * N, D)
        ref = full[part]
        out.append((f"attn window gather grid {g} vs fp32", err(o_gat, ref), tol_bf16(ref, 2.0)))
    # cross attention shape of the head: Nq != Nk (1 query over K keys), separate tensors
    hd, H, Nq, Nk = 32, 8, 1, 256
    q, k, v = rnd(Nq, H * hd, seed=1), rnd(Nk, H * hd, seed=2), rnd(Nk, H * hd, seed=3)
    od = torch.empty(Nq, H * hd, device=DEV, dtype=BF)
    ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), od, batch=1, heads=H, Nq=Nq, Nk=Nk, head_dim=hd,
                  q_strides=(0, hd, H * hd), k_strides=(0, hd, H * hd), v_strides=(0, hd, H * hd), o_strides=(0, hd, H * hd))
    sp = lambda t, n: t.view(n, H, hd).transpose(0, 1)
    ref = _attn_ref(sp(q, Nq), sp(k, Nk), sp(v, Nk), hd ** -0.5).transpose(0, 1).reshape(Nq, H * hd)
    out.append(("attn cross 1xK hd=32", err(od, ref), tol_bf16(ref, 2.0)))
    return out


def check_decode_attn():
    """Decode step in one launch: RoPE + KV append + one-query attention over the cache, with and without the key split."""
    out = []
    hd, cap = 128, 700
    g = torch.Generator().manual_seed(3)
    ang = torch.rand(cap, hd // 2, generator=g) * 6.28
    cos, sin = ang.cos().float(), ang.sin().float()
    for N, H, pos, split in [(1, 32, 329, True), (1, 32, 329, False), (2, 4, 0, True), (3, 2, 15, True), (2, 8, 16, True), (4, 32, 640, True), (1, 2, 77, True)]:
        D = H * hd
        qkv = rnd(N, 3 * D, seed=pos + N)
        kc, vc = rnd(N, cap, D, seed=pos + 11), rnd(N, cap, D, seed=pos + 12)
        posd = torch.tensor([pos, pos + 1], dtype=torch.int32, device=DEV)
        kd, vd = kc.to(DEV), vc.to(DEV)
        scratch = ops.decode_attn_scratch(N, H, DEV) if split else None
        o = ops.decode_attn(qkv.to(DEV), cos.to(DEV), sin.to(DEV), kd, vd, posd, H, hd, scratch=scratch)
        x = qkv.float().view(N, 3, H, hd)

        def rot(t):
            a, b = t[..., :hd // 2], t[..., hd // 2:]
            return torch.cat([a * cos[pos] - b * sin[pos], b * cos[pos] + a * sin[pos]], -1).to(BF).float()
        q, kn, vn = rot(x[:, 0]), rot(x[:, 1]), x[:, 2]
        kr, vr = kc.clone(), vc.clone()
        kr[:, pos] = kn.reshape(N, D).to(BF)
        vr[:, pos] = vn.reshape(N, D).to(BF)
        K = kr[:, :pos + 1].float().view(N, pos + 1, H, hd).transpose(1, 2)
        V = vr[:, :pos + 1].float().view(N, pos + 1, H, hd).transpose(1, 2)
        ref = _attn_ref(q[:, :, None, :], K, V, hd ** -0.5).reshape(N, D)
        tag = f"N={N} heads={H} pos={pos}" + ("" if split else " one workgroup per head")
        out.append((f"decode_attn out {tag}", err(o, ref), tol_bf16(ref, 2.0)))
        out.append((f"decode_attn k cache {tag}", err(kd, kr), tol_bf16(kr, 1.0)))
        out.append((f"decode_attn v cache {tag}", err(vd, vr), 0.0))
        # the two-launch route (rope_kv_append + attn_fwd with a device-side key count)
        k2, v2, q2 = kc.to(DEV), vc.to(DEV), qkv.to(DEV)
        ops.rope_kv_append_(q2, cos.to(DEV), sin.to(DEV), k2, v2, posd, H, hd)
        out.append((f"decode_attn k cache == rope_kv_append {tag}", err(kd, k2.cpu()), 0.0))
    return out


# ------------------------------------------------------------------------------------------------------------- pointwise
def check_pointwise():
    out = []
    x, w, b = rnd(333, 1280, seed=1, scale=2.0), rnd(1280, seed=2), rnd(1280, seed=3)
    ref = F.layer_norm(x.float(), (1280,), w.float(), b.float(), 1e-6)
    out.append(("layernorm", err(ops.norm(x.to(DEV), w.to(DEV), b.to(DEV), eps=1e-6), ref), tol_bf16(ref)))
    x, w = rnd(100, 4096, seed=4, scale=3.0), rnd(4096, seed=5)
    xf = x.float()
    ref = w.float() * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(BF).float()
    out.append(("rmsnorm", err(ops.norm(x.to(DEV), w.to(DEV), None, eps=1e-6, rms=True), ref), tol_bf16(ref)))
    # short + wide (64 <= rows < 2048, cols >= 2048): the workgroup-per-row kernels, forward and backward (frozen weight, with the residual gradient)
    for rows, cols in ((638, 4096), (70, 2048), (129, 5120)):
        x, w, b = rnd(rows, cols, seed=40, scale=2.0), rnd(cols, seed=41), rnd(cols, seed=42)
        xf = x.float()
        ref = w.float() * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(BF).float()
        out.append((f"rmsnorm {rows}x{cols} (workgroup per row)", err(ops.norm(x.to(DEV), w.to(DEV), None, eps=1e-6, rms=True), ref), tol_bf16(ref)))
        ref = F.layer_norm(xf, (cols,), w.float(), b.float(), 1e-5)
        out.append((f"layernorm {rows}x{cols} (workgroup per row)", err(ops.norm(x.to(DEV), w.to(DEV), b.to(DEV), eps=1e-5), ref), tol_bf16(ref)))
        dy, dres = rnd(rows, cols, seed=43), rnd(rows, cols, seed=44)
        for rms in (True, False):
            xr = xf.clone().requires_grad_(True)
            yr = (w.float() * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6))) if rms else F.layer_norm(xr, (cols,), w.float(), b.float(), 1e-5)
            yr.backward(dy.float())
            ref = xr.grad + dres.float()
            got = ops.norm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), 1e-6 if rms else 1e-5, rms, dres=dres.to(DEV))
            out.append((f"{'rms' if rms else 'layer'}norm backward + residual gradient {rows}x{cols} (workgroup per row)", err(got, ref), tol_bf16(ref)))
    # row_map (window partition): map row r -> 2r+1, others stay zero
    x, w, b = rnd(50, 160, seed=6), rnd(160, seed=7), rnd(160, seed=8)
    rm = (torch.arange(50, dtype=torch.int32) * 2 + 1)
    y = torch.zeros(101, 160, device=DEV, dtype=BF)
    ops.norm(x.to(DEV), w.to(DEV), b.to(DEV), eps=1e-6, out=y, row_map=rm.to(DEV))
    ref = torch.zeros(101, 160)
    ref[rm.long()] = F.layer_norm(x.float(), (160,), w.float(), b.float(), 1e-6)
    out.append(("layernorm row_map", err(y, ref), tol_bf16(ref)))
    # rope on q|k of a packed qkv
    T, nh, hd, N = 37, 2, 128, 3
    qkv = rnd(N * T, 3 * nh * hd, seed=9)
    inv = 1.0 / (10000 ** (torch.arange(0, hd, 2).float() / hd))
    ang = torch.outer(torch.arange(T).float(), inv)
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    d = qkv.to(DEV).clone()
    ops.rope_(d, cos.to(DEV), sin.to(DEV), N * T, T, 2 * nh, hd, 3 * nh * hd)
    xr = qkv.float().view(N, T, 3, nh, hd)
    c2, s2 = torch.cat([cos, cos], -1)[None, :, None], torch.cat([sin, sin], -1)[None, :, None]
    rot = lambda t: torch.cat([-t[..., hd // 2:], t[..., :hd // 2]], -1)
    ref = xr.clone()
    for j in (0, 1):
        ref[:, :, j] = xr[:, :, j] * c2 + rot(xr[:, :, j]) * s2
    out.append(("rope", err(d, ref.view(N * T, -1)), tol_bf16(ref)))
    gu = rnd(77, 2 * 512, seed=10)
    ref = F.silu(gu[:, :512].float()) * gu[:, 512:].float()
    out.append(("swiglu", err(ops.swiglu(gu.to(DEV), 512), ref), tol_bf16(ref)))
    x, a = rnd(3 * 50, 64, seed=11), rnd(50, 64, seed=12)
    ref = x.float() + a.float().repeat(3, 1)
    out.append(("add_rows", err(ops.add_rows(x.to(DEV), a.to(DEV)), ref), tol_bf16(ref)))
    for p, Hh in ((14, 56), (16, 64)):
        img = rnd(2, 3, Hh, Hh, seed=p)
        ld = (3 * p * p + 7) // 8 * 8
        got = ops.patchify(img.to(DEV), p, ld)
        ref = F.unfold(img.float(), p, stride=p).transpose(1, 2).reshape(-1, 3 * p * p)
        pad = got[:, 3 * p * p:].float().abs().max().item() if ld > 3 * p * p else 0.0
        out.append((f"patchify p={p}", max(err(got[:, :3 * p * p], ref), pad), 0.0))
    got = ops.patchify(rnd(2, 3, 28, 28, seed=3).to(DEV), 14, 592, rows_per_img=5, row_off=1)
    ref = torch.zeros(10, 588)
    ref.view(2, 5, 588)[:, 1:] = F.unfold(rnd(2, 3, 28, 28, seed=3).float(), 14, stride=14).transpose(1, 2)
    out.append(("patchify cls-offset", err(got[:, :588], ref), 0.0))
    x = rnd(2, 6, 5, 16, seed=13)
    got = ops.im2col3x3(x.to(DEV), 2, 6, 5, 16)
    ref = F.unfold(x.float().permute(0, 3, 1, 2), 3, padding=1)               # [B, C*9, HW], index c*9 + tap
    ref = ref.view(2, 16, 9, 30).permute(0, 3, 2, 1).reshape(60, 144)         # -> [B*HW, tap*C + c]
    out.append(("im2col3x3", err(got, ref), 0.0))
    N, L, P, Hd, V = 3, 12, 5, 64, 100
    ids = torch.randint(0, V, (N, L), generator=torch.Generator().manual_seed(1))
    ids[:, 2] = -200
    ids[1, 2], ids[1, 7] = 5, -200
    emb, feats = rnd(V, Hd, seed=14), rnd(N, P, Hd, seed=15)
    got = ops.embed_splice(ids.to(DEV), emb.to(DEV), feats.to(DEV), P)
    wide = torch.cat([rnd(N, 1, Hd, seed=17), feats], 1).to(DEV)              # a CLS row in front of every image's block
    got2 = ops.embed_splice(ids.to(DEV), emb.to(DEV), wide.view(-1, Hd)[1:], P, feats_stride_n=(P + 1) * Hd)
    ref = torch.stack([torch.cat([emb[ids[n, :int((ids[n] == -200).nonzero())]], feats[n],
                                  emb[ids[n, int((ids[n] == -200).nonzero()) + 1:]]]) for n in range(N)])
    out.append(("embed_splice", err(got, ref), 0.0))
    out.append(("embed_splice strided feats", err(got2, ref), 0.0))
    x = rnd(40, 256, seed=16)
    idx = torch.tensor([3, 39, 0, 3], dtype=torch.int64)
    out.append(("gather_rows", err(ops.gather_rows(x.to(DEV), idx.to(DEV)), x[idx]), 0.0))
    return out


# ------------------------------------------------------------------------------------------------------------------ head
def check_head():
    out = []
    K, Cc, g, S = 16, 256, 64, 256
    feat = rnd(1, Cc, g, g, seed=1)                                            # reference layout [1,C,g,g]
    gen = torch.Generator().manual_seed(2)
    segs = (torch.rand(K, S, S, generator=gen) > 0.7).to(BF)
    segs[1] = torch.rand(S, S, generator=gen).to(BF)                           # a soft mask
    segs[2] = 0                                                                # an empty proposal
    up = ohead.upsample_feats(feat.float(), S)[0]
    ref = ohead.mask_pooling(up, segs.float())
    got = ops.upsample_maskpool(feat[0].permute(1, 2, 0).reshape(g * g, Cc).contiguous().to(DEV), segs.to(DEV), g, S)
    out.append(("upsample_maskpool", err(got, ref), tol_bf16(ref)))
    # BASELINE configs[4] size: K = 512 proposals; the raw pull-back (segs . U, fp32) and the mask areas against the exact adjoint of
    # F.interpolate obtained by autograd, and the pooled features against the oracle
    K5 = 512
    segs5 = (torch.rand(K5, S, S, generator=gen) > 0.6).to(BF)
    segs5[7] = torch.rand(S, S, generator=gen).to(BF)
    x0 = torch.zeros(1, K5, g, g, requires_grad=True)
    (F.interpolate(x0, size=(S, S), mode="bilinear", align_corners=False)[0] * segs5.float()).sum().backward()
    featc = feat[0].permute(1, 2, 0).reshape(g * g, Cc).contiguous().to(DEV)
    got5, pb5, ws5 = ops.upsample_maskpool(featc, segs5.to(DEV), g, S, want_aux=True)
    out.append(("mask pull-back K=512 vs autograd adjoint", err(pb5.view(K5, g, g), x0.grad[0]), 2e-4 * x0.grad.abs().max().item()))
    out.append(("mask areas K=512", err(ws5, segs5.float().flatten(1).sum(1)), 2e-2))
    ref5 = ohead.mask_pooling(up, segs5.float())
    out.append(("upsample_maskpool K=512", err(got5, ref5), tol_bf16(ref5)))
    Kp, D = 256, 256
    e, t = rnd(Kp, D, seed=3), rnd(D, seed=4)
    ref = ohead.cosine_scores(t.float()[None], e.float())[0]
    out.append(("cosine_scores", err(ops.cosine_scores(t.to(DEV), e.to(DEV)), ref), 2e-4))
    gen = torch.Generator().manual_seed(5)
    gi, gp = torch.rand(Kp, generator=gen), torch.rand(Kp, generator=gen)
    pr = torch.rand(Kp, generator=gen).to(BF)
    ef, tf, pf = e.float().requires_grad_(True), t.float().requires_grad_(True), pr.float().requires_grad_(True)
    la = olosses.softmax_align(ef, tf[None], gi[:, None])
    lr = olosses.iop_regression(pf[:, None], gp[:, None])
    la.backward(); lr.backward()
    o, d_e, d_t, d_p = ops.align_reg_loss(e.to(DEV), t.to(DEV), gi.to(DEV), pr.to(DEV), gp.to(DEV), want_grads=True)
    out.append(("align loss", abs(o[0].item() - la.item()), 1e-3 * max(1.0, abs(la.item()))))
    out.append(("regression loss", abs(o[1].item() - lr.item()), 1e-3 * max(1.0, abs(lr.item()))))
    out.append(("align d_e", err(d_e, ef.grad), 1e-3 * max(1e-3, ef.grad.abs().max().item())))
    out.append(("align d_t", err(d_t, tf.grad), 1e-3 * max(1e-3, tf.grad.abs().max().item())))
    out.append(("regression d_pred", err(d_p, pf.grad), 1e-3 * max(1e-3, pf.grad.abs().max().item())))
    # batched form: three items in one launch == three single launches (bit for bit: same kernel, different block index)
    R = 3
    eb, tb = rnd(R * Kp, D, seed=13).view(R, Kp, D), rnd(R, D, seed=14)
    gib, gpb, prb = torch.rand(R, Kp, generator=gen), torch.rand(R, Kp, generator=gen), torch.rand(R, Kp, generator=gen).to(BF)
    ob, deb, dtb, dpb = ops.align_reg_loss(eb.to(DEV), tb.to(DEV), gib.to(DEV), prb.to(DEV), gpb.to(DEV), want_grads=True)
    worst = 0.0
    for r in range(R):
        o1, de1, dt1, dp1 = ops.align_reg_loss(eb[r].contiguous().to(DEV), tb[r].contiguous().to(DEV), gib[r].contiguous().to(DEV),
                                               prb[r].contiguous().to(DEV), gpb[r].contiguous().to(DEV), want_grads=True)
        worst = max(worst, err(ob[r], o1.cpu()), err(deb[r], de1.cpu()), err(dtb[r], dt1.cpu()), err(dpb[r], dp1.cpu()))
    out.append(("align/regression batched == single", worst, 0.0))
    x = torch.randn(3, 64, 64, generator=gen) * 3
    y = (torch.rand(3, 64, 64, generator=gen) > 0.5).float()
    o = ops.dice_bce(x.to(DEV), y.to(DEV), 3)
    out.append(("dice", abs(o[0].item() - olosses.dice(x, y, 3).item()), 1e-4))
    out.append(("bce", abs(o[1].item() - olosses.sigmoid_ce(x, y, 3).item()), 1e-4))
    N, T, V = 2, 9, 32004
    logits = rnd(N, T, V, seed=6, scale=2.0)
    labels = torch.randint(0, V, (N, T), generator=gen)
    labels[:, :3] = -100
    acc = ops.ce_loss(logits.to(DEV), labels.to(DEV))
    ref = F.cross_entropy(logits[:, :-1].float().reshape(-1, V), labels[:, 1:].reshape(-1), ignore_index=-100)
    out.append(("shifted CE", abs(acc[0].item() / acc[1].item() - ref.item()), 1e-3))
    return out


def check_metric(golden_loader=None):
    """gIoU bookkeeping kernel vs the oracle restatement (and the fixture recorded from the reference's intersectionAndUnionGPU)."""
    from oracle import cases
    out = []
    g = golden_loader("iou_metric.pt")["IUT"] if golden_loader else None
    for n, (pred, tgt) in enumerate(cases.iou_metric_cases()):
        got = ops.intersection_union(pred.to(torch.uint8).to(DEV), tgt.to(torch.uint8).to(DEV)).cpu()
        i, u, t = olosses.intersection_and_union(pred, tgt, 2, 255)
        ref = torch.cat([i, u, t]).long()
        out.append((f"intersection/union case {n}", float((got - ref).abs().max()), 0.0))
        if g is not None:
            out.append((f"intersection/union case {n} vs reference fixture", float((got.float() - g[n].reshape(-1)).abs().max()), 0.0))
    return out


def check_validate_body():
    """Fused union + nearest-resize + I/U kernel vs the oracle restatement of validate_threshold's loop body."""
    from oracle import metric
    out = []
    gen = torch.Generator().manual_seed(5)
    for n, (H, W, K, Hg, Wg) in enumerate([(300, 427, 20, 300, 427), (1024, 768, 50, 512, 384), (64, 64, 3, 1024, 1024)]):
        segs = (torch.rand(H, W, K, generator=gen) > 0.9).to(torch.uint8)
        gt = (torch.rand(Hg, Wg, generator=gen) > 0.6).to(torch.uint8)
        if n == 1:
            gt[torch.rand(Hg, Wg, generator=gen) > 0.9] = 255
        piou = torch.rand(K, generator=gen)
        if n == 2:
            piou[:] = 0.1                                                              # nothing selected: empty prediction
        i, u, t, _ = metric.union_resize_iou(segs, piou, gt)
        got = ops.union_resize_iou(segs.to(DEV), (piou > 0.5).to(torch.uint8).to(DEV), gt.to(DEV)).cpu()
        out.append((f"union+resize+I/U case {n}", float((got - torch.cat([i, u, t]).long()).abs().max()), 0.0))
        # the arg-max variant (`validate`): one proposal, scored at the ground truth's own resolution
        sim = torch.rand(K, generator=gen)
        i, u, t, _ = metric.argmax_iou(segs, sim, gt)
        sel = torch.zeros(K, dtype=torch.uint8)
        sel[int(torch.argmax(sim))] = 1
        got = ops.union_resize_iou(segs.to(DEV), sel.to(DEV), gt.to(DEV), out_size=None).cpu()
        out.append((f"arg-max proposal + resize-to-gt + I/U case {n}", float((got - torch.cat([i, u, t]).long()).abs().max()), 0.0))
    return out


ALL = [check_gemm, check_attention, check_pointwise, check_head, check_metric, check_validate_body]


def check_head_f32():
    """The fp32-activation head kernels (csrc/head_f32.hip; ABI 7) against float64 torch on the host: an fp32 FMA chain differs from the exact
    result by ~sqrt(K) ulps, so the bounds are a few 1e-6 relative to the output scale -- three orders below the bf16 route's."""
    import torch.nn.functional as F
    from llmseg_amd import ops
    res = []
    f64 = lambda t: t.detach().double().cpu()
    for (M, N, K, w_kn, act, use_b, use_r) in ((70, 130, 50, False, ops.ACT_NONE, True, True), (256, 256, 256, False, ops.ACT_RELU, True, False),
                                               (512, 2048, 256, False, ops.ACT_SIGMOID, True, False), (3, 256, 4096, False, ops.ACT_RELU, True, False),
                                               (256, 256, 4096, True, ops.ACT_NONE, False, False), (33, 65, 17, True, ops.ACT_NONE, True, True)):
        x = rnd(M, K, seed=M + N).float().to(DEV)
        w = rnd(*((K, N) if w_kn else (N, K)), seed=K + 1, scale=1.0 / K ** 0.5).to(BF).to(DEV)
        b = rnd(N, seed=5).to(BF).to(DEV) if use_b else None
        r = rnd(M, N, seed=6).float().to(DEV) if use_r else None
        got = ops.linear_f32(x, w, b, act, r, w_kn=w_kn, alpha=0.5 if use_r else 1.0)
        wm = f64(w) if w_kn else f64(w).t()
        ref = (0.5 if use_r else 1.0) * (f64(x) @ wm) + (f64(b) if use_b else 0.0)
        ref = torch.relu(ref) if act == ops.ACT_RELU else torch.sigmoid(ref) if act == ops.ACT_SIGMOID else ref
        ref = ref + (f64(r) if use_r else 0.0)
        res.append((f"linear_f32 {M}x{N}x{K} w_kn={w_kn} act={act}", (f64(got) - ref).abs().max().item(), 4e-6 * max(1.0, ref.abs().max().item())))
    # strided input rows (a column block of a wider matrix, as the head slices q | k | v)
    xw = rnd(40, 96, seed=9).float().to(DEV)
    w = rnd(24, 32, seed=10).to(BF).to(DEV)
    got = ops.linear_f32(xw[:, 32:64], w)
    res.append(("linear_f32 on a strided column block", (f64(got) - f64(xw[:, 32:64]) @ f64(w).t()).abs().max().item(), 4e-6 * 8))
    for rows, D in ((5, 256), (513, 256), (7, 100)):
        x = (rnd(rows, D, seed=rows) * 3 + 1).float().to(DEV)
        w, b = (rnd(D, seed=2) + 1).to(BF).to(DEV), rnd(D, seed=3).to(BF).to(DEV)
        ref = F.layer_norm(f64(x), (D,), f64(w), f64(b), 1e-5)
        res.append((f"layernorm_f32 {rows}x{D}", (f64(ops.layernorm_f32(x, w, b, 1e-5)) - ref).abs().max().item(), 4e-6 * max(1.0, ref.abs().max().item())))
    for (Bn, H, Nq, Nk, hd) in ((2, 8, 256, 256, 32), (3, 8, 1, 300, 32), (1, 4, 130, 70, 64), (2, 8, 512, 512, 32)):
        D = H * hd
        qkv = rnd(Bn * max(Nq, Nk), 3 * D, seed=Nq + Nk).float().to(DEV)
        q = qkv[: Bn * Nq, :D]; k = qkv[: Bn * Nk, D:2 * D]; v = qkv[: Bn * Nk, 2 * D:]
        o = torch.empty((Bn * Nq, D), device=DEV, dtype=torch.float32)
        ops.attention_f32(q, k, v, o, Bn, H, Nq, Nk, hd, (Nq * 3 * D, hd, 3 * D), (Nk * 3 * D, hd, 3 * D), (Nk * 3 * D, hd, 3 * D), (Nq * D, hd, D))
        sp = lambda t, n: f64(t).reshape(Bn, n, H, hd).transpose(1, 2)
        ref = (torch.softmax(sp(q, Nq) @ sp(k, Nk).transpose(-1, -2) / hd ** 0.5, -1) @ sp(v, Nk)).transpose(1, 2).reshape(Bn * Nq, D)
        res.append((f"attention_f32 B={Bn} heads={H} Nq={Nq} Nk={Nk} hd={hd}", (f64(o) - ref).abs().max().item(), 4e-6 * max(1.0, ref.abs().max().item())))
    t, e = rnd(256, seed=1).float().to(DEV), rnd(300, 256, seed=2).float().to(DEV)
    ref = (f64(e) / f64(e).norm(dim=-1, keepdim=True)) @ (f64(t) / f64(t).norm())
    res.append(("cosine_f32 300x256", (f64(ops.cosine_f32(t, e)) - ref).abs().max().item(), 2e-6))
    segs = (rnd(20, 256, 256, seed=3) > 0.2).to(BF).to(DEV)
    feat = rnd(64 * 64, 256, seed=4).to(BF).to(DEV)
    pb, ws = ops.mask_pullback_f32(segs, 64, 256)
    up = F.interpolate(f64(feat).t().reshape(1, 256, 64, 64), size=(256, 256), mode="bilinear", align_corners=False)[0].flatten(1)      # [C, S*S]
    wf = f64(segs).flatten(1)
    ref = (wf @ up.t()) / (wf.sum(-1, keepdim=True) + 1e-8)
    got = ops.linear_f32(pb, feat, w_kn=True) / (ws[:, None] + 1e-8)
    res.append(("fp32 mask pooling (pull-back . features / sum) vs upsample-then-pool in float64", (f64(got) - ref).abs().max().item(), 2e-5 * max(1.0, ref.abs().max().item())))
    return res
