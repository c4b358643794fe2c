"""GPU box: the three fused Llama-layer GEMM epilogues (llmseg_gemm_args.fx) beside the product + pointwise launch each replaces, at the
2-image shapes (M = 638), timed as hipGraph replays of 16 back-to-back calls.  Tuning aid.   python tools/fx_bench.py [M=638]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from llmseg_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 638
H, I, T = 4096, 11008, 319
dev = torch.device("cuda", 0)
BF = torch.bfloat16
r = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(BF)
x, wqkv, a2, w2 = r(M, H), r(3 * H, H, sc=H ** -0.5), r(M, 64, sc=0.3), r(3 * H, 64, sc=0.1)
ang = torch.outer(torch.arange(T, device=dev).float(), 1.0 / (10000 ** (torch.arange(0, 128, 2, device=dev).float() / 128)))
cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
wgu, wd_t = r(2 * I, H, sc=H ** -0.5), r(I, H, sc=H ** -0.5)
qkv, gu, h, dgu, dh = torch.empty(M, 3 * H, device=dev, dtype=BF), torch.empty(M, 2 * I, device=dev, dtype=BF), torch.empty(M, I, device=dev, dtype=BF), \
    torch.empty(M, 2 * I, device=dev, dtype=BF), torch.empty(M, I, device=dev, dtype=BF)
dy = r(M, H, sc=0.5)


def timed(name, fn, reps=16, iters=20):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:58s} {e0.elapsed_time(e1) / (iters * reps) * 1e3:8.2f} us", flush=True)


def qkv_unfused():
    ops.gemm(x, wqkv, a2=a2, w2=w2, out=qkv)
    ops.rope_(qkv, cos, sin, M, T, 64, 128, 3 * H)


def gu_unfused():
    ops.gemm(x, wgu, out=gu)
    ops.swiglu(gu, I, out=h)


def dgu_unfused():
    ops.gemm(dy, wd_t, out=dh)
    ops.swiglu_bwd(gu, dh, I)


for _ in range(2):
    timed("q|k|v (+LoRA tile) then rope", qkv_unfused)
    timed("q|k|v (+LoRA tile) with fx rope", lambda: ops.gemm(x, wqkv, a2=a2, w2=w2, out=qkv, rope=(cos, sin, T, 2 * H)))
    timed("q|k|v (+LoRA tile) alone", lambda: ops.gemm(x, wqkv, a2=a2, w2=w2, out=qkv))
    timed("gate|up then swiglu", gu_unfused)
    timed("gate|up with fx swiglu", lambda: ops.gemm(x, wgu, out=gu, swiglu_out=h))
    timed("gate|up alone", lambda: ops.gemm(x, wgu, out=gu))
    timed("dX(down) then swiglu_bwd", dgu_unfused)
    timed("dX(down) with fx swiglu_bwd", lambda: ops.gemm(dy, wd_t, out=dgu, swiglu_bwd_of=gu))
    timed("dX(down) alone", lambda: ops.gemm(dy, wd_t, out=dh))
