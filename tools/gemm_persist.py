"""(Historical: needs the library of commit 721598e -- the persistent kernel and the variant bits 15-16 were removed again, profiles/r06r.)
Experiment (GPU box): the persistent form of the 256 x 256 GEMM kernel (gemm_bf16_tn_ppp_kernel: several tiles per workgroup, the next tile's first K-tile issued from inside
the epilogue) against the shipped kernel: bit equality, a race screen (repeated runs must reproduce the bits) and time.  tools/gemm_persist.py [set]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from llmseg_amd import _lib, ops  # noqa: E402

lib = _lib.load()
V8, VP = 8 | (1 << 15), 8 | (2 << 15)
SHAPES = [  # M, N, K, epilogue
    (8192, 3840, 1280, "bias"), (8192, 5120, 1280, "gelu"), (8192, 1280, 1280, "res"), (8192, 1280, 5120, "res"),
    (9800, 3840, 1280, "bias"), (81920, 3840, 1280, "bias"), (81920, 5120, 1280, "gelu"), (81920, 1280, 1280, "res"), (81920, 1280, 5120, "res"),
    (7656, 12288, 4096, "none"), (7656, 4096, 4096, "res"), (8192, 8192, 8192, "bias"), (5000, 1000, 448, "res"),
]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    SHAPES = SHAPES[:5] + SHAPES[-1:]
print(f"{'shape':28s} {'epi':5s} {'shipped us':>11s} {'persist us':>11s} {'ratio':>6s}  differing / race")
for M, N, K, epi in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    bias = None if epi == "none" else torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16) if epi == "res" else None
    kw = dict(bias=bias, residual=res, act=ops.ACT_GELU if epi == "gelu" else ops.ACT_NONE)
    o0, o1 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)

    def t(v, out, n=10):
        lib.llmseg_gemm_set_variant(v)
        for _ in range(2):
            ops.gemm(a, w, out=out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.gemm(a, w, out=out, **kw)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    u0 = t(V8, o0); u1 = t(VP, o1); u0b = t(V8, o0); u1b = t(VP, o1)
    diff = int((o0 != o1).sum())
    race = 0
    first = o1.clone()
    for _ in range(10):
        o1.zero_()
        ops.gemm(a, w, out=o1, **kw)
        race += int(not torch.equal(o1, first))
    print(f"{f'{M}x{N}x{K}':28s} {epi:5s} {min(u0, u0b):11.1f} {min(u1, u1b):11.1f} {min(u1, u1b) / min(u0, u0b):6.3f}  {diff} / {race}", flush=True)
lib.llmseg_gemm_set_variant(5 | (1 << 15))
