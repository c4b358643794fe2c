"""GEMM micro-benchmark on the GPU box: TF/s per staging variant on the hot-path shapes (random data, within-process A/B)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from llmseg_amd import _lib, ops  # noqa: E402

SHAPES = [  # (M, N, K, tag)  B=16 images
    (65536, 3840, 1280, "sam qkv (global)"), (78400, 3840, 1280, "sam qkv (windows)"), (65536, 1280, 1280, "sam proj"),
    (65536, 5120, 1280, "sam lin1"), (65536, 1280, 5120, "sam lin2"),
    (5104, 12288, 4096, "llama qkv"), (5104, 4096, 4096, "llama o"), (5104, 22016, 4096, "llama gate_up"), (5104, 4096, 11008, "llama down"),
    (5104, 32004, 4096, "lm_head"), (4112, 3072, 1024, "clip qkv"), (1000, 520, 128, "edge"), (4096, 4096, 4096, "4096^3"), (8192, 8192, 8192, "8192^3"),
]
if os.environ.get("GEMM_SET") == "b2":   # BASELINE configs[2]: 2 images per micro-step (Llama M = 2 x 319, SAM M = 2 x 4096 / 2 x 4900)
    SHAPES = [(8192, 3840, 1280, "sam qkv (global)"), (9800, 3840, 1280, "sam qkv (windows)"), (8192, 1280, 1280, "sam proj"),
              (8192, 5120, 1280, "sam lin1"), (8192, 1280, 5120, "sam lin2"),
              (638, 12288, 4096, "llama qkv"), (638, 4096, 4096, "llama o"), (638, 22016, 4096, "llama gate_up"), (638, 4096, 11008, "llama down"),
              (638, 4096, 12288, "llama dx(qkv)"), (638, 4096, 22016, "llama dx(gate_up)"), (638, 11008, 4096, "llama dx(down)"),
              (638, 32004, 4096, "lm_head"), (638, 4096, 32064, "lm_head dx"), (32064, 4096, 640, "lm_head dw"), (514, 3072, 1024, "clip qkv"),
              (514, 4096, 1024, "clip fc1")]
if os.environ.get("GEMM_SET") == "clip":  # CLIP-L at two sequences per step (M = 2 x 257) and the mask-selection head's larger products
    SHAPES = [(514, 3072, 1024, "clip qkv"), (514, 1024, 1024, "clip out"), (514, 4096, 1024, "clip fc1"), (514, 1024, 4096, "clip fc2"),
              (512, 2048, 256, "head lin1"), (512, 256, 2048, "head lin2"), (512, 768, 256, "head qkv")]
if os.environ.get("GEMM_SET") == "dec":  # decode steps of generation (one token per sequence): weight streams
    SHAPES = [(m, n, k, f"dec {t}") for m in (1, 4) for n, k, t in ((12288, 4096, "qkv"), (4096, 4096, "o"), (22016, 4096, "gate_up"), (4096, 11008, "down"),
                                                                     (32004, 4096, "lm_head"))]
if os.environ.get("GEMM_SET") == "fused":  # the fused accumulation window: 10 micro-batches of 2 images in one pass (Llama M = 20 x 319)
    SHAPES = [(6380, 12288, 4096, "llama qkv"), (6380, 4096, 4096, "llama o"), (6380, 22016, 4096, "llama gate_up"), (6380, 4096, 11008, "llama down"),
              (6380, 4096, 12288, "llama dx(qkv)"), (6380, 4096, 22016, "llama dx(gate_up)"), (6380, 11008, 4096, "llama dx(down)"),
              (81920, 1280, 1280, "sam proj"), (81920, 5120, 1280, "sam lin1")]
RACE_REPEATS = int(os.environ.get("RACE_REPEATS", "0"))
if os.environ.get("GEMM_SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["GEMM_SHAPES"].split(",")]
def _variant(spec):          # "8" | "9:3" (kernel 9, 3 K-slices) | "5" (auto) | "9:0:2" (kernel 9, no split, 128 x 256 kernel form 2 = two phases / three buffers)
    f = spec.split(":")
    return int(f[0]) | (int(f[1]) << 8 if len(f) > 1 and f[1] else 0) | ((int(f[2]) + 1) << 13 if len(f) > 2 and f[2] else 0)


variants = [_variant(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "2", "8"])]
lib = _lib.load()
torch.manual_seed(0)
print("variants: v & 15 = kernel variant, (v >> 4) - 1 = XCD skew (none = default 13)")
print(f"{'shape':32s} " + " ".join(f"{'v%d:%d:%d' % (v & 15, v >> 8 & 31, (v >> 13 & 3) - 1):>9s}" for v in variants) + "   (TFLOP/s; check = max|v - v0|)")
for M, N, K, tag in SHAPES:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    l2 = os.environ.get("GEMM_L2", "")      # experiment: "a" / "w" / "aw" = that operand's rows all alias row 0 (cache-resident: no HBM latency)
    if "a" in l2: a = a[:1].expand(M, K)
    if "w" in l2: w = w[:1].expand(N, K)
    bias = torch.randn(N, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res, ref, chk = [], None, []
    for v in variants:
        lib.llmseg_gemm_set_variant(v)
        try:
            for _ in range(3):
                ops.gemm(a, w, bias=bias, out=out)
        except RuntimeError:
            res.append(0.0); chk.append(-1.0)
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            ops.gemm(a, w, bias=bias, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        res.append(2.0 * M * N * K / ms / 1e9)
        if ref is None:
            ref = out.clone()
        chk.append((out.float() - ref.float()).abs().max().item())
        if RACE_REPEATS and v == variants[-1]:   # determinism screen: any run-to-run difference is a synchronisation bug
            first, bad = out.clone(), 0
            for _ in range(RACE_REPEATS):
                ops.gemm(a, w, bias=bias, out=out)
                bad += int(not torch.equal(out, first))
            chk.append(float(bad))
    extra = ""
    if os.environ.get("WITH_TORCH"):                       # hipBLASLt through torch.matmul (+ bias add), same data
        for _ in range(3):
            torch.addmm(bias, a, w.t(), out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.addmm(bias, a, w.t(), out=out)
        e1.record()
        torch.cuda.synchronize()
        extra = f"   torch.addmm {2.0 * M * N * K / (e0.elapsed_time(e1) / 10) / 1e9:6.0f}"
    if os.environ.get("GEMM_SET") == "dec":
        extra += "   GB/s " + " ".join(f"{r * 1e12 / (2.0 * M * N * K) * (N * K * 2) / 1e9:7.0f}" for r in res)      # weight bytes / time
    print(f"{tag + f' {M}x{N}x{K}':32s} " + " ".join(f"{r:9.0f}" for r in res) + "   check " + " ".join(f"{c:.1e}" for c in chk) + extra, flush=True)
lib.llmseg_gemm_set_variant(5)
