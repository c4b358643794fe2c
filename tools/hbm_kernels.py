"""GPU box: the HBM-bound kernels of the path at the benchmark's shapes -- algorithmic bytes (every operand read / written once),
duration (HIP events, median of repeated launches) and GB/s against the 8 TB/s HBM peak.  Run under `rocprofv3 --pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` (tools/pmc_traffic.sh) the same launches give the counter bytes per kernel; tools/pmc_summary.py joins the two.
Usage: python tools/hbm_kernels.py [images_per_step=24] > gpurun_out/hbm_kernels.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from llmseg_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
REPS = int(os.environ.get("REPS", "20"))
dev = torch.device("cuda", 0)
BF = torch.bfloat16
rows_sam, rows_llm, H, I, V, T = B * 4096, B * 319, 4096, 11008, 32004, 319
out = []


def bench(name, kernel, shape, alg_bytes, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(REPS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    us = sorted(ts)[len(ts) // 2]
    out.append({"name": name, "kernel": kernel, "shape": shape, "alg_bytes": alg_bytes, "us": us, "gbps": alg_bytes / us / 1e3, "frac_hbm_peak": alg_bytes / us / 1e3 / 8000.0,
                "launches": REPS + 3})


r = lambda *s: torch.randn(*s, device=dev).to(BF)
x = r(rows_sam, 1280); w, b = r(1280), r(1280); y = torch.empty_like(x)
bench("SAM LayerNorm", "norm_kernel<3>", f"{rows_sam}x1280", 2 * x.numel() * 2, lambda: ops.norm(x, w, b, eps=1e-6, out=y))
x = r(rows_llm, H); w = r(H); y = torch.empty_like(x)
bench("Llama RMSNorm", "norm_kernel<8>" if rows_llm >= 2048 else "norm_wg_kernel<2>", f"{rows_llm}x{H}", 2 * x.numel() * 2, lambda: ops.norm(x, w, None, eps=1e-6, rms=True, out=y))
dy = r(rows_llm, H)
bench("Llama RMSNorm backward", "norm_bwd_wg_kernel<2>", f"{rows_llm}x{H}", 3 * x.numel() * 2, lambda: ops.norm_bwd(dy, x, w, 1e-6, True))
gu = r(rows_llm, 2 * I); so = torch.empty(rows_llm, I, device=dev, dtype=BF)
bench("SwiGLU", "swiglu_kernel", f"{rows_llm}x{I}", 3 * rows_llm * I * 2, lambda: ops.swiglu(gu, I, out=so))
dso = r(rows_llm, I)
bench("SwiGLU backward", "swiglu_bwd_kernel", f"{rows_llm}x{I}", 5 * rows_llm * I * 2, lambda: ops.swiglu_bwd(gu, dso, I))
qkv = r(rows_llm, 3 * H)
ang = torch.outer(torch.arange(T, device=dev).float(), 1.0 / (10000 ** (torch.arange(0, 128, 2, device=dev).float() / 128)))
cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
bench("RoPE (q|k in place)", "rope_kernel", f"{rows_llm}x{2 * H}", 2 * rows_llm * 2 * H * 2, lambda: ops.rope_(qkv, cos, sin, rows_llm, T, 64, 128, 3 * H))
for K in (256, 512):
    segs = (torch.rand(K, 256, 256, device=dev) > 0.7).to(BF)
    wn = torch.empty(K, 4096, device=dev, dtype=BF)
    from llmseg_amd import _lib
    import ctypes as C
    lib = _lib.load()
    bench(f"mask pull-back K={K}", "mask_pullback_s256_kernel", f"{K}x256x256", K * 65536 * 2 + K * 4096 * 2,
          lambda: _lib.check(lib.llmseg_mask_pullback(C.c_void_p(segs.data_ptr()), C.c_void_p(wn.data_ptr()), None, None, K, 64, 256,
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)), "pullback"))
Nn = max(1, B)
logits = r(Nn, T, V); labels = torch.randint(0, V, (Nn, T), device=dev)
bench("shifted CE (bf16 logits)", "ce_kernel", f"{Nn * T}x{V}", Nn * T * V * 2, lambda: ops.ce_loss(logits, labels))
coef = torch.ones(1, device=dev)
bench("CE backward", "ce_bwd_kernel", f"{Nn * T}x{V}", 2 * Nn * T * V * 2, lambda: ops.ce_bwd(logits, labels, coef))
n = V * H
p = r(n); master = p.float(); g32 = torch.randn(n, device=dev); m1 = torch.zeros(n, device=dev); v1 = torch.zeros(n, device=dev)
bench("AdamW (fp32 master + bf16 copy, fp32 grad)", "adamw_kernel", f"{n}", n * (4 + 4 * 2 + 4 * 2 + 4 * 2 + 2), lambda: ops.adamw_(p, master, g32, m1, v1, 1e-4, 0.9, 0.95, 1e-8, 0.0, 3))
acc = torch.zeros(1, device=dev)
bench("gradient norm (sum of squares, fp32)", "sumsq_kernel", f"{n}", n * 4, lambda: ops.sumsq(g32, acc))
# N2 (round 4): the one-pass proposal kernel -- 256 proposals of one 1024 x 1024 image read once through the order index
from llmseg_amd import targets as T_
Kp = 256
props = (torch.rand(Kp + 44, 1024, 1024, device=dev) > 0.7).to(torch.uint8)
order = torch.argsort(props.flatten(1).sum(1), descending=True, stable=True)[:Kp]
gtm = (torch.rand(1024, 1024, device=dev) > 0.5).to(torch.uint8)
bench("N2 proposals: counts + 256x256 maps in one pass", "proposal_targets_kernel<16>", f"{Kp}x1024x1024", Kp * 1024 * 1024 + 1024 * 1024 + Kp * 256 * 256 * 2,
      lambda: T_.proposal_targets_fused(props, order, [gtm]))
M = rows_llm
slab = torch.randn(3, M, H, device=dev)
json.dump({"images_per_step": B, "kernels": out}, sys.stdout, indent=1)
