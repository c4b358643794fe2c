"""GPU probe: the K-sliced register-staging GEMM path vs an fp32 torch reference over transposed-operand shapes."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from llmseg_amd import ops  # noqa: E402

torch.manual_seed(0)
BF = torch.bfloat16
for (M, N, K), (ta, tw), f32 in itertools.product([(16, 256, 4096), (256, 1024, 8192), (512, 256, 2048), (2, 4096, 4096), (768, 256, 512), (130, 260, 1000), (48, 128, 520), (256, 512, 837), (256, 256, 837), (48, 256, 2048), (32064, 256, 128)],
                                                    [(False, True), (True, False), (True, True)], [False, True]):
    a = torch.randn(M, K, device="cuda").to(BF)
    w = torch.randn(N, K, device="cuda").to(BF)
    ref = a.float() @ w.float().t()
    A = a.t().contiguous() if ta else a
    W = w.t().contiguous() if tw else w
    if (ta and M % 8) or (tw and N % 8) or (K % 8 and not (ta and tw)):
        continue
    bias = torch.randn(N, device="cuda").to(BF) if not f32 else None
    if f32:
        out = torch.full((M, N), 0.5, device="cuda", dtype=torch.float32)
        ops.gemm(A, W, out=out, trans_a=ta, trans_w=tw, accumulate=True)
        want = ref + 0.5
    else:
        out = ops.gemm(A, W, bias=bias, trans_a=ta, trans_w=tw)
        want = ref + bias.float()
    err = (out.float() - want).abs().max().item()
    tol = (2.0 ** -8) * want.abs().max().item() if not f32 else 1e-3 * want.abs().max().item()
    print(f"{M}x{N}x{K} ta={int(ta)} tw={int(tw)} f32acc={int(f32)}: err {err:.3e} tol {tol:.3e} {'OK' if err <= tol else 'BAD'}", flush=True)
