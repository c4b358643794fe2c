"""Which hipBLASLt kernels torch.addmm picks for the hot shapes (run under rocprofv3 --kernel-trace; kernel names encode the tile config)."""
import torch
SH = [(8192, 3840, 1280), (9800, 3840, 1280), (8192, 1280, 1280), (8192, 5120, 1280), (8192, 1280, 5120), (638, 12288, 4096), (638, 4096, 4096),
      (638, 22016, 4096), (638, 4096, 11008), (638, 11008, 4096), (638, 32004, 4096)]
for M, N, K in SH:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    b = torch.randn(N, device="cuda").to(torch.bfloat16); o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        torch.addmm(b, a, w.t(), out=o)
    torch.cuda.synchronize()
