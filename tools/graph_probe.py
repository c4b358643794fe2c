"""GPU-box probe: can a sequence of C-ABI launches (ctypes -> hipLaunchKernelGGL on torch's current stream) be captured in a
torch.cuda.CUDAGraph (hipGraph) and replayed?  Prints timings of eager vs replay for a chain of small GEMMs + norms."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from llmseg_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
M, H = 638, 4096
x0 = (torch.randn(M, H, device=dev) * 0.5).to(torch.bfloat16)
ws = [(torch.randn(H, H, device=dev) / 64).to(torch.bfloat16) for _ in range(8)]
nw = torch.ones(H, device=dev, dtype=torch.bfloat16)


def chain(x):
    for w in ws:
        h = ops.norm(x, nw, None, eps=1e-6, rms=True)
        x = ops.gemm(h, w, residual=x)
    return x


ref = chain(x0).clone()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    y = chain(x0)
torch.cuda.synchronize()
eager_ms = (time.perf_counter() - t0) / 20 * 1e3

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        chain(x0)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    yg = chain(x0)
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
print("graph replay max diff vs eager:", (yg.float() - ref.float()).abs().max().item())
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
graph_ms = (time.perf_counter() - t0) / 20 * 1e3
print(f"16 launches: eager {eager_ms:.3f} ms, graph replay {graph_ms:.3f} ms")
