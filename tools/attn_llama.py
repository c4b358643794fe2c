"""GPU box: the Llama attention of one micro-step (2 sequences x 32 heads x 319 tokens, head_dim 128, causal + key padding) forward and backward,
timed as hipGraph replays of 32 back-to-back calls (kernel time without launch gaps, as inside the replayed micro-step).  Tuning aid.
  python tools/attn_llama.py [N=2] [T=319]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from llmseg_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 319
H, hd = 32, 128
D = H * hd
dev = torch.device("cuda", 0)
torch.manual_seed(0)
qkv = (torch.randn(N * T, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
km = torch.ones(N, T, dtype=torch.uint8, device=dev)
km[-1, T - 19:] = 0
lse = torch.empty(N, H, T, device=dev)
out = torch.empty(N * T, D, device=dev, dtype=torch.bfloat16)
do = (torch.randn(N * T, D, device=dev) * 0.1).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
ld = 3 * D
st = (T * ld, hd, ld)
dst = (T * D, hd, D)
ang = torch.outer(torch.arange(T, device=dev).float(), 1.0 / (10000 ** (torch.arange(0, hd, 2, device=dev).float() / hd)))
cos, sin = ang.cos().contiguous(), ang.sin().contiguous()


def fwd():
    ops.attention_packed(qkv, N, T, H, hd, out=out, causal=True, key_mask=km, lse=lse)


def bwd():
    ops.attention_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], out, do, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], lse, batch=N, heads=H, Nq=T, Nk=T, head_dim=hd,
                      q_strides=st, k_strides=st, v_strides=st, o_strides=dst, do_strides=dst, dq_strides=st, dk_strides=st, dv_strides=st,
                      causal=True, key_mask=km)


def rope():
    ops.rope_(dqkv, cos, sin, N * T, T, 2 * H, hd, ld)


def timed(name, fn, reps=32, iters=20):
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
    print(f"{name:44s} {e0.elapsed_time(e1) / (iters * reps) * 1e3:8.2f} us per call", flush=True)


timed("llama attention fwd (+ lse, key mask)", fwd)
timed("llama attention bwd (delta + dq|dk|dv)", bwd)
timed("rope on the gradient buffer", rope)
