"""Run one attention config a few times (for rocprofv3 --pmc / timing).  args: mode(win|glob|llama|dino) [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llmseg_amd import ops
mode = sys.argv[1]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.manual_seed(0)
def run(batch, n, H, hd, rel=None, grid=(0, 0), causal=False, iters=5):
    qkv = torch.randn(batch * n, 3 * H * hd, device="cuda").to(torch.bfloat16)
    kw = {}
    if rel == "tab":
        th = torch.zeros(32, hd, device="cuda", dtype=torch.bfloat16); tw = torch.zeros_like(th)
        th[:27] = torch.randn(27, hd, device="cuda") * 0.3; tw[:27] = torch.randn(27, hd, device="cuda") * 0.3
        kw = dict(rel_tab_h=th, rel_tab_w=tw, grid_hw=grid)
    elif rel:
        ld = (2 * grid[0] - 1 + 3) // 4 * 4
        kw = dict(rel_h=torch.randn(H, batch * n, ld, device="cuda") * 0.5, rel_w=torch.randn(H, batch * n, ld, device="cuda") * 0.5, rel_ld=ld, grid_hw=grid)
    out = torch.empty(batch * n, H * hd, device="cuda", dtype=torch.bfloat16)
    for _ in range(2):
        ops.attention_packed(qkv, batch, n, H, hd, out=out, causal=causal, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.attention_packed(qkv, batch, n, H, hd, out=out, causal=causal, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 4.0 * batch * H * n * n * hd * (0.5 if causal else 1.0)
    print(f"{mode}: batch={batch} n={n} H={H} hd={hd}: {ms*1e3:.1f} us, {fl/ms/1e9:.0f} TFLOP/s (dense count)")
if mode == "wint":
    from llmseg_amd import _lib
    for v in (1, 2, 1, 2):
        _lib.load().llmseg_attn_set_variant(v)
        print("variant", v, end=" ")
        run(B * 25, 196, 16, 80, rel="tab", grid=(14, 14), iters=20)
elif mode == "wint0":
    # every window reads window 0's q/k/v (batch stride 0): L2-resident inputs -> the kernel's compute-side time
    from llmseg_amd import _lib
    batch, n, H, hd = B * 25, 196, 16, 80
    D = H * hd
    qkv = torch.randn(n, 3 * D, device="cuda").to(torch.bfloat16)
    th = torch.zeros(32, hd, device="cuda", dtype=torch.bfloat16); tw = torch.zeros_like(th)
    th[:27] = torch.randn(27, hd, device="cuda") * 0.3; tw[:27] = torch.randn(27, hd, device="cuda") * 0.3
    out = torch.empty(batch * n, D, device="cuda", dtype=torch.bfloat16)
    st = (0, hd, 3 * D)
    for v in (1, 2):
        _lib.load().llmseg_attn_set_variant(v)
        f = lambda: ops.attention(qkv, qkv[:, D:], qkv[:, 2 * D:], out, batch=batch, heads=H, Nq=n, Nk=n, head_dim=hd, q_strides=st, k_strides=st,
                                  v_strides=st, o_strides=(n * D, hd, D), rel_tab_h=th, rel_tab_w=tw, grid_hw=(14, 14))
        f(); f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print("variant", v, "L2-resident inputs:", e0.elapsed_time(e1) / 20 * 1e3, "us")
elif mode == "win": run(B * 25, 196, 16, 80, rel=True, grid=(14, 14))
elif mode == "glob": run(B, 4096, 16, 80, rel=True, grid=(64, 64))
elif mode == "llama": run(B, 319, 32, 128, causal=True)
elif mode == "dino": run(B, 4097, 16, 64)
elif mode == "globnorel": run(B, 4096, 16, 80)
