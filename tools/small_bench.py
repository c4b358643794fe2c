"""GPU box: latency of the small (launch- / latency-bound) kernels at the 2-images-per-step shapes; tuning aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from llmseg_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
BF = torch.bfloat16
M, H = int(os.environ.get("M", "638")), 4096
r = lambda *s: (torch.randn(*s, device=dev) * 0.1).to(BF)
rng = torch.tensor([1, 1], device=dev, dtype=torch.int64)


def t(name, fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:60s} {e0.elapsed_time(e1) / reps * 1e3:8.1f} us", flush=True)


x, aq, av = r(M, H), r(8, H), r(8, H)
a2 = torch.empty(M, 64, device=dev, dtype=BF)
t("lora_down fwd (2 branches, dropout)", lambda: ops.lora_down(x, aq, out=a2, zero_cols=48, drop=(rng, 0, 0.05), x2=x, w2=av))
t("lora_down fwd (2 branches, no dropout)", lambda: ops.lora_down(x, aq, out=a2, zero_cols=48, x2=x, w2=av))
d = r(M, 3 * H); bq, bv = r(H, 8), r(H, 8)
t2 = torch.empty(M, 64, device=dev, dtype=BF)
t("lora_down bwd (dq.Bq | dv.Bv)", lambda: ops.lora_down(d[:, :H], bq, w_kr=True, alpha=2.0, out=t2, zero_cols=48, x2=d[:, 2 * H:], w2=bv))
bt = torch.empty(16, H, device=dev, dtype=BF)
ops.lora_pack(aq, bq, av, bv, 2.0, bt=bt)
t("lora_down bwd via B^T (MFMA form)", lambda: ops.lora_down(d[:, :H], bt[:8], alpha=2.0, out=t2, zero_cols=48, x2=d[:, 2 * H:], w2=bt[8:]))
g1, g2 = torch.zeros(H, 8, device=dev), torch.zeros(H, 8, device=dev)
t("lora_outer dB (2 products)", lambda: ops.lora_outer(d[:, :H], a2[:, :8], alpha=2.0, out=g1, a2=d[:, 2 * H:], b2=a2[:, 8:16], out2=g2))
g3, g4 = torch.zeros(8, H, device=dev), torch.zeros(8, H, device=dev)
t("lora_outer dA (2 products, dropout)", lambda: ops.lora_outer(x, t2[:, :8], out_rn=True, out=g3, drop=(rng, 0, 0.05), a2=x, b2=t2[:, 8:16], out2=g4))
gq, gv, ga, gb = torch.zeros(H, 8, device=dev), torch.zeros(H, 8, device=dev), torch.zeros(8, H, device=dev), torch.zeros(8, H, device=dev)
t("lora_wgrads (4 products, dropout; LLMSEG_LORA_OUTER_GY=%s)" % os.environ.get("LLMSEG_LORA_OUTER_GY", "-"), lambda: ops.lora_wgrads(d, H, x, a2, t2, gq, gv, ga, gb, 2.0, drop=(rng, 0, 0.05)))
t("lora_outer dA (2 products, no dropout)", lambda: ops.lora_outer(x, t2[:, :8], out_rn=True, out=g3, a2=x, b2=t2[:, 8:16], out2=g4))
dx = r(M, H)
t("lora_apply (2 branches, dropout)", lambda: ops.lora_apply_(dx, t2, aq, w_rn=True, drop=(rng, 0, 0.05), w2=av))
w2b = torch.empty(3 * H, 64, device=dev, dtype=BF)
t("lora_pack w2b", lambda: ops.lora_pack(aq, bq, av, bv, 2.0, w2b=w2b))
# head-sized norms (512 rows x 256) forward / backward with weight grads
xs, ws, bs = r(512, 256), r(256), r(256)
dw, db = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
t("layernorm fwd 512x256", lambda: ops.norm(xs, ws, bs))
t("layernorm bwd 512x256 (+dw, db)", lambda: ops.norm_bwd(xs, xs, ws, 1e-5, False, dw, db))
xl, wl = r(M, H), r(H)
t("rmsnorm fwd 638x4096", lambda: ops.norm(xl, wl, None, eps=1e-6, rms=True))
t("rmsnorm bwd 638x4096", lambda: ops.norm_bwd(xl, xl, wl, 1e-6, True))
gu = r(M, 22016)
t("swiglu 638x11008", lambda: ops.swiglu(gu, 11008))
t("swiglu bwd", lambda: ops.swiglu_bwd(gu, r(M, 11008), 11008), reps=10)
qkv = r(M, 3 * H)
ang = torch.outer(torch.arange(319, device=dev).float(), 1.0 / (10000 ** (torch.arange(0, 128, 2, device=dev).float() / 128)))
cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
t("rope 638 x 8192", lambda: ops.rope_(qkv, cos, sin, M, 319, 64, 128, 3 * H))
lse = torch.empty(2, 32, 319, device=dev)
t("llama attention fwd (2 x 32 heads x 319)", lambda: ops.attention_packed(qkv, 2, 319, 32, 128, causal=True, lse=lse))
