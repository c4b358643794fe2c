"""GPU box: is a failing full-depth gradient seed a kernel problem or one unlucky draw of bf16 rounding noise?  For seed S:
  (A) `python tools/probes/fd_seed_diag.py S full`   HIP micro-step + fp32 oracle + bf16-CPU oracle -> gpurun_out/fd_seed{S}_A.pt (head tensors only)
  (B) `LLMSEG_GEMM_PP2=0 LLMSEG_GEMM_NO_NORM_FUSE=1 LLMSEG_ATTN_BWD_SPLIT=1 LLMSEG_GEMM_G4=0 python tools/probes/fd_seed_diag.py S hip B`
      the SAME arithmetic through other kernels / launch sets (another summation order only) -> gpurun_out/fd_seed{S}_B.pt
`python tools/probes/fd_seed_diag.py S report` (CPU) prints per tensor: err(HIP A), err(HIP B), err(bf16-CPU), |HIP A - HIP B|."""
import sys

import torch

sys.path.insert(0, ".")
seed, mode = int(sys.argv[1]), sys.argv[2]
tag = sys.argv[3] if len(sys.argv) > 3 else "A"
if mode in ("full", "hip"):
    from tests import fulldepth_checks as fc
    d = {}
    res = fc.check_full_depth_gradients(seed=seed, dump=d, hip_only=(mode == "hip"), log=lambda *a: print(*a, flush=True))
    keep = [n for n in d["hip"] if (".lisa_" in n or "text_hidden_fcs" in n) and d["hip"][n].numel() <= 70000][:40]       # small tensors only (gpurun_out/ is capped at 64 MiB)
    out = {k: ({n: v[n] for n in keep if n in v} if isinstance(v, dict) and k in ("hip", "ref", "lo") else v) for k, v in d.items()}
    torch.save(out, f"gpurun_out/fd_seed{seed}_{tag}.pt")
    for n, e, t in res:
        if "loss" in n or "quantile" in n or "median" in n:
            print(f"{n}: {e:.3e} (tol {t:.3e})")
    print("hip losses", d["hip_loss"])
else:
    a = torch.load(f"gpurun_out/fd_seed{seed}_A.pt")
    others = {t: torch.load(f"gpurun_out/fd_seed{seed}_{t}.pt") for t in sys.argv[3:]}
    print("losses: ref", a["ref_loss"], "\n        bf16-CPU", a["lo_loss"], "\n        HIP A", a["hip_loss"])
    for t, o in others.items():
        print(f"        HIP {t}", o["hip_loss"])
    rms = lambda x: float(x.double().pow(2).mean().sqrt())
    print("| tensor | rms(ref) | err bf16-CPU | err HIP A | " + " | ".join(f"err HIP {t} | rms(HIP A - HIP {t})" for t in others) + " |\n|---|---|---|---|" + "---|---|" * len(others))
    for n, r in a["ref"].items():
        ha, lo = a["hip"][n].reshape(r.shape), a["lo"][n].reshape(r.shape)
        cols = " | ".join(f"{rms(o['hip'][n].reshape(r.shape) - r):.2e} | {rms(ha - o['hip'][n].reshape(r.shape)):.2e}" for o in others.values())
        fit = lambda x: float((x.double().flatten() @ r.double().flatten()) / (r.double().flatten() @ r.double().flatten() + 1e-300))
        cosv = lambda x: float((x.double().flatten() @ r.double().flatten()) / (x.double().norm() * r.double().norm() + 1e-300))
        sa, sl = fit(ha), fit(lo)
        print(f"| {n} | {rms(r):.2e} | {rms(lo - r):.2e} | {rms(ha - r):.2e} | {cols} | HIP = {sa:.4f} x ref + {rms(ha - sa * r):.2e} (cos {cosv(ha):.5f}); bf16-CPU = {sl:.4f} x ref + {rms(lo - sl * r):.2e} (cos {cosv(lo):.5f}) |")
