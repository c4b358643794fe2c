"""Run every kernel parity check and print the full table (does not stop at the first failure).
Usage on the GPU box:  python tools/gpu_diag.py > gpurun_out/diag.txt"""
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tests import kernel_checks as kc  # noqa: E402

print("device:", torch.cuda.get_device_name(0))
nbad = 0
for fn in kc.ALL:
    try:
        for name, e, t in fn():
            ok = e <= t
            nbad += (not ok)
            print(f"{'ok  ' if ok else 'FAIL'} {name:40s} err {e:.3e} tol {t:.3e}", flush=True)
        torch.cuda.synchronize()
    except Exception:
        nbad += 1
        print("EXC in", fn.__name__)
        traceback.print_exc()
print("failures:", nbad)

# ---- model-level checks -------------------------------------------------------------------------------------------
import time  # noqa: E402
from tests import model_checks as mc  # noqa: E402
from tests.conftest import GOLDEN  # noqa: E402


def _gl(name):
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=True)


for label, fn in [("sam_small", lambda: mc.check_sam_small_golden(_gl)), ("api", mc.check_reference_api),
                  ("inf dinov2", lambda: mc.check_tiny_inference("dinov2")), ("inf sam", lambda: mc.check_tiny_inference("sam")),
                  ("train dinov2", lambda: mc.check_tiny_train_losses("dinov2")), ("train sam", lambda: mc.check_tiny_train_losses("sam"))]:
    t0 = time.time()
    try:
        for name, e, t in fn():
            print(f"{'ok  ' if e <= t else 'FAIL'} {name:70s} err {e:.3e} tol {t:.3e}", flush=True)
    except Exception:
        print("EXC in", label)
        traceback.print_exc()
    print(f"   [{label}: {time.time() - t0:.1f}s]", flush=True)
