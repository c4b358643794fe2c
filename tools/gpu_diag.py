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
