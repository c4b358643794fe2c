"""Backward-pass diagnostics table (does not stop at the first failure)."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import backward_checks as bc
from tests.conftest import GOLDEN
gl = lambda name: torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=True)
for fn in bc.ALL + [lambda: bc.check_model_grads(gl)]:
    try:
        for name, e, t in fn():
            print(f"{'ok  ' if e <= t else 'FAIL'} {name:60s} err {e:.3e} tol {t:.3e}", flush=True)
    except Exception:
        print("EXC in", getattr(fn, "__name__", "model_grads"))
        traceback.print_exc()
