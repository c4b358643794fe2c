"""Print tools/hbm_kernels.py's JSON (stdin) as one line per kernel."""
import json
import sys

d = json.load(sys.stdin)
print(f"images per step: {d['images_per_step']}")
for k in d["kernels"]:
    print(f"{k['name'][:52]:52s} {k['shape']:>16s} {k['alg_bytes'] / 1e6:9.1f} MB {k['us']:8.1f} us {k['gbps']:7.0f} GB/s {100 * k['frac_hbm_peak']:5.1f} %")
