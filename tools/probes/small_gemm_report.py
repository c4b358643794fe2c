"""usage: python tools/probes/small_gemm_report.py <results.db> <stdout of small_gemm_variants.py>: GPU time per call (all kernels of the call) per group."""
import re
import sqlite3
import sys

rows = sqlite3.connect(sys.argv[1]).cursor().execute("select name, start, end from kernels order by start").fetchall()
groups = [ln.split()[2:] for ln in open(sys.argv[2]) if ln.startswith("GROUP")]
cur, gid, acc = None, -1, {}
for n, s, e in rows:
    if "scan" in n.lower() or "cumsum" in n.lower():
        gid += 1
        acc[gid] = []
        continue
    if gid >= 0 and ("gemm" in n or "splitk" in n):
        acc[gid].append((e - s) / 1e3)
table = {}
for g, (shape, v, *rest) in enumerate(groups):
    if rest or g not in acc or not acc[g]:
        table.setdefault(shape, {})[v] = None
        continue
    d = acc[g]
    per_call = sum(d) / 20.0                      # 20 calls after the marker (kernels of one call: gemm [+ reduce])
    table.setdefault(shape, {})[v] = per_call
vs = []
for t in table.values():
    for v in t:
        if v not in vs:
            vs.append(v)
print("GPU microseconds per call (sum of the call's kernels, mean of 20):")
print(f"{'shape':18s} " + " ".join(f"{v:>8s}" for v in vs))
for shape, t in table.items():
    print(f"{shape:18s} " + " ".join(("%8.1f" % t[v]) if t.get(v) else "       -" for v in vs))
