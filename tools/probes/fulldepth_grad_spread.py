"""Spread of the full-depth gradient comparison over seeds (weights, batch, dropout): tests/fulldepth_checks.py::check_full_depth_gradients for seeds 1..n,
the distribution of RMS err HIP / RMS err bf16-CPU and max ratios per seed -> markdown on stdout (profiles/r05_spread_fulldepth_grads.md).
    python tools/probes/fulldepth_grad_spread.py [n_seeds=2] [first_seed=1]        (~3 min of GPU box time per seed: two oracle forward+backward passes on the host)"""
import sys

sys.path.insert(0, ".")
from tests import fulldepth_checks as fc      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
print("# Full-depth fwd+bwd gradient parity over seeds (BASELINE configs[2] shape; seed 0 is the suite's test: profiles/r05_fulldepth_grads.md)\n")
print("| seed | tensors | median RMS ratio HIP / bf16-CPU | 90 % | max RMS ratio | max of max-error ratio | worst err / tol | failed | loss err (ref) | embed rows outside touched |")
print("|---|---|---|---|---|---|---|---|---|---|")
for seed in range(first, first + n):
    table, logs = [], []
    res = fc.check_full_depth_gradients(seed=seed, table=table, log=logs.append)
    rr, mm, et = [], [], []
    for row in table:
        c = [x.strip() for x in row.strip().strip("|").split("|")]
        rr.append(float(c[4])); mm.append(float(c[7])); et.append(float(c[9]))
    rr_s = sorted(rr)
    q = lambda p: rr_s[min(len(rr_s) - 1, int(p * len(rr_s)))]
    bad = [nme for nme, e, t in res if not e <= t]
    loss = [r for r in res if r[0].startswith("full-depth train loss")][0]
    clean = [r for r in res if "outside the touched rows" in r[0]][0]
    print(f"| {seed} | {len(table)} | {q(0.5):.2f} | {q(0.9):.2f} | {max(rr):.2f} | {max(mm):.2f} | {max(et):.2f} | {len(bad)} | {loss[1]:.2e} ({loss[0].split('ref ')[1].split(',')[0]}) | {clean[1]:.1e} |", flush=True)
    for b in bad:
        print(f"<!-- seed {seed} FAILED: {b[:200]} -->")
