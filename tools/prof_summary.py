"""Summarise a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db or *_kernel_stats.csv) into a short table.
Usage: python tools/prof_summary.py <file> [title] > profiles/<name>.md"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^()]{0,40}>)?)", name)
    s = m.group(1) if m else name[:60]
    if s.startswith("at::native"):
        k = re.search(r"(distribution|vectorized_elementwise|unrolled_elementwise|reduce|index|copy|fill|cat)\w*", name)
        s = "torch:" + (k.group(0) if k else "aten")
    return s[:60]


def main():
    path = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else path
    rows = []
    if path.endswith(".db"):
        cur = sqlite3.connect(path).cursor()
        for n, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            rows.append((short(n), int(calls), float(tot), float(avg), float(pct)))   # rocpd view is already in us
    else:
        for r in csv.DictReader(open(path)):
            rows.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    agg = {}
    for n, c, tot, avg, pct in rows:
        a = agg.setdefault(n, [0, 0.0, 0.0])
        a[0] += c; a[1] += tot; a[2] += pct
    print(f"# {title}\n")
    print("rocprofv3 --kernel-trace --stats; durations in microseconds, aggregated by kernel (torch:* = PyTorch plumbing kernels: RNG init, index/cast glue)\n")
    print("| kernel | calls | total us | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for n, (c, tot, pct) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
        print(f"| `{n}` | {c} | {tot:.0f} | {tot / c:.1f} | {pct:.2f} |")


if __name__ == "__main__":
    main()
