"""List the PyTorch (at::native) kernels of a rocprofv3 kernel-trace result by full name: which glue ops a step still launches.
usage: python tools/torch_glue.py <results.db>"""
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = []
for n, calls, tot in cur.execute("select name,total_calls,total_duration from top_kernels"):
    if "at::native" in n:
        f = re.findall(r"(\w+Functor\w*|\w+_kernel_cuda|\w+_kernel\b|CUDAFunctor\w+|\w+Op\b)", n)
        rows.append((float(tot), int(calls), " ".join(dict.fromkeys(f))[:140] or n[:140]))
for tot, calls, n in sorted(rows, reverse=True)[:25]:
    print(f"{tot:10.0f} us {calls:6d} calls  {n}")
