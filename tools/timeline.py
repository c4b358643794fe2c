"""One micro-step of a rocprofv3 --kernel-trace run as a timeline: every kernel dispatch of the LAST complete step (between the last two
launches of a once-per-step marker kernel) in start order -> CSV (t_start_us, dur_us, gap_us since the latest end of anything before it,
stream / queue id, short kernel name).  For the latency analysis of the launch-bound parts (mask-selection head, Llama small kernels).
usage: python tools/timeline.py <results.db> [marker-substring=embed_splice] [steps-back=1] > step.csv
steps-back: 1 = the last complete step, n = the n-th last (bench.py ends with eager single-stream profiling steps: go back past them to see a
hipGraph replay)."""
import re
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from prof_summary import short  # noqa: E402


def main():
    db, marker = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "embed_splice")
    con = sqlite3.connect(db)
    cur = con.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    rows = None
    for q in ("select name, start, end, stream_id from kernels order by start",
              "select name, start, end, queue_id from kernels order by start",
              "select name, start, end, 0 from kernels order by start"):
        try:
            rows = cur.execute(q).fetchall()
            break
        except sqlite3.Error:
            continue
    if rows is None:
        print("no usable `kernels` view; tables/views:", names, file=sys.stderr)
        for n in names:
            if "kernel" in n.lower():
                print(n, [c[1] for c in cur.execute(f"pragma table_info('{n}')")], file=sys.stderr)
        sys.exit(1)
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 2 + (int(sys.argv[3]) if len(sys.argv) > 3 else 1) - 1:
        print(f"marker {marker!r} seen {len(marks)} times", file=sys.stderr)
        sys.exit(1)
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    a, b = marks[-1 - back], marks[-back]
    step = rows[a:b]
    t0 = step[0][1]
    print("t_start_us,dur_us,gap_us,stream,kernel")
    latest_end = t0
    for n, s, e, q in step:
        print(f"{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.1f},{(s - latest_end) / 1e3:.1f},{q},{short(n)}")
        latest_end = max(latest_end, e)
    print(f"# {len(step)} dispatches, wall {(latest_end - t0) / 1e3:.1f} us, sum of durations {sum(e - s for _, s, e, _ in step) / 1e3:.1f} us", file=sys.stderr)


if __name__ == "__main__":
    main()
