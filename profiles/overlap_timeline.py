#!/usr/bin/env python
"""Steady state of a queued run (bench.py --in-flight N --gate ...) in a rocprofv3 kernel trace (rocpd sqlite): per kernel
name the launches of the last STEPS steps (mean / min / max duration, mean start after the count kernel that is running
when they start), then every kernel that starts between the last two count-kernel starts."""
import glob
import sqlite3
import sys
from collections import defaultdict


def main(path, steps=8):
    if not path.endswith(".db"):
        path = sorted(glob.glob(path + "/**/*.db", recursive=True))[-1]
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = db.execute("select name, start, end, %s from kernels order by start" % q).fetchall()
    cnt = [r for r in rows if "lfq_count_" in r[0]]
    if len(cnt) < steps + 2:
        print("too few count launches", len(cnt))
        return
    t_lo, t_hi = cnt[-steps - 1][1], cnt[-1][1]
    starts = [c[1] for c in cnt]
    print("period (count start to count start), last %d: %s ms" % (steps, " ".join(
        "%.3f" % ((starts[i + 1] - starts[i]) / 1e6) for i in range(len(starts) - steps - 1, len(starts) - 1))))
    print("count kernel durations: %s ms" % " ".join("%.3f" % ((c[2] - c[1]) / 1e6) for c in cnt[-steps - 1:-1]))
    per = defaultdict(list)
    for r in rows:
        if t_lo <= r[1] < t_hi and "rocclr" not in r[0]:
            prev = max(s for s in starts if s <= r[1])
            per[r[0].replace("void ", "")[:44]].append(((r[2] - r[1]) / 1e6, (r[1] - prev) / 1e6))
    print("%-44s %5s %8s %8s %8s %10s" % ("kernel", "n", "mean ms", "min", "max", "start+"))
    for k, v in sorted(per.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        d = [x[0] for x in v]
        print("%-44s %5d %8.3f %8.3f %8.3f %10.3f" % (k, len(v), sum(d) / len(d), min(d), max(d), sum(x[1] for x in v) / len(v)))
    print("\nkernels starting between the last two count starts (ms after the first):")
    a, b = cnt[-2][1], cnt[-1][1]
    for r in rows:
        if a <= r[1] < b and "rocclr" not in r[0]:
            print("  q%-3s %-44s start %7.3f end %7.3f dur %6.3f" % (r[3], r[0].replace("void ", "")[:44], (r[1] - a) / 1e6,
                                                                   (r[2] - a) / 1e6, (r[2] - r[1]) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8)
