#!/usr/bin/env python
"""Timeline of the last bench step in a rocprofv3 kernel trace (rocpd sqlite): start / end / duration of every
kernel between the last two count-kernel launches, in ms from the start of the count kernel."""
import glob
import sqlite3
import sys


def main(path):
    if not path.endswith(".db"):
        path = sorted(glob.glob(path + "/**/*.db", recursive=True))[-1]
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if "lfq_count_" in r[0]]
    if len(idx) < 2:
        print("fewer than two count-kernel launches in the trace")
        return
    i0, t0 = idx[-2], rows[idx[-2]][1]
    for r in rows[i0:idx[-1]]:
        if "rocclr" in r[0]:
            continue
        print("%-48s start %8.3f end %8.3f dur %7.3f" % (r[0].replace("void ", "")[:48], (r[1] - t0) / 1e6,
                                                         (r[2] - t0) / 1e6, (r[2] - r[1]) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1])
