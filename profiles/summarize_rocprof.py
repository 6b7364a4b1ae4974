#!/usr/bin/env python
"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table committed under profiles/."""
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B | scratch B | grid | wg |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s | %s | %s | %s |" % (
            r[0][:70], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total,
            r[6], r[7], r[8], r[9], r[10], r[11]))
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "a").write(text)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
