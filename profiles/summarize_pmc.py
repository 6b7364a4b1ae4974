#!/usr/bin/env python
"""Summarise rocprofv3 --pmc runs (one sqlite db per counter pass, profiles/run_pmc.sh) into gpurun_out/<tag>_pmc.md.
(bench.py measures roofline.traffic itself, with its own counter passes; nothing reads these files back.)

FETCH_SIZE / WRITE_SIZE are in KiB.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on
gfx950 reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read: the count kernel's
reads are of that kind and are doubled; the other kernels' byte-granular loads are left uncorrected and
flagged as uncalibrated."""
import glob
import json
import os
import sqlite3
import sys


def counters(db_path):
    db = sqlite3.connect(db_path)
    out = {}
    for name, cname, val, n in db.execute(
            "select kernel_name, counter_name, sum(value), count(*) from counters_collection "
            "group by kernel_name, counter_name"):
        short = name.split("(")[0].replace("void ", "")
        out.setdefault(short, {})[cname] = (val, n)
    return out


def main(root, tag):
    allc = {}
    for path in sorted(glob.glob(root + "/pmc_*/**/*.db", recursive=True)):
        for k, d in counters(path).items():
            allc.setdefault(k, {}).update(d)
    lines = ["# PMC summary (%s)" % tag, "",
             "Separate `rocprofv3 --pmc <one counter> --kernel-trace` passes of `bench.py --pmc-child` (2 steps, LFQ_SINGLE_STREAM=1)",
             "(C3, 1 MI355X).  One counter per pass: FETCH_SIZE and WRITE_SIZE together exceed what the hardware collects at once",
             "(rocprofv3 aborts with error 38 and then sits until it is killed -- the 'hang' of the earlier attempts).",
             "`lfq_count_lean_kernel<one BQ threshold, columns per workgroup, chunks in flight per lane>` is what `bench.py` runs (packed nt: 1.5 B per",
             "observation read; record-only counts left to the columns that emit); `lfq_count_fast_kernel` = byte layout / dense strand counts.", "",
             "Counter columns are sums over the launches; bytes are per launch.", "",
             "| kernel | launches | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes / launch (corrected) | SQ_INSTS_VALU | "
             "SQ_WAVE_CYCLES | SQ_BUSY_CYCLES | GRBM_GUI_ACTIVE |", "|---|---|---|---|---|---|---|---|---|"]
    traffic = {}
    for k in sorted(allc):
        if not k.startswith("lfq_"):
            continue
        d = allc[k]
        f, n = d.get("FETCH_SIZE", (0, 1))
        w, _ = d.get("WRITE_SIZE", (0, 1))
        wide = k.startswith(("lfq_count_kernel", "lfq_count_fast_kernel", "lfq_count_lean_kernel", "lfq_synth_kernel"))
        byt = ((2.0 if wide else 1.0) * f + w) * 1024.0 / max(n, 1)
        traffic[k] = byt
        lines.append("| %s | %d | %.0f | %.0f | %.4g %s | %.4g | %.4g | %.4g | %.4g |" % (
            k, n, f, w, byt, "(FETCH x2: wide coalesced reads)" if wide else "(uncalibrated)",
            d.get("SQ_INSTS_VALU", (0, 1))[0], d.get("SQ_WAVE_CYCLES", (0, 1))[0], d.get("SQ_BUSY_CYCLES", (0, 1))[0],
            d.get("GRBM_GUI_ACTIVE", (0, 1))[0]))
    text = "\n".join(lines) + "\n"
    os.makedirs("gpurun_out", exist_ok=True)
    open("gpurun_out/%s_pmc.md" % tag, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
