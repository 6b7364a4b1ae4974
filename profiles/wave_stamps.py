#!/usr/bin/env python
"""Lifetimes of the lean count kernel's wavefronts, from a library built with -DLFQ_COUNT_STAMP (every wavefront leaves
its start, header, loop-end and record times -- 10 ns ticks of the constant clock -- and its HW_ID / XCC_ID in the
record-only fields of its column's dense entry).  Two contexts on one C3 batch: `alone` = one batch at a time,
`beside` = context B's count kernel while context A's DP kernels of the batch before run (no gate)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lofreq_amd as la  # noqa: E402


def stats(counts, label):
    raw = counts["alt_raw_counts"].astype(np.uint32).astype(np.uint64)
    fw = counts["alt_fw"].astype(np.uint32).astype(np.int64)
    t0 = raw[:, 0] | (raw[:, 1] << np.uint64(32))
    hdr, loop, rec = raw[:, 2].astype(np.int64), fw[:, 0], fw[:, 1]
    hw = counts["alt_fw"][:, 2].astype(np.uint32)
    xcc = counts["ref_fw"].astype(np.uint32) & 0xF
    ok = counts["kmax"] < 12                        # (the heavy columns' record-only fields were overwritten by the strand kernel)
    raw, fw, t0, hdr, loop, rec, hw, xcc = raw[ok], fw[ok], t0[ok], hdr[ok], loop[ok], rec[ok], hw[ok], xcc[ok]
    cyc = counts["ref_rv"].astype(np.uint32).astype(np.int64)[ok]
    t0 = (t0 - t0.min()).astype(np.int64)
    span = (t0 + rec).max()
    mhz = cyc / np.maximum(rec, 1) * 100.0          # cycles per 10 ns tick -> MHz
    print("   shader clock over a wavefront's life: mean %.0f MHz, p10 %.0f, p90 %.0f" % (mhz.mean(), np.percentile(mhz, 10), np.percentile(mhz, 90)))
    print("== %s: %d wavefronts, kernel span %.3f ms" % (label, len(t0), span / 1e5))
    for name, v in (("start -> header known", hdr), ("header -> loop done", loop - hdr), ("loop done -> record", rec - loop),
                    ("wavefront life", rec)):
        print("   %-24s mean %7.2f us  median %7.2f  p10 %7.2f  p90 %7.2f" % (
            name, v.mean() / 100, np.median(v) / 100, np.percentile(v, 10) / 100, np.percentile(v, 90) / 100))
    # wavefronts resident per SIMD over time: sum of lives / (span x SIMDs)
    cu = ((hw >> 8) & 0xF).astype(np.int64) | (((hw >> 12) & 1).astype(np.int64) << 4) | (((hw >> 13) & 7).astype(np.int64) << 5) \
        | (xcc.astype(np.int64) << 8)
    simd = (hw >> 4) & 3
    print("   distinct CUs seen %d, resident count wavefronts per SIMD (time average) %.2f" % (
        len(np.unique(cu)), rec.sum() / float(span) / (len(np.unique(cu)) * 4)))
    # in ten slices of the kernel: how many wavefronts started, their mean life
    edges = np.linspace(0, span, 11)
    which = np.digitize(t0, edges) - 1
    print("   per tenth of the span: started / mean life us: " + "  ".join(
        "%d/%.1f" % ((which == k).sum(), rec[which == k].mean() / 100 if (which == k).any() else 0) for k in range(10)))
    return simd


def main():
    dev = torch.device("cuda", 0)
    ncols, depth = 1000000, 10000
    a, b = la.SnvCaller(0), la.SnvCaller(0)
    for c in (a, b):
        c.set_dense_strand_counts(False)
        c.set_batch_gate("none")
    batch = a.synth_batch(0x9E3779B97F4A7C15 ^ (3 << 32), depth, ncols, plant_period=997)
    bufs = []
    for _ in range(2):
        bufs.append((torch.zeros(ncols * 64, dtype=torch.uint8, device=dev), torch.zeros(ncols * 128, dtype=torch.uint8, device=dev)))
    # alone
    for _ in range(3):
        b.snv_batch_device(batch, la.VarcallConf(), bufs[1][0], bufs[1][1], ncols)
        b.batch_finish()
    torch.cuda.synchronize()
    stats(bufs[1][0].cpu().numpy().view(la.COL_COUNTS_DTYPE), "alone")
    # beside: A's batch is submitted, then B's; B's count kernel runs beside A's DP kernels once A's count kernel is done
    # -- so queue A, B, A, B ... and look at the last B
    for _ in range(4):
        a.snv_batch_device(batch, la.VarcallConf(), bufs[0][0], bufs[0][1], ncols)
        b.snv_batch_device(batch, la.VarcallConf(), bufs[1][0], bufs[1][1], ncols)
        a.batch_finish()
        a.snv_batch_device(batch, la.VarcallConf(), bufs[0][0], bufs[0][1], ncols)
        b.batch_finish()
        a.batch_finish()
    torch.cuda.synchronize()
    stats(bufs[1][0].cpu().numpy().view(la.COL_COUNTS_DTYPE), "beside the DP kernels of the batch before")


if __name__ == "__main__":
    main()
