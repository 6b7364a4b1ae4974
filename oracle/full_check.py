"""Whole-batch oracle runs on all host cores -- TEST INFRASTRUCTURE ONLY (the checker of bench.py's full-size
concordance block and of tests/test_gpu_parity.py::test_config_full_batch; never the thing measured).

The reference's own full-size check is a whole-VCF comparison (tests/parallel.sh:40-51).  Here the synthetic workload
of include/lofreq_synth.h is regenerated on the host range by range (oracle/synth_ref.c), every range runs through the
restated call_snvs loop (orc_call_batch, lofreq_call.c:735-879) in its own process, and the results are compared with
what the device returned for the SAME columns: the integer outputs of plp_to_errprobs for every column, every
reported record field by field, the VCF text line by line.

Running Bonferroni factor of a range (lofreq_call.c:794-801): a range that starts after `P` tested columns starts from
bonf_subst = 3 P (1 when P = 0).  P comes from the device's `tested` flags -- and every range's own oracle flags are
compared with them, so by induction over the ranges every prefix used was the oracle's own.
"""
import ctypes as C
import os
import time

import numpy as np


def _conf_for(orc, conf_kw, prefix_tested):
    conf = orc.default_conf(**(conf_kw or {}))
    if conf.bonf_dynamic and prefix_tested > 0:
        conf.bonf_subst = 3 * int(prefix_tested)
    conf.num_snv_tests = 3 * int(prefix_tested)
    return conf


def _run_chunk(args):
    """worker: [(col_begin, ncols, tested columns before col_begin)] -> per range the dense integer outputs and
    the rows of the columns that emitted a record"""
    seed, depth, plant_period, conf_kw, ranges = args
    import pyoracle as orc                  # sys.path of the parent travels with the spawn
    out = []
    for begin, n, prefix in ranges:
        host = orc.synth_fill(seed, depth, plant_period, begin, n)
        conf = _conf_for(orc, conf_kw, prefix)
        res, _ = orc.call_batch(host["nt"], host["bq"], host["baq"], host["mq"], None, host["col_off"],
                                host["ref_base"], conf)
        em = np.nonzero(res["emitted"].any(axis=1))[0]
        # the exact tails of the emitting columns (80-bit linear-space recurrence, orc_tail_truth): what the DEVICE's
        # p-values are held to, next to the oracle's own (log-space, i.e. the reference's noise) values
        truth = np.full((len(em), 3), np.nan)
        for i, c in enumerate(em):
            try:
                truth[i] = orc.col_tail_truth(host, int(c), conf)[1]
            except RuntimeError:
                pass
        out.append({
            "emit_truth_log": truth,
            "begin": begin, "n": n,
            "n_err_probs": res["n_err_probs"].copy(), "alt_counts": res["alt_counts"].copy(),
            "alt_raw_counts": res["alt_raw_counts"].copy(), "tested": res["tested"].astype(np.uint8),
            "emit_cols": em + begin, "emit_rows": res[em].copy(), "emit_ref": host["ref_base"][em].copy(),
            "n_tested": int(res["tested"].sum()),
        })
    return out


def cpu_budget():
    """CPUs this process may really use: the affinity mask, cut down to the cgroup's CPU quota when there is one (a
    container that shows 256 CPUs may be allowed the time of a dozen)"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def default_procs():
    n = cpu_budget()
    return max(1, n - 2 if n > 8 else n)


def run_ranges(seed, depth, plant_period, ranges, conf_kw=None, procs=None, chunk_cols=None):
    """ranges: [(col_begin, ncols, tested columns before col_begin)].  -> list of per-range dicts, in input order.
    Ranges are dealt to `procs` worker processes in chunks of about `chunk_cols` columns."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    procs = procs or default_procs()
    if chunk_cols is None:
        chunk_cols = max(64, min(1000, 40000000 // max(depth, 1)))       # <= 160 MB of tracks per task
    tasks, cur, cur_n = [], [], 0
    for r in ranges:
        b, n, p = int(r[0]), int(r[1]), int(r[2])
        if n > chunk_cols:                      # the prefix of a later piece is unknown here: the caller cuts
            raise ValueError("range longer than chunk_cols: cut it with split_range()")
        cur.append((b, n, p))
        cur_n += n
        if cur_n >= chunk_cols:
            tasks.append(cur)
            cur, cur_n = [], 0
    if cur:
        tasks.append(cur)
    args = [(seed, depth, plant_period, conf_kw, t) for t in tasks]
    if procs == 1 or len(tasks) == 1:
        res = [_run_chunk(a) for a in args]
    else:
        ctx = mp.get_context("spawn")           # the parent may hold a HIP runtime: no fork
        with ProcessPoolExecutor(max_workers=min(procs, len(tasks)), mp_context=ctx) as ex:
            res = list(ex.map(_run_chunk, args))
    return [r for chunk in res for r in chunk]


def split_range(begin, n, tested_prefix, chunk_cols):
    """cut [begin, begin + n) into pieces of <= chunk_cols columns; tested_prefix = exclusive prefix sum of the tested
    flags over the whole batch (len ncols + 1) -> [(begin, n, prefix)]"""
    out = []
    b = begin
    while b < begin + n:
        m = min(chunk_cols, begin + n - b)
        out.append((b, m, int(tested_prefix[b])))
        b += m
    return out


def _log_of(pv):
    return float(np.log(np.longdouble(pv)))


LDBL_MAX = np.finfo(np.longdouble).max
LDBL_MIN = np.finfo(np.longdouble).tiny


def check_batch(orc, seed, depth, plant_period, ncols, gpu_counts, gpu_recs, gpu_vcf_text=None, conf_kw=None,
                default_filter=False, procs=None, columns=None, chrom="synth", pv_tol=1e-10, pv_deep_log=600.0,
                pv_noise_a=0.16, pv_truth_tol=2e-11, chunk_cols=None, lazy_raw=False):
    """Compare one device batch of the synthetic workload with the oracle.

    gpu_counts  COL_COUNTS_DTYPE[ncols] (dense device output: n_err_probs, alt_counts, alt_raw_counts, tested)
    lazy_raw    the batch ran with lfq_set_dense_strand_counts(ctx, 0): a column's dense alt_raw_counts are the oracle's or
                all 0 (the library counts them, like the strand fields, only where a record can use them -- and every
                record's alt_raw_count is compared below); "dense_raw_columns" says how many were filled
    gpu_recs    SNV_RECORD_DTYPE[] of the same batch (column order)
    columns     None = every column; else a sorted array of column indices (each its own one-column range)
    p-values: against the oracle 1e-10 up to |log p| = pv_deep_log and, beyond, the noise bound of the reference's own
    log-space arithmetic, max(pv_tol, pv_noise_a * ulp(|log p|) * depth) (tests/util.py::pv_deep_bound, measured in
    tests/test_oracle_kat.py); against the exact 80-bit recurrence (every record whose tail is inside the long-double
    range) pv_truth_tol.
    -> dict (JSON-able summary; "identical" is the verdict)"""
    t0 = time.perf_counter()
    tested = np.asarray(gpu_counts["tested"]).astype(np.int64)
    prefix = np.zeros(ncols + 1, np.int64)
    prefix[1:] = np.cumsum(tested)
    chunk = chunk_cols or max(64, min(1000, 40000000 // max(depth, 1)))
    if columns is None:
        ranges = split_range(0, ncols, prefix, chunk)
    else:
        columns = np.asarray(columns, np.int64)
        ranges = [(int(c), 1, int(prefix[c])) for c in columns]
    res = run_ranges(seed, depth, plant_period, ranges, conf_kw=conf_kw, procs=procs, chunk_cols=chunk)
    t_oracle = time.perf_counter() - t0

    L = orc.lib()
    n_cols_cmp = 0
    n_raw_filled = 0
    bad_counts = []
    exp = []                                    # (col, allele index, row, ref base)
    for r in res:
        b, n = r["begin"], r["n"]
        g = gpu_counts[b:b + n]
        n_cols_cmp += n
        for f in ("n_err_probs", "alt_counts"):
            if not np.array_equal(g[f], r[f]):
                bad_counts.append((int(b), f))
        same_raw = (np.asarray(g["alt_raw_counts"]) == np.asarray(r["alt_raw_counts"])).reshape(n, -1).all(axis=1)
        zero_raw = (np.asarray(g["alt_raw_counts"]) == 0).reshape(n, -1).all(axis=1)
        if not (same_raw | zero_raw).all() if lazy_raw else not same_raw.all():
            bad_counts.append((int(b), "alt_raw_counts"))
        n_raw_filled += int(same_raw.sum())
        if not np.array_equal(g["tested"].astype(np.uint8), r["tested"]):
            bad_counts.append((int(b), "tested"))
        for c, row, ref, tr in zip(r["emit_cols"], r["emit_rows"], r["emit_ref"], r["emit_truth_log"]):
            for a in range(3):
                if row["emitted"][a]:
                    exp.append((int(c), a, row, int(ref), float(tr[a])))
    in_scope = gpu_recs if columns is None else gpu_recs[np.isin(gpu_recs["col"], columns)]
    # records field by field
    mism = []
    max_d = {"le": [0.0, 0], "gt": [0.0, 0, 0.0], "truth": [0.0, 0]}
    n_sentinel = 0
    if len(exp) != len(in_scope):
        mism.append("record count: oracle %d, device %d" % (len(exp), len(in_scope)))
    for (c, a, row, ref, truth_log), g in zip(exp, in_scope):
        rcode = b"ACGT".index(bytes([ref]))
        alt = int(row["alt_base"][a])
        acode = b"ACGT".index(bytes([alt]))
        e = dict(col=c, qual=int(row["qual"][a]), dp=depth, alt_raw_count=int(row["alt_raw_counts"][a]),
                 ref_fw=int(row["fw"][rcode]), ref_rv=int(row["rv"][rcode]), alt_fw=int(row["fw"][acode]),
                 alt_rv=int(row["rv"][acode]), hqa=int(row["alt_counts"][a]))
        e["sb"] = int(L.orc_sb_phred(e["ref_fw"], e["ref_rv"], e["alt_fw"], e["alt_rv"]))
        for k, v in e.items():
            if int(g[k]) != v:
                mism.append("col %d allele %d: %s oracle %d device %d" % (c, a, k, v, int(g[k])))
        if g["ref"] != bytes([ref]) or g["alt"] != bytes([alt]):
            mism.append("col %d: alleles" % c)
        pv_o, pv_g = np.longdouble(row["pvalue"][a]), np.longdouble(g["pvalue"])
        if pv_o in (LDBL_MAX, LDBL_MIN) or pv_g in (LDBL_MAX, LDBL_MIN):
            n_sentinel += 1
            if pv_o != pv_g:
                mism.append("col %d allele %d: sentinel p-value" % (c, a))
            continue
        lp = _log_of(pv_o)
        d = abs(_log_of(pv_g) - lp)
        deep = abs(lp) > pv_deep_log
        tol = max(pv_tol, pv_noise_a * float(np.spacing(abs(lp))) * depth) if deep else pv_tol
        st = max_d["gt" if deep else "le"]
        st[0] = max(st[0], d)
        st[1] += 1
        if deep:
            st[2] = max(st[2], d / tol)
        if d > tol:
            mism.append("col %d allele %d: |dlog p| %.3g (bound %.3g)" % (c, a, d, tol))
        if np.isfinite(truth_log) and truth_log > -11300.0:         # (below: the 80-bit recurrence's own range ends)
            dt = abs(_log_of(pv_g) - truth_log)
            max_d["truth"][0] = max(max_d["truth"][0], dt)
            max_d["truth"][1] += 1
            if dt > pv_truth_tol:
                mism.append("col %d allele %d: device %.3g off the exact tail" % (c, a, dt))
    # final filter + VCF text, as `lofreq call` ends (lofreq_call.c:1506-1538)
    vcf_identical = None
    n_lines = None
    if gpu_vcf_text is not None and columns is None:
        conf_end = _conf_for(orc, conf_kw, int(prefix[ncols]))
        if conf_end.bonf_dynamic and prefix[ncols] == 0:
            conf_end.bonf_subst = 1
        thr = L.orc_snvqual_thresh(conf_end.sig, conf_end.bonf_subst) if conf_end.bonf_dynamic else 0
        rows = []
        for (c, a, row, ref, _t) in exp:
            rcode = b"ACGT".index(bytes([ref]))
            acode = b"ACGT".index(bytes([int(row["alt_base"][a])]))
            rows.append((c, ref, int(row["alt_base"][a]), int(row["qual"][a]), depth, int(row["alt_raw_counts"][a]),
                         int(row["fw"][rcode]), int(row["rv"][rcode]), int(row["fw"][acode]), int(row["rv"][acode]),
                         int(row["alt_counts"][a])))
        ne = len(rows)
        arr = lambda i: (C.c_int * max(ne, 1))(*[r[i] for r in rows])
        sbs = [int(L.orc_sb_phred(r[6], r[7], r[8], r[9])) for r in rows]
        keep = (C.c_int * max(ne, 1))()
        L.orc_default_filter(arr(3), arr(4), (C.c_int * max(ne, 1))(*sbs), arr(8), arr(9), ne, thr,
                             1 if default_filter else 0, keep)
        lines = []
        buf = C.create_string_buffer(512)
        for i, r in enumerate(rows):
            if not keep[i]:
                continue
            n = L.orc_format_snv(buf, 512, chrom.encode(), r[0], bytes([r[1]]), bytes([r[2]]), r[3], r[4], r[5], sbs[i],
                                 r[6], r[7], r[8], r[9], r[10], 1, b"PASS")
            lines.append(buf.raw[:n].decode())
        exp_text = "".join(lines)
        vcf_identical = exp_text == gpu_vcf_text
        n_lines = len(lines)
        if not vcf_identical:
            mism.append("VCF text differs (%d oracle lines, %d device lines)" % (n_lines, gpu_vcf_text.count("\n")))
    return {
        "columns_compared": int(n_cols_cmp), "counts_identical": not bad_counts, "dense_raw_columns": int(n_raw_filled),
        "reference_records": len(exp), "gpu_records": int(len(in_scope)), "records_compared": min(len(exp), len(in_scope)),
        "sentinel_pvalues": n_sentinel,
        "max_dlogp_upto_600": max_d["le"][0], "n_pvalues_upto_600": max_d["le"][1],
        "max_dlogp_beyond_600": max_d["gt"][0], "n_pvalues_beyond_600": max_d["gt"][1],
        "max_dlogp_beyond_600_over_bound": max_d["gt"][2],
        "bound_beyond_600": "max(%g, %g * ulp(|log p|) * depth): the reference's own log-space noise (tests/test_oracle_kat.py)" % (pv_tol, pv_noise_a),
        "max_dlogp_device_vs_80bit_truth": max_d["truth"][0], "n_pvalues_vs_80bit_truth": max_d["truth"][1],
        "tol_device_vs_80bit_truth": pv_truth_tol,
        "vcf_lines": n_lines, "vcf_text_identical": vcf_identical,
        "identical": (not bad_counts) and (not mism), "mismatches": (["counts: %s" % (bad_counts[:5],)] if bad_counts else []) + mism[:10],
        "oracle_s": t_oracle, "oracle_processes": procs or default_procs(), "total_s": time.perf_counter() - t0,
    }
