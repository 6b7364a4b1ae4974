"""The whole reads -> VCF path in the ORACLE (test infrastructure): orc_baq_idaq_read per read (bam_md_ext.c:260-491) ->
orc_pileup_region (compile_plp_col, plp.c:797-1288) -> orc_call_indels_batch + orc_call_batch (call_vars:
lofreq_call.c:887-935, indels before SNVs in a column, no SNVs where the consensus is an indel :928-931) -> the
epilogue of main_call (QUAL thresholds from the final factors, lofreq_call.c:1519-1538) -> VCF text.  Pinned on the
2.1.4 binary's own VCFs by tests/test_orc_pileup.py; the GPU tests of the C4 / C5 shapes compare the device chain with
it at sizes no fixture covers."""
import numpy as np


def add_alnqual_tags(orc, reads, ref, extended=True, idaq=True, procs=1):
    """lb (and ai / ad) of every read by the oracle's BAQ -- what mplp_func computes on the fly (plp.c:667-683)"""
    if procs > 1 and len(reads) > 2000:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        chunks = [reads[i::procs] for i in range(procs)]
        with ProcessPoolExecutor(max_workers=procs, mp_context=mp.get_context("spawn")) as ex:
            res = list(ex.map(_tags_chunk, [(c, ref, extended, idaq) for c in chunks]))
        for i in range(procs):
            for r, t in zip(reads[i::procs], res[i]):
                r["lb"], r["ai"], r["ad"] = t
        return
    for r, t in zip(reads, _tags_chunk((reads, ref, extended, idaq))):
        r["lb"], r["ai"], r["ad"] = t


def _tags_chunk(args):
    reads, ref, extended, idaq = args
    import pyoracle as orc
    out = []
    for r in reads:
        if idaq:
            lb, ai, ad = orc.baq_idaq_read(r["pos0"], r["cigar"], r["seq"], r["qual"], ref, extended)
        else:
            lb, ai, ad = orc.baq_read(r["pos0"], r["cigar"], r["seq"], r["qual"], ref, extended), None, None
        out.append((lb, ai, ad))
    return out


def call_region(orc, reads, ref, begin, end, conf_kw, call_indels=True, only_indels=False, chrom="chr1",
                no_default_filter=True, raw_counts_after_minbq=0, min_plp_bq=3, min_plp_idq=0):
    """-> dict(lines = VCF records in position order (indels first within a column), col_pos, snv = per-column oracle
    results, indel_tests, n_snv_tests, n_indel_tests, conf)"""
    use_baq = bool(conf_kw.get("flag", 3) & 1)
    P = reads if isinstance(reads, dict) else orc.pack_reads(reads, ref)       # packed arrays pass through
    plp = orc.pileup_region(P, begin, end, min_plp_bq=min_plp_bq, min_plp_idq=min_plp_idq, use_baq=use_baq)
    col_pos, host, flat = plp["col_pos"], plp["host"], plp["flat"]
    kw = dict(conf_kw)
    conf = orc.default_conf(raw_counts_after_minbq=raw_counts_after_minbq, **kw)
    L = orc.lib()
    items = []
    tests = np.zeros(0, orc.INDEL_TEST_DTYPE)
    if call_indels:
        tests = orc.call_indels_batch(flat, conf)
    res = None
    if not only_indels:
        nb = host["num_bases"].copy()
        nb[plp["cons_indel"] != 0] = 0              # lofreq_call.c:928-931: the other half of the same gate
        host = dict(host, num_bases=nb, sq=None)
        res, _ = orc.call_batch(host["nt"], host["bq"], host["baq"], host["mq"], None, host["col_off"], host["ref_base"],
                                conf, coverage_plp=host["coverage_plp"], num_bases=nb)
    dynamic = bool(conf.bonf_dynamic)
    # indel lines
    if call_indels:
        thr_i = L.orc_snvqual_thresh(conf.sig, conf.bonf_indel) if dynamic else 0
        keys = [[flat["key_chars"][s][flat["key_off"][s][i]:flat["key_off"][s][i + 1]].decode()
                 for i in range(len(flat["key_off"][s]) - 1)] for s in (0, 1)]
        for t in tests[tests["emitted"] == 1]:
            if thr_i > 0 and t["qual"] < thr_i:
                continue
            if not no_default_filter and t["dp"] < 10:
                continue
            c = int(t["col"])
            r = chr(int(host["ref_base"][c]))
            key = keys[int(t["side"])][int(t["event"])]
            ref_s, alt_s = (r, r + key) if int(t["side"]) == 0 else (r + key, r)
            items.append((int(col_pos[c]), 0, orc.format_indel(chrom, int(col_pos[c]), ref_s, alt_s, t, "PASS").rstrip("\n")))
    if res is not None:
        import ctypes as C
        thr_s = L.orc_snvqual_thresh(conf.sig, conf.bonf_subst) if dynamic else 0
        recs = []
        for c in np.nonzero(res["emitted"].any(axis=1))[0]:
            ref_c = int(host["ref_base"][c])
            rc = b"ACGT".index(bytes([ref_c]))
            for a in range(3):
                if res["emitted"][c, a]:
                    ac = b"ACGT".index(bytes([int(res["alt_base"][c, a])]))
                    d = dict(col=int(c), qual=int(res["qual"][c, a]), dp=int(host["coverage_plp"][c]),
                             raw=int(res["alt_raw_counts"][c, a]), ref_fw=int(res["fw"][c, rc]), ref_rv=int(res["rv"][c, rc]),
                             alt_fw=int(res["fw"][c, ac]), alt_rv=int(res["rv"][c, ac]), ref=ref_c,
                             alt=int(res["alt_base"][c, a]), hqa=int(res["alt_counts"][c, a]))
                    d["sb"] = L.orc_sb_phred(d["ref_fw"], d["ref_rv"], d["alt_fw"], d["alt_rv"])
                    recs.append(d)
        n = len(recs)
        keep = (C.c_int * max(n, 1))()
        arr = lambda k: (C.c_int * max(n, 1))(*[r[k] for r in recs])
        L.orc_default_filter(arr("qual"), arr("dp"), arr("sb"), arr("alt_fw"), arr("alt_rv"), n, thr_s,
                             0 if no_default_filter else 1, keep)
        buf = C.create_string_buffer(512)
        for i, r in enumerate(recs):
            if keep[i]:
                m = L.orc_format_snv(buf, 512, chrom.encode(), int(col_pos[r["col"]]), bytes([r["ref"]]), bytes([r["alt"]]),
                                     r["qual"], r["dp"], r["raw"], r["sb"], r["ref_fw"], r["ref_rv"], r["alt_fw"], r["alt_rv"],
                                     r["hqa"], 0, b"PASS")
                items.append((int(col_pos[r["col"]]), 1, buf.raw[:m].decode().rstrip("\n")))
    items.sort(key=lambda t: (t[0], t[1]))
    return dict(lines=[t[2] for t in items], col_pos=col_pos, snv=res, indel_tests=tests, plp=plp,
                n_snv_tests=int(conf.num_snv_tests), n_indel_tests=int(conf.num_indel_tests), conf=conf)


def call_targets(orc, P, ref, targets, conf_kw, chrom="chr1", min_plp_bq=3):
    """`lofreq call -l targets.bed` in the oracle, SNVs only: the targets in genome order, one conf carried from target to
    target (the running Bonferroni factor does not restart), then the epilogue of main_call.
    -> dict(lines, emitted = [(pos0, pvalue)] of every reported allele, n_snv_tests, n_columns, conf)"""
    import ctypes as C
    conf = orc.default_conf(**conf_kw)
    use_baq = bool(conf_kw.get("flag", 3) & 1)
    L = orc.lib()
    recs, emitted, n_cols = [], [], 0
    for _, b, e in sorted(targets, key=lambda t: t[1]):
        plp = orc.pileup_region(P, b, e, min_plp_bq=min_plp_bq, use_baq=use_baq)
        host, col_pos = plp["host"], plp["col_pos"]
        n_cols += len(col_pos)
        nb = host["num_bases"].copy()
        nb[plp["cons_indel"] != 0] = 0
        res, _ = orc.call_batch(host["nt"], host["bq"], host["baq"], host["mq"], None, host["col_off"], host["ref_base"],
                                conf, coverage_plp=host["coverage_plp"], num_bases=nb)
        for c in np.nonzero(res["emitted"].any(axis=1))[0]:
            ref_c = int(host["ref_base"][c])
            rc = b"ACGT".index(bytes([ref_c]))
            for a in range(3):
                if res["emitted"][c, a]:
                    ac = b"ACGT".index(bytes([int(res["alt_base"][c, a])]))
                    d = dict(pos=int(col_pos[c]), qual=int(res["qual"][c, a]), dp=int(host["coverage_plp"][c]),
                             raw=int(res["alt_raw_counts"][c, a]), ref_fw=int(res["fw"][c, rc]), ref_rv=int(res["rv"][c, rc]),
                             alt_fw=int(res["fw"][c, ac]), alt_rv=int(res["rv"][c, ac]), ref=ref_c,
                             alt=int(res["alt_base"][c, a]), hqa=int(res["alt_counts"][c, a]))
                    d["sb"] = L.orc_sb_phred(d["ref_fw"], d["ref_rv"], d["alt_fw"], d["alt_rv"])
                    recs.append(d)
                    emitted.append((int(col_pos[c]), res["pvalue"][c, a]))
    thr = L.orc_snvqual_thresh(conf.sig, conf.bonf_subst) if conf.bonf_dynamic else 0
    n = len(recs)
    keep = (C.c_int * max(n, 1))()
    arr = lambda k: (C.c_int * max(n, 1))(*[r[k] for r in recs])
    L.orc_default_filter(arr("qual"), arr("dp"), arr("sb"), arr("alt_fw"), arr("alt_rv"), n, thr, 0, keep)
    buf = C.create_string_buffer(512)
    lines = []
    for i, r in enumerate(recs):
        if keep[i]:
            m = L.orc_format_snv(buf, 512, chrom.encode(), r["pos"], bytes([r["ref"]]), bytes([r["alt"]]), r["qual"], r["dp"],
                                 r["raw"], r["sb"], r["ref_fw"], r["ref_rv"], r["alt_fw"], r["alt_rv"], r["hqa"], 0, b"PASS")
            lines.append(buf.raw[:m].decode().rstrip("\n"))
    return dict(lines=lines, emitted=emitted, n_snv_tests=int(conf.num_snv_tests), n_columns=n_cols, conf=conf)
