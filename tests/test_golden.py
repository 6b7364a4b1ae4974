"""CPU tests: the oracle against golden vectors produced by the reference itself (its lofreq 2.1.4
binary run in the build container, see oracle/make_golden.py): VCF records byte-for-byte (modulo the
;HQA= field HEAD added) and the substitution-test count."""
import numpy as np
import pytest

import golden_util as gu


@pytest.mark.parametrize("path", gu.fixtures(), ids=lambda p: p.split("/")[-1])
def test_oracle_reproduces_reference_binary(oracle, path):
    fx, host = gu.load(path)
    kw, no_default_filter = gu.conf_kwargs(fx["call_args"])
    conf = oracle.default_conf(raw_counts_after_minbq=1, **kw)     # 2.1.4 counted raw alts after min_bq
    res, _ = oracle.call_batch(host["nt"], host["bq"], host["baq"], host["mq"], None, host["col_off"],
                               host["ref_base"], conf)
    if not fx.get("column_subset"):      # deep fixtures store a subset of the run's columns (fixed Bonferroni factor)
        assert conf.num_snv_tests == fx["num_snv_tests"]
    L = oracle.lib()
    recs = []
    for c in range(len(res)):
        ref = int(host["ref_base"][c])
        if chr(ref) not in "ACGT":
            continue
        rcode = "ACGT".index(chr(ref))
        for a in range(3):
            if not res["emitted"][c, a]:
                continue
            acode = "ACGT".index(chr(int(res["alt_base"][c, a])))
            dp = int(host["col_off"][c + 1] - host["col_off"][c])
            d = dict(col=c, qual=int(res["qual"][c, a]), dp=dp, raw=int(res["alt_raw_counts"][c, a]),
                     ref_fw=int(res["fw"][c, rcode]), ref_rv=int(res["rv"][c, rcode]),
                     alt_fw=int(res["fw"][c, acode]), alt_rv=int(res["rv"][c, acode]), ref=ref,
                     alt=int(res["alt_base"][c, a]), hqa=int(res["alt_counts"][c, a]))
            d["sb"] = L.orc_sb_phred(d["ref_fw"], d["ref_rv"], d["alt_fw"], d["alt_rv"])
            recs.append(d)
    # what `lofreq call` does after the loop (lofreq_call.c:1402-1551)
    dynamic = bool(conf.bonf_dynamic)
    direct = no_default_filter and not dynamic
    keep = np.ones(len(recs), bool)
    filt = "."
    if not direct:
        import ctypes as C
        thr = L.orc_snvqual_thresh(conf.sig, conf.bonf_subst) if dynamic else 0
        arr = lambda k: (C.c_int * len(recs))(*[r[k] for r in recs])
        k = (C.c_int * max(len(recs), 1))()
        L.orc_default_filter(arr("qual"), arr("dp"), arr("sb"), arr("alt_fw"), arr("alt_rv"), len(recs), thr,
                             0 if no_default_filter else 1, k)
        keep = np.array([bool(k[i]) for i in range(len(recs))])
        filt = "PASS"
    lines = []
    for r, kp in zip(recs, keep):
        if not kp:
            continue
        buf = C_buf = bytearray(512)
        import ctypes as C
        cb = (C.c_char * 512).from_buffer(buf)
        n = L.orc_format_snv(cb, 512, b"chr1", fx["columns"][r["col"]]["pos0"], bytes([r["ref"]]), bytes([r["alt"]]),
                             r["qual"], r["dp"], r["raw"], r["sb"], r["ref_fw"], r["ref_rv"], r["alt_fw"],
                             r["alt_rv"], r["hqa"], 0, filt.encode())
        lines.append(bytes(buf[:n]).decode().rstrip("\n"))
    assert lines == fx["vcf"]


@pytest.mark.parametrize("path", gu.indel_fixtures(), ids=lambda p: p.split("/")[-1])
def test_oracle_reproduces_reference_binary_indels(oracle, path):
    """call_indels restatement vs the 2.1.4 binary: indel VCF records byte-for-byte and the test count"""
    from lofreq_amd.indel import IndelColumns
    fx, dicts = gu.load_indels(path)
    cols = IndelColumns.from_columns(dicts)
    kw, no_default_filter = gu.conf_kwargs(fx["call_args"])
    kw.pop("bonf_subst", None)                       # -b does not touch bonf_indel (lofreq_call.c, snpcaller.c:642)
    conf = oracle.default_conf(**kw)
    tests = oracle.call_indels_batch(cols.flat(), conf)
    assert conf.num_indel_tests == fx["num_indel_tests"] == len(tests)
    dynamic = bool(conf.bonf_dynamic)
    direct = no_default_filter and not dynamic
    thr = oracle.lib().orc_snvqual_thresh(conf.sig, conf.bonf_indel) if dynamic else 0
    lines = []
    for t in tests[tests["emitted"] == 1]:
        if not direct:
            if thr > 0 and t["qual"] < thr:
                continue
            if not no_default_filter and t["dp"] < 10:
                continue
        ref, alt = cols.ref_alt(int(t["col"]), int(t["side"]), int(t["event"]))
        lines.append(oracle.format_indel("chr1", fx["columns"][int(t["col"])]["pos0"], ref, alt, t,
                                         None if direct else "PASS").rstrip("\n"))
    assert lines == fx["vcf"]
