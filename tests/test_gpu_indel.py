"""-m gpu parity of the indel path (lfq_call_indels_batch) against the oracle's call_indels restatement."""
import numpy as np
import pytest

import util

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180)]


def _run_both(la, caller, oracle, dicts, okw=None, **kw):
    cols = la.IndelColumns.from_columns(dicts)
    oconf = oracle.default_conf()
    conf = la.VarcallConf(**kw)
    for k, v in kw.items():
        setattr(oconf, k, v)
    tests = oracle.call_indels_batch(cols.flat(), oconf)
    recs, ntests = la.call_indels(caller, cols, conf)
    assert ntests == len(tests)
    assert conf.bonf_indel == oconf.bonf_indel and conf.num_indel_tests == oconf.num_indel_tests
    exp = tests[tests["emitted"] == 1]
    assert len(recs) == len(exp), (len(recs), len(exp))
    for r, t in zip(recs, exp):
        ctx = "col %d side %d event %d" % (t["col"], t["side"], t["event"])
        for k in ("col", "side", "event", "qual", "dp", "sb", "ref_fw", "ref_rv", "alt_fw", "alt_rv", "hrun", "count"):
            assert int(r[k]) == int(t[k]), (ctx, k, r[k], t[k])
        assert int(r["bonf"]) == int(t["bonf_used"]), ctx
        assert r["af"] == t["af"], ctx
        util.assert_pvalue_close(r["pvalue"], t["pvalue"], ctx=ctx)
        line = la.format_indel_record("chr1", int(t["col"]), cols, r)
        ref, alt = cols.ref_alt(int(t["col"]), int(t["side"]), int(t["event"]))
        buf = oracle.format_indel("chr1", int(t["col"]), ref, alt, t)
        assert line == buf
    return cols, tests, recs


@pytest.mark.parametrize("seed,lo,hi,n", [(1, 20, 400, 120), (2, 1500, 2500, 30), (3, 1, 40, 200)])
def test_indel_default_conf_random(caller, oracle, seed, lo, hi, n):
    import lofreq_amd as la
    rng = np.random.default_rng(seed)
    _, tests, recs = _run_both(la, caller, oracle, util.random_indel_columns(rng, n, lo, hi))
    assert len(tests) > 0


def test_indel_every_test_compared(caller, oracle):
    """sig = 1 and a fixed factor of 1: every test is emitted, so every p-value is compared"""
    import lofreq_amd as la
    rng = np.random.default_rng(7)
    _, tests, recs = _run_both(la, caller, oracle, util.random_indel_columns(rng, 80, 30, 900),
                               bonf_dynamic=0, bonf_indel=1, sig=1.0)
    assert len(recs) == len(tests) > 50


@pytest.mark.parametrize("kw", [
    dict(flag=3), dict(flag=2), dict(flag=8), dict(flag=15), dict(flag=0), dict(min_cov=150),
    dict(bonf_dynamic=0, bonf_indel=12345), dict(sig=1e-4), dict(min_bq=30, min_alt_bq=30, min_jq=20),
])
def test_indel_conf_variants(caller, oracle, kw):
    """flag bits: 1 BAQ (ignored here), 2 MQ, 4 SQ, 8 IDAQ; the SNV base filters must not leak in"""
    import lofreq_amd as la
    rng = np.random.default_rng(21)
    _run_both(la, caller, oracle, util.random_indel_columns(rng, 100, 20, 300), **kw)


def test_indel_polyat_rule_and_batches(caller, oracle):
    import lofreq_amd as la
    rng = np.random.default_rng(9)
    dicts = util.random_indel_columns(rng, 150, 100, 300, p_event=0.9, polyat=True)
    cols, tests, _ = _run_both(la, caller, oracle, dicts)
    # the rule must have fired somewhere: fewer tests than events
    assert len(tests) < len(cols.keys[0]) + len(cols.keys[1])
    # running factor carried across two calls == one call
    conf = la.VarcallConf()
    a = la.IndelColumns.from_columns(dicts[:70])
    b = la.IndelColumns.from_columns(dicts[70:])
    ra, na = la.call_indels(caller, a, conf)
    rb, nb = la.call_indels(caller, b, conf)
    one = la.VarcallConf()
    rall, nall = la.call_indels(caller, cols, one)
    assert na + nb == nall and conf.bonf_indel == one.bonf_indel
    assert len(ra) + len(rb) == len(rall)
    assert (np.concatenate([ra["qual"], rb["qual"]]) == rall["qual"]).all()
    assert (np.concatenate([ra["bonf"], rb["bonf"]]) == rall["bonf"]).all()


def test_indel_empty(caller):
    import lofreq_amd as la
    cols = la.IndelColumns.from_columns([])
    conf = la.VarcallConf()
    recs, n = la.call_indels(caller, cols, conf)
    assert len(recs) == 0 and n == 0 and conf.bonf_indel == 1


def test_indel_golden_reference_binary_vcf(caller):
    """the HIP path alone against the VCFs the reference's own 2.1.4 binary wrote (tests/golden/indel_*.json)"""
    import golden_util as gu
    import lofreq_amd as la
    paths = gu.indel_fixtures()
    assert len(paths) >= 3
    for path in paths:
        fx, dicts = gu.load_indels(path)
        cols = la.IndelColumns.from_columns(dicts)
        kw, no_default_filter = gu.conf_kwargs(fx["call_args"])
        kw.pop("bonf_subst", None)
        conf = la.VarcallConf(**kw)
        recs, ntests = la.call_indels(caller, cols, conf)
        assert ntests == fx["num_indel_tests"] == conf.num_indel_tests, path
        dynamic = bool(conf.bonf_dynamic)
        direct = no_default_filter and not dynamic
        if direct:
            keep = np.ones(len(recs), bool)
        else:
            thr = la.snvqual_thresh(conf.sig, conf.bonf_indel) if dynamic else 0
            keep = la.filter_indel_records(recs, thr, apply_defaults=not no_default_filter)
        lines = [la.format_indel_record("chr1", fx["columns"][int(r["col"])]["pos0"], cols, r,
                                        None if direct else "PASS").rstrip("\n")
                 for r, k in zip(recs, keep) if k]
        assert lines == fx["vcf"], path


def test_indel_approx_threshold_gate(caller, oracle):
    """the same gate in front of the indel tests (lofreq_call.c:319-320, 384-385: snpcaller(..., conf->approx_threshold_n))"""
    import lofreq_amd as la
    rng = np.random.default_rng(77)
    dicts = util.random_indel_columns(rng, 200, 100, 900)
    _, tests_off, recs_off = _run_both(la, caller, oracle, dicts)
    _, tests, recs = _run_both(la, caller, oracle, dicts, approx_threshold_n=150)
    assert len(tests) == len(tests_off) and len(recs) <= len(recs_off)
    assert caller.dp_work()["n_approx_pruned"] > 0
