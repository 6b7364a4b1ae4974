"""-m gpu parity tests: the HIP path (through the C ABI) against the oracle on the same inputs.

Integer outputs (counts, tested flags, Bonferroni factors, QUAL, DP4, SB, AF numerators) must be
bit-exact; p-values within 1e-10 relative (BASELINE.json north_star).
"""
import numpy as np
import pytest

import util

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180)]


def _compare_records(la, recs, ores, host, tol=None):      # None: per record, 1e-10 up to |log p| = 600, 1e-9 beyond
    exp = [(c, a) for c in range(len(ores)) for a in range(3) if ores["emitted"][c, a]]
    assert len(recs) == len(exp), (len(recs), len(exp))
    for r, (c, a) in zip(recs, exp):
        assert int(r["col"]) == c
        assert r["alt"] == bytes([int(ores["alt_base"][c, a])])
        assert r["ref"] == bytes([int(host["ref_base"][c])])
        assert int(r["qual"]) == int(ores["qual"][c, a]), (c, a, r["qual"], ores["qual"][c, a])
        assert int(r["alt_raw_count"]) == int(ores["alt_raw_counts"][c, a])
        assert int(r["hqa"]) == int(ores["alt_counts"][c, a])
        util.assert_pvalue_close(r["pvalue"], ores["pvalue"][c, a], tol, ctx="col %d allele %d" % (c, a),
                                 n_obs=int(host["col_off"][c + 1] - host["col_off"][c]))


@pytest.mark.parametrize("seed,depth_lo,depth_hi,ncols", [(1, 0, 300, 400), (2, 900, 1100, 200), (3, 1, 70, 300),
                                                         (4, 2500, 4000, 60), (5, 250, 320, 200)])      # several rounds of the shared-wavefront count kernel
def test_default_conf_random(caller, oracle, seed, depth_lo, depth_hi, ncols):
    import lofreq_amd as la
    rng = np.random.default_rng(seed)
    planted = {c: af for c, af in zip(range(5, ncols, 37), [0.01, 0.03, 0.1, 0.3, 0.6, 0.02, 0.05, 0.9])}
    host = util.random_batch(rng, ncols, depth_lo, depth_hi, planted=planted, ref_n_frac=0.02)
    ores, oconf = util.run_oracle(oracle, host)
    conf = la.VarcallConf()
    recs, counts, st = caller.call_snvs(util.to_pileup_batch(la, host), conf, want_counts=True)
    util.assert_counts_equal(counts, ores, host)
    assert conf.bonf_subst == oconf.bonf_subst
    assert conf.num_snv_tests == oconf.num_snv_tests
    assert st.n_tested == int(ores["tested"].sum())
    _compare_records(la, recs, ores, host)
    # SB / DP4 / AF text against the oracle's formatter
    for r in recs:
        c = int(r["col"])
        sb = oracle.lib().orc_sb_phred(int(r["ref_fw"]), int(r["ref_rv"]), int(r["alt_fw"]), int(r["alt_rv"]))
        assert sb == int(r["sb"])


def test_all_pvalues_no_pruning(caller, oracle):
    """sig=1, fixed bonf=1: nothing is pruned, every allele of every tested column is comparable."""
    import lofreq_amd as la
    rng = np.random.default_rng(11)
    planted = {c: af for c, af in zip(range(0, 120, 9), np.linspace(0.01, 0.7, 14))}
    host = util.random_batch(rng, 120, 50, 1500, planted=planted)
    kw = dict(bonf_dynamic=0, bonf_subst=1, sig=1.0)
    ores, _ = util.run_oracle(oracle, host, **kw)
    conf = la.VarcallConf(**kw)
    counts, pvals, st = util.run_layer1(la, caller, host, conf)
    util.assert_counts_equal(counts, ores, host)
    got = {int(p["col"]): p for p in pvals}
    ncmp = 0
    for c in range(len(ores)):
        if not ores["tested"][c]:
            assert c not in got
            continue
        kmax = ores["alt_counts"][c].max()
        main_pv = min(ores["pvalue"][c])
        if main_pv * 1 > 1.0:  # oracle bailed out (p ~ 1 rounding above sig): device may prune too
            continue
        assert c in got, c
        p = got[c]
        for a in range(3):
            if ores["alt_counts"][c, a] == 0:
                assert p["status"][a] == la.LFQ_PV_NONE
                continue
            ref_logp = ores["logp"][c, a]
            assert abs(p["logp"][a] - ref_logp) <= util.PV_LOG_TOL * max(1.0, 0.0) + 0, (c, a, p["logp"][a], ref_logp)
            ncmp += 1
    assert ncmp > 100


def test_fe_clamp_table(caller, oracle):
    """the errno/fenv clamp quirks (LDBL_MIN -> QUAL 49314, lost minor alleles)."""
    import lofreq_amd as la
    cases = [(10000, (60, 40, 0)), (10000, (1000, 40, 0)), (10000, (1000, 60, 0)), (10000, (5000, 0, 0)),
             (10000, (9000, 0, 0)), (1000, (100, 12, 0)), (1000, (300, 12, 0)), (1000, (300, 12, 5))]
    host = util.concat_batches([util.uniform_p_column(n, c) for n, c in cases])
    kw = dict(bonf_dynamic=0, bonf_subst=3000000, min_bq=0, min_alt_bq=0)
    ores, _ = util.run_oracle(oracle, host, **kw)
    conf = la.VarcallConf(**kw)
    recs, counts, st = caller.call_snvs(util.to_pileup_batch(la, host), conf, want_counts=True)
    util.assert_counts_equal(counts, ores, host)
    # the table's landmarks, straight from the reference probe
    assert ores["pvalue"][1, 1] == util.LDBL_MAX and ores["pvalue"][2, 1] == util.LDBL_MIN
    assert ores["qual"][2, 1] == 49314 and ores["qual"][3, 0] == 49314
    # tolerance per record (util.assert_pvalue_close): 1e-10 up to |log p| = 600; the p-values at |log p| up to 3670,
    # where the reference's own log-space rounding noise is ~3e-10 (DESIGN.md "tolerance"), at 1e-9; sentinels and
    # QUALs exact
    _compare_records(la, recs, ores, host)


@pytest.mark.parametrize("kw", [
    dict(min_jq=15), dict(min_alt_jq=20), dict(min_jq=10, min_alt_jq=25), dict(def_alt_bq=-1),
    dict(def_alt_bq=20), dict(def_alt_jq=25), dict(min_bq=0, min_alt_bq=0), dict(min_bq=10, min_alt_bq=20),
    dict(flag=2), dict(flag=1), dict(flag=0), dict(flag=7), dict(min_cov=40),
])
def test_conf_variants(caller, oracle, kw):
    import lofreq_amd as la
    rng = np.random.default_rng(5)
    planted = {c: af for c, af in zip(range(3, 150, 11), np.linspace(0.02, 0.5, 14))}
    host = util.random_batch(rng, 150, 10, 400, planted=planted, with_sq=True)
    ores, oconf = util.run_oracle(oracle, host, **kw)
    conf = la.VarcallConf(**kw)
    recs, counts, st = caller.call_snvs(util.to_pileup_batch(la, host), conf, want_counts=True)
    util.assert_counts_equal(counts, ores, host)
    assert conf.bonf_subst == oconf.bonf_subst
    _compare_records(la, recs, ores, host)


def test_edge_cases(caller, oracle):
    import lofreq_amd as la
    conf = la.VarcallConf()
    # empty batch
    empty = dict(nt=np.zeros(0, np.uint8), bq=np.zeros(0, np.uint8), baq=None, mq=np.zeros(0, np.uint8), sq=None,
                 col_off=np.zeros(1, np.uint64), ref_base=np.zeros(0, np.uint8))
    recs, _, st = caller.call_snvs(util.to_pileup_batch(la, empty), conf)
    assert len(recs) == 0 and conf.bonf_subst == 1
    # all-empty columns, a column of only N, a 100 % alt column, ref N
    cols = [util.uniform_p_column(0, (0, 0, 0)), util.uniform_p_column(50, (50, 0, 0)),
            util.uniform_p_column(40, (0, 0, 0)), util.uniform_p_column(30, (10, 10, 10), ref=b"N")]
    cols[2]["nt"][:] = 4
    host = util.concat_batches(cols)
    ores, oconf = util.run_oracle(oracle, host)
    recs, counts, st = caller.call_snvs(util.to_pileup_batch(la, host), conf, want_counts=True)
    util.assert_counts_equal(counts, ores, host)
    assert conf.bonf_subst == oconf.bonf_subst
    _compare_records(la, recs, ores, host)


def test_dynamic_bonferroni_across_batches(caller, oracle):
    """Splitting a run into batches must not change anything (running factor carried in conf)."""
    import lofreq_amd as la
    rng = np.random.default_rng(21)
    host = util.random_batch(rng, 300, 200, 600, planted={7: 0.05, 150: 0.02, 290: 0.2})
    ores, oconf = util.run_oracle(oracle, host)
    conf = la.VarcallConf()
    recs = []
    for lo, hi in ((0, 100), (100, 101), (101, 300)):
        o0, o1 = int(host["col_off"][lo]), int(host["col_off"][hi])
        part = {k: (None if host[k] is None else host[k][o0:o1]) for k in ("nt", "bq", "baq", "mq", "sq")}
        part["col_off"] = host["col_off"][lo:hi + 1] - np.uint64(o0)
        part["ref_base"] = host["ref_base"][lo:hi]
        r, _, _ = caller.call_snvs(util.to_pileup_batch(la, part), conf)
        r["col"] += lo
        recs.append(r)
    recs = np.concatenate(recs)
    assert conf.bonf_subst == oconf.bonf_subst and conf.num_snv_tests == oconf.num_snv_tests
    _compare_records(la, recs, ores, host)


def _unpack_nt(packed, n):
    """LFQ_TRACKS_NT_PACKED (include/lofreq_amd.h): byte k of a group's 4 bytes = observation k | observation 4 + k << 4"""
    g = packed[: (n + 7) // 8 * 4].reshape(-1, 4)
    return np.concatenate([g & 15, g >> 4], axis=1).reshape(-1)[:n]


@pytest.mark.parametrize("nt_packed,depth", [(False, 10000), (True, 10000), (True, 9999), (True, 37), (False, 37)])
def test_synthetic_workload_matches_cpu_generator(caller, oracle, nt_packed, depth):
    """Device generator == CPU generator byte for byte (both nt layouts, columns that do not start on a group of 8);
    calls match the oracle."""
    import lofreq_amd as la
    ncols = 48 if depth > 1000 else 700
    batch = caller.synth_batch(seed=99, depth=depth, ncols=ncols, plant_period=5, nt_packed=nt_packed)
    host = oracle.synth_fill(99, depth, 5, 0, ncols)
    n = depth * ncols
    for k in ("nt", "bq", "baq", "mq"):
        dev = getattr(batch, k).cpu().numpy()
        if k == "nt" and nt_packed:
            dev = _unpack_nt(dev, n)
        assert np.array_equal(dev[:n], host[k]), k
    assert np.array_equal(batch.ref_base.cpu().numpy()[:ncols], host["ref_base"])
    host["sq"] = None
    ores, oconf = util.run_oracle(oracle, host)
    conf = la.VarcallConf()
    recs, counts, st = caller.call_snvs(batch, conf, want_counts=True)
    util.assert_counts_equal(counts, ores, host)
    assert conf.bonf_subst == oconf.bonf_subst
    # planted 5 % / 50 % columns reach |log p| > 1500: reference log-space noise, see DESIGN.md
    _compare_records(la, recs, ores, host)


def test_underflow_shortcut_extreme_columns(caller, oracle):
    """Columns whose main p-value is below the 80-bit range take the mu^K/K! shortcut in the big-column
    kernel; minor alleles, sentinels and QUALs must still match the reference exactly."""
    import lofreq_amd as la
    cases = [(10000, (6000, 3000, 5)), (10000, (6000, 200, 3)), (10000, (7000, 0, 1)), (10000, (5000, 60, 0)),
             (12000, (6000, 5990, 10)), (10000, (2500, 30, 2)), (10000, (3500, 3400, 0))]
    host = util.concat_batches([util.uniform_p_column(n, c) for n, c in cases])
    # mix in realistic qualities for half of the columns so that mu is not a round number
    rng = np.random.default_rng(3)
    n0 = int(host["col_off"][3])
    host["bq"][:n0] = np.clip(np.round(rng.normal(33, 5, n0)), 6, 41).astype(np.uint8)
    kw = dict(bonf_dynamic=0, bonf_subst=3000000, min_bq=0, min_alt_bq=0)
    ores, _ = util.run_oracle(oracle, host, **kw)
    conf = la.VarcallConf(**kw)
    recs, counts, st = caller.call_snvs(util.to_pileup_batch(la, host), conf, want_counts=True)
    util.assert_counts_equal(counts, ores, host)
    assert (ores["pvalue"] == util.LDBL_MIN).sum() >= 7
    _compare_records(la, recs, ores, host)


def test_golden_reference_binary_vcf(caller, oracle):
    """HIP path end to end against VCFs written by the reference's own lofreq 2.1.4 binary
    (tests/golden, oracle/make_golden.py).  Fields the two known 2.1.4-vs-HEAD deltas do not touch
    (;HQA= suffix, raw-alt counting before the BQ filter -> AF) must be byte-identical."""
    import lofreq_amd as la
    import golden_util as gu
    for path in gu.fixtures():
        fx, host = gu.load(path)
        kw, no_default_filter = gu.conf_kwargs(fx["call_args"])
        conf = la.VarcallConf(**kw)
        recs, counts, st = caller.call_snvs(util.to_pileup_batch(la, host), conf, want_counts=True)
        if not fx.get("column_subset"):
            assert conf.num_snv_tests == fx["num_snv_tests"], path
        dynamic = bool(conf.bonf_dynamic)
        if no_default_filter and not dynamic:
            keep, filt = np.ones(len(recs), bool), None
        else:
            thr = la.snvqual_thresh(conf.sig, conf.bonf_subst) if dynamic else 0
            keep, filt = la.filter_records(recs, thr, apply_defaults=not no_default_filter), "PASS"
        pos0 = np.array([fx["columns"][int(r["col"])]["pos0"] for r in recs], np.int64)
        text = la.format_vcf(recs, "chr1", pos0=pos0, keep=keep, filter_str=filt)
        got = [gu.strip_hqa(l) for l in text.splitlines()]
        assert len(got) == len(fx["vcf"]), path

        def no_af(line):
            f = line.split("\t")
            f[7] = ";".join(x for x in f[7].split(";") if not x.startswith("AF="))
            return "\t".join(f)
        # 2.1.4 raw alt counts (after the BQ filter) from the oracle's compat switch
        oc = oracle.default_conf(raw_counts_after_minbq=1, **kw)
        ores, _ = oracle.call_batch(host["nt"], host["bq"], host["baq"], host["mq"], None, host["col_off"],
                                    host["ref_base"], oc)
        kept = recs[keep]
        for g, e, r in zip(got, fx["vcf"], kept):
            assert no_af(g) == no_af(e), (path, g, e)
            c = int(r["col"])
            a = [bytes([int(x)]) for x in ores["alt_base"][c]].index(r["alt"])
            if int(r["alt_raw_count"]) == int(ores["alt_raw_counts"][c, a]):
                assert g == e, (path, g, e)


def test_row_split_long_columns(caller, oracle):
    """Deep columns take the row-split route (segments from the identity distribution + convolution fold,
    lfq_dp_segw/segb/combine kernels): K from ~40 to ~2400 at depth 6000..10000, every allele compared."""
    import lofreq_amd as la
    rng = np.random.default_rng(31)
    afs = [0.004, 0.008, 0.012, 0.02, 0.03, 0.05, 0.08, 0.12, 0.2, 0.24, 0.0, 0.0]
    planted = {c: af for c, af in enumerate(afs * 3) if af > 0}
    host = util.random_batch(rng, len(afs) * 3, 6000, 10000, planted=planted)
    for kw in (dict(bonf_dynamic=0, bonf_subst=1, sig=1.0), dict()):
        ores, oconf = util.run_oracle(oracle, host, **kw)
        conf = la.VarcallConf(**kw)
        counts, pvals, st = util.run_layer1(la, caller, host, conf)
        util.assert_counts_equal(counts, ores, host)
        got = {int(p["col"]): p for p in pvals}
        ncmp = 0
        for c in range(len(ores)):
            if not ores["tested"][c]:
                continue
            bonf = int(ores["bonf_used"][c])
            for a in range(3):
                pv = ores["pvalue"][c, a]
                if ores["alt_counts"][c, a] == 0 or pv == util.LDBL_MAX:
                    continue
                if not kw and not (pv * bonf < np.float32(conf.sig)):
                    continue            # not significant: the device may have pruned it
                assert c in got, (c, a)
                p = got[c]
                gpv = la.pvalue_from_log(p["logp"][a], int(p["status"][a]))
                util.assert_pvalue_close(gpv, pv, ctx="col %d allele %d" % (c, a))
                ncmp += 1
        assert ncmp >= (60 if kw else 20), ncmp
    # and through layer 2 (records, QUAL, dynamic Bonferroni)
    ores, oconf = util.run_oracle(oracle, host)
    conf = la.VarcallConf()
    recs, _, _ = caller.call_snvs(util.to_pileup_batch(la, host), conf)
    _compare_records(la, recs, ores, host)
    assert conf.bonf_subst == oconf.bonf_subst


def test_ragged_deep_mix(caller, oracle):
    """Very different depths in one batch (1 .. 40000), every DP class and route at once: quad kernel, retry,
    mid kernel with and without row split, split and unsplit big columns (K > 2016 and short-but-wide)."""
    import lofreq_amd as la
    rng = np.random.default_rng(77)
    parts = []
    specs = [(40, 1, 60, {}), (6, 30000, 40000, {0: 0.004, 1: 0.02, 2: 0.1}), (20, 2000, 2600, {3: 0.3, 7: 0.9, 11: 0.05}),
             (30, 100, 700, {5: 0.6, 6: 0.95}), (4, 9000, 11000, {1: 0.3})]
    for ncols, lo, hi, planted in specs:
        parts.append(util.random_batch(rng, ncols, lo, hi, planted=planted, ref_n_frac=0.03))
    host = util.concat_batches(parts)
    ores, oconf = util.run_oracle(oracle, host)
    conf = la.VarcallConf()
    recs, counts, st = caller.call_snvs(util.to_pileup_batch(la, host), conf, want_counts=True)
    util.assert_counts_equal(counts, ores, host)
    assert conf.bonf_subst == oconf.bonf_subst and conf.num_snv_tests == oconf.num_snv_tests
    _compare_records(la, recs, ores, host)
    assert len(recs) >= 8


@pytest.mark.parametrize("kind", ["certain_errors", "noisy"])
@pytest.mark.parametrize("af", [0.004, 0.1, 0.23, 0.7])
def test_cells_below_double_range_at_the_left_end(caller, oracle, kind, af):
    """Columns whose LOW counts are impossible: P(X < 4) drops below 2^-1024 -- after twenty-odd observations with error
    probability 1 (BAQ 0: the base is certainly misaligned) or with more than ~700 expected errors (10 000 reads at Q8).
    Lane 0 of a strip has no left neighbour; its exponent difference to "nothing" once overflowed to inf and 0 * inf
    poisoned the column (log p = -inf, a call with QUAL INT_MIN).  All K classes: light / mid (wave kernels), row split,
    and the unsplit multi-pass kernel (K > 2016)."""
    import lofreq_amd as la
    rng = np.random.default_rng(int(af * 1000) + (7 if kind == "noisy" else 0))
    host = util.random_batch(rng, 6, 9500, 10500, planted={1: af, 4: af}, low_bq_frac=0.0)
    if kind == "certain_errors":
        for c in (1, 4):
            a, b = int(host["col_off"][c]), int(host["col_off"][c + 1])
            idx = rng.choice(np.arange(a, b), 40, replace=False)
            host["baq"][idx] = 0
    else:
        host["bq"][:] = np.clip(np.round(rng.normal(8, 2, len(host["bq"]))), 2, 14).astype(np.uint8)
    ores, oconf = util.run_oracle(oracle, host)
    conf = la.VarcallConf()
    recs, counts, st = caller.call_snvs(util.to_pileup_batch(la, host), conf, want_counts=True)
    util.assert_counts_equal(counts, ores, host)
    assert conf.bonf_subst == oconf.bonf_subst and conf.num_snv_tests == oconf.num_snv_tests
    _compare_records(la, recs, ores, host)
    # every p-value the device produced is a number
    counts2, pvals, st2 = util.run_layer1(la, caller, host, la.VarcallConf())
    for p in pvals:
        for a in range(3):
            if p["status"][a] == 1:
                assert np.isfinite(p["logp"][a]), (int(p["col"]), a, p["logp"])


@pytest.mark.parametrize("depth_lo,depth_hi", [(30, 400), (3000, 9000)])
def test_lazy_strand_counts_same_records(caller, oracle, depth_lo, depth_hi):
    """call_snvs without the dense counts leaves the strand planes out of the count kernel and counts DP4 only for
    the columns of the sparse output (lfq_strand_*): records byte-identical to the dense route, on host batches and
    on both device layouts; with lfq_set_dense_strand_counts(0) layer 1 does the same"""
    import torch
    import lofreq_amd as la
    rng = np.random.default_rng(17)
    n = 400 if depth_hi < 1000 else 40
    host = util.random_batch(rng, n, depth_lo, depth_hi, planted={3: 0.2, 11: 0.02, 20: 0.5, 33: 0.004})
    batch = util.to_pileup_batch(la, host)
    full, counts, _ = caller.call_snvs(batch, la.VarcallConf(), want_counts=True)
    lazy, _, _ = caller.call_snvs(batch, la.VarcallConf())
    assert len(full) > 0 and full.tobytes() == lazy.tobytes()
    assert (full["alt_fw"] + full["alt_rv"] == full["alt_raw_count"]).all() and (full["ref_fw"] + full["ref_rv"] > 0).all()
    for packed in (False, True):
        sb = caller.synth_batch(seed=5, depth=2000, ncols=300, plant_period=7, nt_packed=packed)
        f2, _, _ = caller.call_snvs(sb, la.VarcallConf(), want_counts=True)
        l2, _, _ = caller.call_snvs(sb, la.VarcallConf())
        assert len(f2) > 0 and f2.tobytes() == l2.tobytes()
    # layer 1 with the option off: sparse records complete, dense strand fields untouched (0)
    sb = caller.synth_batch(seed=5, depth=2000, ncols=300, plant_period=7)
    dev = torch.device("cuda", 0)
    d_counts = torch.zeros(300 * 64, dtype=torch.uint8, device=dev)
    d_pvals = torch.zeros(300 * 128, dtype=torch.uint8, device=dev)
    caller.set_dense_strand_counts(False)
    try:
        conf = la.VarcallConf()
        caller.snv_batch_device(sb, conf, d_counts, d_pvals, 300)
        st = caller.batch_finish()
    finally:
        caller.set_dense_strand_counts(True)
    pv = d_pvals[: st.n_pvals * 128].cpu().numpy().view(la.COL_PVALS_DTYPE)
    recs = la.finalize_pvals(conf, pv, None)
    assert recs.tobytes() == f2.tobytes()
    dense = d_counts.cpu().numpy().view(la.COL_COUNTS_DTYPE)
    quiet = np.setdiff1d(np.arange(300), pv["col"])      # (heavy columns that did not emit got their strands as well)
    assert (dense["ref_fw"][quiet] == 0).mean() > 0.5 and (dense["alt_fw"][quiet] == 0).all(axis=1).mean() > 0.5


@pytest.mark.parametrize("kw", [dict(def_alt_bq=-1), dict(min_jq=15, min_alt_jq=20), dict(min_bq=20, min_alt_bq=25),
                                dict(flag=2), dict(bonf_dynamic=0, bonf_subst=1000, sig=0.05)])
def test_packed_nt_layout_equals_byte_layout(caller, kw):
    """LFQ_TRACKS_NT_PACKED against the byte layout on the same synthetic columns, through the fast count path, the
    general (per-observation) path and every DP class: dense counts and records identical"""
    import lofreq_amd as la
    for depth, ncols, period in ((3001, 64, 3), (150, 900, 11)):
        res = []
        for packed in (False, True):
            b = caller.synth_batch(seed=21, depth=depth, ncols=ncols, plant_period=period, nt_packed=packed)
            recs, counts, st = caller.call_snvs(b, la.VarcallConf(**kw), want_counts=True)
            res.append((recs.tobytes(), counts.tobytes(), st.n_tested))
        assert res[0] == res[1], (depth, kw)
        assert res[0][2] > 0


@pytest.mark.parametrize("seed,lo,hi,n,with_sq", [(5, 0, 300, 400, False), (6, 2000, 9000, 30, True), (7, 1, 17, 500, False)])
def test_host_tracks_packed_nt_equal_byte_tracks(caller, oracle, seed, lo, hi, n, with_sq):
    """a HOST producer that nibble-packs its nt track (what the plp_proc_func shim does as the columns arrive;
    lfq_pack_nt_track): lfq_call_snvs_batch(tracks_on_device = 0, LFQ_TRACKS_NT_PACKED) returns the records and dense
    counts of the byte tracks -- and both equal the oracle.  Ragged columns, empty ones, observation counts that are not
    multiples of 8."""
    import lofreq_amd as la
    rng = np.random.default_rng(seed)
    host = util.random_batch(rng, n, lo, hi, with_sq=with_sq, planted={c: 0.2 for c in range(3, n, 29)}, ref_n_frac=0.02)
    kw = dict(flag=7) if with_sq else {}
    ores, oconf = util.run_oracle(oracle, host, **kw)
    b = util.to_pileup_batch(la, host)
    res = []
    for batch in (b, b.packed()):
        conf = la.VarcallConf(**kw)
        recs, counts, st = caller.call_snvs(batch, conf, want_counts=True)
        res.append((recs.tobytes(), counts.tobytes(), int(st.n_tested), int(conf.bonf_subst)))
        util.assert_counts_equal(counts, ores, host)
        _compare_records(la, recs, ores, host)
    assert res[0] == res[1] and res[0][3] == oconf.bonf_subst
    assert b.packed().nt.nbytes < b.nt.nbytes // 2 + 32


def test_config_c2_depth1000_default_filter(caller, oracle):
    """BASELINE.json configs[1] (C2) at test size: the synthetic generator at depth 1000, dynamic Bonferroni, QUAL
    threshold from the final factor and the DEFAULT filter (DP >= 10, strand-bias FDR) -- device tracks through layer 2,
    lfq_snvqual_thresh and lfq_filter_records against the oracle's call loop + orc_default_filter on 12 000 columns.
    Planted variants every 97th column so that the filter has something to decide; shallow batch = the
    four-columns-per-wavefront count kernel and the screen kernel's unaligned windows (depth 1000 is not a multiple
    of 16)."""
    import ctypes as C
    import lofreq_amd as la
    seed, depth, ncols, period = 0x9E3779B97F4A7C15 ^ (2 << 32), 1000, 12000, 97
    batch = caller.synth_batch(seed, depth, ncols, plant_period=period)
    host = oracle.synth_fill(seed, depth, period, 0, ncols)
    host["sq"] = None
    ores, oconf = util.run_oracle(oracle, host)
    conf = la.VarcallConf()
    recs, counts, st = caller.call_snvs(batch, conf, want_counts=True)
    util.assert_counts_equal(counts, ores, host)
    assert conf.bonf_subst == oconf.bonf_subst and conf.num_snv_tests == oconf.num_snv_tests
    assert st.n_tested == int(ores["tested"].sum())
    _compare_records(la, recs, ores, host)
    assert len(recs) >= 50
    # the epilogue of main_call: threshold from the final factor, then `lofreq filter` with its defaults
    thr = la.snvqual_thresh(conf.sig, conf.bonf_subst)
    L = oracle.lib()
    assert thr == L.orc_snvqual_thresh(oconf.sig, oconf.bonf_subst)
    keep = la.filter_records(recs, thr, apply_defaults=True)
    n = len(recs)
    arr = lambda k: (C.c_int * n)(*[int(x) for x in recs[k]])
    k = (C.c_int * n)()
    L.orc_default_filter(arr("qual"), arr("dp"), arr("sb"), arr("alt_fw"), arr("alt_rv"), n, thr, 1, k)
    okeep = np.array([bool(k[i]) for i in range(n)])
    assert np.array_equal(keep, okeep)
    assert 0 < keep.sum() <= n
    # VCF text of the kept records: every line is well-formed and carries PASS
    text = la.format_vcf(recs, "synth", keep=keep, filter_str="PASS")
    lines = text.strip().split("\n")
    assert len(lines) == int(keep.sum()) and all(ln.split("\t")[6] == "PASS" for ln in lines)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("cfg,depth", [(3, 10000), (2, 1000)], ids=["C3", "C2"])
def test_config_full_batch(caller, oracle, cfg, depth):
    """BASELINE.json configs[2] (C3) and configs[1] (C2: 1000x, the shared-wavefront count kernel, every big column unsplit)
    at FULL size.  C3: the resident 10^6-column x 10 000x batch the benchmark times -- the
    work distribution that only exists there (eight per-XCD queues over 997 k light columns, the screen kernel's KREG
    instantiation chosen from the previous batch's histogram: <8> on a context's first step, <10> afterwards, the
    4096-segment pool budgets).  Step 1 and step 2 of the same context must return the same records; the batch is
    compared with the oracle run on all host cores (oracle/full_check.py): every column's integer outputs, every record,
    the VCF text -- or, on a host with few cores, every planted column plus every 10th (50th) column."""
    import os
    import full_check as fc
    import lofreq_amd as la
    import torch
    seed, ncols, period = 0x9E3779B97F4A7C15 ^ (cfg << 32), 1000000, 997
    own = la.SnvCaller(0)                               # a fresh context: its first step is really a first step
    own.set_dense_strand_counts(False)
    try:
        batch = own.synth_batch(seed, depth, ncols, plant_period=period)
        steps = []
        for _ in range(2):
            conf = la.VarcallConf()
            recs, _, st = own.call_snvs(batch, conf, records_capacity=1 << 16)
            steps.append((recs, int(st.n_tested), int(conf.bonf_subst), own.dp_work()))
        assert steps[0][0].tobytes() == steps[1][0].tobytes() and steps[0][1:3] == steps[1][1:3]
        recs, n_tested, bonf = steps[1][:3]
        assert len(recs) >= 500 and bonf == 3 * n_tested
        thr = la.snvqual_thresh(conf.sig, conf.bonf_subst)
        text = la.format_vcf(recs, "synth", keep=la.filter_records(recs, thr, apply_defaults=(cfg == 2)), filter_str="PASS")
        # dense integer outputs of the same (lazy strand count) instantiation: layer 1 on the resident batch
        dev = torch.device("cuda", 0)
        d_counts = torch.zeros(ncols * 64, dtype=torch.uint8, device=dev)
        d_pvals = torch.zeros(ncols * 128, dtype=torch.uint8, device=dev)
        own.snv_batch_device(batch, la.VarcallConf(), d_counts, d_pvals, ncols)
        st = own.batch_finish()
        assert int(st.n_tested) == n_tested
        counts = d_counts.cpu().numpy().view(la.COL_COUNTS_DTYPE).copy()
        del d_counts, d_pvals
    finally:
        del batch
        own.close()
        torch.cuda.empty_cache()
    procs = fc.default_procs()
    cols = None
    if procs < 48 * depth // 10000 or os.environ.get("LFQ_C3_SAMPLE") == "1":
        # ~1.6 ms of oracle per 10 000x column and core: all 10^6 columns need ~27 core-minutes (the bench hosts grant 16
        # cores); at 1000x every column fits
        stride = 10 if procs >= 12 else 50
        cols = np.union1d(np.arange(0, ncols, period), np.arange(0, ncols, stride))
    out = fc.check_batch(oracle, seed, depth, period, ncols, counts, recs, gpu_vcf_text=text, columns=cols, procs=procs,
                         default_filter=(cfg == 2), lazy_raw=True)
    print("C%d full batch: %s" % (cfg, {k: v for k, v in out.items() if k != "mismatches"}))
    assert out["identical"], out["mismatches"]
    # every emitted record's p-value against the exact (80-bit) tail too, not a sample: the sentinels have no finite tail
    assert out["n_pvalues_vs_80bit_truth"] >= out["records_compared"] - out["sentinel_pvalues"] - 5
    assert out["max_dlogp_device_vs_80bit_truth"] <= 2e-11
    assert out["records_compared"] == (len(recs) if cols is None else int(np.isin(recs["col"], cols).sum()))
    assert out["columns_compared"] == (ncols if cols is None else len(cols))
    if cols is None:
        assert out["vcf_text_identical"] and out["vcf_lines"] > 0


def test_deep_tail_against_80bit_truth(caller, oracle):
    """The tolerance story of DESIGN 5, on the DEVICE's values: every p-value beyond |log p| = 600 of deep columns
    (K from ~300 to ~1800 per allele at 10 000x: the mid and big classes, row-split and unsplit routes) against the 80-bit linear-space
    recurrence (orc_tail_truth) -- bar 2e-11 -- next to the reference's own log-space chain (the oracle), whose distance
    from the same truth is what the oracle-relative bar of 1e-9 pays for."""
    import lofreq_amd as la
    rng = np.random.default_rng(2024)
    afs = [0.1, 0.15, 0.2, 0.25, 0.3, 0.35, 0.4, 0.45, 0.5, 0.55, 0.0]      # random_batch spreads a planted rate over the three alt bases
    planted = {c: af for c, af in enumerate(afs * 3) if af > 0}
    host = util.random_batch(rng, len(afs) * 3, 9000, 10000, planted=planted)
    kw = dict(bonf_dynamic=0, bonf_subst=1, sig=1.0)           # nothing pruned: every allele has a value
    ores, oconf = util.run_oracle(oracle, host, **kw)
    counts, pvals, st = util.run_layer1(la, caller, host, la.VarcallConf(**kw))
    util.assert_counts_equal(counts, ores, host)
    worst_dev, worst_ref, n = 0.0, 0.0, 0
    for p in pvals:
        c = int(p["col"])
        tails, tlog, cnt = oracle.col_tail_truth(host, c, oconf)
        assert cnt == [int(x) for x in ores["alt_counts"][c]]
        for a in range(3):
            if cnt[a] == 0 or int(p["status"][a]) != la.LFQ_PV_LOG:
                continue
            if not np.isfinite(tlog[a]) or abs(tlog[a]) <= util.PV_DEEP_LOG or tlog[a] < -11300.0:
                continue
            d_dev = abs(float(p["logp"][a]) - tlog[a])
            d_ref = abs(float(ores["logp"][c, a]) - tlog[a])
            worst_dev, worst_ref, n = max(worst_dev, d_dev), max(worst_ref, d_ref), n + 1
            assert d_dev <= 2e-11, (c, a, float(p["logp"][a]), tlog[a], d_dev)
            # and the p-value the host hands out (expl of that log) against expl of the truth
            gpv = la.pvalue_from_log(p["logp"][a], int(p["status"][a]))
            assert abs(util.log_of(gpv) - tlog[a]) <= 2e-11
    print("deep tail vs 80-bit truth over %d p-values: device %.3g, reference log-space chain %.3g" % (n, worst_dev, worst_ref))
    assert n >= 50, n
    assert worst_ref <= 1e-9


@pytest.mark.parametrize("thr,kw", [(50, {}), (400, dict(def_alt_bq=-1)), (50, dict(min_jq=12, flag=7)),
                                    (50, dict(bonf_dynamic=0, bonf_subst=1000000))])
def test_approx_threshold_gate(caller, oracle, thr, kw):
    """-t / --approx-threshold (snpcaller.c:1128-1142): columns with more than thr error probabilities pass the Poisson gate
    first.  Against the oracle's restatement of the gate (pinned to scipy, not to GSL: parity unpinned); counts, the
    Bonferroni factors (the gate does not touch them) and every record bit-exact / 1e-10 as without the gate.  Marginal
    planted frequencies and many low qualities (Poisson variance above the Poisson-binomial one), so that the gate gives up
    columns the exact test calls."""
    import lofreq_amd as la
    rng = np.random.default_rng(31 + thr)
    ncols = 700
    planted = {c: float(af) for c, af in zip(range(1, ncols, 2), rng.uniform(0.05, 0.4, ncols))}
    host = util.random_batch(rng, ncols, 20, 1500, planted=planted, low_bq_frac=0.35, with_sq=True)
    conf_off = la.VarcallConf(**kw)
    recs_off, _, _ = caller.call_snvs(util.to_pileup_batch(la, host), conf_off, want_counts=True)
    ores, oconf = util.run_oracle(oracle, host, approx_threshold_n=thr, **kw)
    conf = la.VarcallConf(approx_threshold_n=thr, **kw)
    recs, counts, st = caller.call_snvs(util.to_pileup_batch(la, host), conf, want_counts=True)
    work = caller.dp_work()
    util.assert_counts_equal(counts, ores, host)
    assert conf.bonf_subst == oconf.bonf_subst == conf_off.bonf_subst
    assert conf.num_snv_tests == oconf.num_snv_tests
    assert st.n_tested == int(ores["tested"].sum())
    _compare_records(la, recs, ores, host)
    assert work["n_approx_pruned"] > 50
    assert work["n_approx_pruned"] + work["n_light"] + work["n_mid"] + work["n_big"] == st.n_tested
    assert len(recs) < len(recs_off)            # the gate cost calls here (lofreq_call.c:994: "might decrease number of calls")
    assert {(int(r["col"]), r["alt"]) for r in recs} <= {(int(r["col"]), r["alt"]) for r in recs_off}
    # deep columns below the threshold are not looked at
    conf_hi = la.VarcallConf(approx_threshold_n=100000, **kw)
    recs_hi, _, _ = caller.call_snvs(util.to_pileup_batch(la, host), conf_hi, want_counts=True)
    assert caller.dp_work()["n_approx_pruned"] == 0 and len(recs_hi) == len(recs_off)


@pytest.mark.parametrize("depth", [150, 300, 600, 1000, 3000])
def test_sparse_dense_entries(oracle, depth):
    """lfq_set_dense_counts(0): the shared-wavefront count kernel stores the dense entry of a TESTED column only -- those
    bit-identical to the dense run, the others untouched (a sentinel survives) --, class flags, work lists and the sparse
    output are those of the dense run (same p-value records, same tested count)."""
    import lofreq_amd as la
    import torch
    seed, ncols = 0x1234ABCD ^ (depth << 20), 40000 + depth        # not a multiple of 64: a last pass with idle lanes
    c = la.SnvCaller(0)
    try:
        c.set_dense_strand_counts(False)
        batch = c.synth_batch(seed, depth, ncols, plant_period=97)
        dev = torch.device("cuda", 0)
        out = []
        for dense in (True, False):
            c.set_dense_counts(dense)
            d_counts = torch.full((ncols * 64,), 0xA5, dtype=torch.uint8, device=dev)
            d_pvals = torch.zeros(ncols * 128, dtype=torch.uint8, device=dev)
            c.snv_batch_device(batch, la.VarcallConf(), d_counts, d_pvals, ncols)
            st = c.batch_finish()
            counts = d_counts.cpu().numpy().view(la.COL_COUNTS_DTYPE).copy()
            pv = d_pvals.cpu().numpy().view(la.COL_PVALS_DTYPE)[: st.n_pvals].copy()
            out.append((counts, pv[np.argsort(pv["col"], kind="stable")], int(st.n_tested), c.dp_work()))
        (cd, pd_, nd, wd), (cs, ps, ns, ws) = out
        assert nd == ns and nd > 0 and pd_.tobytes() == ps.tobytes()
        tested = cd["tested"] != 0
        assert int(tested.sum()) == nd and 0 < nd < ncols
        assert cs[tested].tobytes() == cd[tested].tobytes()
        assert np.all(cs[~tested].view(np.uint8) == 0xA5), "an untested column's entry was written"
        assert ws["bytes_written_count"] == ncols + 64 * nd and wd["bytes_written_count"] == 65 * ncols
    finally:
        c.close()
        torch.cuda.empty_cache()
