"""CPU tests: the oracle (C restatement) against every known-answer vector available for this path:
the reference's own source comments, SURVEY.md App. C (produced by the compiled HEAD reference during
the survey) and exact / high-precision arithmetic."""
import ctypes as C
import math

import numpy as np
import pytest

LDBL_MAX = np.finfo(np.longdouble).max
LDBL_MIN = np.finfo(np.longdouble).tiny


def test_merge_quals_app_c1(oracle):
    L = oracle.lib()
    table = [(-1, 60, 40, 30, "0x1.20981316a9bcbp-10", 29), (-1, 0, -1, 20, "0x1.028f5c28f5c29p-1", 2),
             (-1, -1, -1, 30, "0x1.0624dd2f1a9fcp-10", 29), (20, 60, 93, 41, "0x1.4a4a0e449c059p-7", 19),
             (-1, 255, 93, 2, "0x1.430cd75104027p-1", 1), (-1, 20, 20, 20, "0x1.e69f05ea24cc7p-6", 15),
             (-1, 60, -1, 6, "0x1.0137cabe5231ap-2", 5)]
    for sq, mq, baq, bq, hexval, q in table:
        v = L.orc_merge_quals(sq, mq, baq, bq)
        assert v == float.fromhex(hexval), (sq, mq, baq, bq, v.hex())
        assert L.orc_prob_to_phred_safe(v) == q


def test_log_sum_app_c2(oracle):
    L = oracle.lib()
    assert L.orc_log_sum(-1, -2) == -0.68673831248177719
    assert L.orc_log_sum(-1000, -3) == -3
    assert L.orc_log_sum(-5, -5) == -4.3068528194400546


def _mixed_vector():
    libm = C.CDLL("libm.so.6")
    libm.pow.restype = C.c_double
    libm.pow.argtypes = [C.c_double, C.c_double]
    return np.array(sorted(libm.pow(10.0, -(20 + (7 * i % 21)) / 10.0) for i in range(5000)))


def test_poissbin_app_c3(oracle):
    rows = [([0.001] * 10, 1, -4.6096683083639114, "0.00995511979025179"),
            ([0.01] * 100, 5, -5.674518399020565, "0.00343232158775454"),
            ([0.001] * 1000, 9, -13.726026845941917, "1.09360952047986e-06"),
            ([0.001] * 1000, 1, -0.45838407809203952, "0.632304575229035"),
            ([0.001] * 10000, 30, -15.217298556132013, "2.46156442964619e-07"),
            ([0.0001] * 10000, 12, -20.913200269145964, "8.27013357612479e-10"),
            ([0.001] * 10000, 1000, -3670.2412275820329, "1.08264882105079e-1594"),
            ([0.05] * 200, 40, -30.381759208595003, "6.38806939302205e-14")]
    for ep, k, logv, pv in rows:
        vec, p, _ = oracle.poissbin(ep, k)
        assert vec[k] == logv, (k, repr(vec[k]))
        assert abs(p / np.longdouble(pv) - 1) < 1e-13
    mixed = _mixed_vector()
    for k, logv in [(5, -9.3543434228839269e-06), (20, -0.53142912354294658), (60, -27.50242589174858)]:
        vec, p, _ = oracle.poissbin(mixed, k)
        assert vec[k] == logv
    # pruning example: early exit returns a lower bound (snpcaller.c:1017)
    vec, p, rows_done = oracle.poissbin(mixed, 5, 3000000, 0.01)
    assert vec[5] == -19.485958181295725 and rows_done < 5000
    pv, _, _ = oracle.snpcaller(mixed, (60, 20, 5), 3000000, 0.01)
    for got, exp in zip(pv, ["1.13722970790175e-12", "0.587764381254103", "0.999990645700225"]):
        assert abs(got / np.longdouble(exp) - 1) < 1e-13


def test_reference_comment_kat_poibin(oracle):
    """snpcaller.c:1219-1232: R poibin, ppoibin(kk=9, pp=rep(0.999,10)) = 0.00995512"""
    _, p, _ = oracle.poissbin([0.001] * 10, 1)
    assert abs(float(p) - 0.0099551198) < 5e-11


def test_fe_clamp_table_app_a6(oracle):
    ep = [0.001] * 10000
    pv, _, _ = oracle.snpcaller(ep, (60, 40, 0), 3000000, 0.01)
    assert oracle.prob_to_phred(pv[0]) == 262 and oracle.prob_to_phred(pv[1]) == 121
    pv, _, _ = oracle.snpcaller(ep, (1000, 40, 0), 3000000, 0.01)
    assert oracle.prob_to_phred(pv[0]) == 15939 and pv[1] == LDBL_MAX
    pv, _, _ = oracle.snpcaller(ep, (1000, 60, 0), 3000000, 0.01)
    assert pv[1] == LDBL_MIN and oracle.prob_to_phred(pv[1]) == 49314
    for c in ((5000, 0, 0), (9000, 0, 0)):
        pv, _, _ = oracle.snpcaller(ep, c, 3000000, 0.01)
        assert pv[0] == LDBL_MIN
    ep = [0.001] * 1000
    pv, _, _ = oracle.snpcaller(ep, (100, 12, 0), 3000000, 0.01)
    assert oracle.prob_to_phred(pv[0]) == 1605 and oracle.prob_to_phred(pv[1]) == 91
    pv, _, _ = oracle.snpcaller(ep, (300, 12, 0), 3000000, 0.01)
    assert oracle.prob_to_phred(pv[0]) == 6365 and pv[1] == LDBL_MAX


def test_threshold_arithmetic_app_c4(oracle):
    L = oracle.lib()
    assert L.orc_snvqual_thresh(0.01, 3000000) == 84
    assert L.orc_snvqual_thresh(0.01, 29727) == 64      # the denv2 factor, tests/bonf_auto_vs_dyn.sh:17
    assert oracle.prob_to_phred(LDBL_MIN) == 49314


def test_strand_bias_app_c5(oracle):
    l, r, t = C.c_double(), C.c_double(), C.c_double()
    oracle.lib().orc_fisher_exact(138, 140, 11, 13, l, r, t)
    assert abs(t.value - 0.8323994524) < 5e-11
    assert oracle.lib().orc_sb_phred(138, 140, 11, 13) == 0
    assert oracle.lib().orc_sb_phred(0, 0, 5, 0) == 2147483647          # lofreq_call.c:122-123


def _fdr(oracle, p, alpha, ntests=0):
    a = np.asarray(p, np.float64)
    rej = (C.c_long * len(a))()
    n = oracle.lib().orc_fdr(a.ctypes.data_as(C.POINTER(C.c_double)), len(a), alpha, ntests, rej)
    return n, sorted(rej[i] for i in range(n))


# R p.adjust(BH) cross-check vector of multtest.c:219-241
P50 = [2.354054e-07, 2.101590e-05, 2.576842e-05, 9.814783e-05, 1.052610e-04, 1.241481e-04, 1.325988e-04,
       1.568503e-04, 2.254557e-04, 3.795380e-04, 6.114943e-04, 1.613954e-03, 3.302430e-03, 3.538342e-03,
       5.236997e-03, 6.831909e-03, 7.059226e-03, 8.805129e-03, 9.401040e-03, 1.129798e-02, 2.115017e-02,
       4.922736e-02, 6.053298e-02, 6.262239e-02, 7.395153e-02, 8.281103e-02, 8.633331e-02, 1.190654e-01,
       1.890796e-01, 2.058494e-01, 2.209214e-01, 2.856000e-01, 3.048895e-01, 4.660682e-01, 4.830809e-01,
       4.921755e-01, 5.319453e-01, 5.751550e-01, 5.783195e-01, 6.185894e-01, 6.363620e-01, 6.448587e-01,
       6.558414e-01, 6.885884e-01, 7.189864e-01, 8.179539e-01, 8.274487e-01, 8.971300e-01, 9.118680e-01,
       9.437890e-01]
# multtest.c:334-340 (biostathandbook set, also tests/fdr.sh) and multtest.c:409-415
P25 = [0.000001, 0.008, 0.039, 0.041, 0.042, 0.06, 0.074, 0.205, 0.212, 0.216, 0.222, 0.251, 0.269, 0.275,
       0.34, 0.341, 0.384, 0.569, 0.594, 0.696, 0.762, 0.94, 0.942, 0.975, 0.986]
P10 = [0.010, 0.013, 0.014, 0.190, 0.350, 0.500, 0.630, 0.670, 0.750, 0.810]


def test_multtest_kats(oracle):
    L = oracle.lib()
    dp = C.POINTER(C.c_double)
    # multtest.c:486-508 / multiple_testing.py:52-53, 83-84
    p = np.array([0.01, 0.01, 0.03, 0.05, 0.005])
    b = p.copy()
    L.orc_bonf_corr(b.ctypes.data_as(dp), 5, 0)
    assert np.allclose(b, [0.05, 0.05, 0.15, 0.25, 0.025], rtol=0, atol=1e-15)
    b = p.copy()
    L.orc_bonf_corr(b.ctypes.data_as(dp), 5, 999)
    assert np.allclose(b, [9.99, 9.99, 29.97, 49.95, 4.995], rtol=1e-15, atol=0)
    h = p.copy()
    L.orc_holm_bonf_corr(h.ctypes.data_as(dp), 5, 1.0, 0)
    assert np.allclose(h, [0.04, 0.04, 0.06, 0.05, 0.025], rtol=0, atol=1e-15)
    h = p.copy()
    L.orc_holm_bonf_corr(h.ctypes.data_as(dp), 5, 100.0, 999)
    assert np.allclose(h, [9.98, 9.98, 29.88, 49.75, 4.995], rtol=1e-15, atol=0)
    # fdr.py:35-43 / multtest.c:503-506
    pv = [0.6, 0.07, 0.49, 0.2, 0.48, 0.74, 0.68, 0.01, 0.97, 0.38, 0.032, 0.07]
    n, idx = _fdr(oracle, pv, 0.20)
    assert n == 2 and sorted(pv[i] for i in idx) == [0.01, 0.032]
    # multtest.c:219-241: R p.adjust(p, "BH", n) cross-check
    for ntests, alpha, exp in [(50, 0.05, 20), (1000, 0.05, 10), (100, 0.001, 3), (10000, 1.0, 11)]:
        assert _fdr(oracle, P50, alpha, ntests)[0] == exp
    # multtest.c:334-340 (exp_sig = 5 at alpha 0.25) and :409-415 (exp_sig = 3 at alpha 0.05)
    assert _fdr(oracle, P25, np.float32(0.25), 25)[0] == 5
    assert _fdr(oracle, P10, np.float32(0.05), 10)[0] == 3


def test_exp_underflow_threshold():
    """The constant the DP kernel uses to predict FE_UNDERFLOW inside the reference's log_sum chain:
    glibc exp() raises it exactly when the result drops below DBL_MIN."""
    libm = C.CDLL("libm.so.6")
    libm.exp.restype = C.c_double
    libm.exp.argtypes = [C.c_double]
    libm.feclearexcept.argtypes = [C.c_int]
    libm.fetestexcept.argtypes = [C.c_int]
    FE_UNDERFLOW, FE_ALL = 0x10, 0x3D

    def raises(x):
        libm.feclearexcept(FE_ALL)
        libm.exp(x)
        return bool(libm.fetestexcept(FE_UNDERFLOW))

    t = -708.3964185322641
    assert t == math.log(2.0 ** -1022) or abs(t - math.log(2.0 ** -1022)) < 2e-13
    assert not raises(np.nextafter(t, 0.0) + 1e-9)
    assert raises(t - 1e-9)
    assert raises(-745.0) and raises(-1000.0) and not raises(-700.0)


def test_linear_dp_truth_and_reference_noise(oracle):
    """The tolerance story of DESIGN.md section 5: the reference's log-space DP carries its own rounding noise -- N row
    updates, each rounding numbers of magnitude |log p|; an 80-bit linear-space recurrence (orc_tail_truth, checked here
    against a numpy long-double loop) is the ground truth both the oracle and the device are compared with.  This test
    MEASURES the noise as a = |d log p| / (ulp(|log p|) * N) and pins the constant of tests/util.py::pv_deep_bound to it:
    identical probabilities are the worst case (the roundings of consecutive rows line up and add linearly: a = 0.06 ..
    0.12, whatever N), ragged ones stay two orders of magnitude below (a random walk, ~sqrt(N))."""
    import util

    def truth_numpy(ep, k):
        v = np.zeros(k + 1, np.longdouble)
        v[0] = 1
        for p in ep:
            p = np.longdouble(p)
            q = 1 - p
            tail = v[k] + v[k - 1] * p
            v[1:k] = v[1:k] * q + v[0:k - 1] * p
            v[0] *= q
            v[k] = tail
        return float(np.log(v[k]))

    ep = np.full(2000, 0.001)
    assert abs(oracle.tail_truth(ep, [10, 0, 0])[1][0] - truth_numpy(ep, 10)) < 1e-14
    rng = np.random.default_rng(5)
    seen = {}
    for n, k, p, ragged in [(2000, 10, 0.001, False), (10000, 100, 0.001, False), (10000, 1000, 0.001, False),
                            (10000, 300, 0.001, True), (10000, 2000, 0.001, True), (30000, 1500, 0.002, True),
                            (45000, 1500, 0.01, False), (45000, 2500, 0.02, False), (45000, 3000, 0.02, False)]:
        ep = np.sort(np.clip(p * 10 ** rng.normal(0, 0.5, n), 1e-6, 0.3)) if ragged else np.full(n, p)
        vec, _, _ = oracle.poissbin(ep, k)
        logp = oracle.tail_truth(ep, [k, 0, 0])[1][0]
        d = abs(vec[k] - logp)
        seen[(n, k, ragged)] = (logp, d, d / (np.spacing(abs(logp)) * n))
        if abs(logp) > util.PV_DEEP_LOG:
            assert d <= util.pv_deep_bound(logp, n) / 1.3, (n, k, logp, d)       # the bound holds with its margin
        else:
            assert d < util.PV_LOG_TOL, (n, k, logp, d)                          # below the line 1e-10 is met outright
    # the anchors DESIGN.md quotes
    assert 2.0e-11 < seen[(10000, 100, False)][1] < 3.2e-11 and 2.2e-10 < seen[(10000, 1000, False)][1] < 3.2e-10
    assert 5.0e-10 < seen[(45000, 2500, False)][1] < 7.5e-10
    worst_a = max(v[2] for v in seen.values())
    assert 0.08 < worst_a <= util.PV_NOISE_A / 1.3, worst_a
    assert max(v[2] for key, v in seen.items() if key[2]) < 0.005                # ragged probabilities: far below


def test_poisson_cdf_of_the_approximation_gate_against_scipy(oracle):
    """-t / --approx-threshold (snpcaller.c:1128-1142) needs gsl_cdf_poisson_P, and GSL is neither in the reference tree
    nor in this image: the gate is PARITY UNPINNED.  What can be pinned is the function itself -- the oracle's long-double
    evaluation of P(X <= k), X ~ Poisson(mu), against scipy.special.pdtr / pdtrc (cephes: another implementation of the
    same definition) over the (k, mu) range a pileup produces."""
    import ctypes as C
    from scipy import special
    L = oracle.lib()
    L.orc_poisson_cdf.restype = C.c_double
    L.orc_poisson_cdf.argtypes = [C.c_uint, C.c_double]
    rng = np.random.default_rng(11)
    worst_abs = worst_rel_tail = 0.0
    n_tail = 0
    for _ in range(4000):
        mu = float(10 ** rng.uniform(-3, 4.5))
        k = int(max(0, mu + rng.normal() * 7 * np.sqrt(mu) + rng.integers(0, 6)))
        cdf = L.orc_poisson_cdf(k, mu)
        worst_abs = max(worst_abs, abs(cdf - special.pdtr(k, mu)))
        sf = special.pdtrc(k, mu)
        if sf > 1e-13:                      # (1 - cdf keeps 16 digits of 1: below that the reference's own value is noise)
            n_tail += 1
            worst_rel_tail = max(worst_rel_tail, abs((1.0 - cdf) - sf) / sf - 1.2e-16 / sf)
    assert worst_abs < 1e-13, worst_abs
    assert n_tail > 1000 and worst_rel_tail < 1e-10, (n_tail, worst_rel_tail)
    assert np.isnan(L.orc_poisson_cdf(3, 0.0))          # GSL: domain error


def test_approximation_gate_semantics(oracle):
    """orc_approx_gate: only above the threshold (:1131), mean = plain sum (:1133-1135), never a call the exact test
    would not make (the gate only gives columns up), off for approx_threshold_n <= 0 (:1131)"""
    import ctypes as C
    L = oracle.lib()
    L.orc_approx_gate.restype = C.c_int
    L.orc_approx_gate.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.c_longlong, C.c_double, C.c_int,
                                  C.POINTER(C.c_longdouble)]
    ep = np.sort(np.full(400, 1e-3))
    p = ep.ctypes.data_as(C.POINTER(C.c_double))
    out = C.c_longdouble()
    # mean 0.4: one mismatch is unremarkable (tail 0.33), thirty are not
    assert L.orc_approx_gate(p, 400, 1, 3, 0.01, 100, C.byref(out)) == 1
    assert abs(float(out.value) - (1 - np.exp(-0.4))) < 1e-15
    assert L.orc_approx_gate(p, 400, 30, 3, 0.01, 100, C.byref(out)) == 0
    assert L.orc_approx_gate(p, 400, 1, 3, 0.01, 400, None) == 0         # n_ep > threshold is strict
    assert L.orc_approx_gate(p, 400, 1, 3, 0.01, 399, None) == 1
    assert L.orc_approx_gate(p, 400, 1, 3, 0.01, 0, None) == 0
    assert L.orc_approx_gate(p, 400, 1, 3, 0.01, -1, None) == 0
    # a column the gate gives up produces no p-values, its Bonferroni bump stays (lofreq_call.c:794-801 runs before)
    rng = np.random.default_rng(3)
    import util
    host = util.random_batch(rng, 60, 300, 900, planted={c: 0.012 for c in range(2, 60, 3)}, low_bq_frac=0.3)
    off, _ = util.run_oracle(oracle, host)
    on, conf_on = util.run_oracle(oracle, host, approx_threshold_n=100)
    assert np.array_equal(off["tested"], on["tested"]) and conf_on.bonf_subst == 3 * int(on["tested"].sum())
    assert (on["emitted"] <= off["emitted"]).all()
