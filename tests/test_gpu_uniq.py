"""-m gpu: `lofreq uniq --use-det-lim` on the device (lfq_uniq_detlim_batch) against the UNIQ
flags of the reference binary and, on seeded deep columns, against the oracle (flags equal, p-values within 1e-10)."""
import numpy as np
import pytest

import golden_util as gu
import util

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


@pytest.mark.parametrize("path", gu.uniq_fixtures(), ids=lambda p: p.split("/")[-1])
def test_uniq_detlim_matches_reference_binary(caller, path):
    import lofreq_amd as la
    fx, host, af = gu.load_uniq(path)
    det, pv = caller.uniq_detlim(util.to_pileup_batch(la, host), af)
    assert det.astype(bool).tolist() == [v["uniq"] for v in fx["variants"]]


@pytest.mark.parametrize("seed,lo,hi,n", [(1, 20, 400, 300), (2, 3000, 9000, 40), (3, 1, 30, 200)])
def test_uniq_detlim_random_vs_oracle(caller, oracle, seed, lo, hi, n):
    import lofreq_amd as la
    rng = np.random.default_rng(seed)
    host = util.random_batch(rng, n, lo, hi)
    host["baq"] = None                          # uniq's mpileup carries no BAQ (lofreq_uniq.c:465)
    # incl. AFs out of bounds: the reference logs them and RESETS them (af < 0 -> 0.01, af > 1 -> 1.0, lofreq_uniq.c:262-268)
    af = rng.choice(np.array([0.0, 0.0005, 0.002, 0.005, 0.01, 0.02, 0.05, 0.1, 0.25, 0.5, 0.9, 1.0, -0.3, 1.7], np.float32), n)
    flag, opv = oracle.uniq_detlim_batch(host["nt"], host["bq"], None, host["mq"], None, host["col_off"],
                                         host["ref_base"], af)
    det, pv = caller.uniq_detlim(util.to_pileup_batch(la, host), af)
    assert det.tolist() == flag.tolist()
    assert 0 < int(flag.sum()) < n
    ncmp = 0
    for c in range(n):
        if det[c]:                              # emitted: the exact value is there
            util.assert_pvalue_close(pv[c], opv[c], ctx="col %d af %g" % (c, af[c]))
            ncmp += 1
    assert ncmp > 10


@pytest.mark.parametrize("path", gu.uniq_binom_fixtures(), ids=lambda p: p.split("/")[-1])
def test_uniq_binom_matches_reference_binary(caller, path):
    """default mode of `lofreq uniq`: device base counts + host binomial test + MTC == the 2.1.4 binary's UQ= values
    and PASS / uq_fdr decisions"""
    import lofreq_amd as la
    fx, host, af = gu.load_uniq(path)
    alt = "".join(v["alt"] for v in fx["variants"])
    uq, pv = caller.uniq_binom(util.to_pileup_batch(la, host), af, alt)
    assert uq.tolist() == [(-1 if v["uq"] is None else v["uq"]) for v in fx["variants"]]
    assert la.uniq_mtc(uq, fx["mtc"], fx["alpha"], 0).tolist() == [v["filter"] == "PASS" for v in fx["variants"]]


@pytest.mark.parametrize("seed,lo,hi,n", [(1, 0, 400, 300), (2, 3000, 9000, 40)])
def test_uniq_binom_random_vs_oracle(caller, oracle, seed, lo, hi, n):
    import lofreq_amd as la
    rng = np.random.default_rng(seed)
    host = util.random_batch(rng, n, lo, hi, planted={c: float(rng.choice([0.01, 0.1, 0.4])) for c in range(0, n, 3)})
    host["baq"] = None
    af = rng.choice(np.array([0.0, 0.002, 0.01, 0.05, 0.1, 0.25, 0.5, 0.9, 1.0, -2.0, 1.25], np.float32), n)
    alt = "".join(rng.choice(list("ACGTN"), n))
    ouq, opv = oracle.uniq_binom_batch(host["nt"], host["col_off"], af, alt)
    uq, pv = caller.uniq_binom(util.to_pileup_batch(la, host), af, alt)
    assert uq.tolist() == ouq.tolist()
    oob = (af < 0) | (af > 1)
    assert (oob.sum() > 5 or seed > 2) and (ouq[oob & (np.diff(host["col_off"]).astype(np.int64) > 0)] >= 0).all()      # reset, not dropped (tests/stress_gpu.py calls this with other seeds)
    ok = ouq >= 0
    assert np.allclose(pv[ok], opv[ok], rtol=1e-11, atol=1e-300)
    for mtc in ("fdr", "holm"):
        assert la.uniq_mtc(uq, mtc, 0.001, 0).tolist() == oracle.uniq_mtc(ouq, mtc, 0.001, 0).tolist()
