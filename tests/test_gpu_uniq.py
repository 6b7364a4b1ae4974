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
    af = rng.choice(np.array([0.0, 0.0005, 0.002, 0.005, 0.01, 0.02, 0.05, 0.1, 0.25, 0.5, 0.9, 1.0], np.float32), n)
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
