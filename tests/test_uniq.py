"""`lofreq uniq --use-det-lim`: the oracle's restatement of uniq_snv's detection-limit branch
(lofreq_uniq.c:274-333) against the UNIQ flags the reference binary itself assigns."""
import numpy as np
import pytest

import golden_util as gu


@pytest.mark.parametrize("path", gu.uniq_fixtures(), ids=lambda p: p.split("/")[-1])
def test_oracle_uniq_detlim_matches_reference_binary(oracle, path):
    fx, host, af = gu.load_uniq(path)
    flag, pv = oracle.uniq_detlim_batch(host["nt"], host["bq"], None, host["mq"], None, host["col_off"], host["ref_base"], af)
    want = [v["uniq"] for v in fx["variants"]]
    assert flag.astype(bool).tolist() == want
    assert 20 < sum(want) < len(want) - 20
