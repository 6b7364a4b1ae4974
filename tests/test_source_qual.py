"""Source quality: the oracle's restatement of source_qual / count_cigar_ops (plp.c:427-593,
samutils.c:437-614) against the SQ track the reference binary itself prints (`lofreq plpsummary -s`), read by read
through the pileup."""
import json

import numpy as np
import pytest

import golden_util as gu


def expected_sq_per_read(fx, reads):
    """the binary's SQ of every read that shows up in at least one column (all its entries must agree)"""
    plp = gu.py_pileup(reads)
    sq = {}
    for c in fx["columns"]:
        for letter, o in c["obs"].items():
            members = plp[c["pos0"]].get(letter, [])
            vals = gu.dec(o["sq"])
            assert len(members) == len(vals), (c["pos0"], letter)
            for (ri, _), v in zip(members, vals.tolist()):
                v = 49314 if v < 0 else v
                assert sq.setdefault(ri, v) == v, (ri, c["pos0"])
    return sq


@pytest.mark.parametrize("path", gu.srcq_fixtures(), ids=lambda p: p.split("/")[-1])
def test_oracle_source_qual_matches_plpsummary(oracle, path):
    fx, reads, nmq, ign = gu.load_srcq(path)
    want = expected_sq_per_read(fx, reads)
    assert len(want) > 150
    ref = fx["genome"].encode()
    seen = set()
    for ri, v in want.items():
        r = reads[ri]
        got = oracle.source_qual(r["pos0"], r["cigar"], r["seq"], r["qual"], ref, nonmatch_qual=nmq, min_bq=6, ign=ign)
        assert max(got, 0) == v, (ri, got, v)               # mplp_func stores max(sq, 0) (plp.c:731-733)
        seen.add(v)
    assert len(seen) >= 4                                   # 49314, 0 and a few real phred values


def test_oracle_source_qual_edges(oracle):
    ref = b"ACGTACGTACGTACGTACGT"
    seq = np.array([0, 1, 2, 3] * 3, np.uint8)
    q = np.full(12, 30, np.uint8)
    # perfect read, one mismatch: PROB_TO_PHREDQUAL(LDBL_MIN) (plp.c:517-524)
    assert oracle.source_qual(0, [("M", 12)], seq, q, ref) == 49314
    s1 = seq.copy(); s1[5] = 3
    assert oracle.source_qual(0, [("M", 12)], s1, q, ref) == 49314
    # every base below min_bq: count_cigar_ops returns 0 -> NA (plp.c:468-474)
    assert oracle.source_qual(0, [("M", 12)], seq, np.full(12, 5, np.uint8), ref) == -1
    # X ops are mismatches whatever the bases (samutils.c:489); two of them leave K = 1
    v = oracle.source_qual(0, [("M", 5), ("X", 2), ("M", 5)], seq, q, ref)
    assert 0 <= v < 100
    # an indel counts as one non-match of quality 45 (samutils.c:562-575)
    v2 = oracle.source_qual(0, [("M", 4), ("I", 4), ("M", 2), ("D", 2), ("M", 2)], seq, q, ref)
    assert 0 <= v2 < 100
    # the ignore list drops mismatches and indels at listed positions (samutils.c:505-519, 535-556)
    ign = np.zeros(len(ref), np.uint8); ign[5] = 1; ign[6] = 1
    assert oracle.source_qual(0, [("M", 5), ("X", 2), ("M", 5)], seq, q, ref, ign=ign) == 49314
