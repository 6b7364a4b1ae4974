"""The pure-Python restatement of compile_plp_col's indel part (golden_util.py_indel_pileup), which the GPU tests use
on random reads, is itself pinned here against the reference binary's column dump (tests/golden/plpindel_*.json)."""
import pytest

import golden_util as gu


@pytest.mark.parametrize("path", gu.plpindel_fixtures(), ids=lambda p: p.split("/")[-1])
def test_py_indel_pileup_matches_plpsummary(path):
    fx, reads = gu.load_plpindel(path)
    got = gu.py_indel_pileup(reads, fx["genome"])
    n_ev = 0
    for e in fx["columns"]:
        c = got[e["pos0"]]
        assert (c["cov"], c["tails"], c["non_indels"], c["n_ins"], c["n_dels"]) == (
            e["coverage_plp"], e["num_tails"], e["num_non_indels"], e["num_ins"], e["num_dels"]), e["pos0"]
        for sd, sn in enumerate(("ins", "dels")):
            E = e[sn]
            assert (c["non_fw"][sd], c["non_rv"][sd]) == (E["non_fw"], E["non_rv"])
            assert sorted(c["ne"][sd]) == sorted(zip(gu.dec(E["ne_q"]).tolist(), E["ne_mq"]))
            assert list(c["ev"][sd].keys()) == [ev["key"] for ev in E["events"]]
            for ev in E["events"]:
                m = c["ev"][sd][ev["key"]]
                assert [x[0] for x in m] == gu.dec(ev["q"]).tolist() and [x[1] for x in m] == gu.dec(ev["aq"]).tolist()
                assert [x[2] for x in m] == ev["mq"]
                assert (len(m) - sum(x[4] for x in m), sum(x[4] for x in m)) == (ev["fw"], ev["rv"])
                n_ev += 1
    assert n_ev >= 20
    for p, c in got.items():
        if p not in {e["pos0"] for e in fx["columns"]}:
            assert not c["ev"][0] and not c["ev"][1]
