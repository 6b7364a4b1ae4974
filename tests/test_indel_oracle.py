"""CPU checks of the oracle's indel path (orc_call_indels_batch) -- hand-computed expectations."""
import numpy as np

from util import random_indel_columns


def _cols(la, dicts):
    from lofreq_amd.indel import IndelColumns
    return IndelColumns.from_columns(dicts)


def test_single_insertion_matches_snpcaller(oracle):
    """one column, one insertion event: the test is snpcaller over all reads with counts={n,0,0}"""
    from lofreq_amd.indel import IndelColumns
    n_ne, n_ev = 300, 12
    col = {"ref": "G", "coverage_plp": n_ne + n_ev, "num_tails": 0, "num_non_indels": n_ne, "hrun": 3,
           "ins": {"non_fw": 150, "non_rv": 150, "ne_q": [45] * n_ne, "ne_mq": [60] * n_ne,
                   "events": [{"key": "AC", "fw": 7, "rv": 5, "q": [45] * n_ev, "aq": [30] * n_ev,
                               "mq": [60] * n_ev}]},
           "dels": {"non_fw": 156, "non_rv": 156, "ne_q": [45] * (n_ne + n_ev), "ne_mq": [60] * (n_ne + n_ev),
                    "events": []}}
    cols = IndelColumns.from_columns([col])
    conf = oracle.default_conf()
    tests = oracle.call_indels_batch(cols.flat(), conf)
    assert len(tests) == 1 and conf.bonf_indel == 2 and conf.num_indel_tests == 1
    t = tests[0]
    assert t["n_err_probs"] == n_ne + n_ev and t["count"] == n_ev and t["side"] == 0
    ep = [oracle.lib().orc_merge_quals(-1, 60, -1, 45)] * n_ne + [oracle.lib().orc_merge_quals(-1, 60, 30, 45)] * n_ev
    pv, lp, _ = oracle.snpcaller(np.sort(np.asarray(ep)), [n_ev, 0, 0], 2, 0.01)
    assert t["pvalue"] == pv[0] and t["emitted"] == 1
    assert t["dp"] == n_ne + n_ev and t["af"] == np.float32(n_ev) / np.float32(n_ne + n_ev)
    assert (t["ref_fw"], t["ref_rv"], t["alt_fw"], t["alt_rv"], t["hrun"]) == (150, 150, 7, 5, 3)
    assert cols.ref_alt(0, 0, 0) == ("G", "GAC")


def test_gates_and_polyat_rule(oracle):
    from lofreq_amd.indel import IndelColumns

    def col(ref, ins_keys, del_keys, depth=200, cnt=4):
        d = {"ref": ref, "coverage_plp": depth, "num_tails": 0, "hrun": 1}
        used = 0
        for sn, keys in (("ins", ins_keys), ("dels", del_keys)):
            ev = [{"key": k, "fw": cnt // 2, "rv": cnt - cnt // 2, "q": [40] * cnt, "aq": [40] * cnt, "mq": [60] * cnt}
                  for k in keys]
            n_ne = depth - cnt * len(keys)
            used += cnt * len(keys)
            d[sn] = {"non_fw": n_ne // 2, "non_rv": n_ne - n_ne // 2, "ne_q": [40] * n_ne, "ne_mq": [60] * n_ne,
                     "events": ev}
        d["num_non_indels"] = depth - used
        return d

    dicts = [
        col("N", ["A"], []),               # 'N' reference: skipped
        col("C", ["A"], ["A"]),            # 1-bp A insertion AND deletion, both < 5%: both ignored
        col("C", ["A"], ["T"]),            # different base: both tested
        col("C", ["A", "AA"], ["A"]),      # ins A ignored, ins AA tested, del A ignored
        col("C", ["A"], ["A"], cnt=20),    # 10% AF: tested
        col("C", ["G"], []),
    ]
    cols = IndelColumns.from_columns(dicts)
    conf = oracle.default_conf()
    tests = oracle.call_indels_batch(cols.flat(), conf)
    got = [(int(t["col"]), int(t["side"]), cols.keys[t["side"]][t["event"]]) for t in tests]
    assert got == [(2, 0, "A"), (2, 1, "T"), (3, 0, "AA"), (4, 0, "A"), (4, 1, "A"), (5, 0, "G")]
    assert [int(t["bonf_used"]) for t in tests] == [2, 3, 4, 5, 6, 7]
    # min_cov on num_non_indels + num_ins + num_dels
    conf = oracle.default_conf()
    conf.min_cov = 201
    assert len(oracle.call_indels_batch(cols.flat(), conf)) == 0


def test_random_runs_and_fixed_bonf(oracle):
    from lofreq_amd.indel import IndelColumns
    rng = np.random.default_rng(5)
    cols = IndelColumns.from_columns(random_indel_columns(rng, 40))
    conf = oracle.default_conf()
    conf.bonf_dynamic = 0
    conf.bonf_indel = 1000
    tests = oracle.call_indels_batch(cols.flat(), conf)
    assert conf.bonf_indel == 1000 and conf.num_indel_tests == len(tests) > 10
    assert (tests["bonf_used"] == 1000).all()
    assert tests["emitted"].sum() > 0 and (tests["emitted"] == 0).sum() > 0
