"""CPU tests of oracle/full_check.py (the whole-batch checker of bench.py and test_config_full_batch): ranges run in
separate processes from their tested-column prefix reproduce ONE sequential run of the restated call_snvs loop
(lofreq_call.c:735-879) over the whole batch -- running Bonferroni factor included -- and a deviation is reported."""
import numpy as np

import full_check as fc

SEED, DEPTH, PLANT, NCOLS = 0x9E3779B97F4A7C15 ^ (3 << 32), 600, 7, 900


def _whole(oracle):
    host = oracle.synth_fill(SEED, DEPTH, PLANT, 0, NCOLS)
    conf = oracle.default_conf()
    res, _ = oracle.call_batch(host["nt"], host["bq"], host["baq"], host["mq"], None, host["col_off"],
                               host["ref_base"], conf)
    return host, conf, res


def _as_device(oracle, host, res):
    """the whole sequential run, in the shapes the device returns (dense counts + records in column order)"""
    import lofreq_amd._lib as _lib
    L = oracle.lib()
    counts = np.zeros(NCOLS, _lib.COL_COUNTS_DTYPE)
    for f in ("n_err_probs", "alt_counts", "alt_raw_counts"):
        counts[f] = res[f]
    counts["tested"] = res["tested"]
    recs = []
    for c in range(NCOLS):
        ref = int(host["ref_base"][c])
        for a in range(3):
            if res["emitted"][c, a]:
                rc, ac = b"ACGT".index(bytes([ref])), b"ACGT".index(bytes([int(res["alt_base"][c, a])]))
                r = np.zeros(1, _lib.SNV_RECORD_DTYPE)
                r["col"], r["qual"], r["dp"] = c, res["qual"][c, a], DEPTH
                r["alt_raw_count"], r["hqa"] = res["alt_raw_counts"][c, a], res["alt_counts"][c, a]
                r["ref_fw"], r["ref_rv"] = res["fw"][c, rc], res["rv"][c, rc]
                r["alt_fw"], r["alt_rv"] = res["fw"][c, ac], res["rv"][c, ac]
                r["sb"] = L.orc_sb_phred(int(r["ref_fw"][0]), int(r["ref_rv"][0]), int(r["alt_fw"][0]), int(r["alt_rv"][0]))
                r["ref"], r["alt"] = bytes([ref]), bytes([int(res["alt_base"][c, a])])
                r["pvalue"] = res["pvalue"][c, a]
                recs.append(r)
    return counts, np.concatenate(recs)


def test_ranges_in_processes_equal_the_sequential_run(oracle):
    host, conf, res = _whole(oracle)
    counts, recs = _as_device(oracle, host, res)
    assert len(recs) > 50 and conf.bonf_subst == 3 * int(res["tested"].sum())
    out = fc.check_batch(oracle, SEED, DEPTH, PLANT, NCOLS, counts, recs, procs=3, chunk_cols=100)     # 9 tasks, 3 processes
    assert out["identical"], out["mismatches"]
    assert out["columns_compared"] == NCOLS and out["records_compared"] == len(recs) == out["reference_records"]
    # a sample of single columns, each from its own prefix
    cols = np.arange(0, NCOLS, 7)
    out = fc.check_batch(oracle, SEED, DEPTH, PLANT, NCOLS, counts, recs, procs=2, columns=cols)
    assert out["identical"], out["mismatches"]
    assert out["columns_compared"] == len(cols) and out["records_compared"] == int(np.isin(recs["col"], cols).sum()) > 5


def test_a_deviation_is_reported(oracle):
    host, conf, res = _whole(oracle)
    counts, recs = _as_device(oracle, host, res)
    bad = recs.copy()
    bad["qual"][5] += 1
    out = fc.check_batch(oracle, SEED, DEPTH, PLANT, NCOLS, counts, bad, procs=1)
    assert not out["identical"] and "qual" in out["mismatches"][0]
    c2 = counts.copy()
    c2["alt_counts"][17, 0] += 1
    out = fc.check_batch(oracle, SEED, DEPTH, PLANT, NCOLS, c2, recs, procs=1)
    assert not out["identical"] and not out["counts_identical"]
    # a wrong prefix (one tested flag dropped) shifts the Bonferroni factor of everything behind it: caught by the flags
    c3 = counts.copy()
    c3["tested"][np.nonzero(counts["tested"])[0][3]] = 0
    out = fc.check_batch(oracle, SEED, DEPTH, PLANT, NCOLS, c3, recs, procs=1)
    assert not out["identical"]
