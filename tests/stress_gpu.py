"""Not a pytest module: a longer randomised parity run for the GPU box (`python tests/stress_gpu.py [n_seeds]`).
Same comparisons as tests/test_gpu_parity.py::test_default_conf_random and tests/test_gpu_baq.py, over many more seeds
and shapes; prints what it covered and exits non-zero on the first difference."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))

import util                                   # noqa: E402
import test_gpu_parity as tp                  # noqa: E402
import test_gpu_baq as tb                     # noqa: E402


def main(n_seeds):
    import lofreq_amd as la
    import pyoracle as oracle
    oracle.build()
    caller = la.SnvCaller(0)
    n_rec = n_cols = 0
    shapes = [(0, 300, 300), (900, 1100, 150), (1, 70, 300), (2000, 6000, 40), (9000, 11000, 12), (100, 4000, 80),
              (30000, 60000, 3), (90000, 110000, 2), (9000, 11000, 6)]
    confs = [dict(), dict(), dict(), dict(min_jq=15), dict(min_alt_jq=20), dict(def_alt_bq=-1), dict(def_alt_bq=20),
             dict(def_alt_jq=25), dict(flag=2), dict(flag=1), dict(flag=0), dict(flag=7), dict(min_cov=40),
             dict(bonf_dynamic=0, bonf_subst=3000), dict(sig=0.05), dict(sig=1e-6)]
    for seed in range(100, 100 + n_seeds):
        rng = np.random.default_rng(seed)
        lo, hi, ncols = shapes[seed % len(shapes)]
        afs = rng.choice([0.003, 0.01, 0.02, 0.05, 0.1, 0.3, 0.5, 0.9, 1.0], 8)
        planted = {c: float(af) for c, af in zip(range(int(rng.integers(0, 9)), ncols, int(rng.integers(1, 41))), afs)}
        host = util.random_batch(rng, ncols, lo, hi, planted=planted, ref_n_frac=0.02, with_sq=bool(seed % 3 == 0),
                                 with_baq=bool(seed % 7 != 0), low_bq_frac=float(rng.choice([0.0, 0.02, 0.3])))
        if seed % 11 == 0:                      # a noisy run: qualities around Q10
            host["bq"][:] = np.clip(np.round(rng.normal(10, 4, len(host["bq"]))), 0, 41).astype(np.uint8)
        kw = dict(confs[(seed // 3) % len(confs)])
        if seed % 5 == 0:
            kw.update(min_bq=int(rng.integers(0, 20)), min_alt_bq=int(rng.integers(0, 25)))
        ores, oconf = util.run_oracle(oracle, host, **kw)
        conf = la.VarcallConf(**kw)
        recs, counts, st = caller.call_snvs(util.to_pileup_batch(la, host), conf, want_counts=True)
        util.assert_counts_equal(counts, ores, host)
        assert conf.bonf_subst == oconf.bonf_subst and conf.num_snv_tests == oconf.num_snv_tests, seed
        try:
            tp._compare_records(la, recs, ores, host)
        except AssertionError:
            print("FAILED at seed", seed, "shape", (lo, hi, ncols), "planted", planted, "conf", kw)
            raise
        n_rec += len(recs)
        n_cols += ncols
    print("snv: %d seeds, %d columns, %d records identical" % (n_seeds, n_cols, n_rec))
    n_reads = 0
    for seed in range(200, 200 + max(n_seeds // 4, 2)):
        rng = np.random.default_rng(seed)
        genome = "".join(rng.choice(list("ACGT"), 4000))
        reads = tb._random_reads(rng, genome, 500, 20, int(rng.choice([100, 160, 260])))
        for extended in (True, False):
            out = la.baq_batch(caller, reads, genome.encode(), extended=extended, idaq=True)
            for r, (lb, ai, ad) in zip(reads, out):
                elb, eai, ead = oracle.baq_idaq_read(r["pos0"], r["cigar"], r["seq"], r["qual"], genome.encode(), extended)
                assert lb.tobytes() == elb.tobytes(), (seed, r["pos0"], r["cigar"])
                assert (ai is None) == (eai is None) and (ad is None) == (ead is None)
                assert ai is None or ai.tobytes() == eai.tobytes()
                assert ad is None or ad.tobytes() == ead.tobytes()
        n_reads += len(reads)
    print("baq: %d reads x 2 modes identical (lb, ai, ad)" % n_reads)
    # the seed-parameterised bodies of the other GPU parity tests, over more seeds and shapes
    import test_gpu_indel as ti
    import test_gpu_uniq as tu
    n_other = 0
    for seed in range(300, 300 + max(n_seeds // 6, 3)):
        rng = np.random.default_rng(seed)
        lo = int(rng.choice([1, 20, 300, 1500, 6000]))
        hi = lo + int(rng.integers(10, 3000))
        n = int(max(40, 40000 // hi))
        for name, fn in (("indel", ti.test_indel_default_conf_random), ("uniq detlim", tu.test_uniq_detlim_random_vs_oracle),
                         ("uniq binom", tu.test_uniq_binom_random_vs_oracle)):
            try:
                fn(caller, oracle, seed, lo, hi, n)
            except AssertionError:
                print("FAILED", name, "seed", seed, "depth", (lo, hi), "n", n)
                raise
            n_other += 1
    print("indel / uniq: %d seeded runs identical to the oracle" % n_other)
    print("p-value deviations:", dict(util.PV_ERR_MAX) if hasattr(util, "PV_ERR_MAX") else "")
    caller.close()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 60)
