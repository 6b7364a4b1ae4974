"""-m gpu: BASELINE.json configs[3] and configs[4] as WORKLOADS, against the oracle (not against another run of the
device code): reads -> BAQ / IDAQ -> both pileups -> SNV + indel calls -> VCF text.

C4 shape: a region at 500x with planted SNVs and insertions / deletions, `--call-indels`; the device chain on a resident
read set against the whole oracle chain (oracle/orc_pileup.c + the pinned BAQ, call_snvs and call_indels restatements;
tests/oracle_chain.py), tags, test counts and every VCF line.
C5 shape: ragged BED targets at 200x, cut and dealt to two ranks by shard.plan_regions the way call-parallel cuts a BED
file (lofreq2_call_pparallel.py:569-613), merged through the shard exchange; rank 0's VCF against the oracle's
single-process loop over the targets with ONE running Bonferroni factor."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util as gu
import oracle_chain as oc
import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


def _procs():
    import full_check as fc
    return max(1, min(fc.default_procs(), 32))


def _snv_lines(la, recs, col_pos, conf, chrom="chr1"):
    keep = la.filter_records(recs, la.snvqual_thresh(conf.sig, conf.bonf_subst), apply_defaults=False)
    out = []
    for r, k in zip(recs, keep):
        if k:
            p0 = int(col_pos[int(r["col"])])
            out.append((p0, 1, gu.strip_hqa(la.format_vcf(np.array([r]), chrom, pos0=np.array([p0]), filter_str="PASS").rstrip("\n"))))
    return out


def test_config_c4_shape_reads_to_vcf_with_indels(caller, oracle):
    import lofreq_amd as la
    glen, depth = 60000, 500
    R = util.make_region_reads(404, glen, depth)
    assert R["n"] >= 190000 and R["n_indel_reads"] > 1500
    # ---- device: one resident read set, nothing leaves HBM between the steps
    rs = la.ReadSet.from_arrays(caller, R)
    rs.baq(extended=True, idaq=True)
    conf = la.VarcallConf(flag=la.LFQ_USE_BAQ | la.LFQ_USE_MQ | la.LFQ_USE_IDAQ)
    cols, col_pos = rs.pileup_indels(0, glen)
    irecs, n_indel_tests = la.call_indels(caller, cols, conf)
    thr_i = la.snvqual_thresh(conf.sig, conf.bonf_indel)
    ikeep = la.filter_indel_records(irecs, thr_i, apply_defaults=False)
    lines = [(int(col_pos[int(r["col"])]), 0, la.format_indel_record("chr1", int(col_pos[int(r["col"])]), cols, r, "PASS").rstrip("\n"))
             for r, k in zip(irecs, ikeep) if k]
    dt = rs.pileup_snv(0, glen)
    assert dt.col_pos.tolist() == col_pos.tolist() and len(col_pos) >= 50000
    la.skip_snv_columns(caller, cols.cons_indel)
    recs, _, st = caller.call_snvs(dt, conf)
    lines += _snv_lines(la, recs, dt.col_pos, conf)
    lines = [l[2] for l in sorted(lines, key=lambda t: (t[0], t[1]))]
    lb, ai, ad, fl = rs.fetch_tags(idaq=True)
    rs.close()
    # ---- oracle: BAQ / IDAQ of every read, compile_plp_col, call_indels + call_snvs, the epilogue of main_call
    P = dict(R)
    oracle.baq_idaq_reads(P, extended=True, idaq=True, procs=_procs())
    nb = int(R["seq_off"][-1])
    assert lb[:nb].tobytes() == P["lb"].tobytes()                                     # 30 M bases of lb, bit for bit
    has_ai, has_ad = (P["flags"][: R["n"]] & 4) != 0, (P["flags"][: R["n"]] & 8) != 0
    assert ((fl[: R["n"]] & 1) != 0).tolist() == has_ai.tolist() and ((fl[: R["n"]] & 2) != 0).tolist() == has_ad.tolist()
    per_base = lambda m: np.repeat(m, np.diff(R["seq_off"]))
    assert ai[:nb][per_base(has_ai)].tobytes() == P["ai"][per_base(has_ai)].tobytes()
    assert ad[:nb][per_base(has_ad)].tobytes() == P["ad"][per_base(has_ad)].tobytes()
    assert has_ai.sum() + has_ad.sum() >= R["n_indel_reads"]
    kw = dict(flag=la.LFQ_USE_BAQ | la.LFQ_USE_MQ | la.LFQ_USE_IDAQ)
    out = oc.call_region(oracle, P, R["ref"], 0, glen, kw, call_indels=True)
    assert out["col_pos"].tolist() == col_pos.tolist()
    assert conf.num_snv_tests == out["n_snv_tests"] and n_indel_tests == out["n_indel_tests"] and n_indel_tests > 30
    assert conf.bonf_subst == out["conf"].bonf_subst and conf.bonf_indel == out["conf"].bonf_indel
    assert lines == out["lines"]
    n_ind = sum(1 for l in lines if "INDEL" in l)
    assert n_ind >= 25 and len(lines) - n_ind >= 100, (n_ind, len(lines))
    # p-values (QUAL only shows their integer part)
    ores = out["snv"]
    exp = [(c, a) for c in np.nonzero(ores["emitted"].any(axis=1))[0] for a in range(3) if ores["emitted"][c, a]]
    assert len(exp) == len(recs)
    for r, (c, a) in zip(recs, exp):
        assert int(r["col"]) == c
        util.assert_pvalue_close(r["pvalue"], ores["pvalue"][c, a], ctx="col %d" % c)
    ot = out["indel_tests"]
    ot = ot[ot["emitted"] == 1]
    assert len(ot) == len(irecs)
    for r, t in zip(irecs, ot):
        assert (int(r["col"]), int(r["side"]), int(r["count"]), int(r["qual"])) == (int(t["col"]), int(t["side"]), int(t["count"]), int(t["qual"]))
        util.assert_pvalue_close(r["pvalue"], t["pvalue"], ctx="indel col %d" % int(r["col"]))


# ---- C5 shape: BED targets over two ranks -------------------------------------------------------------------------------

GLEN5, DEPTH5, SEED5 = 200000, 200, 505


def _targets():
    """ragged exome-like targets: 60 intervals of 150 .. 3500 bases with gaps, about 58 000 target positions"""
    rng = np.random.default_rng(55)
    t, x = [], 500
    while len(t) < 60 and x < GLEN5 - 5000:
        l = int(rng.choice([150, 300, 600, 1200, 2000, 3500], p=[0.2, 0.25, 0.2, 0.15, 0.12, 0.08]))
        t.append(("chr1", x, x + l))
        x += l + int(rng.integers(200, 3500))
    return t


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _c5_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lofreq_amd as la
    from lofreq_amd import shard
    import util as U
    dev = torch.device("cuda", 0)
    caller = la.SnvCaller(0)
    caller.set_dense_strand_counts(False)
    R = U.make_region_reads(SEED5, GLEN5, DEPTH5)
    rs = la.ReadSet.from_arrays(caller, R)
    rs.baq(extended=True, idaq=False)
    targets = _targets()
    bins, owner = shard.plan_regions(targets, lambda c, b, e: float(e - b), world)
    mine, pos_of = [], {}
    for i, ((_, lo, hi), o) in enumerate(zip(bins, owner)):
        if o != rank:
            continue
        conf = la.VarcallConf()
        dt = rs.pileup_snv(lo, hi)
        n = dt.ncols
        d_counts = torch.zeros(max(n, 1) * 64, dtype=torch.uint8, device=dev)
        d_pvals = torch.zeros(max(n, 1) * 128, dtype=torch.uint8, device=dev)
        caller.snv_batch_device(dt, conf, d_counts, d_pvals, max(n, 1))
        st = caller.batch_finish()
        pv = d_pvals[: st.n_pvals * 128].cpu().numpy().view(la.COL_PVALS_DTYPE).copy()
        pv["col"] = dt.col_pos[pv["col"]] - lo                # column of the bin -> offset from the bin's start ...
        mine.append((i, lo, pv, int(st.n_tested)))            # ... which finish_bins turns into the genome position
    conf = la.VarcallConf()
    recs, total = shard.finish_bins(conf, mine, len(bins), dist, None)
    if rank == 0:
        np.save(out, recs.view(np.uint8))
        np.save(out + ".meta", np.array([total, conf.bonf_subst, conf.num_snv_tests, len(bins), sum(o == 0 for o in owner)]))
    rs.close()
    caller.close()
    dist.destroy_process_group()


def test_config_c5_shape_bed_targets_two_ranks(tmp_path, oracle):
    import lofreq_amd as la
    out = str(tmp_path / "recs.npy")
    mp.spawn(_c5_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out).view(la.SNV_RECORD_DTYPE)
    total, bonf, ntests, nbins, nbins0 = np.load(out + ".meta.npy")
    assert nbins >= 4 and 0 < nbins0 < nbins
    # the oracle's single-process loop over the targets (`lofreq call -l targets.bed`): one running Bonferroni factor
    R = util.make_region_reads(SEED5, GLEN5, DEPTH5)
    P = dict(R)
    oracle.baq_idaq_reads(P, extended=True, idaq=False, procs=_procs())
    kw = dict(flag=la.LFQ_USE_BAQ | la.LFQ_USE_MQ | la.LFQ_USE_IDAQ)
    ref = oc.call_targets(oracle, P, R["ref"], _targets(), kw)
    assert ref["n_columns"] >= 50000
    assert total * 3 == ref["n_snv_tests"] == ntests and bonf == ref["conf"].bonf_subst
    conf = la.VarcallConf()
    conf.c.bonf_subst, conf.c.num_snv_tests = int(bonf), int(ntests)
    lines = [l[2] for l in _snv_lines(la, got, np.arange(GLEN5), conf)]       # finish_bins made `col` the genome position
    assert lines == ref["lines"]
    assert len(lines) >= 60
    exp = ref["emitted"]
    assert len(exp) == len(got)
    for r, (p0, pv) in zip(got, exp):
        assert int(r["col"]) == p0
        util.assert_pvalue_close(r["pvalue"], pv, ctx="pos %d" % p0)
