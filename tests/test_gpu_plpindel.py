"""-m gpu: the indel fields of the pileup on the device (lfq_pileup_indel_columns) against the
reference binary's own column dump (`lofreq plpsummary`), then reads -> columns -> indel calls against
`lofreq call --call-indels --only-indels`, and the whole chain reads -> BAQ/IDAQ -> both pileups -> SNV + indel calls
against `lofreq call --call-indels` (tests/golden/plpindel_*.json)."""
import os

import numpy as np
import pytest

import golden_util as gu

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def _indel_lines(la, caller, cols, col_pos, conf):
    recs, ntests = la.call_indels(caller, cols, conf)
    thr = la.snvqual_thresh(conf.sig, conf.bonf_indel)                  # lofreq_call.c:1529-1534
    keep = la.filter_indel_records(recs, thr, apply_defaults=False)
    lines = [(int(col_pos[int(r["col"])]), 0, la.format_indel_record("chr1", int(col_pos[int(r["col"])]), cols, r,
                                                                      "PASS").rstrip("\n"))
             for r, k in zip(recs, keep) if k]
    return lines, ntests


@pytest.mark.parametrize("path", gu.plpindel_fixtures(), ids=lambda p: p.split("/")[-1])
def test_indel_columns_match_plpsummary(caller, path):
    import lofreq_amd as la
    fx, reads = gu.load_plpindel(path)
    ref = fx["genome"].encode()
    cols, col_pos = la.pileup_indel_columns(caller, reads, ref, 0, len(ref))
    col_of = {int(p): i for i, p in enumerate(col_pos)}
    n_ev = 0
    for e in fx["columns"]:
        c = col_of[e["pos0"]]
        ctx = "pos0 %d" % e["pos0"]
        assert chr(cols.ref_base[c]) == e["ref"], ctx
        assert bool(cols.cons_indel[c]) == (e["cons"][0] in "+-"), (ctx, e["cons"])        # plp.c:1236-1270
        for k in ("coverage_plp", "num_tails", "num_non_indels", "num_ins", "num_dels", "hrun"):
            assert int(getattr(cols, k)[c]) == e[k], (ctx, k)
        for sd, sn in enumerate(("ins", "dels")):
            S, E = cols.sides[sd], e[sn]
            assert int(S["non_fw"][c]) == E["non_fw"] and int(S["non_rv"][c]) == E["non_rv"], (ctx, sn)
            a, b = int(S["ne_off"][c]), int(S["ne_off"][c + 1])
            # position-sorted reads: the column-major kernel writes the arrays in pileup order = the reference's order
            got, want = list(zip(S["ne_q"][a:b].tolist(), S["ne_mq"][a:b].tolist())), list(zip(gu.dec(E["ne_q"]).tolist(), E["ne_mq"]))
            if os.environ.get("LFQ_PILEUP_ATOMIC"):
                got, want = sorted(got), sorted(want)
            assert got == want, (ctx, sn)
            e0, e1 = int(S["ev_off"][c]), int(S["ev_off"][c + 1])
            assert [cols.keys[sd][i] for i in range(e0, e1)] == [ev["key"] for ev in E["events"]], (ctx, sn)
            for i, ev in zip(range(e0, e1), E["events"]):
                assert int(S["ev_fw"][i]) == ev["fw"] and int(S["ev_rv"][i]) == ev["rv"], (ctx, ev["key"])
                r0, r1 = int(S["rd_off"][i]), int(S["rd_off"][i + 1])
                for name, want in (("rd_q", gu.dec(ev["q"]).tolist()), ("rd_aq", gu.dec(ev["aq"]).tolist()),
                                   ("rd_mq", ev["mq"]), ("rd_sq", gu.dec(ev["sq"]).tolist())):
                    assert S[name][r0:r1].tolist() == want, (ctx, ev["key"], name)
                n_ev += 1
    assert n_ev >= 20
    # columns the binary printed no event for have none here either
    with_ev = {e["pos0"] for e in fx["columns"]}
    for c, p in enumerate(col_pos):
        if int(p) not in with_ev:
            assert not cols.cons_indel[c]
            assert cols.sides[0]["ev_off"][c] == cols.sides[0]["ev_off"][c + 1]
            assert cols.sides[1]["ev_off"][c] == cols.sides[1]["ev_off"][c + 1]


@pytest.mark.parametrize("path", gu.plpindel_fixtures(), ids=lambda p: p.split("/")[-1])
def test_reads_to_indel_vcf(caller, path):
    import lofreq_amd as la
    fx, reads = gu.load_plpindel(path)
    ref = fx["genome"].encode()
    cols, col_pos = la.pileup_indel_columns(caller, reads, ref, 0, len(ref))
    kw, _ = gu.conf_kwargs(fx["call_args"])
    conf = la.VarcallConf(**kw)
    lines, ntests = _indel_lines(la, caller, cols, col_pos, conf)
    assert ntests == fx["only_indels"]["num_tests"]["indel"]
    assert [l[2] for l in lines] == fx["only_indels"]["vcf"]


@pytest.mark.parametrize("path", gu.plpindel_fixtures(), ids=lambda p: p.split("/")[-1])
def test_reads_to_full_vcf_device_chain(caller, path):
    """nothing taken from `lofreq alnqual`: lb / ai / ad come from lfq_baq_idaq_batch, both pileups from the device"""
    import lofreq_amd as la
    fx, reads = gu.load_plpindel(path, with_alnqual_tags=False)
    ref = fx["genome"].encode()
    tags = la.baq_batch(caller, reads, ref, extended=True, idaq=True)
    for r, (lb, ai, ad) in zip(reads, tags):
        r["ai"], r["ad"] = ai, ad
    kw, _ = gu.conf_kwargs(fx["call_args"])
    conf = la.VarcallConf(**kw)
    cols, col_pos = la.pileup_indel_columns(caller, reads, ref, 0, len(ref))
    ilines, ntests = _indel_lines(la, caller, cols, col_pos, conf)
    dt = la.pileup_snv_tracks(caller, reads, ref, 0, len(ref), lb=[t[0] for t in tags])
    assert dt.col_pos.tolist() == col_pos.tolist()
    la.skip_snv_columns(caller, cols.cons_indel)    # call_vars: no SNVs where the consensus is an indel (:928-931)
    recs, _, st = caller.call_snvs(dt, conf)
    assert conf.num_snv_tests == fx["all"]["num_tests"]["snv"] and ntests == fx["all"]["num_tests"]["indel"]
    keep = la.filter_records(recs, la.snvqual_thresh(conf.sig, conf.bonf_subst), apply_defaults=False)
    slines = []
    for r, k in zip(recs, keep):
        if k:
            p0 = int(dt.col_pos[int(r["col"])])
            slines.append((p0, 1, gu.strip_hqa(la.format_vcf(np.array([r]), "chr1", pos0=np.array([p0]),
                                                             filter_str="PASS").rstrip("\n"))))
    merged = [l[2] for l in sorted(ilines + slines, key=lambda t: (t[0], t[1]))]     # indels first within a column (:896)
    assert merged == fx["all"]["vcf"]


def _random_indel_reads(rng, n, glen, genome):
    reads = []
    for i in range(n):
        pos = int(rng.integers(0, glen - 400))
        cigar, seq, x = [], [], pos
        if rng.random() < 0.2:
            k = int(rng.integers(1, 6)); cigar.append(("S", k)); seq.extend(rng.integers(0, 4, k).tolist())
        nops = int(rng.integers(1, 6))
        for j in range(nops):
            l = int(rng.integers(1, 40))
            cigar.append((str(rng.choice(["M", "M", "M", "=", "X"])), l))
            seq.extend(int(genome[x + t]) if rng.random() > 0.02 else int(rng.integers(0, 5)) for t in range(l))
            x += l
            if j + 1 < nops:
                u = rng.random()
                if u < 0.35:
                    # (one inserted base in twelve is an ambiguity code, 5..15: its own letter in the insertion's key)
                    k = int(rng.integers(1, 4)); cigar.append(("I", k))
                    seq.extend(int(b) if rng.random() > 0.08 else int(rng.integers(4, 16)) for b in rng.integers(0, 4, k))
                elif u < 0.7:
                    k = int(rng.integers(1, 4)); cigar.append(("D", k)); x += k
                    if rng.random() < 0.15:        # a deletion directly followed by an insertion
                        k = int(rng.integers(1, 3)); cigar.append(("I", k)); seq.extend(rng.integers(0, 4, k).tolist())
                elif u < 0.8:
                    k = int(rng.integers(5, 30)); cigar.append(("N", k)); x += k
                elif u < 0.85:
                    cigar.append(("P", 1)); k = int(rng.integers(1, 3)); cigar.append(("I", k))
                    seq.extend(rng.integers(0, 4, k).tolist())
        if rng.random() < 0.2:
            k = int(rng.integers(1, 6)); cigar.append(("S", k)); seq.extend(rng.integers(0, 4, k).tolist())
        if rng.random() < 0.1:
            cigar.append(("H", 2))
        m = len(seq)
        tag = lambda lo, hi: rng.integers(33 + lo, 33 + hi, m).astype(np.uint8)
        reads.append({"pos0": pos, "cigar": cigar, "seq": np.asarray(seq, np.uint8),
                      "qual": rng.integers(2, 42, m).astype(np.uint8), "mapq": int(rng.choice([60, 60, 30, 0, 255])),
                      "reverse": bool(rng.random() < 0.5),
                      "bi": tag(10, 50) if rng.random() < 0.9 else None, "bd": tag(10, 50) if rng.random() < 0.9 else None,
                      "ai": tag(0, 60) if rng.random() < 0.7 else None, "ad": tag(0, 60) if rng.random() < 0.7 else None,
                      "sq": int(rng.choice([0, 3, 12, 49314])) if rng.random() < 0.5 else None})
    reads.sort(key=lambda r: r["pos0"])
    return reads


@pytest.mark.parametrize("min_idq,begin,end,n_reads", [(0, 0, 3000, 1500), (25, 0, 3000, 1500), (0, 700, 1900, 1500),
                                                       (10, 101, 2503, 7000)])
def test_indel_columns_random_reads_vs_plain_restatement(caller, min_idq, begin, end, n_reads):
    """every CIGAR operation incl. N, P and D followed by I, missing tags, the min_plp_idq gate, a sub-region that starts
    and ends inside a tile of the counter kernel, piles deeper than one round of its reads:
    all fields against the pure-Python restatement of compile_plp_col's indel part (tests/golden_util.py)"""
    import lofreq_amd as la
    rng = np.random.default_rng(5)
    glen = 3000
    genome = rng.integers(0, 4, glen).astype(np.uint8)
    genome[1000:1012] = 2                                   # a homopolymer for hrun
    ref = "".join("ACGT"[c] for c in genome)
    reads = _random_indel_reads(rng, n_reads, glen, genome)
    want = gu.py_indel_pileup(reads, ref, min_plp_idq=min_idq)
    cols, col_pos = la.pileup_indel_columns(caller, reads, ref.encode(), begin, end, min_plp_idq=min_idq)
    assert col_pos.tolist() == sorted(p for p in want if begin <= p < end)
    n_ev = 0
    for c, p in enumerate(col_pos.tolist()):
        w = want[p]
        got = {k: int(getattr(cols, k)[c]) for k in ("coverage_plp", "num_tails", "num_non_indels", "num_ins", "num_dels")}
        assert got == {"coverage_plp": w["cov"], "num_tails": w["tails"], "num_non_indels": w["non_indels"],
                       "num_ins": w["n_ins"], "num_dels": w["n_dels"]}, p
        for sd in range(2):
            S = cols.sides[sd]
            assert (int(S["non_fw"][c]), int(S["non_rv"][c])) == (w["non_fw"][sd], w["non_rv"][sd]), (p, sd)
            e0, e1 = int(S["ev_off"][c]), int(S["ev_off"][c + 1])
            assert [cols.keys[sd][i] for i in range(e0, e1)] == list(w["ev"][sd].keys()), (p, sd)
            a, b = int(S["ne_off"][c]), int(S["ne_off"][c + 1])
            if w["ev"][0] or w["ev"][1]:
                got_ne, want_ne = list(zip(S["ne_q"][a:b].tolist(), S["ne_mq"][a:b].tolist())), list(w["ne"][sd])
                if os.environ.get("LFQ_PILEUP_ATOMIC"):
                    got_ne, want_ne = sorted(got_ne), sorted(want_ne)
                assert got_ne == want_ne, (p, sd)
            else:
                assert a == b
            for i, key in zip(range(e0, e1), w["ev"][sd]):
                r0, r1 = int(S["rd_off"][i]), int(S["rd_off"][i + 1])
                members = w["ev"][sd][key]
                assert S["rd_q"][r0:r1].tolist() == [m[0] for m in members], (p, key)
                assert S["rd_aq"][r0:r1].tolist() == [m[1] for m in members], (p, key)
                assert S["rd_mq"][r0:r1].tolist() == [m[2] for m in members], (p, key)
                assert S["rd_sq"][r0:r1].tolist() == [min(m[3], 32767) for m in members], (p, key)
                assert int(S["ev_rv"][i]) == sum(m[4] for m in members) and int(S["ev_fw"][i]) == len(members) - int(S["ev_rv"][i])
                n_ev += 1
    assert n_ev > 300


def test_indel_columns_empty_inputs(caller):
    import lofreq_amd as la
    cols, col_pos = la.pileup_indel_columns(caller, [], b"ACGT", 0, 4)
    assert cols.ncols == 0 and len(col_pos) == 0
    r = [{"pos0": 1, "cigar": [("M", 2)], "seq": np.array([1, 2], np.uint8), "qual": np.array([30, 30], np.uint8),
          "mapq": 60, "reverse": False}]
    cols, col_pos = la.pileup_indel_columns(caller, r, b"ACGT", 2, 2)          # empty region
    assert cols.ncols == 0
    cols, col_pos = la.pileup_indel_columns(caller, r, b"ACGT", 0, 4)          # no tags at all, no events
    assert col_pos.tolist() == [1, 2] and cols.num_non_indels.tolist() == [1, 1] and cols.num_tails.tolist() == [0, 1]
    assert len(cols.keys[0]) == len(cols.keys[1]) == 0 and not cols.cons_indel.any()


@pytest.mark.parametrize("path", gu.plpindel_fixtures(), ids=lambda p: p.split("/")[-1])
def test_resident_readset_chain(caller, path):
    """the same chain on a resident read set (lfq_readset_*): one upload, BAQ / IDAQ -> both pileups -> calls with the
    per-base arrays staying in HBM; identical VCF, and the fetched tags equal the host-buffer entry point's"""
    import lofreq_amd as la
    fx, reads = gu.load_plpindel(path, with_alnqual_tags=False)
    ref = fx["genome"].encode()
    rs = la.ReadSet(caller, reads, ref)
    rs.baq(extended=True, idaq=True)
    kw, _ = gu.conf_kwargs(fx["call_args"])
    conf = la.VarcallConf(**kw)
    cols, col_pos = rs.pileup_indels(0, len(ref))
    ilines, ntests = _indel_lines(la, caller, cols, col_pos, conf)
    dt = rs.pileup_snv(0, len(ref))
    la.skip_snv_columns(caller, cols.cons_indel)
    recs, _, st = caller.call_snvs(dt, conf)
    assert conf.num_snv_tests == fx["all"]["num_tests"]["snv"] and ntests == fx["all"]["num_tests"]["indel"]
    keep = la.filter_records(recs, la.snvqual_thresh(conf.sig, conf.bonf_subst), apply_defaults=False)
    slines = [(int(dt.col_pos[int(r["col"])]), 1,
               gu.strip_hqa(la.format_vcf(np.array([r]), "chr1", pos0=np.array([int(dt.col_pos[int(r["col"])])]),
                                          filter_str="PASS").rstrip("\n"))) for r, k in zip(recs, keep) if k]
    assert [l[2] for l in sorted(ilines + slines, key=lambda t: (t[0], t[1]))] == fx["all"]["vcf"]
    lb, ai, ad, fl = rs.fetch_tags(idaq=True)
    tags = la.baq_batch(caller, reads, ref, extended=True, idaq=True)
    for i, (tlb, tai, tad) in enumerate(tags):
        a, b = int(rs.seq_off[i]), int(rs.seq_off[i + 1])
        assert lb[a:b].tobytes() == tlb.tobytes()
        assert (tai is None) == (not fl[i] & 1) and (tad is None) == (not fl[i] & 2)
        if tai is not None:
            assert ai[a:b].tobytes() == tai.tobytes()
        if tad is not None:
            assert ad[a:b].tobytes() == tad.tobytes()
    rs.close()


def test_resident_readset_source_quality(caller):
    """sq computed on the read set feeds the sq track of its SNV pileup: same calls as the host-buffer route"""
    import lofreq_amd as la
    fx, reads, nmq, ign = gu.load_srcq(gu.srcq_fixtures()[0])
    ref = fx["genome"].encode()
    rs = la.ReadSet(caller, reads, ref)
    sq = rs.source_qual(def_nm_q=nmq, min_bq=6, ign=ign)
    sq2, sqb = la.source_qual_batch(caller, reads, ref, def_nm_q=nmq, min_bq=6, ign=ign)
    assert sq.tolist() == sq2.tolist()
    kw, _ = gu.conf_kwargs(fx["call_args"] + ["-B"] + fx["args"])
    recs, _, _ = caller.call_snvs(rs.pileup_snv(0, len(ref)), la.VarcallConf(**kw))
    dt = la.pileup_snv_tracks(caller, reads, ref, 0, len(ref), lb=None, sq=sqb)
    recs2, _, _ = caller.call_snvs(dt, la.VarcallConf(**kw))
    assert recs.tobytes() == recs2.tobytes() and len(recs) >= 3


def test_indel_calls_device_packing_equals_host_packing(caller):
    """columns straight from the device pileup are packed into pseudo-columns on the device (lfq_indel_pack_kernel,
    quality arrays still resident); a copy of the same columns goes through the host packing: identical records"""
    import lofreq_amd as la
    fx, reads = gu.load_plpindel(gu.plpindel_fixtures()[0])
    ref = fx["genome"].encode()
    for flag in (la.LFQ_USE_MQ | la.LFQ_USE_IDAQ, la.LFQ_USE_MQ, 0, la.LFQ_USE_MQ | la.LFQ_USE_IDAQ | la.LFQ_USE_SQ):
        cols, col_pos = la.pileup_indel_columns(caller, reads, ref, 0, len(ref))
        assert cols._c_ptr is not None
        conf_d = la.VarcallConf(flag=flag, bonf_dynamic=0, bonf_indel=1, sig=1.0)       # every test is emitted
        dev, nt_d = la.call_indels(caller, cols, conf_d)
        cols._c_ptr = None                                                            # same data, host route
        conf_h = la.VarcallConf(flag=flag, bonf_dynamic=0, bonf_indel=1, sig=1.0)
        host, nt_h = la.call_indels(caller, cols, conf_h)
        assert nt_d == nt_h > 20 and len(dev) == len(host) > 20
        for k in dev.dtype.names:
            assert dev[k].tobytes() == host[k].tobytes(), (flag, k)


def test_indel_columns_device_only_arrays(caller):
    """lfq_set_indel_arrays_on_host(0): the ins_quals / del_quals arrays never leave the device (NULL in the struct, the
    consensus flag from the kernel's quality sums) -- same consensus flags, same indel records"""
    import lofreq_amd as la
    if os.environ.get("LFQ_PILEUP_ATOMIC") or os.environ.get("LFQ_INDEL_HOST_PACK"):
        pytest.skip("needs the column-major kernels (quality sums) and the device-side packing")
    for path in gu.plpindel_fixtures():
        fx, reads = gu.load_plpindel(path)
        ref = fx["genome"].encode()
        kw, _ = gu.conf_kwargs(fx["call_args"])
        cols, col_pos = la.pileup_indel_columns(caller, reads, ref, 0, len(ref))
        want, nt_w = la.call_indels(caller, cols, la.VarcallConf(**kw))
        caller.set_indel_arrays_on_host(False)
        try:
            cols2, col_pos2 = la.pileup_indel_columns(caller, reads, ref, 0, len(ref))
            got, nt_g = la.call_indels(caller, cols2, la.VarcallConf(**kw))
        finally:
            caller.set_indel_arrays_on_host(True)
        assert len(cols2.sides[0]["ne_q"]) == 0 and int(cols2.sides[0]["ne_off"][-1]) > 0
        assert cols2.cons_indel.tolist() == cols.cons_indel.tolist() and col_pos2.tolist() == col_pos.tolist()
        assert nt_g == nt_w and len(got) == len(want) > 0
        for k in got.dtype.names:
            assert got[k].tobytes() == want[k].tobytes(), k


def _chain_digest_script():
    return r'''
import sys, hashlib, json
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import lofreq_amd as la
import test_gpu_plpindel as T
rng = np.random.default_rng(77)
glen = 12000
genome = rng.integers(0, 4, glen).astype(np.uint8)
ref = "".join("ACGT"[c] for c in genome).encode()
reads = T._random_indel_reads(rng, 6000, glen, genome)
for r in reads:
    r["ai"] = r["ad"] = None
caller = la.SnvCaller(0)
h = hashlib.sha256()
rs = la.ReadSet(caller, reads, ref)
rs.baq(extended=True, idaq=True)
lb, ai, ad, fl = rs.fetch_tags(idaq=True)
for a in (lb, ai, ad, fl):
    h.update(np.ascontiguousarray(a).tobytes())
cols, col_pos = rs.pileup_indels(0, glen)
h.update(col_pos.tobytes())
for k in ("coverage_plp", "num_tails", "num_non_indels", "num_ins", "num_dels", "hrun", "ref_base", "cons_indel"):
    h.update(np.ascontiguousarray(getattr(cols, k)).tobytes())
for sd in range(2):
    S = cols.sides[sd]
    for k in ("non_fw", "non_rv", "ne_off", "ne_q", "ne_mq", "ev_off", "ev_fw", "ev_rv", "rd_off", "rd_q", "rd_aq", "rd_mq", "rd_sq"):
        h.update(np.ascontiguousarray(S[k]).tobytes())
    h.update("|".join(cols.keys[sd]).encode())
# indel tests on the columns while they are this context's current ones (pseudo-columns packed on the device; the scan
# of the gates over the columns is one of the threaded host loops)
irecs, n_itests = la.call_indels(caller, cols, la.VarcallConf(flag=la.LFQ_USE_MQ | la.LFQ_USE_IDAQ, bonf_dynamic=0, bonf_indel=1, sig=1.0))
assert n_itests > 500 and len(irecs) > 100
h.update(irecs.tobytes()); h.update(str(n_itests).encode())
dt = rs.pileup_snv(0, glen)
recs, _, st = caller.call_snvs(dt, la.VarcallConf())
h.update(recs.tobytes()); h.update(dt.col_pos.tobytes())
# the same columns when the indel pileup is called while the BAQ kernels are still running (nothing fetched in
# between): its counter and scatter kernels then run beside them on another stream
snap = {(sd, k): np.array(cols.sides[sd][k]) for sd in range(2) for k in ("ne_off", "ne_q", "ne_mq", "ev_off", "rd_off", "rd_q", "rd_aq")}
cov = np.array(cols.coverage_plp)
rs2 = la.ReadSet(caller, reads, ref)
rs2.baq(extended=True, idaq=True)
cols2, col_pos2 = rs2.pileup_indels(0, glen)
assert np.array_equal(col_pos2, col_pos) and np.array_equal(cols2.coverage_plp, cov)
for (sd, k), v in snap.items():
    assert np.array_equal(cols2.sides[sd][k], v), (sd, k)
print(json.dumps({"digest": h.hexdigest(), "ncols": int(cols.ncols), "events": [len(cols.keys[0]), len(cols.keys[1])], "recs": len(recs)}))
'''


def test_host_loops_parallel_equals_serial(tmp_path):
    """the host-side loops of the read-set steps (BAQ geometry + launch order, events from the CIGARs, column assembly,
    event tables, SNV column prefix) split over threads from LFQ_HOST_PAR_MIN reads / positions on (200 000 by default:
    no other test is that large).  The same 6 000-read chain with the threshold at 500 and at 'never': identical tags,
    columns, event tables and calls."""
    import subprocess, sys, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for par_min in ("500", "1000000000"):
        env = dict(os.environ, LFQ_HOST_PAR_MIN=par_min,          # a knob of the tuning build (lfq_knobs(), -DLFQ_TUNE)
                   LFQ_AMD_LIB=os.path.join(root, "lofreq_amd", "liblofreq_amd_tune.so"))
        r = subprocess.run([sys.executable, "-c", _chain_digest_script()], cwd=root, env=env, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1], outs
    assert outs[0]["ncols"] > 5000 and sum(outs[0]["events"]) > 1000
    # lfq_readset_create hands uploads of 8 MB and more to a helper thread and returns (the BAQ geometry runs under
    # them); LFQ_SYNC_UPLOAD=2 does that for this small set too, =1 never: same results
    for mode in ("2", "1"):
        env = dict(os.environ, LFQ_SYNC_UPLOAD=mode)
        r = subprocess.run([sys.executable, "-c", _chain_digest_script()], cwd=root, env=env, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert json.loads(r.stdout.strip().splitlines()[-1]) == outs[0], mode


def test_context_closed_before_its_read_set():
    """a read set hands its device allocations back to its context on destruction: SnvCaller.close() therefore closes
    the context's read sets first, and closing them again afterwards is a no-op"""
    import lofreq_amd as la
    rng = np.random.default_rng(5)
    glen = 2000
    genome = rng.integers(0, 4, glen).astype(np.uint8)
    ref = "".join("ACGT"[c] for c in genome).encode()
    reads = _random_indel_reads(rng, 200, glen, genome)
    own = la.SnvCaller(0)
    rs = la.ReadSet(own, reads, ref)
    rs.baq(extended=True, idaq=True)
    rs2 = la.ReadSet(own, reads, ref)               # takes over nothing yet: rs is still alive
    rs2.close()
    rs3 = la.ReadSet(own, reads, ref)               # ... and this one the allocations rs2 gave back
    rs3.baq(extended=True, idaq=True)
    a = rs.fetch_tags(idaq=True)
    b = rs3.fetch_tags(idaq=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    own.close()
    assert rs.h is None and rs3.h is None
    rs.close()
    rs3.close()
