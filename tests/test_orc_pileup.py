"""CPU: the oracle's column builder (oracle/orc_pileup.c, restating compile_plp_col, plp.c:797-1288, over htslib's
pileup entries) pinned on outputs of the reference itself: `lofreq plpsummary` column dumps of its 2.1.4 binary
(element by element, in pileup order) and -- as the last stage of the whole oracle chain, BAQ / IDAQ included -- the
VCFs `lofreq call --call-indels` wrote from the same reads."""
import json

import numpy as np
import pytest

import golden_util as gu
import oracle_chain as oc


@pytest.mark.parametrize("path", gu.pileup_fixtures(), ids=lambda p: p.split("/")[-1])
def test_snv_tracks_match_plpsummary(oracle, path):
    fx = json.load(open(path))
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    reads = [{"pos0": r[0], "cigar": gu.parse_cigar(r[3]), "seq": np.array([code.get(c, 4) for c in r[4].upper()], np.uint8),
              "qual": np.array([ord(c) - 33 for c in r[5]], np.uint8), "mapq": r[2], "reverse": bool(r[1] & 16),
              "lb": np.frombuffer(r[6].encode(), np.uint8)} for r in fx["reads"]]
    P = oracle.pack_reads(reads, fx["genome"].encode())
    out = oracle.pileup_region(P, 0, len(fx["genome"]), min_plp_bq=3, use_baq=True)
    h, col_pos = out["host"], out["col_pos"]
    exp = {c["pos0"]: c for c in fx["columns"]}
    checked = 0
    for ci, p0 in enumerate(col_pos.tolist()):
        a, b = int(h["col_off"][ci]), int(h["col_off"][ci + 1])
        e = exp.get(p0)
        if e is None:
            assert a == b
            continue
        assert chr(h["ref_base"][ci]) == e["ref"]
        nt = h["nt"][a:b]
        for c, letter in enumerate("ACGTN"):
            sel = (nt & 7) == c
            o = e["obs"].get(letter)
            got = list(zip(h["bq"][a:b][sel].tolist(), h["baq"][a:b][sel].tolist(), h["mq"][a:b][sel].tolist()))
            want = [] if not o else list(zip(gu.dec(o["bq"]).tolist(), [(255 if v < 0 else v) for v in gu.dec(o["baq"]).tolist()],
                                             o["mq"]))
            assert got == want, (p0, letter)                  # pileup order = the order of the reference's arrays
            fw = int((sel & ((nt & 8) == 0)).sum())
            assert [fw, int(sel.sum()) - fw] == e["fwrv"][letter], (p0, letter)
            checked += len(got)
    assert checked > 20000


@pytest.mark.parametrize("path", gu.plpindel_fixtures(), ids=lambda p: p.split("/")[-1])
def test_indel_fields_match_plpsummary(oracle, path):
    fx, reads = gu.load_plpindel(path)
    ref = fx["genome"].encode()
    out = oracle.pileup_region(oracle.pack_reads(reads, ref), 0, len(ref), use_baq=True)
    f, col_pos = out["flat"], out["col_pos"]
    col_of = {int(p): i for i, p in enumerate(col_pos)}
    keys = [[f["key_chars"][s][f["key_off"][s][i]:f["key_off"][s][i + 1]].decode() for i in range(len(f["key_off"][s]) - 1)]
            for s in (0, 1)]
    n_ev = 0
    for e in fx["columns"]:
        c = col_of[e["pos0"]]
        assert chr(f["ref_base"][c]) == e["ref"]
        assert bool(out["cons_indel"][c]) == (e["cons"][0] in "+-"), (e["pos0"], e["cons"])          # plp.c:1236-1270
        for k in ("coverage_plp", "num_tails", "num_non_indels", "num_ins", "num_dels", "hrun"):
            assert int(f[k][c]) == e[k], (e["pos0"], k)
        for sd, sn in enumerate(("ins", "dels")):
            E = e[sn]
            assert (int(f["non_fw"][sd][c]), int(f["non_rv"][sd][c])) == (E["non_fw"], E["non_rv"])
            a, b = int(f["ne_off"][sd][c]), int(f["ne_off"][sd][c + 1])
            assert list(zip(f["ne_q"][sd][a:b].tolist(), f["ne_mq"][sd][a:b].tolist())) == \
                list(zip(gu.dec(E["ne_q"]).tolist(), E["ne_mq"])), (e["pos0"], sn)
            e0, e1 = int(f["ev_off"][sd][c]), int(f["ev_off"][sd][c + 1])
            assert keys[sd][e0:e1] == [ev["key"] for ev in E["events"]]
            for i, ev in zip(range(e0, e1), E["events"]):
                assert (int(f["ev_fw"][sd][i]), int(f["ev_rv"][sd][i])) == (ev["fw"], ev["rv"])
                r0, r1 = int(f["rd_off"][sd][i]), int(f["rd_off"][sd][i + 1])
                for name, want in (("rd_q", gu.dec(ev["q"]).tolist()), ("rd_aq", gu.dec(ev["aq"]).tolist()),
                                   ("rd_mq", ev["mq"]), ("rd_sq", gu.dec(ev["sq"]).tolist())):
                    assert f[name][sd][r0:r1].tolist() == want, (e["pos0"], ev["key"], name)
                n_ev += 1
    assert n_ev >= 20


@pytest.mark.parametrize("path", gu.plpindel_fixtures(), ids=lambda p: p.split("/")[-1])
def test_whole_oracle_chain_reproduces_the_binary_vcf(oracle, path):
    """nothing taken from `lofreq alnqual`: reads -> oracle BAQ / IDAQ -> oracle pileup -> oracle calls = the VCF of
    `lofreq call --call-indels` (and of --only-indels), test counts included"""
    fx, reads = gu.load_plpindel(path, with_alnqual_tags=False)
    ref = fx["genome"].encode()
    oc.add_alnqual_tags(oracle, reads, ref, extended=True, idaq=True)
    kw, ndf = gu.conf_kwargs(fx["call_args"])
    out = oc.call_region(oracle, reads, ref, 0, len(ref), kw, call_indels=True, raw_counts_after_minbq=1,
                         no_default_filter=ndf)
    assert out["n_snv_tests"] == fx["all"]["num_tests"]["snv"] and out["n_indel_tests"] == fx["all"]["num_tests"]["indel"]
    assert out["lines"] == fx["all"]["vcf"]
    only = oc.call_region(oracle, reads, ref, 0, len(ref), kw, call_indels=True, only_indels=True, no_default_filter=ndf)
    assert only["n_indel_tests"] == fx["only_indels"]["num_tests"]["indel"] and only["lines"] == fx["only_indels"]["vcf"]


@pytest.mark.parametrize("path", gu.chain_fixtures(), ids=lambda p: p.split("/")[-1])
def test_whole_oracle_chain_snv_fixtures(oracle, path):
    """reads -> oracle BAQ -> oracle pileup -> oracle SNV calls = the VCF `lofreq call` wrote from the same SAM"""
    fx = json.load(open(path))
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    reads = [{"pos0": r[0], "cigar": gu.parse_cigar(r[3]), "seq": np.array([code.get(c, 4) for c in r[4].upper()], np.uint8),
              "qual": np.array([ord(c) - 33 for c in r[5]], np.uint8), "mapq": r[2], "reverse": bool(r[1] & 16)}
             for r in fx["reads"]]
    ref = fx["genome"].encode()
    kw, ndf = gu.conf_kwargs(fx["call_args"])
    if kw["flag"] & 1:
        oc.add_alnqual_tags(oracle, reads, ref, extended=True, idaq=False)
    out = oc.call_region(oracle, reads, ref, 0, len(ref), kw, call_indels=False, raw_counts_after_minbq=1,
                         no_default_filter=ndf)
    assert out["n_snv_tests"] == fx["num_snv_tests"]
    assert out["lines"] == fx["vcf"]
