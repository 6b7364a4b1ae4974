"""-m gpu: raw reads -> BAQ (lfq_baq_batch) -> pileup columns -> SNV calls (lfq_call_snvs_batch) -> VCF text, against
the VCF the reference's own 2.1.4 binary wrote from the same SAM with its on-the-fly BAQ (tests/golden/chain_*.json).
The pileup itself (compile_plp_col, plp.c) is out of scope and done here in numpy for all-M reads."""
import numpy as np
import pytest

import golden_util as gu

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180)]


def _pileup(fx, lb):
    """columns of the all-M reads of a fixture: what compile_plp_col hands to the callback (plp.c:797-1017):
    per read base nt4 code + strand, BQ, BAQ (lb byte - 33), MQ; bases below min_plp_bq = 3 are dropped"""
    import json
    glen = len(fx["genome"])
    per_col = [[] for _ in range(glen)]
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    for (pos0, flag, mapq, cigar, seq, qual), tag in zip(fx["reads"], lb):
        assert cigar == "%dM" % len(seq)
        strand = 1 if flag & 16 else 0
        for j, (b, q) in enumerate(zip(seq, qual)):
            bq = ord(q) - 33
            if bq < 3:
                continue
            per_col[pos0 + j].append((code.get(b, 4) | (strand << 3), bq, int(tag[j]) - 33, mapq))
    cols = [c for c in range(glen) if per_col[c]]
    nt = np.array([o[0] for c in cols for o in per_col[c]], np.uint8)
    bq = np.array([o[1] for c in cols for o in per_col[c]], np.uint8)
    baq = np.array([o[2] for c in cols for o in per_col[c]], np.uint8)
    mq = np.array([o[3] for c in cols for o in per_col[c]], np.uint8)
    off = np.zeros(len(cols) + 1, np.uint64)
    off[1:] = np.cumsum([len(per_col[c]) for c in cols])
    ref = np.frombuffer("".join(fx["genome"][c] for c in cols).encode(), np.uint8).copy()
    return cols, dict(nt=nt, bq=bq, baq=baq, mq=mq, sq=None, col_off=off, ref_base=ref)


@pytest.mark.parametrize("path", gu.chain_fixtures(), ids=lambda p: p.split("/")[-1])
def test_reads_to_vcf_matches_reference_binary(caller, path):
    import json
    import lofreq_amd as la
    import util
    fx = json.load(open(path))
    reads = [{"pos0": r[0], "cigar": gu.parse_cigar(r[3]), "seq": la.encode_seq(r[4]),
              "qual": np.array([ord(c) - 33 for c in r[5]], np.uint8)} for r in fx["reads"]]
    lb = la.baq_batch(caller, reads, fx["genome"].encode(), extended=True)          # lofreq call: extended BAQ
    cols, host = _pileup(fx, lb)
    kw, no_default_filter = gu.conf_kwargs(fx["call_args"])
    conf = la.VarcallConf(**kw)
    recs, _, st = caller.call_snvs(util.to_pileup_batch(la, host), conf)
    assert conf.num_snv_tests == fx["num_snv_tests"]
    thr = la.snvqual_thresh(conf.sig, conf.bonf_subst)
    keep = la.filter_records(recs, thr, apply_defaults=not no_default_filter)
    pos0 = np.array([cols[int(r["col"])] for r in recs], np.int64)
    text = la.format_vcf(recs, "chr1", pos0=pos0, keep=keep, filter_str="PASS")
    got = [gu.strip_hqa(l) for l in text.splitlines()]
    assert got == fx["vcf"]
