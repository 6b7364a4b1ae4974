"""-m gpu: source quality on the device (lfq_source_qual_batch) against the oracle's restatement
of source_qual (plp.c:427-593), against the SQ track of the reference binary (`lofreq plpsummary -s`), and the chain
reads -> source quality -> pileup with the sq track -> SNV calls against `lofreq call -s`."""
import json

import numpy as np
import pytest

import golden_util as gu
from test_source_qual import expected_sq_per_read

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


@pytest.mark.parametrize("path", gu.srcq_fixtures(), ids=lambda p: p.split("/")[-1])
def test_source_qual_matches_reference_binary(caller, path):
    import lofreq_amd as la
    fx, reads, nmq, ign = gu.load_srcq(path)
    want = expected_sq_per_read(fx, reads)
    sq, sqb = la.source_qual_batch(caller, reads, fx["genome"].encode(), def_nm_q=nmq, min_bq=6, ign=ign)
    for ri, v in want.items():
        assert max(int(sq[ri]), 0) == v, (ri, int(sq[ri]), v)
        assert int(sqb[ri]) == min(v, 254)


@pytest.mark.parametrize("path", gu.srcq_fixtures(), ids=lambda p: p.split("/")[-1])
def test_source_qual_chain_matches_lofreq_call_s(caller, path):
    """reads -> sq -> device pileup (sq track) -> calls == the VCF of `lofreq call -s` (2.1.4; modulo ;HQA=)"""
    import lofreq_amd as la
    fx, reads, nmq, ign = gu.load_srcq(path)
    ref = fx["genome"].encode()
    _, sqb = la.source_qual_batch(caller, reads, ref, def_nm_q=nmq, min_bq=6, ign=ign)
    dt = la.pileup_snv_tracks(caller, reads, ref, 0, len(ref), lb=None, min_plp_bq=3, sq=sqb)
    kw, no_default = gu.conf_kwargs(fx["call_args"] + ["-B"] + fx["args"])
    conf = la.VarcallConf(**kw)
    recs, _, st = caller.call_snvs(dt, conf)
    assert conf.num_snv_tests == fx["num_snv_tests"]
    pos0 = np.array([dt.col_pos[int(r["col"])] for r in recs], np.int64)
    text = la.format_vcf(recs, "chr1", pos0=pos0)

    def no_af(line):            # 2.1.4 counts raw alts after the min_bq filter, HEAD before: AF only
        f = line.split("\t")
        f[7] = ";".join(x for x in f[7].split(";") if not x.startswith("AF="))
        return "\t".join(f)
    assert [no_af(gu.strip_hqa(l)) for l in text.splitlines()] == [no_af(l) for l in fx["vcf"]]
    assert len(fx["vcf"]) >= 3


def test_source_qual_random_reads_vs_oracle(caller, oracle):
    """seeded reads with every CIGAR operation, qualities 0..60, N bases, long reads whose K exceeds the LDS
    cells (scratch path), def_nm_q on and off, an ignore mask: bit-equal to the oracle"""
    import lofreq_amd as la
    rng = np.random.default_rng(7)
    glen = 6000
    genome = rng.integers(0, 4, glen).astype(np.uint8)
    # ambiguity letters in the reference and in the reads: count_cigar_ops compares LETTERS (samutils.c:486-489), so a read's
    # R on a reference R is a match and on a reference N a mismatch (base codes 5..15, include/lofreq_amd.h)
    amb = rng.random(glen) < 0.02
    genome[amb] = rng.integers(4, 16, int(amb.sum()))
    genome[genome == 5] = 4                                  # ('=' is not a reference letter)
    ref = bytes(gu.SEQ_LETTERS[c].encode()[0] for c in genome)
    reads = []
    for i in range(600):
        long_read = i % 50 == 0
        rl = int(rng.integers(2500, 4000)) if long_read else int(rng.integers(1, 260))
        pos = int(rng.integers(0, glen - rl - 50)) if rl < glen - 60 else 0
        rate = float(rng.choice([0.0, 0.0, 0.005, 0.02, 0.1, 0.45])) if not long_read else 0.5
        cigar, seq, x = [], [], pos
        left = rl
        if rng.random() < 0.2:
            k = int(rng.integers(1, 8)); cigar.append(("S", k)); seq.extend(rng.integers(0, 4, k).tolist())
        if rng.random() < 0.05:
            cigar.insert(0, ("H", 3))
        while left > 0:
            l = int(min(left, rng.integers(1, 80)))
            op = "X" if rng.random() < 0.05 else "M"
            for j in range(l):
                b = int(genome[x + j]) if x + j < glen else 0
                if rng.random() < rate:
                    b = int(rng.integers(0, 5)) if rng.random() < 0.8 else int(rng.integers(5, 16))
                seq.append(b)
            cigar.append((op, l)); x += l; left -= l
            u = rng.random()
            if left > 0 and u < 0.15:
                k = int(rng.integers(1, 5)); cigar.append(("I", k)); seq.extend(rng.integers(0, 4, k).tolist())
            elif left > 0 and u < 0.3:
                k = int(rng.integers(1, 5)); cigar.append(("D", k)); x += k
            elif left > 0 and u < 0.33:
                k = int(rng.integers(1, 30)); cigar.append(("N", k)); x += k
        if rng.random() < 0.2:
            k = int(rng.integers(1, 8)); cigar.append(("S", k)); seq.extend(rng.integers(0, 4, k).tolist())
        qual = rng.integers(0, 61, len(seq)).astype(np.uint8) if i % 3 else np.full(len(seq), 40, np.uint8)
        reads.append({"pos0": pos, "cigar": cigar, "seq": np.asarray(seq, np.uint8), "qual": qual})
    ign = (rng.random(glen) < 0.05).astype(np.uint8)
    n_values = 0
    for nmq, mask, min_bq in ((-1, None, 6), (-1, ign, 6), (25, None, 6), (-1, None, 0), (0, ign, 20)):
        sq, sqb = la.source_qual_batch(caller, reads, ref, def_nm_q=nmq, min_bq=min_bq, ign=mask)
        for i, r in enumerate(reads):
            exp = oracle.source_qual(r["pos0"], r["cigar"], r["seq"], r["qual"], ref, nonmatch_qual=nmq, min_bq=min_bq,
                                     ign=mask)
            assert int(sq[i]) == exp, (nmq, min_bq, i, int(sq[i]), exp)
            n_values += 0 < exp < 49314
    assert n_values > 100


def test_source_qual_empty_and_edge_reads(caller, oracle):
    import lofreq_amd as la
    ref = b"ACGTACGTACGTACGTACGT"
    sq, sqb = la.source_qual_batch(caller, [], ref)
    assert len(sq) == 0
    seq = np.array([0, 1, 2, 3] * 3, np.uint8)
    reads = [{"pos0": 0, "cigar": [("M", 12)], "seq": seq, "qual": np.full(12, 30, np.uint8)},
             {"pos0": 0, "cigar": [("M", 12)], "seq": seq, "qual": np.full(12, 5, np.uint8)},
             {"pos0": 0, "cigar": [("M", 5), ("X", 2), ("M", 5)], "seq": seq, "qual": np.full(12, 30, np.uint8)},
             {"pos0": 14, "cigar": [("M", 12)], "seq": seq, "qual": np.full(12, 30, np.uint8)},      # runs off the contig
             {"pos0": 0, "cigar": [("=", 12)], "seq": seq, "qual": np.full(12, 30, np.uint8)},       # nothing counted
             {"pos0": 0, "cigar": [("M", 12)], "seq": np.full(12, 3, np.uint8), "qual": np.zeros(12, np.uint8)}]
    sq, sqb = la.source_qual_batch(caller, reads, ref, min_bq=0)
    exp = [oracle.source_qual(r["pos0"], r["cigar"], r["seq"], r["qual"], ref, min_bq=0) for r in reads]
    assert sq.tolist() == exp
    assert sq[0] == 49314 and sqb[0] == 254 and sq[4] == -1 and sqb[4] == 0
