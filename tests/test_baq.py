"""CPU tests of the BAQ oracle: the restated profile HMM against the reference's own
kprobaln_ext.c object (compiled unmodified, oracle/_ref), bit for bit."""
import numpy as np
import pytest


def _random_case(rng, lq, indel=0, n_frac=0.02):
    """a read of length lq against a reference window; `indel` = reference minus query length"""
    lr = lq + indel + int(rng.integers(0, 8))
    ref = rng.integers(0, 4, lr).astype(np.uint8)
    start = int(rng.integers(0, max(lr - lq - indel, 0) + 1))
    if indel >= 0:
        src = np.concatenate([ref[start:start + lq // 2], ref[start + lq // 2 + indel:start + lq + indel]])
    else:
        src = np.concatenate([ref[start:start + lq // 2], rng.integers(0, 4, -indel).astype(np.uint8),
                              ref[start + lq // 2:start + lq + indel]])
    query = src[:lq].copy()
    if len(query) < lq:
        query = np.concatenate([query, rng.integers(0, 4, lq - len(query)).astype(np.uint8)])
    mism = rng.random(lq) < 0.03
    query[mism] = (query[mism] + rng.integers(1, 4, int(mism.sum()))) % 4
    query[rng.random(lq) < n_frac] = 4
    ref[rng.random(lr) < n_frac / 2] = 4
    qual = np.clip(np.round(rng.normal(32, 8, lq)), 0, 60).astype(np.uint8)
    return ref, query, qual


@pytest.mark.parametrize("seed", range(6))
def test_hmm_restatement_equals_reference_object(oracle, seed):
    if oracle.ref_parts() is None or not hasattr(oracle.ref_parts(), "kpa_ext_glocal"):
        pytest.skip("oracle/_ref/libref_parts.so (built from /root/reference) not present")
    rng = np.random.default_rng(seed)
    n = 0
    for _ in range(60):
        lq = int(rng.integers(1, 160))
        indel = int(rng.choice([0, 0, 0, 1, 3, -2, 12, -9]))
        if lq + indel < 1:
            indel = 0
        ref, query, qual = _random_case(rng, lq, indel)
        bw = 7 if abs(len(ref) - lq) <= 7 else abs(len(ref) - lq) + 3
        for d, e in ((0.00001, 0.4), (0.001, 0.1), (0.1, 0.4)):      # illumina, samtools' default, pacbio (kprobaln_ext.c:48-51)
            a = oracle.kpa_glocal(ref, query, qual, d, e, bw, use_reference=True)
            b = oracle.kpa_glocal(ref, query, qual, d, e, bw, use_reference=False)
            assert a[0] == b[0]
            assert (a[1] == b[1]).all() and (a[2] == b[2]).all()
            n += len(query)
    assert n > 5000


def test_baq_read_perfect_match_and_mismatch(oracle):
    """a perfectly matching read far from the ends keeps high BAQ; extended BAQ is the min of the running
    maxima from both ends of each match block (bam_md_ext.c:431-451)"""
    rng = np.random.default_rng(1)
    genome = "".join(rng.choice(list("ACGT"), 400))
    pos, l = 100, 80
    seq = np.array(["ACGT".index(c) for c in genome[pos:pos + l]], np.uint8)
    qual = np.full(l, 35, np.uint8)
    out = oracle.baq_read(pos, [("M", l)], seq, qual, genome.encode())
    assert out is not None and len(out) == l
    assert out.max() <= 93 + 33 and out.min() >= 33
    plain = oracle.baq_read(pos, [("M", l)], seq, qual, genome.encode(), extended=False)
    assert (out >= plain).all()           # the extension can only raise a base's BAQ
    # soft-clipped and inserted bases are not aligned by the HMM: their BAQ is their base quality (:391-392)
    seq2 = np.concatenate([np.array([0, 0, 0], np.uint8), seq])
    out2 = oracle.baq_read(pos, [("S", 3), ("M", l)], seq2, np.full(l + 3, 35, np.uint8), genome.encode())
    assert (out2[:3] == 35 + 33).all()


@pytest.mark.parametrize("path", __import__("golden_util").baq_fixtures(), ids=lambda p: p.split("/")[-1])
def test_oracle_reproduces_alnqual_lb_tags(oracle, path):
    """orc_baq_read vs the `lb` tags the reference's 2.1.4 binary (`lofreq alnqual`) wrote, byte for byte"""
    import golden_util as gu
    fx, reads = gu.load_baq(path)
    extended = "-e" not in fx["alnqual_args"]
    genome = fx["genome"].encode()
    nbytes = 0
    for r in reads:
        out = oracle.baq_read(r["pos0"], r["cigar"], r["seq"], r["qual"], genome, extended=extended)
        assert r["lb"] is not None and out is not None
        assert out.tobytes() == r["lb"].tobytes(), (r["pos0"], r["cigar"])
        nbytes += len(out)
    assert nbytes > 10000


@pytest.mark.parametrize("path", __import__("golden_util").baq_fixtures(), ids=lambda p: p.split("/")[-1])
def test_oracle_reproduces_alnqual_ai_ad_tags(oracle, path):
    """the indel alignment qualities (idaq, bam_md_ext.c:73-248) vs the ai / ad tags of the 2.1.4 binary"""
    import json
    import golden_util as gu
    fx, reads = gu.load_baq(path)
    raw = json.load(open(path))["reads"]
    extended = "-e" not in fx["alnqual_args"]
    genome = fx["genome"].encode()
    n_tags = 0
    for r, rr in zip(reads, raw):
        lb, ai, ad = oracle.baq_idaq_read(r["pos0"], r["cigar"], r["seq"], r["qual"], genome, extended=extended)
        assert lb.tobytes() == r["lb"].tobytes()
        assert (ai is None) == (rr["ai"] is None) and (ad is None) == (rr["ad"] is None), (r["pos0"], r["cigar"])
        if ai is not None:
            assert ai.tobytes() == rr["ai"].encode(), (r["pos0"], r["cigar"])
            n_tags += 1
        if ad is not None:
            assert ad.tobytes() == rr["ad"].encode(), (r["pos0"], r["cigar"])
            n_tags += 1
    assert n_tags >= 80
