"""-m gpu parity of the BAQ kernel (lfq_baq_batch): bit-exact against the oracle (itself bit-identical to the
reference's kprobaln_ext.c object) and against the `lb` tags the reference's 2.1.4 binary wrote."""
import numpy as np
import pytest

import golden_util as gu

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180)]


@pytest.mark.parametrize("path", gu.baq_fixtures(), ids=lambda p: p.split("/")[-1])
def test_baq_golden_alnqual_tags(caller, path):
    import lofreq_amd as la
    fx, reads = gu.load_baq(path)
    extended = "-e" not in fx["alnqual_args"]
    out = la.baq_batch(caller, reads, fx["genome"].encode(), extended=extended)
    assert len(out) == len(reads)
    for r, o in zip(reads, out):
        assert o.tobytes() == r["lb"].tobytes(), (r["pos0"], r["cigar"])


def _random_reads(rng, genome, n, rl_lo, rl_hi):
    reads = []
    glen = len(genome)
    for _ in range(n):
        rl = int(rng.integers(rl_lo, rl_hi + 1))
        pos = int(rng.integers(0, glen - rl - 20))
        seq, cigar, gp, run = [], [], pos, 0
        if rng.random() < 0.2:
            k = int(rng.integers(1, 6))
            seq.extend(rng.choice(list("ACGT"), k))
            cigar.append(("S", k))
        while len(seq) < rl and gp < glen - 1:
            seq.append(genome[gp] if rng.random() > 0.02 else str(rng.choice(list("ACGTN"))))
            run += 1
            gp += 1
            u = rng.random()
            if run > 3 and len(seq) < rl - 4 and u < 0.01:
                cigar.append(("M", run)); run = 0
                k = int(rng.integers(1, 14))
                seq.extend(rng.choice(list("ACGT"), k)); cigar.append(("I", k))
            elif run > 3 and len(seq) < rl - 4 and u < 0.02:
                cigar.append(("M", run)); run = 0
                k = int(rng.integers(1, 14))
                cigar.append(("D", k)); gp += k
        if run:
            cigar.append(("M", run))
        if cigar[-1][0] != "M":
            continue
        import lofreq_amd as la
        reads.append({"pos0": pos, "cigar": cigar, "seq": la.encode_seq("".join(seq)),
                      "qual": np.clip(np.round(rng.normal(32, 8, len(seq))), 0, 60).astype(np.uint8)})
    return reads


@pytest.mark.parametrize("extended", [True, False])
def test_baq_random_reads_vs_oracle(caller, oracle, extended):
    """ragged read lengths (more than one wavefront, lanes of a wavefront with different lengths and bands),
    soft clips, insertions and deletions up to 13 bp, N bases, reads at the contig ends"""
    import lofreq_amd as la
    rng = np.random.default_rng(5)
    genome = "".join(rng.choice(list("ACGT"), 3000))
    reads = _random_reads(rng, genome, 400, 20, 160)
    # reads hanging over both contig ends
    reads.append({"pos0": 0, "cigar": [("M", 50)], "seq": la.encode_seq(genome[:50]), "qual": np.full(50, 30, np.uint8)})
    reads.append({"pos0": 2950, "cigar": [("M", 50)], "seq": la.encode_seq(genome[2950:]), "qual": np.full(50, 30, np.uint8)})
    out = la.baq_batch(caller, reads, genome.encode(), extended=extended)
    nb = 0
    for r, o in zip(reads, out):
        exp = oracle.baq_read(r["pos0"], r["cigar"], r["seq"], r["qual"], genome.encode(), extended=extended)
        assert o.tobytes() == exp.tobytes(), (r["pos0"], r["cigar"])
        nb += len(o)
    assert nb > 20000


@pytest.mark.parametrize("rl,n", [(150, 320), (250, 192), (320, 70)])
def test_baq_uniform_length_wavefronts(caller, oracle, rl, n):
    """What the register kernel's interior path sees: whole wavefronts of plain reads of one length (every row between
    the ends takes the branch-free body), with N bases in some reads and a stretch of N in the reference (the body
    variant with the N case, chosen per row for the whole wavefront), mismatches, and -- at 320 bp -- reference windows
    beyond the register kernel's limit, which the host sends to the all-HBM kernel."""
    import lofreq_amd as la
    rng = np.random.default_rng(100 + rl)
    g = rng.choice(list("ACGT"), 4000)
    g[1200:1260] = "N"
    genome = "".join(g)
    reads = []
    for i in range(n):
        pos = int(rng.integers(0, len(genome) - rl - 1))
        seq = list(genome[pos:pos + rl])
        for j in np.nonzero(rng.random(rl) < 0.01)[0]:
            seq[j] = str(rng.choice(list("ACGT")))
        if i % 7 == 0:
            seq[int(rng.integers(0, rl))] = "N"
        reads.append({"pos0": pos, "cigar": [("M", rl)], "seq": la.encode_seq("".join(seq)),
                      "qual": np.clip(np.round(rng.normal(33, 6, rl)), 2, 41).astype(np.uint8)})
    for extended in (True, False):
        out = la.baq_batch(caller, reads, genome.encode(), extended=extended)
        for r, o in zip(reads, out):
            exp = oracle.baq_read(r["pos0"], r["cigar"], r["seq"], r["qual"], genome.encode(), extended=extended)
            assert o.tobytes() == exp.tobytes(), (rl, r["pos0"], extended)


@pytest.mark.parametrize("dlen", [1, 2, 3, 9])
def test_baq_deletion_bands(caller, oracle, dlen):
    """Whole wavefronts of reads with one deletion of `dlen` bases: odd lengths widen the band to 8 (the register kernel's
    second instantiation: its interior path needs every read of a wavefront at band 8), 2 keeps band 7, 9 goes to the
    all-HBM kernel.  lb, ai and ad against the oracle."""
    import lofreq_amd as la
    rng = np.random.default_rng(300 + dlen)
    genome = "".join(rng.choice(list("ACGT"), 3000))
    reads = []
    for _ in range(200):
        rl = 150
        pos = int(rng.integers(10, len(genome) - rl - dlen - 10))
        cut = int(rng.integers(40, 110))
        seq = list(genome[pos:pos + cut] + genome[pos + cut + dlen:pos + dlen + rl])
        for j in np.nonzero(rng.random(rl) < 0.01)[0]:
            seq[j] = str(rng.choice(list("ACGT")))
        reads.append({"pos0": pos, "cigar": [("M", cut), ("D", dlen), ("M", rl - cut)], "seq": la.encode_seq("".join(seq)),
                      "qual": np.clip(np.round(rng.normal(33, 6, rl)), 2, 41).astype(np.uint8)})
    out = la.baq_batch(caller, reads, genome.encode(), extended=True, idaq=True)
    n_ad = 0
    for r, (lb, ai, ad) in zip(reads, out):
        elb, eai, ead = oracle.baq_idaq_read(r["pos0"], r["cigar"], r["seq"], r["qual"], genome.encode(), True)
        assert lb.tobytes() == elb.tobytes(), (dlen, r["pos0"])
        assert (ai is None) == (eai is None) and (ad is None) == (ead is None), (dlen, r["pos0"])
        if ad is not None:
            assert ad.tobytes() == ead.tobytes(), (dlen, r["pos0"])
            n_ad += 1
    assert n_ad > 100


def test_baq_empty(caller):
    import lofreq_amd as la
    assert la.baq_batch(caller, [], b"ACGT") == []


@pytest.mark.parametrize("path", gu.baq_fixtures(), ids=lambda p: p.split("/")[-1])
def test_idaq_golden_ai_ad_tags(caller, path):
    """indel alignment qualities against the ai / ad tags of the reference's 2.1.4 binary"""
    import json
    import lofreq_amd as la
    fx, reads = gu.load_baq(path)
    raw = json.load(open(path))["reads"]
    extended = "-e" not in fx["alnqual_args"]
    out = la.baq_batch(caller, reads, fx["genome"].encode(), extended=extended, idaq=True)
    n_tags = 0
    for r, rr, (lb, ai, ad) in zip(reads, raw, out):
        assert lb.tobytes() == r["lb"].tobytes()
        assert (ai is None) == (rr["ai"] is None) and (ad is None) == (rr["ad"] is None), (r["pos0"], r["cigar"])
        if ai is not None:
            assert ai.tobytes() == rr["ai"].encode(), (r["pos0"], r["cigar"])
            n_tags += 1
        if ad is not None:
            assert ad.tobytes() == rr["ad"].encode(), (r["pos0"], r["cigar"])
            n_tags += 1
    assert n_tags >= 80


def test_idaq_random_reads_vs_oracle(caller, oracle):
    import lofreq_amd as la
    rng = np.random.default_rng(6)
    # homopolymer-rich genome so that the repeat scan of idaq (bam_md_ext.c:132-145, 192-205) has work to do
    genome = "".join(rng.choice(list("ACGT"), 3000))
    g = list(genome)
    for p0 in range(30, 2900, 41):
        g[p0:p0 + int(rng.integers(3, 9))] = g[p0] * 8
    genome = "".join(g[:3000])
    reads = _random_reads(rng, genome, 300, 30, 160)
    out = la.baq_batch(caller, reads, genome.encode(), extended=True, idaq=True)
    n_tags = 0
    for r, (lb, ai, ad) in zip(reads, out):
        elb, eai, ead = oracle.baq_idaq_read(r["pos0"], r["cigar"], r["seq"], r["qual"], genome.encode(), True)
        assert lb.tobytes() == elb.tobytes(), (r["pos0"], r["cigar"])
        assert (ai is None) == (eai is None) and (ad is None) == (ead is None), (r["pos0"], r["cigar"])
        if ai is not None:
            assert ai.tobytes() == eai.tobytes(), (r["pos0"], r["cigar"])
            n_tags += 1
        if ad is not None:
            assert ad.tobytes() == ead.tobytes(), (r["pos0"], r["cigar"])
            n_tags += 1
    assert n_tags > 100


@pytest.mark.parametrize("d,e", [(0.1, 0.4), (0.001, 0.1)])
def test_baq_idaq_other_hmm_parameters(caller, oracle, d, e):
    """lfq_set_baq_hmm_params: the HMM of a -DPACBIO_REALN build (kpa_ext_par_lofreq_pacbio = { 0.1, 0.4 },
    kprobaln_ext.c:51, bam_md_ext.c:268-273) and samtools' own { 0.001, 0.1 } (:48); lb / ai / ad against the oracle, whose
    HMM with these parameters is pinned bitwise against the reference's kprobaln_ext.c object (tests/test_baq.py)."""
    import lofreq_amd as la
    rng = np.random.default_rng(16)
    genome = "".join(rng.choice(list("ACGT"), 3000))
    g = list(genome)
    for p0 in range(30, 2900, 53):
        g[p0:p0 + 6] = g[p0] * 6
    genome = "".join(g[:3000])
    reads = _random_reads(rng, genome, 400, 30, 160)
    default = la.baq_batch(caller, reads, genome.encode(), extended=True, idaq=True)
    caller.set_baq_hmm_params(d, e)
    oracle.set_baq_hmm_params(d, e)
    try:
        out = la.baq_batch(caller, reads, genome.encode(), extended=True, idaq=True)
        n_tags = n_diff = 0
        for r, (lb, ai, ad), (lb0, _, _) in zip(reads, out, default):
            elb, eai, ead = oracle.baq_idaq_read(r["pos0"], r["cigar"], r["seq"], r["qual"], genome.encode(), True)
            assert lb.tobytes() == elb.tobytes(), (r["pos0"], r["cigar"])
            assert (ai is None) == (eai is None) and (ad is None) == (ead is None), (r["pos0"], r["cigar"])
            if ai is not None:
                assert ai.tobytes() == eai.tobytes(), (r["pos0"], r["cigar"])
                n_tags += 1
            if ad is not None:
                assert ad.tobytes() == ead.tobytes(), (r["pos0"], r["cigar"])
                n_tags += 1
            n_diff += lb.tobytes() != lb0.tobytes()
        assert n_tags > 100
        assert n_diff > 50                  # the parameters do reach the kernels
    finally:
        caller.set_baq_hmm_params()
        oracle.set_baq_hmm_params()
    for bad in ((0.0, 0.4), (0.5, 0.4), (1e-5, 1.0), (float("nan"), 0.4)):
        with pytest.raises(Exception):
            caller.set_baq_hmm_params(*bad)
    again = la.baq_batch(caller, reads, genome.encode(), extended=True, idaq=True)
    assert all(a[0].tobytes() == b[0].tobytes() for a, b in zip(again, default))


def test_baq_wavefronts_with_and_without_n(caller, oracle):
    """The plain narrow-band launches run two instantiations of the register kernel, chosen per wavefront by
    lfq_baq_nflag_kernel (an N among the bases or in the reference window of any of its 64 reads).  Uniform reads in input
    order, so that whole wavefronts fall on either side: stretches of the contig with N / lower-case / IUPAC letters, a few
    reads with N bases elsewhere, wavefronts with neither."""
    import lofreq_amd as la
    rng = np.random.default_rng(23)
    glen = 12000
    g = list(rng.choice(list("ACGT"), glen))
    for p0 in (500, 3100, 3101, 7777):
        g[p0] = "N"
    g[5000:5040] = list("acgtn" * 8)                   # lower case counts as its base, n as N
    g[9000] = "R"                                      # any other letter is an N to the HMM
    genome = "".join(g)
    reads = []
    for i in range(64 * 20):
        pos = int(i * (glen - 200) / (64 * 20))
        seq = [genome[pos + k].upper() if genome[pos + k].upper() in "ACGT" else "A" for k in range(100)]
        for k in np.nonzero(rng.random(100) < 0.02)[0]:
            seq[k] = "ACGT"[int(rng.integers(0, 4))]
        if i in (700, 701, 1100):                      # N bases in wavefronts whose reference windows have none
            seq[int(rng.integers(0, 100))] = "N"
        reads.append({"pos0": pos, "cigar": [("M", 100)], "seq": la.encode_seq("".join(seq)),
                      "qual": np.clip(np.round(rng.normal(32, 8, 100)), 2, 60).astype(np.uint8)})
    for extended in (True, False):
        out = la.baq_batch(caller, reads, genome.encode(), extended=extended)
        for r, o in zip(reads, out):
            exp = oracle.baq_read(r["pos0"], r["cigar"], r["seq"], r["qual"], genome.encode(), extended=extended)
            assert o.tobytes() == exp.tobytes(), r["pos0"]


def test_readset_baq_large_set_chunked_upload(caller):
    """More than sixteen rounds of reads (one round = a wavefront of 64 reads on every SIMD) from pinned arrays:
    lfq_readset_create sends bases and qualities in chunks of 1, 2, 4 rounds and thirds of the rest, lfq_readset_baq ramps
    its launches the same way and every launch waits for its chunk on the device.  The same reads in pieces small enough
    for one plain copy and full-size launches must give the same lb / ai / ad bytes and tag flags."""
    import torch
    from bench import make_reads
    from lofreq_amd.pileup import ReadSet
    R = make_reads(1_150_000, 600_000, indel_frac=0.04)
    n, rl = R["n"], R["rl"]
    assert n >= 16 * 65536 and n * rl >= 64 << 20
    keep, P = {}, dict(R)
    for k in ("seq", "qual", "bi", "bd", "pos", "cig_off", "cig", "seq_off", "mapq", "rev"):
        keep[k] = torch.from_numpy(R[k]).pin_memory()
        P[k] = keep[k].numpy()
    rs = ReadSet.from_arrays(caller, P)
    rs.baq(extended=True, idaq=True)
    lb, ai, ad, fl = rs.fetch_tags(idaq=True)
    rs.close()
    assert (lb != 0).mean() > 0.9
    for a in range(0, n, 300_000):
        b = min(n, a + 300_000)
        c0, c1 = int(R["cig_off"][a]), int(R["cig_off"][b])
        Q = {"n": b - a, "ref": R["ref"], "pos": R["pos"][a:b], "cig_off": R["cig_off"][a:b + 1] - c0, "cig": R["cig"][c0:c1],
             "seq_off": R["seq_off"][a:b + 1] - a * rl, "mapq": R["mapq"][a:b], "rev": R["rev"][a:b]}
        for k in ("seq", "qual", "bi", "bd"):
            Q[k] = R[k][a * rl:b * rl]
        ps = ReadSet.from_arrays(caller, Q)
        ps.baq(extended=True, idaq=True)
        plb, pai, pad, pfl = ps.fetch_tags(idaq=True)
        ps.close()
        assert plb.tobytes() == lb[a * rl:b * rl].tobytes(), a
        assert pai.tobytes() == ai[a * rl:b * rl].tobytes(), a
        assert pad.tobytes() == ad[a * rl:b * rl].tobytes(), a
        assert pfl[:b - a].tobytes() == fl[a:b].tobytes(), a
