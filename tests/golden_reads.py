"""Seeded read sets for the REAL-SIZE goldens (tests/golden/big_*.json, written by oracle/make_golden.py --big-only from the
reference's own 2.1.4 binary): test infrastructure, numpy only.

The fixtures of this family do not hold the reads (a C1-shaped run is 70 MB of SAM): they hold the generator's parameters,
its version and the SHA-256 of the SAM text the binary was given.  A test regenerates the reads with `make()`, checks
`sam_sha256()` against the fixture (same numpy, same image: byte-identical) and compares its VCF with the binary's.

Shapes (BASELINE.json): C1 = denv2-like, 10.7 kb, depth 1 000 .. 5 000 varying along the genome, all-M reads, `lofreq call`
defaults (tests/bonf_auto_vs_dyn.sh:10-30); C4-like = 500x with insertions / deletions and BI / BD tags for `--call-indels`.
"""
import hashlib

import numpy as np

GENERATOR_VERSION = 1

_OPS = "MIDNSHP=X"


def make(seed, glen, depth_lo, depth_hi, rl=150, min_q=6, snv_every=20, indel_every=0, mapq_mix=True, low_q_frac=0.0):
    """Position-sorted reads over a random genome of `glen` bases.

    depth: the read starts follow a density that swings between depth_lo and depth_hi with a period of glen / 3 bases;
    SNV sites every `snv_every` bases, allele frequency cycling 0.5, 1, 2, 5, 10, 25, 50 %; with indel_every > 0 an
    insertion / deletion site (1..3 bases, alternating) every `indel_every` bases at 3 / 10 / 30 % of the reads that cover it
    with 12 bases to spare on both sides, and BI / BD tags (30..49) on every read; base qualities ~ N(34, 6) clipped to
    [min_q, 41], a fraction low_q_frac of them replaced by U{2..5} (below `lofreq call`'s min_bq 6); sequencing errors by the
    quality of the base; mapping quality 60 (92 %), else 0..59 or 255; random strand.
    -> dict of flat arrays in the layout of lfq_pileup_reads / oracle/pyoracle.py::pack_reads."""
    rng = np.random.default_rng([GENERATOR_VERSION, seed])
    genome = rng.integers(0, 4, glen).astype(np.uint8)
    x = np.arange(glen - rl - 8)
    dens = depth_lo + (depth_hi - depth_lo) * (0.5 - 0.5 * np.cos(2 * np.pi * x / (glen / 3.0)))
    n = int(dens.sum() / rl)
    cdf = np.cumsum(dens)
    pos = np.sort(np.searchsorted(cdf, rng.random(n) * cdf[-1])).astype(np.int64)
    pos = np.minimum(pos, glen - rl - 9)
    snv_sites = np.arange(snv_every // 2, glen, snv_every)
    snv_af = np.array([0.005, 0.01, 0.02, 0.05, 0.1, 0.25, 0.5])[np.arange(len(snv_sites)) % 7]
    snv_alt = (genome[snv_sites] + 1 + (np.arange(len(snv_sites)) % 3)) % 4
    seqs = np.empty((n, rl), np.uint8)
    ref_idx = pos[:, None] + np.arange(rl)[None, :]
    rpos = ref_idx.copy()
    cig = np.zeros((n, 3), np.uint32)
    ncig = np.ones(n, np.int64)
    cig[:, 0] = rl << 4
    carry = np.full(n, -1, np.int64)
    if indel_every:
        ind_sites = np.arange(indel_every // 3, glen - rl - 8, indel_every)
        k = np.arange(len(ind_sites))
        ind_af, ind_len, ind_ins = np.array([0.03, 0.1, 0.3])[k % 3], 1 + (k % 3), (k % 2) == 0
        ins_seq = rng.integers(0, 4, (len(ind_sites), 3)).astype(np.uint8)
        si = np.searchsorted(ind_sites, pos + 12)
        for d in range(2):
            j = si + d
            ok = j < len(ind_sites)
            jj = np.where(ok, j, 0)
            inside = ok & (ind_sites[jj] >= pos + 12) & (ind_sites[jj] <= pos + rl - 16) & (carry < 0)
            take = inside & (rng.random(n) < ind_af[jj])
            carry[take] = jj[take]
    else:
        ind_sites = np.zeros(0, np.int64)
    plain = carry < 0
    seqs[plain] = genome[ref_idx[plain]]
    for i in np.nonzero(~plain)[0]:
        s = int(carry[i])
        c = int(ind_sites[s] - pos[i]) + 1
        L = int(ind_len[s])
        if ind_ins[s]:
            seqs[i, :c] = genome[pos[i]:pos[i] + c]
            seqs[i, c:c + L] = ins_seq[s, :L]
            seqs[i, c + L:] = genome[pos[i] + c:pos[i] + rl - L]
            rpos[i, c:c + L] = -1
            rpos[i, c + L:] = np.arange(pos[i] + c, pos[i] + rl - L)
            cig[i] = [(c << 4), (L << 4) | 1, ((rl - c - L) << 4)]
        else:
            seqs[i, :c] = genome[pos[i]:pos[i] + c]
            seqs[i, c:] = genome[pos[i] + c + L:pos[i] + rl + L]
            rpos[i, c:] = np.arange(pos[i] + c + L, pos[i] + rl + L)
            cig[i] = [(c << 4), (L << 4) | 2, ((rl - c) << 4)]
        ncig[i] = 3
    on = np.isin(rpos, snv_sites)
    ri, ci = np.nonzero(on)
    sidx = np.searchsorted(snv_sites, rpos[ri, ci])
    flip = rng.random(len(ri)) < snv_af[sidx]
    seqs[ri[flip], ci[flip]] = snv_alt[sidx[flip]]
    qual = np.clip(np.round(rng.normal(34, 6, seqs.shape)), min_q, 41).astype(np.uint8)
    if low_q_frac > 0:
        low = rng.random(seqs.shape) < low_q_frac
        qual[low] = rng.integers(2, 6, int(low.sum())).astype(np.uint8)
    err = rng.random(seqs.shape) < np.power(10.0, -(qual.astype(np.float64)) / 10.0)
    seqs[err] = (seqs[err] + 1 + rng.integers(0, 3, int(err.sum()))) % 4
    cig_off = np.zeros(n + 1, np.int64)
    cig_off[1:] = np.cumsum(ncig)
    if mapq_mix:
        u = rng.random(n)
        mapq = np.where(u < 0.92, 60, np.where(u < 0.995, rng.integers(0, 60, n), 255)).astype(np.uint8)
    else:
        mapq = np.full(n, 60, np.uint8)
    R = {
        "n": n, "rl": rl, "glen": glen, "ref": np.frombuffer(b"ACGT", np.uint8)[genome].tobytes(),
        "pos": pos.astype(np.int32), "cig_off": cig_off, "cig": np.ascontiguousarray(cig[np.arange(3)[None, :] < ncig[:, None]]),
        "seq_off": np.arange(n + 1, dtype=np.int64) * rl, "seq": np.ascontiguousarray(seqs.reshape(-1)),
        "qual": np.ascontiguousarray(qual.reshape(-1)),
        "bi": None, "bd": None, "ai": None, "ad": None, "lb": None, "sq": None, "flags": np.zeros(max(n, 1), np.uint8),
        "mapq": mapq, "rev": (rng.random(n) < 0.5).astype(np.uint8),
        "n_indel_reads": int((~plain).sum()), "snv_sites": snv_sites, "indel_sites": ind_sites,
    }
    if indel_every:
        R["bi"] = rng.integers(33 + 30, 33 + 50, n * rl).astype(np.uint8)
        R["bd"] = rng.integers(33 + 30, 33 + 50, n * rl).astype(np.uint8)
        R["flags"] = np.full(max(n, 1), 3, np.uint8)
    return R


def make_from_fixture(fx):
    g = fx["generator"]
    assert g["version"] == GENERATOR_VERSION, "fixture written by another version of tests/golden_reads.py"
    return make(**g["params"])


def sam_lines(R, chrom="chr1"):
    """the SAM text of a read set, one bytes object per line (header first), as oracle/make_golden.py hands it to the binary"""
    yield b"@HD\tVN:1.0\tSO:coordinate\n"
    yield ("@SQ\tSN:%s\tLN:%d\n" % (chrom, R["glen"])).encode()
    letters = np.frombuffer(b"ACGTN", np.uint8)
    seq = letters[R["seq"]]
    qual = (R["qual"] + 33).astype(np.uint8)
    so, co = R["seq_off"], R["cig_off"]
    ch = chrom.encode()
    for i in range(R["n"]):
        a, b = int(so[i]), int(so[i + 1])
        cg = "".join("%d%s" % (int(w) >> 4, _OPS[int(w) & 15]) for w in R["cig"][co[i]:co[i + 1]]).encode()
        f = [b"r%d" % i, b"16" if R["rev"][i] else b"0", ch, b"%d" % (int(R["pos"][i]) + 1), b"%d" % int(R["mapq"][i]), cg,
             b"*", b"0", b"0", seq[a:b].tobytes(), qual[a:b].tobytes()]
        if R.get("bi") is not None:
            f.append(b"BI:Z:" + R["bi"][a:b].tobytes())
            f.append(b"BD:Z:" + R["bd"][a:b].tobytes())
        yield b"\t".join(f) + b"\n"


def write_sam(R, path, chrom="chr1"):
    h = hashlib.sha256()
    with open(path, "wb") as f:
        for ln in sam_lines(R, chrom):
            f.write(ln)
            h.update(ln)
    return h.hexdigest()


def sam_sha256(R, chrom="chr1"):
    h = hashlib.sha256()
    for ln in sam_lines(R, chrom):
        h.update(ln)
    return h.hexdigest()
