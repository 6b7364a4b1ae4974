"""Shared helpers for the parity tests: seeded random pileup batches and oracle/HIP comparison."""
import ctypes as C

import numpy as np

LDBL_MAX = np.finfo(np.longdouble).max
LDBL_MIN = np.finfo(np.longdouble).tiny

# tolerance stated by BASELINE.json north_star: p-values within 1e-10 relative (= 1e-10 absolute on
# the natural log of the p-value)
PV_LOG_TOL = 1e-10


def random_batch(rng, ncols, depth_lo, depth_hi, alt_rate=0.002, with_sq=False, with_baq=True,
                 low_bq_frac=0.02, n_frac=0.01, planted=None, ref_n_frac=0.0):
    """Packed host tracks with ragged depths, N bases, low BQ, MQ 0/255, missing BAQ."""
    depths = rng.integers(depth_lo, depth_hi + 1, size=ncols)
    off = np.zeros(ncols + 1, np.uint64)
    off[1:] = np.cumsum(depths)
    n = int(off[-1])
    ref_code = rng.integers(0, 4, size=ncols)
    ref_base = np.frombuffer(b"ACGT", np.uint8)[ref_code].copy()
    if ref_n_frac > 0:
        ref_base[rng.random(ncols) < ref_n_frac] = ord("N")
    col_of = np.repeat(np.arange(ncols), depths)
    code = ref_code[col_of].copy()
    rate = np.full(n, alt_rate)
    if planted:
        for c, af in planted.items():
            rate[off[c]:off[c + 1]] = af
    is_alt = rng.random(n) < rate
    shift = rng.integers(1, 4, size=n)
    code[is_alt] = (code[is_alt] + shift[is_alt]) % 4
    code[rng.random(n) < n_frac] = 4
    strand = rng.integers(0, 2, size=n)
    nt = (code | (strand << 3)).astype(np.uint8)
    bq = np.clip(np.round(rng.normal(33, 6, n)), 0, 93).astype(np.uint8)
    low = rng.random(n) < low_bq_frac
    bq[low] = rng.integers(0, 8, size=int(low.sum()))
    mq = np.where(rng.random(n) < 0.9, 60, rng.integers(0, 61, size=n)).astype(np.uint8)
    mq[rng.random(n) < 0.01] = 255
    mq[rng.random(n) < 0.01] = 0
    baq = None
    if with_baq:
        baq = np.where(rng.random(n) < 0.85, 93, rng.integers(0, 94, size=n)).astype(np.uint8)
        baq[rng.random(n) < 0.02] = 255
    sq = None
    if with_sq:
        sq = rng.integers(5, 60, size=n).astype(np.uint8)
        sq[rng.random(n) < 0.05] = 255
    return dict(nt=nt, bq=bq, baq=baq, mq=mq, sq=sq, col_off=off, ref_base=ref_base)


def uniform_p_column(n_obs, counts, ref=b"A", bq=30):
    """A column whose every merged error probability is exactly 10^(-bq/10) (BAQ missing, MQ NA),
    with the given (c0, c1, c2) alt counts -- the construction behind SURVEY App. A.6."""
    code = np.zeros(n_obs, np.int64)
    pos = 0
    for a, c in enumerate(counts):
        code[pos:pos + c] = a + 1
        pos += c
    nt = code.astype(np.uint8)
    nt[1::2] |= 8
    return dict(nt=nt, bq=np.full(n_obs, bq, np.uint8), baq=np.full(n_obs, 255, np.uint8),
                mq=np.full(n_obs, 255, np.uint8), sq=None,
                col_off=np.array([0, n_obs], np.uint64), ref_base=np.frombuffer(ref, np.uint8).copy())


def concat_batches(batches):
    out = {}
    for k in ("nt", "bq", "baq", "mq", "sq"):
        vals = [b[k] for b in batches]
        out[k] = None if any(v is None for v in vals) else np.concatenate(vals)
    offs = [np.zeros(1, np.uint64)]
    base = 0
    for b in batches:
        offs.append(b["col_off"][1:] + np.uint64(base))
        base += int(b["col_off"][-1])
    out["col_off"] = np.concatenate(offs)
    out["ref_base"] = np.concatenate([b["ref_base"] for b in batches])
    return out


def to_pileup_batch(la, host):
    return la.PileupBatch(host["nt"], host["bq"], host["mq"], host["col_off"], host["ref_base"],
                          baq=host["baq"], sq=host["sq"], coverage_plp=host.get("coverage_plp"),
                          num_bases=host.get("num_bases"))


def oracle_conf(orc, **kw):
    return orc.default_conf(**kw)


def run_oracle(orc, host, **conf_kw):
    oc = orc.default_conf(**conf_kw)
    res, _ = orc.call_batch(host["nt"], host["bq"], host["baq"], host["mq"], host["sq"], host["col_off"],
                            host["ref_base"], oc, coverage_plp=host.get("coverage_plp"),
                            num_bases=host.get("num_bases"))
    return res, oc


def run_layer1(la, caller, host, conf):
    """Upload a host batch with torch, run the kernels only, return (counts, pvals sorted by col, stats)."""
    import torch
    dev = torch.device("cuda", caller.device)

    def up(a, dtype=torch.uint8):
        if a is None:
            return None
        a = np.ascontiguousarray(a)
        t = torch.zeros((a.size + 15) // 16 * 16 + 16, dtype=dtype, device=dev)
        if a.size:
            t[: a.size] = torch.from_numpy(a.view(np.uint8) if dtype == torch.uint8 else a).to(dev)
        return t

    ncols = len(host["col_off"]) - 1
    off = torch.from_numpy(host["col_off"].astype(np.int64)).to(dev)
    b = la.PileupBatch(up(host["nt"]), up(host["bq"]), up(host["mq"]), off, up(host["ref_base"]),
                       baq=up(host["baq"]), sq=up(host["sq"]), on_device=True)
    b.ncols = ncols
    d_counts = torch.zeros(max(ncols, 1) * 64, dtype=torch.uint8, device=dev)
    d_pvals = torch.zeros(max(ncols, 1) * 128, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    caller.snv_batch_device(b, conf, d_counts, d_pvals, max(ncols, 1))
    st = caller.batch_finish()
    counts = d_counts.cpu().numpy().view(la.COL_COUNTS_DTYPE)[:ncols]
    pvals = d_pvals.cpu().numpy().view(la.COL_PVALS_DTYPE)[: st.n_pvals]
    pvals = pvals[np.argsort(pvals["col"], kind="stable")]
    return counts, pvals, st


def assert_counts_equal(counts, ores, host):
    """Integer outputs of plp_to_errprobs + the Bonferroni 'tested' flag + strand counts: bit-exact."""
    for f in ("n_err_probs", "alt_counts", "alt_raw_counts"):
        assert np.array_equal(counts[f], ores[f]), f
    assert np.array_equal(counts["tested"].astype(np.int32), ores["tested"]), "tested"
    ng = np.nonzero(counts["gated"] == 0)[0]
    for c in ng:
        ref = b"ACGT".find(bytes([int(host["ref_base"][c])]))
        alts = [x for x in range(4) if x != ref]
        assert counts["ref_fw"][c] == ores["fw"][c, ref] and counts["ref_rv"][c] == ores["rv"][c, ref], c
        for a, x in enumerate(alts):
            assert counts["alt_fw"][c, a] == ores["fw"][c, x], (c, a)
            assert counts["alt_raw_counts"][c, a] - counts["alt_fw"][c, a] == ores["rv"][c, x], (c, a)


def log_of(pv):
    """natural log of an np.longdouble p-value as float (80-bit log, then narrowed)."""
    return float(np.log(np.longdouble(pv)))


# Tolerance per record (north_star: 1e-10 relative) -- a BOUND, not a constant.  Every p-value with |log p| <= 600, i.e.
# every one that decides a call or a QUAL below 2600, is held to 1e-10.  Beyond that the reference's own log-space
# recurrence is the limit: each of its N row updates rounds numbers of magnitude ~|log p|, and against an exact (80-bit,
# linear-space) recurrence it is off by up to a * ulp(|log p|) * N.  The constant is MEASURED in
# test_oracle_kat.py::test_linear_dp_truth_and_reference_noise (the oracle = the reference's arithmetic against
# orc_tail_truth).  Identical error probabilities are the worst case -- the roundings of consecutive rows line up and add
# linearly: a = 0.06 .. 0.12 for N = 1e4 .. 4.5e4, |log p| = 144 .. 3670 (2.7e-10 at N = 1e4, log p = -3670; 6.3e-10 at
# N = 45 000, log p = -988) --, ragged ones stay two orders of magnitude below (a random walk).  PV_NOISE_A = 0.16 is the
# worst measured value with a margin of 1.3, asserted there.  No independent implementation can agree with the reference
# better than the reference's own noise; the DEVICE's values are held to 2e-11 of the exact recurrence by
# test_deep_tail_against_80bit_truth and, for every record of the full C3 / C2 batches, by oracle/full_check.py.
PV_DEEP_LOG = 600.0
PV_NOISE_A = 0.16
PV_ERR_MAX = {"|log p| <= 600": [0.0, 0], "|log p| > 600": [0.0, 0, 0.0]}     # [max observed |dlog|, records compared(, max |dlog| / bound)]


def pv_deep_bound(logp, n_obs):
    """the bar for a p-value beyond |log p| = 600 of a column with n_obs error probabilities"""
    return max(PV_LOG_TOL, PV_NOISE_A * float(np.spacing(abs(float(logp)))) * max(float(n_obs), 1.0))


def pv_tol_for(pv_ref, n_obs=10000):
    lp = log_of(pv_ref)
    return pv_deep_bound(lp, n_obs) if abs(lp) > PV_DEEP_LOG else PV_LOG_TOL


def assert_pvalue_close(pv_gpu, pv_ref, tol=None, ctx="", n_obs=None):
    """Sentinels must match exactly; finite values within `tol` (default: per record -- 1e-10 up to |log p| = 600, the
    noise bound of the reference's own arithmetic beyond, pv_deep_bound).
    n_obs: the column's depth (number of error probabilities; an upper bound such as the coverage will do).  Where a
    caller does not know it, 10 000 is assumed -- deeper columns than that only exist in tests that pass it."""
    pv_gpu, pv_ref = np.longdouble(pv_gpu), np.longdouble(pv_ref)
    if pv_ref == LDBL_MAX or pv_ref == LDBL_MIN or pv_gpu == LDBL_MAX or pv_gpu == LDBL_MIN:
        assert pv_gpu == pv_ref, "sentinel mismatch %s: gpu=%r ref=%r" % (ctx, pv_gpu, pv_ref)
        return
    lp = log_of(pv_ref)
    d = abs(log_of(pv_gpu) - lp)
    deep = abs(lp) > PV_DEEP_LOG
    if tol is None:
        # a caller that knows the column's depth passes it and gets the measured bound for that depth; one that does not gets
        # the bound of a 10 000-deep column, capped at 1e-9 (a default must not loosen with |log p|)
        tol = (pv_deep_bound(lp, n_obs) if n_obs is not None else min(pv_deep_bound(lp, 10000), 1e-9)) if deep else PV_LOG_TOL
    st = PV_ERR_MAX["|log p| > 600" if deep else "|log p| <= 600"]
    st[0] = max(st[0], d)
    st[1] += 1
    if deep:
        st[2] = max(st[2], d / tol)
    assert d <= tol, "p-value mismatch %s: gpu=%r ref=%r |dlog|=%g (tolerance %g)" % (ctx, pv_gpu, pv_ref, d, tol)


def random_indel_columns(rng, ncols, depth_lo=20, depth_hi=400, p_event=0.5, polyat=False):
    """Random indel fields of `ncols` pileup columns (dicts for lofreq_amd.indel.IndelColumns.from_columns)."""
    cols = []
    for _ in range(ncols):
        depth = int(rng.integers(depth_lo, depth_hi + 1))
        col = {"ref": str(rng.choice(list("ACGTN"), p=[0.24, 0.24, 0.24, 0.24, 0.04])), "hrun": int(rng.integers(0, 9))}
        tails = int(rng.integers(0, 3))
        tot_ev = [0, 0]
        for sd, sn in enumerate(("ins", "dels")):
            events = []
            left = depth
            if rng.random() < p_event:
                n_ev = int(rng.integers(1, 4))
                keys = set()
                for _e in range(n_ev):
                    klen = 1 if (polyat or rng.random() < 0.5) else int(rng.integers(2, 6))
                    key = "".join(rng.choice(list("AT" if polyat else "ACGT"), klen))
                    if key in keys:
                        continue
                    keys.add(key)
                    frac = float(rng.choice([0.005, 0.02, 0.04, 0.1, 0.4]))
                    cnt = max(1, min(left - 1, int(round(frac * depth * rng.uniform(0.5, 1.5)))))
                    if cnt <= 0 or left - cnt < 1:
                        continue
                    left -= cnt
                    fw = int(rng.integers(0, cnt + 1))
                    events.append({
                        "key": key, "fw": fw, "rv": cnt - fw,
                        "q": rng.integers(20, 61, cnt).tolist(),
                        "aq": rng.choice([-1, 10, 25, 40, 60], cnt).tolist(),
                        "mq": rng.choice([0, 20, 42, 60, 255], cnt, p=[0.02, 0.08, 0.1, 0.78, 0.02]).tolist(),
                        "sq": rng.choice([-1, 30, 50], cnt).tolist(),
                    })
            n_ne = left
            nfw = int(rng.integers(0, n_ne + 1))
            col[sn] = {"non_fw": nfw, "non_rv": n_ne - nfw,
                       "ne_q": rng.integers(25, 61, n_ne).tolist(),
                       "ne_mq": rng.choice([0, 20, 42, 60, 255], n_ne, p=[0.02, 0.08, 0.1, 0.78, 0.02]).tolist(),
                       "events": events}
            tot_ev[sd] = depth - left
        col["coverage_plp"] = depth + tails
        col["num_tails"] = tails
        col["num_non_indels"] = max(depth - tot_ev[0] - tot_ev[1], 0)
        cols.append(col)
    return cols


def make_region_reads(seed, glen, depth, rl=150, snv_every=400, indel_every=1500, err=0.003, lo=0, hi=None):
    """Position-sorted synthetic reads over [lo, hi) of a random genome of length glen, with PLANTED variants so that the
    callers have something to report: an SNV site every `snv_every` bases (allele frequency cycling 1 %, 3 %, 10 %, 50 %) and
    an insertion / deletion site (1-3 bases, alternating) every `indel_every` bases (AF cycling 3 %, 10 %, 30 %) carried by
    the reads that cover it with at least 12 bases on both sides; sequencing errors at rate `err`, qualities ~N(34, 5)
    clipped to 2..41, BI / BD 30..49, mapping quality 60 (8 %: 20..59), random strands.  The genome depends on the seed
    only, so regions of one genome can be generated separately.
    -> dict of flat arrays in the layout of lfq_pileup_reads / orc_reads (keys as oracle/pyoracle.py::pack_reads)."""
    hi = glen if hi is None else hi
    genome = np.random.default_rng(seed).integers(0, 4, glen).astype(np.uint8)
    rng = np.random.default_rng([seed, lo, hi])
    n = int(depth * (hi - lo) / rl)
    pos = np.sort(rng.integers(max(lo - rl + 1, 0), max(min(hi, glen - rl - 8), 1), n)).astype(np.int64)
    snv_sites = np.arange(snv_every // 2, glen, snv_every)
    snv_af = np.array([0.01, 0.03, 0.1, 0.5])[np.arange(len(snv_sites)) % 4]
    snv_alt = (genome[snv_sites] + 1 + (np.arange(len(snv_sites)) % 3)) % 4
    ind_sites = np.arange(indel_every // 3, glen - rl, indel_every)
    ind_af = np.array([0.03, 0.1, 0.3])[np.arange(len(ind_sites)) % 3]
    ind_len = 1 + (np.arange(len(ind_sites)) % 3)
    ind_ins = (np.arange(len(ind_sites)) % 2) == 0
    ins_seq = np.random.default_rng(seed + 1).integers(0, 4, (len(ind_sites), 3)).astype(np.uint8)
    # which reads carry which planted indel (at most one per read)
    carry = np.full(n, -1, np.int64)
    si = np.searchsorted(ind_sites, pos + 12)
    for k in range(2):                                          # a read spans at most a couple of sites
        j = si + k
        ok = (j < len(ind_sites))
        jj = np.where(ok, j, 0)
        inside = ok & (ind_sites[jj] >= pos + 12) & (ind_sites[jj] <= pos + rl - 16) & (carry < 0)
        take = inside & (rng.random(n) < ind_af[jj])
        carry[take] = jj[take]
    seqs = np.empty((n, rl), np.uint8)
    ref_idx = pos[:, None] + np.arange(rl)[None, :]
    plain = carry < 0
    seqs[plain] = genome[ref_idx[plain]]
    cig = np.zeros((n, 3), np.uint32)
    ncig = np.ones(n, np.int64)
    cig[:, 0] = rl << 4
    # reference coordinate of every base (for the planted SNVs), -1 for inserted bases
    rpos = ref_idx.copy()
    for i in np.nonzero(~plain)[0]:
        s = int(carry[i])
        c = int(ind_sites[s] - pos[i]) + 1                      # bases before the event (the event follows base c - 1)
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
    # planted SNVs: a read base on a site becomes the alt allele with the site's frequency
    on = np.isin(rpos, snv_sites)
    if on.any():
        ri, ci = np.nonzero(on)
        sidx = np.searchsorted(snv_sites, rpos[ri, ci])
        flip = rng.random(len(ri)) < snv_af[sidx]
        seqs[ri[flip], ci[flip]] = snv_alt[sidx[flip]]
    mism = rng.random(seqs.shape) < err
    seqs[mism] = (seqs[mism] + 1 + rng.integers(0, 3, int(mism.sum()))) % 4
    qual = np.clip(np.round(rng.normal(34, 5, seqs.shape)), 2, 41).astype(np.uint8)
    cig_off = np.zeros(n + 1, np.int64)
    cig_off[1:] = np.cumsum(ncig)
    mapq = np.where(rng.random(n) < 0.92, 60, rng.integers(20, 60, n)).astype(np.uint8)
    return {
        "n": n, "rl": rl, "glen": glen, "ref": np.frombuffer(b"ACGT", np.uint8)[genome].tobytes(),
        "pos": pos.astype(np.int32), "cig_off": cig_off, "cig": np.ascontiguousarray(cig[np.arange(3)[None, :] < ncig[:, None]]),
        "seq_off": np.arange(n + 1, dtype=np.int64) * rl, "seq": np.ascontiguousarray(seqs.reshape(-1)),
        "qual": np.ascontiguousarray(qual.reshape(-1)),
        "bi": rng.integers(33 + 30, 33 + 50, n * rl).astype(np.uint8), "bd": rng.integers(33 + 30, 33 + 50, n * rl).astype(np.uint8),
        "ai": None, "ad": None, "lb": None, "sq": None, "flags": np.full(max(n, 1), 3, np.uint8),
        "mapq": mapq, "rev": (rng.random(n) < 0.5).astype(np.uint8),
        "n_indel_reads": int((~plain).sum()), "snv_sites": snv_sites, "indel_sites": ind_sites,
    }
