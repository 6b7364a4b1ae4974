"""-m gpu: the device-side pileup (lfq_pileup_snv_tracks) against the reference binary's own column dump
(`lofreq plpsummary`) of the same reads, incl. insertions, deletions and low base qualities."""
import json
import os

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180)]


def _fetch(ptr, nbytes):
    """device memory at a raw pointer -> numpy (through torch's allocator-free path: hipMemcpy via ctypes)"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    out = np.zeros(nbytes, np.uint8)
    assert hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(ptr), C.c_size_t(nbytes), 2) == 0   # DeviceToHost
    return out


def _fetch_nt(t, n_obs):
    """the nt track of device tracks, one code per observation whatever the layout (LFQ_TRACKS_NT_PACKED: observation o
    sits in byte (o >> 3) * 4 + (o & 3), nibble (o & 7) >> 2)"""
    if not (t.flags & 1):
        return _fetch(t.nt, max(n_obs, 1))[:n_obs]
    raw = _fetch(t.nt, (n_obs + 7) // 8 * 4 + 4)
    o = np.arange(n_obs)
    return ((raw[(o >> 3) * 4 + (o & 3)] >> (4 * ((o & 7) >> 2))) & 15).astype(np.uint8)


@pytest.mark.parametrize("packed", [True, False], ids=["nt_packed", "nt_bytes"])
@pytest.mark.parametrize("path", gu.pileup_fixtures(), ids=lambda p: p.split("/")[-1])
def test_pileup_matches_plpsummary(caller, path, packed):
    import lofreq_amd as la
    caller.set_pileup_nt_packed(packed)
    try:
        _pileup_matches_plpsummary(caller, path, packed)
    finally:
        caller.set_pileup_nt_packed(True)


def _pileup_matches_plpsummary(caller, path, packed):
    import lofreq_amd as la
    fx = json.load(open(path))
    reads = [{"pos0": r[0], "cigar": gu.parse_cigar(r[3]), "seq": la.encode_seq(r[4]),
              "qual": np.array([ord(c) - 33 for c in r[5]], np.uint8), "mapq": r[2], "reverse": bool(r[1] & 16)}
             for r in fx["reads"]]
    lb = [np.frombuffer(r[6].encode(), np.uint8) for r in fx["reads"]]
    glen = len(fx["genome"])
    dt = la.pileup_snv_tracks(caller, reads, fx["genome"].encode(), 0, glen, lb=lb, min_plp_bq=3)
    t = dt._tracks()
    ncols = dt.ncols
    off = _fetch(t.col_off, (ncols + 1) * 8).view(np.uint64)
    n_obs = int(off[-1])
    assert bool(t.flags & 1) == packed           # the layout the context was asked for (default: packed)
    nt = _fetch_nt(t, n_obs)
    bq, baq, mq = (_fetch(p, n_obs) for p in (t.bq, t.baq, t.mq))
    ref = _fetch(t.ref_base, ncols)
    exp = {c["pos0"]: c for c in fx["columns"]}
    assert ncols >= len(exp)
    checked = 0
    for ci in range(ncols):
        p0 = int(dt.col_pos[ci])
        a, b = int(off[ci]), int(off[ci + 1])
        e = exp.get(p0)
        if e is None:               # plpsummary prints nothing for columns without a single base (all deleted)
            assert a == b
            continue
        assert chr(ref[ci]) == e["ref"]
        for code, letter in enumerate("ACGTN"):
            sel = (nt[a:b] & 7) == code
            o = e["obs"].get(letter)
            # position-sorted reads take the column-major kernels: the observations come out in pileup order, i.e.
            # exactly the order of the reference's per-nucleotide arrays (LFQ_PILEUP_ATOMIC=1: any order)
            got = list(zip(bq[a:b][sel].tolist(), baq[a:b][sel].tolist(), mq[a:b][sel].tolist()))
            want = [] if not o else list(zip(gu.dec(o["bq"]).tolist(),
                                             [(255 if v < 0 else v) for v in gu.dec(o["baq"]).tolist()], o["mq"]))
            if os.environ.get("LFQ_PILEUP_ATOMIC"):
                got, want = sorted(got), sorted(want)
            assert got == want, (p0, letter)
            fw = int((sel & ((nt[a:b] & 8) == 0)).sum())
            assert [fw, int(sel.sum()) - fw] == e["fwrv"][letter], (p0, letter)
            checked += len(got)
    assert checked > 20000


def test_pileup_then_call_device_resident(caller, oracle):
    """the tracks stay in HBM: pileup -> lfq_call_snvs_batch(tracks_on_device) gives the same records as packing
    the same columns on the host"""
    import lofreq_amd as la
    import util
    fx = json.load(open(gu.chain_fixtures()[0]))
    reads = [{"pos0": r[0], "cigar": gu.parse_cigar(r[3]), "seq": la.encode_seq(r[4]),
              "qual": np.array([ord(c) - 33 for c in r[5]], np.uint8), "mapq": r[2], "reverse": bool(r[1] & 16)}
             for r in fx["reads"]]
    lb = la.baq_batch(caller, reads, fx["genome"].encode(), extended=True)
    dt = la.pileup_snv_tracks(caller, reads, fx["genome"].encode(), 0, len(fx["genome"]), lb=lb)
    conf = la.VarcallConf()
    recs, _, st = caller.call_snvs(dt, conf)
    assert conf.num_snv_tests == fx["num_snv_tests"]
    thr = la.snvqual_thresh(conf.sig, conf.bonf_subst)
    keep = la.filter_records(recs, thr, apply_defaults=True)
    pos0 = np.array([dt.col_pos[int(r["col"])] for r in recs], np.int64)
    text = la.format_vcf(recs, "chr1", pos0=pos0, keep=keep, filter_str="PASS")
    assert [gu.strip_hqa(l) for l in text.splitlines()] == fx["vcf"]


def test_pileup_unsorted_reads_take_the_read_major_kernels(caller):
    """reads that are not position-sorted cannot use the window search: the read-major (atomic) kernels run instead;
    same columns, same observations per column as for the sorted list (any order), same calls"""
    import lofreq_amd as la
    fx = json.load(open(gu.pileup_fixtures()[0]))
    reads = [{"pos0": r[0], "cigar": gu.parse_cigar(r[3]), "seq": la.encode_seq(r[4]),
              "qual": np.array([ord(c) - 33 for c in r[5]], np.uint8), "mapq": r[2], "reverse": bool(r[1] & 16)}
             for r in fx["reads"]]
    lb = [np.frombuffer(r[6].encode(), np.uint8) for r in fx["reads"]]
    ref = fx["genome"].encode()
    rng = np.random.default_rng(1)
    perm = rng.permutation(len(reads))
    res = []
    # refused by default, like a file that is not coordinate-sorted is by mpileup (plp.c:1406-1447) ...
    if not os.environ.get("LFQ_PILEUP_ATOMIC"):     # (the tuning build's knob sends every read set to the read-major kernels)
        with pytest.raises(RuntimeError):
            la.pileup_snv_tracks(caller, [reads[i] for i in perm], ref, 0, len(ref), lb=[lb[i] for i in perm])
        with pytest.raises(RuntimeError):
            la.pileup_indel_columns(caller, [reads[i] for i in perm], ref, 0, len(ref))
    caller.set_pileup_unsorted(True)            # ... taken when the caller asks for it (lfq_set_pileup_unsorted)
    for order in (np.arange(len(reads)), perm):
        dt = la.pileup_snv_tracks(caller, [reads[i] for i in order], ref, 0, len(ref), lb=[lb[i] for i in order])
        t = dt._tracks()
        off = _fetch(t.col_off, (dt.ncols + 1) * 8).view(np.uint64)
        n_obs = int(off[-1])
        tr = [_fetch_nt(t, n_obs)] + [_fetch(p, n_obs) for p in (t.bq, t.baq, t.mq)]
        cols = [sorted(zip(*(x[int(off[c]):int(off[c + 1])].tolist() for x in tr))) for c in range(dt.ncols)]
        recs, _, _ = caller.call_snvs(dt, la.VarcallConf())
        res.append((dt.col_pos.tolist(), cols, recs.tobytes()))
        icols, ipos = la.pileup_indel_columns(caller, [reads[i] for i in order], ref, 0, len(ref))
        res[-1] += (ipos.tolist(), icols.num_non_indels.tolist(), icols.num_tails.tolist(), icols.coverage_plp.tolist())
    caller.set_pileup_unsorted(False)
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1] and res[0][2] == res[1][2]
    assert res[0][3:] == res[1][3:]


@pytest.mark.parametrize("n_reads", [1, 2, 63, 64, 65, 66, 129, 130, 4224, 4225, 4226, 4300])
def test_window_search_at_its_round_boundaries(caller, n_reads):
    """the column-major kernels find a position's window of reads with 64 probes per round (lfq_wave_first_above): read
    counts at the edges of one, two and three rounds (64, 65, 65 * 65 = 4225), windows at both ends of the read list.
    Same columns and the same observations per column as the read-major kernels (which do not search) on the shuffled list."""
    import lofreq_amd as la
    import test_gpu_plpindel as T
    rng = np.random.default_rng(1000 + n_reads)
    glen = 900
    genome = rng.integers(0, 4, glen).astype(np.uint8)
    ref = "".join("ACGT"[c] for c in genome).encode()
    reads = T._random_indel_reads(rng, n_reads, glen, genome)
    perm = rng.permutation(n_reads)
    if n_reads > 1 and all(reads[perm[i]]["pos0"] <= reads[perm[i + 1]]["pos0"] for i in range(n_reads - 1)):
        perm = perm[::-1].copy()                            # (two reads: make sure the second list is not sorted)
    res = []
    caller.set_pileup_unsorted(True)
    for order in (np.arange(n_reads), perm):
        rd = [reads[i] for i in order]
        dt = la.pileup_snv_tracks(caller, rd, ref, 0, glen)
        t = dt._tracks()
        off = _fetch(t.col_off, (dt.ncols + 1) * 8).view(np.uint64)
        n_obs = int(off[-1])
        tr = [_fetch_nt(t, n_obs)] + [_fetch(p, max(n_obs, 1))[:n_obs] for p in (t.bq, t.mq)]
        cols = [sorted(zip(*(x[int(off[c]):int(off[c + 1])].tolist() for x in tr))) for c in range(dt.ncols)]
        icols, ipos = la.pileup_indel_columns(caller, rd, ref, 0, glen)
        res.append((dt.col_pos.tolist(), cols, ipos.tolist(), icols.coverage_plp.tolist(), icols.num_non_indels.tolist(),
                    icols.num_tails.tolist(), icols.num_ins.tolist(), icols.num_dels.tolist()))
    caller.set_pileup_unsorted(False)
    assert res[0] == res[1]
    assert len(res[0][0]) > 0


def _hard_reads(rng, glen, n):
    """reads that exercise the tile kernel's phase A: two indels within 64 positions, insertions too long for a row,
    = / X / N / S / H operations, reads that start and end inside a tile, deep piles with ragged starts"""
    import lofreq_amd as la
    reads = []
    starts = np.sort(np.concatenate([rng.integers(0, glen - 400, n), rng.integers(300, 330, n // 3)]))
    for pos in starts.tolist():
        kind = rng.integers(0, 8)
        cigar = []
        if rng.random() < 0.3:
            cigar.append(("S", int(rng.integers(1, 12))))
        body = int(rng.integers(20, 180))
        if kind == 0:                                           # two indels close together
            a, b = int(rng.integers(5, 30)), int(rng.integers(2, 25))
            cigar += [("M", a), ("I" if rng.random() < 0.5 else "D", int(rng.integers(1, 5))), ("M", b),
                      ("D" if rng.random() < 0.5 else "I", int(rng.integers(1, 6))), ("M", body)]
        elif kind == 1:                                         # an insertion longer than a row's slack
            cigar += [("M", int(rng.integers(3, 40))), ("I", int(rng.integers(18, 60))), ("M", body)]
        elif kind == 2:                                         # = and X next to each other, a skipped region
            cigar += [("=", int(rng.integers(3, 50))), ("X", 1), ("=", int(rng.integers(3, 50))), ("N", int(rng.integers(1, 90))),
                      ("M", body)]
        elif kind == 3:                                         # three indels within a few bases
            cigar += [("M", 4), ("I", 1), ("M", 3), ("D", 2), ("M", 2), ("I", 2), ("M", body)]
        elif kind == 4:                                         # a long deletion: whole tiles inside it
            cigar += [("M", int(rng.integers(10, 40))), ("D", int(rng.integers(60, 200))), ("M", body)]
        else:
            cigar += [("M", body)]
        if rng.random() < 0.2:
            cigar.append(("S", int(rng.integers(1, 9))))
        if rng.random() < 0.1:
            cigar.append(("H", 5))
        ql = sum(l for o, l in cigar if o in "MIS=X")
        reads.append({"pos0": int(pos), "cigar": cigar, "seq": rng.integers(0, 5, ql).astype(np.uint8),
                      "qual": rng.integers(0, 61, ql).astype(np.uint8), "mapq": int(rng.integers(0, 61)),
                      "reverse": bool(rng.integers(0, 2)), "lb": rng.integers(33, 127, ql).astype(np.uint8)})
    return reads


@pytest.mark.parametrize("begin,end", [(0, 2000), (37, 1201), (250, 400)])
def test_tile_kernel_hard_reads_against_oracle_pileup(caller, oracle, begin, end):
    """lfq_pileup_tiles_kernel (both passes) on reads built against its fast path, for regions that start and end inside
    tiles: every track byte, in pileup order, against the oracle's column builder (oracle/orc_pileup.c)"""
    import lofreq_amd as la
    rng = np.random.default_rng(5 + begin)
    glen = 2000
    genome = "".join(rng.choice(list("ACGT"), glen)).encode()
    reads = _hard_reads(rng, glen, 900)
    ends = [r["pos0"] + sum(l for o, l in r["cigar"] if o in "MDN=X") for r in reads]
    reads = [r for r, e in zip(reads, ends) if e <= glen]
    caller.set_pileup_nt_packed(False)
    try:
        dt = la.pileup_snv_tracks(caller, reads, genome, begin, end, lb=[r["lb"] for r in reads], min_plp_bq=3)
        t = dt._tracks()
        ncols = dt.ncols
        off = _fetch(t.col_off, (ncols + 1) * 8).view(np.uint64)
        n_obs = int(off[-1])
        got = {k: _fetch(p, max(n_obs, 1))[:n_obs] for k, p in (("nt", t.nt), ("bq", t.bq), ("baq", t.baq), ("mq", t.mq))}
        cov = _fetch(t.coverage_plp, ncols * 4).view(np.int32)
        nb = _fetch(t.num_bases, ncols * 4).view(np.int32)
        col_pos = np.asarray(dt.col_pos[:ncols])
    finally:
        caller.set_pileup_nt_packed(True)
    out = oracle.pileup_region(oracle.pack_reads(reads, genome), begin, end, min_plp_bq=3, use_baq=True)
    h = out["host"]
    assert np.array_equal(col_pos, out["col_pos"])
    assert np.array_equal(off, h["col_off"])
    assert np.array_equal(cov, h["coverage_plp"]) and np.array_equal(nb, h["num_bases"])
    if os.environ.get("LFQ_PILEUP_ATOMIC"):         # the read-major kernels: a column's observations in any order
        pack = lambda d: (d["nt"].astype(np.uint32) | (d["bq"].astype(np.uint32) << 8) | (d["baq"].astype(np.uint32) << 16)
                          | (d["mq"].astype(np.uint32) << 24))
        a, b = pack(got), pack({k: h[k][:n_obs] for k in ("nt", "bq", "baq", "mq")})
        for c in range(ncols):
            assert np.array_equal(np.sort(a[int(off[c]):int(off[c + 1])]), np.sort(b[int(off[c]):int(off[c + 1])])), c
    else:
        for k in ("nt", "bq", "baq", "mq"):
            assert np.array_equal(got[k], h[k][:n_obs]), k
    assert n_obs > 20000 and int(np.diff(off.astype(np.int64)).max()) > 256      # more than one round of reads per tile


@pytest.mark.parametrize("begin,end", [(0, 21000), (4095, 12289), (5000, 5001), (8200, 20500)])
def test_columns_assembled_on_the_device_over_tiles_and_gaps(caller, oracle, begin, end):
    """lfq_plp_compact_* (column index, offsets, reference bytes, compacted counts from the per-position counters): regions of
    several 4096-position tiles that start and end inside tiles, with stretches no read covers (whole tiles among them) and
    reference letters other than ACGT -- every column array and every track byte against the oracle's column builder"""
    import lofreq_amd as la
    rng = np.random.default_rng(77 + begin)
    glen = 21000
    g = rng.choice(list("ACGT"), glen)
    g[rng.integers(0, glen, 300)] = rng.choice(list("NRYacgtn"), 300)     # plp.c:818-823: anything but ACGTN reads as N
    genome = "".join(g).encode()
    reads = []
    for lo, hi, n in ((0, 3000, 300), (3900, 4300, 150), (9000, 9050, 40), (12280, 12300, 30), (16500, 20800, 500)):
        for pos in np.sort(rng.integers(lo, hi, n)).tolist():
            body = int(rng.integers(30, 150))
            cigar = [("M", body)] if rng.random() < 0.8 else [("M", body // 2), ("D", int(rng.integers(1, 30))), ("M", body - body // 2)]
            ql = sum(l for o, l in cigar if o in "MIS=X")
            reads.append({"pos0": int(pos), "cigar": cigar, "seq": rng.integers(0, 5, ql).astype(np.uint8),
                          "qual": rng.integers(0, 61, ql).astype(np.uint8), "mapq": int(rng.integers(0, 61)),
                          "reverse": bool(rng.integers(0, 2)), "lb": rng.integers(33, 127, ql).astype(np.uint8)})
    reads.sort(key=lambda r: r["pos0"])
    reads = [r for r in reads if r["pos0"] + sum(l for o, l in r["cigar"] if o in "MDN=X") <= glen]
    caller.set_pileup_nt_packed(False)
    try:
        dt = la.pileup_snv_tracks(caller, reads, genome, begin, end, lb=[r["lb"] for r in reads], min_plp_bq=3)
        t = dt._tracks()
        ncols = dt.ncols
        off = _fetch(t.col_off, (ncols + 1) * 8).view(np.uint64)
        n_obs = int(off[-1])
        got = {k: _fetch(p, max(n_obs, 1))[:n_obs] for k, p in (("nt", t.nt), ("bq", t.bq), ("baq", t.baq), ("mq", t.mq))}
        cov = _fetch(t.coverage_plp, max(ncols, 1) * 4).view(np.int32)[:ncols]
        nb = _fetch(t.num_bases, max(ncols, 1) * 4).view(np.int32)[:ncols]
        ref = _fetch(t.ref_base, max(ncols, 1))[:ncols]
        col_pos = np.asarray(dt.col_pos[:ncols])
        max_obs = int(t.max_col_obs)
    finally:
        caller.set_pileup_nt_packed(True)
    out = oracle.pileup_region(oracle.pack_reads(reads, genome), begin, end, min_plp_bq=3, use_baq=True)
    h = out["host"]
    assert ncols == len(out["col_pos"]) and np.array_equal(col_pos, out["col_pos"])
    assert np.array_equal(off, h["col_off"])
    assert np.array_equal(cov, h["coverage_plp"]) and np.array_equal(nb, h["num_bases"])
    assert np.array_equal(ref, np.frombuffer(bytes(h["ref_base"][:ncols]), dtype=np.uint8))
    assert max_obs == (int(nb.max()) if ncols else 0)
    if not os.environ.get("LFQ_PILEUP_ATOMIC"):
        for k in ("nt", "bq", "baq", "mq"):
            assert np.array_equal(got[k], h[k][:n_obs]), k
    if end - begin > 1:
        assert 0 < ncols < end - begin                  # gaps inside the region
