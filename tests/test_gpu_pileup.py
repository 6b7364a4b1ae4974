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
    assert res[0] == res[1]
    assert len(res[0][0]) > 0
