"""Device-side pileup: reads of a region -> packed SNV tracks in HBM (`lfq_pileup_snv_tracks`),
what `compile_plp_col` (plp.c:797-1017) builds per column on the CPU."""
import ctypes as C

import numpy as np

from . import _lib
from .baq import _OPS


class DeviceTracks:
    """Tracks owned by the caller's context (valid until its next pileup call); quacks like a device PileupBatch."""

    on_device = True

    def __init__(self, tracks, col_pos):
        self._t = tracks
        self.ncols = int(tracks.ncols)
        self.max_col_obs = int(tracks.max_col_obs)
        self.col_pos = col_pos

    def _tracks(self):
        return self._t


def pileup_snv_tracks(caller, reads, ref, begin, end, lb=None, min_plp_bq=3, sq=None):
    """reads: list of dicts {pos0, cigar [(op, len)], seq (codes 0..4), qual (phred), mapq, reverse};
    lb: list of the reads' lb tag bytes (from baq_batch) or None; sq: the reads' source-quality bytes (second
    result of source_qual_batch) or None.  -> DeviceTracks"""
    n = len(reads)
    pos = np.asarray([r["pos0"] for r in reads], np.int32)
    cig_off = np.zeros(n + 1, np.int64)
    seq_off = np.zeros(n + 1, np.int64)
    cig, seqs, quals = [], [], []
    for i, r in enumerate(reads):
        cig.extend((l << 4) | _OPS.index(o) for o, l in r["cigar"])
        cig_off[i + 1] = len(cig)
        seqs.append(np.asarray(r["seq"], np.uint8))
        quals.append(np.asarray(r["qual"], np.uint8))
        seq_off[i + 1] = seq_off[i] + len(seqs[-1])
    cig = np.asarray(cig if cig else [0], np.uint32)
    seq = np.concatenate(seqs) if seqs else np.zeros(1, np.uint8)
    qual = np.concatenate(quals) if quals else np.zeros(1, np.uint8)
    baq = None if lb is None else np.concatenate([np.asarray(x, np.uint8) for x in lb])
    mapq = np.asarray([r["mapq"] for r in reads] or [0], np.uint8)
    rev = np.asarray([1 if r["reverse"] else 0 for r in reads] or [0], np.uint8)
    ref = bytes(ref)
    rd = _lib.PileupReads()
    rd.n_reads = n
    rd.pos = pos.ctypes.data
    rd.cigar_off = cig_off.ctypes.data
    rd.cigar = cig.ctypes.data
    rd.seq_off = seq_off.ctypes.data
    rd.seq = seq.ctypes.data
    rd.qual = qual.ctypes.data
    rd.baq = baq.ctypes.data if baq is not None else None
    rd.mapq = mapq.ctypes.data
    rd.reverse = rev.ctypes.data
    rd.ref = C.cast(C.c_char_p(ref), C.c_void_p)
    rd.ref_len = len(ref)
    if sq is not None:
        sq = np.ascontiguousarray(sq, np.uint8)
        assert len(sq) == n
        rd.sq = sq.ctypes.data
    t = _lib.Tracks()
    col_pos = np.zeros(max(end - begin, 1), np.int64)
    _lib.check(_lib.load().lfq_pileup_snv_tracks(caller.h, C.byref(rd), int(begin), int(end), int(min_plp_bq),
                                                 C.byref(t), col_pos.ctypes.data), "lfq_pileup_snv_tracks")
    return DeviceTracks(t, col_pos[: int(t.ncols)].copy())


def _pack_reads(reads, ref, lb=None, sq=None):
    n = len(reads)
    keep = {"pos": np.asarray([r["pos0"] for r in reads] or [0], np.int32)}
    cig_off = np.zeros(n + 1, np.int64)
    seq_off = np.zeros(n + 1, np.int64)
    cig, seqs, quals = [], [], []
    for i, r in enumerate(reads):
        cig.extend((l << 4) | _OPS.index(o) for o, l in r["cigar"])
        cig_off[i + 1] = len(cig)
        seqs.append(np.asarray(r["seq"], np.uint8))
        quals.append(np.asarray(r["qual"], np.uint8))
        seq_off[i + 1] = seq_off[i] + len(seqs[-1])
    keep["cig_off"], keep["seq_off"] = cig_off, seq_off
    keep["cig"] = np.asarray(cig if cig else [0], np.uint32)
    keep["seq"] = np.concatenate(seqs) if seqs else np.zeros(1, np.uint8)
    keep["qual"] = np.concatenate(quals) if quals else np.zeros(1, np.uint8)
    keep["mapq"] = np.asarray([r["mapq"] for r in reads] or [0], np.uint8)
    keep["rev"] = np.asarray([1 if r["reverse"] else 0 for r in reads] or [0], np.uint8)
    keep["ref"] = bytes(ref)
    rd = _lib.PileupReads()
    rd.n_reads = n
    rd.pos = keep["pos"].ctypes.data
    rd.cigar_off = cig_off.ctypes.data
    rd.cigar = keep["cig"].ctypes.data
    rd.seq_off = seq_off.ctypes.data
    rd.seq = keep["seq"].ctypes.data
    rd.qual = keep["qual"].ctypes.data
    rd.mapq = keep["mapq"].ctypes.data
    rd.reverse = keep["rev"].ctypes.data
    rd.ref = C.cast(C.c_char_p(keep["ref"]), C.c_void_p)
    rd.ref_len = len(keep["ref"])
    return rd, keep


def pileup_indel_columns(caller, reads, ref, begin, end, min_plp_idq=0):
    """The indel fields of the pileup (`lfq_pileup_indel_columns`).  reads: as for pileup_snv_tracks, plus the
    optional per-read entries "bi", "bd", "ai", "ad" (tag bytes, uint8 arrays of the read's length, or None) and
    "sq" (int).  -> (IndelColumns for call_indels / format_indel_record, positions of the columns)"""
    from .indel import IndelColumns, _I32
    n = len(reads)
    rd, keep = _pack_reads(reads, ref)
    n_bases = int(keep["seq_off"][-1])
    tags = _lib.PileupIndelTags()
    flags = np.zeros(max(n, 1), np.uint8)
    for bit, name in enumerate(("bi", "bd", "ai", "ad")):
        if any(r.get(name) is not None for r in reads):
            arr = np.full(max(n_bases, 1), 33, np.uint8)
            for i, r in enumerate(reads):
                if r.get(name) is not None:
                    arr[keep["seq_off"][i]:keep["seq_off"][i + 1]] = np.asarray(r[name], np.uint8)
                    flags[i] |= 1 << bit
            keep[name] = arr
            setattr(tags, name, arr.ctypes.data)
    keep["flags"] = flags
    tags.tag_flags = flags.ctypes.data
    if any(r.get("sq") is not None for r in reads):
        keep["sq"] = np.asarray([(-1 if r.get("sq") is None else r["sq"]) for r in reads], np.int32)
        tags.sq = keep["sq"].ctypes.data
    out = C.POINTER(_lib.IndelColumnsC)()
    col_pos = np.zeros(max(end - begin, 1), np.int64)
    _lib.check(_lib.load().lfq_pileup_indel_columns(caller.h, C.byref(rd), C.byref(tags), int(begin), int(end),
                                                    int(min_plp_idq), C.byref(out), col_pos.ctypes.data),
               "lfq_pileup_indel_columns")
    return _indel_columns_from_c(out, col_pos, caller)


def _indel_columns_from_c(out, col_pos, caller=None):
    """copy a context-owned lfq_indel_columns into an IndelColumns; with `caller`, remember the original so that
    call_indels can hand it back (its quality arrays are still resident on the device)"""
    from .indel import IndelColumns, _I32
    cs = out.contents
    ncols = int(cs.ncols)

    def arr(ptr, count, dt):
        if count == 0 or not ptr:
            return np.zeros(0, dt)
        return np.frombuffer(C.string_at(ptr, count * np.dtype(dt).itemsize), dtype=dt).copy()

    o = IndelColumns()
    o.ncols = ncols
    o.ref_base = arr(cs.ref_base, ncols, np.uint8)
    for name in _I32:
        setattr(o, name, arr(getattr(cs, name), ncols, np.int32))
    for sd in range(2):
        S = cs.side[sd]
        ne_off = arr(S.ne_off, ncols + 1, np.int64)
        ev_off = arr(S.ev_off, ncols + 1, np.int64)
        nev = int(ev_off[-1]) if ncols else 0
        key_off = arr(S.key_off, nev + 1, np.int64)
        rd_off = arr(S.rd_off, nev + 1, np.int64)
        nrd = int(rd_off[-1]) if nev else 0
        nne = int(ne_off[-1]) if ncols else 0
        key_chars = arr(S.key_chars, (int(key_off[-1]) if nev else 0) + 1, np.uint8)
        o.sides[sd] = {
            "non_fw": arr(S.non_fw, ncols, np.int32), "non_rv": arr(S.non_rv, ncols, np.int32), "ne_off": ne_off,
            "ne_q": arr(S.ne_q, nne, np.int16), "ne_mq": arr(S.ne_mq, nne, np.int16), "ev_off": ev_off,
            "key_off": key_off, "key_chars": key_chars, "ev_fw": arr(S.ev_fw, nev, np.int32),
            "ev_rv": arr(S.ev_rv, nev, np.int32), "rd_off": rd_off, "rd_q": arr(S.rd_q, nrd, np.int16),
            "rd_aq": arr(S.rd_aq, nrd, np.int16), "rd_mq": arr(S.rd_mq, nrd, np.int16),
            "rd_sq": arr(S.rd_sq, nrd, np.int16)}
        kc = key_chars.tobytes()
        o.keys[sd] = [kc[key_off[e]:key_off[e + 1]].decode() for e in range(nev)]
    o.cons_indel = arr(cs.cons_indel, ncols, np.uint8)
    if caller is not None:
        caller._indel_gen = getattr(caller, "_indel_gen", 0) + 1
        o._c_ptr, o._c_gen, o._c_caller = out, caller._indel_gen, caller      # valid until the context's next indel pileup
    return o, col_pos[:ncols].copy()

def skip_snv_columns(caller, skip):
    """call_vars' gate (lofreq_call.c:928-931): take the columns with skip[col] != 0 (IndelColumns.cons_indel) out of
    the SNV tracks last returned by pileup_snv_tracks"""
    skip = np.ascontiguousarray(skip, np.uint8)
    _lib.check(_lib.load().lfq_pileup_skip_snv_columns(caller.h, skip.ctypes.data, len(skip)),
               "lfq_pileup_skip_snv_columns")


class ReadSet:
    """The reads of one contig region resident on the device (`lfq_readset`): upload once, then BAQ / IDAQ, source
    quality, both pileups and the calls without the per-base arrays leaving HBM.

        rs = ReadSet(caller, reads, ref)          # reads: dicts as for pileup_snv_tracks (+ "bi", "bd")
        rs.baq(idaq=True); rs.source_qual()       # optional steps, results stay resident
        tracks = rs.pileup_snv(0, len(ref)); cols, col_pos = rs.pileup_indels(0, len(ref))
    """

    def __init__(self, caller, reads, ref):
        self.caller = caller
        self.L = _lib.load()
        n = len(reads)
        self.n = n
        rd, keep = _pack_reads(reads, ref)
        n_bases = int(keep["seq_off"][-1])
        tags = _lib.PileupIndelTags()
        flags = np.zeros(max(n, 1), np.uint8)
        for bit, name in enumerate(("bi", "bd", "ai", "ad")):
            if any(r.get(name) is not None for r in reads):
                arr = np.full(max(n_bases, 1), 33, np.uint8)
                for i, r in enumerate(reads):
                    if r.get(name) is not None:
                        arr[keep["seq_off"][i]:keep["seq_off"][i + 1]] = np.asarray(r[name], np.uint8)
                        flags[i] |= 1 << bit
                keep[name] = arr
                setattr(tags, name, arr.ctypes.data)
        keep["flags"] = flags
        tags.tag_flags = flags.ctypes.data
        if any(r.get("lb") is not None for r in reads):
            keep["lb"] = np.concatenate([np.asarray(r["lb"], np.uint8) for r in reads])
            rd.baq = keep["lb"].ctypes.data
        self._keep = keep                           # the host arrays must outlive the read set
        self.seq_off = keep["seq_off"]
        h = C.c_void_p()
        _lib.check(self.L.lfq_readset_create(caller.h, C.byref(rd), C.byref(tags), C.byref(h)), "lfq_readset_create")
        self.h = h
        if not hasattr(caller, "_readsets"):
            import weakref
            caller._readsets = weakref.WeakSet()
        caller._readsets.add(self)                  # SnvCaller.close() closes its read sets first

    @classmethod
    def from_arrays(cls, caller, R):
        """the same from flat arrays (no per-read Python objects): R = dict with n, pos (int32), cig_off / seq_off (int64),
        cig (uint32, BAM encoding), seq (codes 0..4), qual, mapq, rev (uint8 per read), ref (bytes), and optionally
        bi / bd (tag bytes per base), lb (tag bytes per base), flags (uint8 per read, bits 0 / 1 = has BI / BD)"""
        self = cls.__new__(cls)
        self.caller = caller
        self.L = _lib.load()
        self.n = int(R["n"])
        keep = {k: np.ascontiguousarray(R[k], dt) for k, dt in (("pos", np.int32), ("cig_off", np.int64), ("cig", np.uint32),
                                                                ("seq_off", np.int64), ("seq", np.uint8), ("qual", np.uint8),
                                                                ("mapq", np.uint8), ("rev", np.uint8))}
        keep["ref"] = bytes(R["ref"])
        rd = _lib.PileupReads()
        rd.n_reads = self.n
        rd.pos, rd.cigar_off, rd.cigar = keep["pos"].ctypes.data, keep["cig_off"].ctypes.data, keep["cig"].ctypes.data
        rd.seq_off, rd.seq, rd.qual = keep["seq_off"].ctypes.data, keep["seq"].ctypes.data, keep["qual"].ctypes.data
        rd.mapq, rd.reverse = keep["mapq"].ctypes.data, keep["rev"].ctypes.data
        rd.ref = C.cast(C.c_char_p(keep["ref"]), C.c_void_p)
        rd.ref_len = len(keep["ref"])
        if R.get("lb") is not None:
            keep["lb"] = np.ascontiguousarray(R["lb"], np.uint8)
            rd.baq = keep["lb"].ctypes.data
        tags = _lib.PileupIndelTags()
        for name in ("bi", "bd"):
            if R.get(name) is not None:
                keep[name] = np.ascontiguousarray(R[name], np.uint8)
                setattr(tags, name, keep[name].ctypes.data)
        if R.get("flags") is not None:
            keep["flags"] = np.ascontiguousarray(R["flags"], np.uint8)
            tags.tag_flags = keep["flags"].ctypes.data
        self._keep = keep
        self.seq_off = keep["seq_off"]
        h = C.c_void_p()
        _lib.check(self.L.lfq_readset_create(caller.h, C.byref(rd), C.byref(tags), C.byref(h)), "lfq_readset_create")
        self.h = h
        if not hasattr(caller, "_readsets"):
            import weakref
            caller._readsets = weakref.WeakSet()
        caller._readsets.add(self)
        return self

    def close(self):
        if getattr(self, "h", None):
            self.L.lfq_readset_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def baq(self, extended=True, idaq=False):
        _lib.check(self.L.lfq_readset_baq(self.caller.h, self.h, 1 if extended else 0, 1 if idaq else 0), "lfq_readset_baq")

    def source_qual(self, def_nm_q=-1, min_bq=6, ign=None):
        sq = np.zeros(max(self.n, 1), np.int32)
        if ign is not None:
            ign = np.ascontiguousarray(ign, np.uint8)
        _lib.check(self.L.lfq_readset_source_qual(self.caller.h, self.h, int(def_nm_q), int(min_bq),
                                                  ign.ctypes.data if ign is not None else None, sq.ctypes.data),
                   "lfq_readset_source_qual")
        return sq[: self.n]

    def fetch_tags(self, idaq=False):
        nb = max(int(self.seq_off[-1]), 1)
        lb = np.zeros(nb, np.uint8)
        ai = np.zeros(nb, np.uint8) if idaq else None
        ad = np.zeros(nb, np.uint8) if idaq else None
        fl = np.zeros(max(self.n, 1), np.uint8) if idaq else None
        p = lambda a: a.ctypes.data if a is not None else None
        _lib.check(self.L.lfq_readset_fetch_tags(self.caller.h, self.h, p(lb), p(ai), p(ad), p(fl)), "lfq_readset_fetch_tags")
        return lb, ai, ad, fl

    def pileup_snv(self, begin, end, min_plp_bq=3, sync=False):
        """lfq_readset_pileup_snv returns when the scatter pass is queued: the tracks are complete in stream order (the calls
        that take them are queued behind); sync=True waits, for code that reads the device memory itself"""
        t = _lib.Tracks()
        col_pos = np.zeros(max(end - begin, 1), np.int64)
        _lib.check(self.L.lfq_readset_pileup_snv(self.caller.h, self.h, int(begin), int(end), int(min_plp_bq), C.byref(t),
                                                 col_pos.ctypes.data), "lfq_readset_pileup_snv")
        if sync:
            self.caller.synchronize()
        return DeviceTracks(t, col_pos[: int(t.ncols)].copy())

    def pileup_indels(self, begin, end, min_plp_idq=0):
        out = C.POINTER(_lib.IndelColumnsC)()
        col_pos = np.zeros(max(end - begin, 1), np.int64)
        _lib.check(self.L.lfq_readset_pileup_indels(self.caller.h, self.h, int(begin), int(end), int(min_plp_idq),
                                                    C.byref(out), col_pos.ctypes.data), "lfq_readset_pileup_indels")
        return _indel_columns_from_c(out, col_pos, self.caller)
