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
