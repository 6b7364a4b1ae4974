"""Host mirror of the reference's source-quality pre-step (`source_qual`, plp.c:427-593, computed per read in
mplp_func when `lofreq call -s` is given): a batch of reads of one contig -> their source qualities, through
`lfq_source_qual_batch`."""
import ctypes as C

import numpy as np

from . import _lib
from .baq import _OPS


def source_qual_batch(caller, reads, ref, def_nm_q=-1, min_bq=6, ign=None):
    """reads: list of dicts {pos0, cigar [(op, len)], seq (codes 0..4), qual (phred)}; ref: the contig (bytes,
    upper case); def_nm_q: -T/--def-nm-q; ign: uint8 mask over the contig (positions of the -S/--ign-vcf list).
    -> (int32 array: what source_qual returns per read, uint8 array: the byte for the packed sq track)"""
    n = len(reads)
    pos = np.asarray([r["pos0"] for r in reads] or [0], np.int32)
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
    ref = bytes(ref)
    rd = _lib.BaqReads()
    rd.n_reads = n
    rd.pos = pos.ctypes.data
    rd.cigar_off = cig_off.ctypes.data
    rd.cigar = cig.ctypes.data
    rd.seq_off = seq_off.ctypes.data
    rd.seq = seq.ctypes.data
    rd.qual = qual.ctypes.data
    rd.ref = C.cast(C.c_char_p(ref), C.c_void_p)
    rd.ref_len = len(ref)
    if ign is not None:
        ign = np.ascontiguousarray(ign, np.uint8)
        assert len(ign) == len(ref)
    sq = np.zeros(max(n, 1), np.int32)
    sqb = np.zeros(max(n, 1), np.uint8)
    _lib.check(_lib.load().lfq_source_qual_batch(caller.h, C.byref(rd), int(def_nm_q), int(min_bq),
                                                 ign.ctypes.data if ign is not None else None, sq.ctypes.data,
                                                 sqb.ctypes.data), "lfq_source_qual_batch")
    return sq[:n], sqb[:n]
