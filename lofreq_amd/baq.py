"""Host mirror of the reference's BAQ pre-step (`bam_prob_realn_core_ext`, bam_md_ext.c:260-491; what
`lofreq alnqual` / `lofreq call` compute per read before the pileup): a batch of reads of one contig ->
the bytes of their `lb:Z` tags, through `lfq_baq_batch`."""
import ctypes as C

import numpy as np

from . import _lib

_OPS = "MIDNSHP=X"
_CODE = np.full(256, 4, np.uint8)
for _i, _c in enumerate("ACGTN=MRSVWYHKDB"):          # LFQ_SEQ_LETTERS: htslib's seq_nt16_str with A, C, G, T, N in front
    _CODE[ord(_c)] = _i
    _CODE[ord(_c.lower())] = _i


def encode_seq(s):
    """ASCII bases -> the library's base codes: 0..3 = A, C, G, T, 4 = N (seq_nt16_int of the BAM base) and 5..15 = the other
    letters a BAM base can be ("=MRSVWYHKDB"), which behave like N except where the reference compares or prints the letter"""
    return _CODE[np.frombuffer(s.encode() if isinstance(s, str) else s, np.uint8)]


def baq_batch(caller, reads, ref, extended=True, idaq=False):
    """reads: list of dicts {pos0, cigar [(op, len)], seq (codes 0..4), qual (phred)}; ref: the contig (bytes).
    -> list of uint8 arrays: the `lb` tag bytes (BAQ + 33) of every read; with idaq=True a list of
    (lb, ai or None, ad or None) per read (indel alignment qualities, `lfq_baq_idaq_batch`)."""
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
    out = np.zeros(max(int(seq_off[-1]), 1), np.uint8)
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
    if not idaq:
        _lib.check(_lib.load().lfq_baq_batch(caller.h, C.byref(rd), 1 if extended else 0, out.ctypes.data),
                   "lfq_baq_batch")
        return [out[seq_off[i]:seq_off[i + 1]].copy() for i in range(n)]
    ai = np.zeros_like(out)
    ad = np.zeros_like(out)
    fl = np.zeros(max(n, 1), np.uint8)
    _lib.check(_lib.load().lfq_baq_idaq_batch(caller.h, C.byref(rd), 1 if extended else 0, out.ctypes.data,
                                              ai.ctypes.data, ad.ctypes.data, fl.ctypes.data), "lfq_baq_idaq_batch")
    res = []
    for i in range(n):
        a, b = seq_off[i], seq_off[i + 1]
        res.append((out[a:b].copy(), ai[a:b].copy() if fl[i] & 1 else None, ad[a:b].copy() if fl[i] & 2 else None))
    return res
