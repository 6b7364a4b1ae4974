"""ctypes binding of oracle/liblofreq_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; nothing under lofreq_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblofreq_oracle.so")


def build(force=False):
    """Compile the C restatement (and, where the reference tree is mounted, oracle/_ref)."""
    src_newer = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("lofreq_oracle.c", "lofreq_oracle.h", "synth_ref.c", "orc_pileup.c")
    )
    if force or src_newer:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    if os.path.isdir("/root/reference/src/lofreq"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])


class Conf(C.Structure):
    _fields_ = [
        ("min_bq", C.c_int32), ("min_alt_bq", C.c_int32), ("def_alt_bq", C.c_int32),
        ("min_jq", C.c_int32), ("min_alt_jq", C.c_int32), ("def_alt_jq", C.c_int32),
        ("bonf_dynamic", C.c_int32), ("min_cov", C.c_int32),
        ("bonf_subst", C.c_int64), ("sig", C.c_float), ("flag", C.c_int32),
        ("raw_counts_after_minbq", C.c_int32), ("num_snv_tests", C.c_int64),
        ("bonf_indel", C.c_int64), ("num_indel_tests", C.c_int64),
        ("approx_threshold_n", C.c_int32), ("pad_", C.c_int32),
    ]


class ColResult(C.Structure):
    _fields_ = [
        ("n_err_probs", C.c_int32), ("alt_counts", C.c_int32 * 3),
        ("alt_raw_counts", C.c_int32 * 3), ("alt_base", C.c_int32 * 3),
        ("tested", C.c_int32), ("fw", C.c_int32 * 5), ("rv", C.c_int32 * 5),
        ("bonf_used", C.c_int64), ("logp", C.c_double * 3),
        ("pvalue", C.c_longdouble * 3), ("emitted", C.c_int32 * 3),
        ("qual", C.c_int32 * 3), ("dp_rows", C.c_int32), ("pad_", C.c_int32),
    ]


COL_RESULT_DTYPE = np.dtype([
    ("n_err_probs", "i4"), ("alt_counts", "i4", 3), ("alt_raw_counts", "i4", 3), ("alt_base", "i4", 3),
    ("tested", "i4"), ("fw", "i4", 5), ("rv", "i4", 5), ("bonf_used", "i8"), ("logp", "f8", 3),
    ("pvalue", np.longdouble, 3), ("emitted", "i4", 3), ("qual", "i4", 3), ("dp_rows", "i4"),
    ("pad_", "i4")], align=True)
assert COL_RESULT_DTYPE.itemsize == C.sizeof(ColResult), (COL_RESULT_DTYPE.itemsize, C.sizeof(ColResult))

LDBL_MAX = np.finfo(np.longdouble).max
LDBL_MIN = np.finfo(np.longdouble).tiny
_ldp = C.POINTER(C.c_longdouble)


_i32p, _i16p, _i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int16), C.POINTER(C.c_int64)


class IndelBatch(C.Structure):
    _fields_ = [
        ("ncols", C.c_int64), ("ref_base", C.POINTER(C.c_uint8)),
        ("coverage_plp", _i32p), ("num_tails", _i32p), ("num_non_indels", _i32p), ("num_ins", _i32p),
        ("num_dels", _i32p), ("hrun", _i32p),
        ("non_fw", _i32p * 2), ("non_rv", _i32p * 2),
        ("ne_off", _i64p * 2), ("ne_q", _i16p * 2), ("ne_mq", _i16p * 2),
        ("ev_off", _i64p * 2), ("key_off", _i64p * 2), ("key_chars", C.c_char_p * 2),
        ("ev_fw", _i32p * 2), ("ev_rv", _i32p * 2),
        ("rd_off", _i64p * 2), ("rd_q", _i16p * 2), ("rd_aq", _i16p * 2), ("rd_mq", _i16p * 2),
        ("rd_sq", _i16p * 2),
    ]


class IndelTest(C.Structure):
    _fields_ = [
        ("col", C.c_int64), ("side", C.c_int32), ("event", C.c_int32), ("n_err_probs", C.c_int32),
        ("count", C.c_int32), ("bonf_used", C.c_int64), ("logp", C.c_double), ("pvalue", C.c_longdouble),
        ("emitted", C.c_int32), ("qual", C.c_int32), ("dp", C.c_int32), ("sb", C.c_int32),
        ("ref_fw", C.c_int32), ("ref_rv", C.c_int32), ("alt_fw", C.c_int32), ("alt_rv", C.c_int32),
        ("hrun", C.c_int32), ("af", C.c_float),
    ]


INDEL_TEST_DTYPE = np.dtype([
    ("col", "i8"), ("side", "i4"), ("event", "i4"), ("n_err_probs", "i4"), ("count", "i4"), ("bonf_used", "i8"),
    ("logp", "f8"), ("pvalue", np.longdouble), ("emitted", "i4"), ("qual", "i4"), ("dp", "i4"), ("sb", "i4"),
    ("ref_fw", "i4"), ("ref_rv", "i4"), ("alt_fw", "i4"), ("alt_rv", "i4"), ("hrun", "i4"), ("af", "f4")],
    align=True)
assert INDEL_TEST_DTYPE.itemsize == C.sizeof(IndelTest), (INDEL_TEST_DTYPE.itemsize, C.sizeof(IndelTest))


class Timing(C.Structure):
    _fields_ = [("t_merge", C.c_double), ("t_sort", C.c_double), ("t_dp", C.c_double)]


class SynthSpec(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("depth", C.c_uint32), ("plant_period", C.c_uint32),
                ("err_thresh", C.c_uint64 * 64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u8p = C.POINTER(C.c_uint8)
        L.orc_conf_init.argtypes = [C.POINTER(Conf)]
        L.orc_phred_to_prob.restype = C.c_double
        L.orc_phred_to_prob.argtypes = [C.c_int]
        L.orc_prob_to_phred_p.restype = C.c_int
        L.orc_prob_to_phred_p.argtypes = [_ldp]
        L.orc_prob_to_phred_safe.restype = C.c_int
        L.orc_prob_to_phred_safe.argtypes = [C.c_double]
        L.orc_merge_quals.restype = C.c_double
        L.orc_merge_quals.argtypes = [C.c_int] * 4
        L.orc_int_median.restype = C.c_int
        L.orc_int_median.argtypes = [C.POINTER(C.c_int), C.c_int]
        L.orc_log_sum.restype = C.c_double
        L.orc_log_sum.argtypes = [C.c_double, C.c_double]
        L.orc_poissbin.restype = C.POINTER(C.c_double)
        L.orc_poissbin.argtypes = [_ldp, C.POINTER(C.c_double), C.c_int, C.c_int,
                                   C.c_longlong, C.c_double, C.POINTER(C.c_int)]
        L.orc_snpcaller.restype = C.c_int
        L.orc_snpcaller.argtypes = [_ldp, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.c_int, C.POINTER(C.c_int), C.c_longlong, C.c_double,
                                    C.POINTER(C.c_int)]
        L.orc_call_batch.restype = C.c_int
        L.orc_call_batch.argtypes = [u8p, u8p, u8p, u8p, u8p, C.POINTER(C.c_uint64), u8p,
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int64,
                                     C.POINTER(Conf), C.POINTER(ColResult), C.POINTER(Timing)]
        L.orc_fisher_exact.restype = C.c_double
        L.orc_fisher_exact.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_double)] * 3
        L.orc_sb_phred.restype = C.c_int
        L.orc_sb_phred.argtypes = [C.c_int] * 4
        L.orc_format_snv.restype = C.c_int
        L.orc_format_snv.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_long, C.c_char, C.c_char,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_char_p]
        L.orc_bonf_corr.argtypes = [C.POINTER(C.c_double), C.c_long, C.c_long]
        L.orc_holm_bonf_corr.argtypes = [C.POINTER(C.c_double), C.c_long, C.c_double, C.c_long]
        L.orc_fdr.restype = C.c_long
        L.orc_fdr.argtypes = [C.POINTER(C.c_double), C.c_long, C.c_double, C.c_long, C.POINTER(C.c_long)]
        L.orc_snvqual_thresh.restype = C.c_int
        L.orc_snvqual_thresh.argtypes = [C.c_float, C.c_longlong]
        L.orc_default_filter.restype = C.c_int
        L.orc_default_filter.argtypes = [C.POINTER(C.c_int)] * 5 + [C.c_long, C.c_int, C.c_int,
                                                                    C.POINTER(C.c_int)]
        L.orc_call_indels_batch.restype = C.c_int
        L.orc_call_indels_batch.argtypes = [C.POINTER(IndelBatch), C.POINTER(Conf), C.POINTER(IndelTest), C.c_int64,
                                            C.POINTER(C.c_int64)]
        L.orc_format_indel.restype = C.c_int
        L.orc_format_indel.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_long, C.c_char_p, C.c_char_p, C.c_int,
                                       C.c_int, C.c_float] + [C.c_int] * 6 + [C.c_char_p]
        L.orc_synth_init_spec.argtypes = [C.POINTER(SynthSpec), C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_synth_fill.argtypes = [C.POINTER(SynthSpec), C.c_int64, C.c_int64, u8p, u8p, u8p, u8p,
                                     C.POINTER(C.c_uint64), u8p]
        _lib = L
    return _lib


def default_conf(**kw):
    c = Conf()
    lib().orc_conf_init(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _u8(a):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


def poissbin(err_probs, k, bonf=1, sig=1.0):
    """-> (probvec[K] as float, pvalue as np.longdouble, rows_done)"""
    ep = np.ascontiguousarray(err_probs, dtype=np.float64)
    pv = np.zeros(1, np.longdouble)
    rows = C.c_int()
    p = lib().orc_poissbin(pv.ctypes.data_as(_ldp), ep.ctypes.data_as(C.POINTER(C.c_double)), len(ep),
                           int(k), int(bonf), float(sig), C.byref(rows))
    vec = np.array([p[i] for i in range(k + 1)])
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(C.cast(p, C.c_void_p))
    return vec, pv[0], rows.value


def snpcaller(err_probs, counts, bonf, sig):
    ep = np.ascontiguousarray(err_probs, dtype=np.float64)
    pv = np.zeros(3, np.longdouble)
    lp = (C.c_double * 3)()
    cnt = (C.c_int * 3)(*[int(x) for x in counts])
    rows = C.c_int()
    rc = lib().orc_snpcaller(pv.ctypes.data_as(_ldp), lp, ep.ctypes.data_as(C.POINTER(C.c_double)),
                             len(ep), cnt, int(bonf), float(sig), C.byref(rows))
    assert rc == 0
    return pv, [lp[i] for i in range(3)], rows.value


def tail_truth(err_probs, counts):
    """orc_tail_truth: the tail probabilities P(X >= counts[i]) by the 80-bit linear-space recurrence (ground
    truth of the tolerance story, not reference code) -> (tails np.longdouble[3], natural logs float[3])"""
    ep = np.ascontiguousarray(err_probs, dtype=np.float64)
    tails = np.zeros(3, np.longdouble)
    logs = (C.c_double * 3)()
    cnt = (C.c_int * 3)(*[int(x) for x in counts])
    L = lib()
    L.orc_tail_truth.restype = C.c_int
    L.orc_tail_truth.argtypes = [_ldp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
    rc = L.orc_tail_truth(tails.ctypes.data_as(_ldp), logs, ep.ctypes.data_as(C.POINTER(C.c_double)), len(ep), cnt)
    if rc:
        raise RuntimeError("orc_tail_truth failed (all counts zero?)")
    return tails, [logs[i] for i in range(3)]


def col_tail_truth(host, col, conf):
    """orc_col_tail_truth on column `col` of packed host tracks (dict as tests/util.py builds them)
    -> (tails np.longdouble[3], natural logs float[3], filtered alt counts[3])"""
    a, b = int(host["col_off"][col]), int(host["col_off"][col + 1])
    keep, ptrs = [], []
    for k in ("nt", "bq", "baq", "mq", "sq"):
        v = host.get(k)
        if v is None:
            ptrs.append(None)
        else:
            arr = np.ascontiguousarray(v[a:b], np.uint8)
            keep.append(arr)
            ptrs.append(arr.ctypes.data)
    tails = np.zeros(3, np.longdouble)
    logs = (C.c_double * 3)()
    cnt = (C.c_int * 3)()
    L = lib()
    L.orc_col_tail_truth.restype = C.c_int
    L.orc_col_tail_truth.argtypes = [_ldp, C.POINTER(C.c_double), C.POINTER(C.c_int)] + [C.c_void_p] * 5 + \
        [C.c_int64, C.c_char, C.POINTER(Conf)]
    rc = L.orc_col_tail_truth(tails.ctypes.data_as(_ldp), logs, cnt, ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4],
                              b - a, bytes([int(host["ref_base"][col])]), C.byref(conf))
    if rc:
        raise RuntimeError("orc_col_tail_truth failed")
    return tails, [logs[i] for i in range(3)], [cnt[i] for i in range(3)]


def prob_to_phred(p):
    """PROB_TO_PHREDQUAL on an np.longdouble without losing the 80-bit range."""
    a = np.array([p], np.longdouble)
    return lib().orc_prob_to_phred_p(a.ctypes.data_as(_ldp))


def call_batch(nt, bq, baq, mq, sq, col_off, ref_base, conf, coverage_plp=None, num_bases=None,
               timing=False):
    """Run the restated call_snvs loop; returns (structured ndarray, Timing|None). conf is mutated."""
    col_off = np.ascontiguousarray(col_off, dtype=np.uint64)
    ncols = len(col_off) - 1
    keep = []
    ptrs = []
    for a in (nt, bq, baq, mq, sq):
        if a is None:
            ptrs.append(None)
        else:
            arr, p = _u8(a)
            keep.append(arr)
            ptrs.append(p)
    rb, rbp = _u8(np.frombuffer(ref_base, dtype=np.uint8) if isinstance(ref_base, (bytes, bytearray))
                  else ref_base)
    cov_p = nb_p = None
    if coverage_plp is not None:
        cov = np.ascontiguousarray(coverage_plp, dtype=np.int32)
        cov_p = cov.ctypes.data_as(C.POINTER(C.c_int32))
    if num_bases is not None:
        nb = np.ascontiguousarray(num_bases, dtype=np.int32)
        nb_p = nb.ctypes.data_as(C.POINTER(C.c_int32))
    out = (ColResult * max(ncols, 1))()
    tm = Timing()
    rc = lib().orc_call_batch(ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4],
                              col_off.ctypes.data_as(C.POINTER(C.c_uint64)), rbp, cov_p, nb_p, ncols,
                              C.byref(conf), out, C.byref(tm))
    if rc != 0:
        raise RuntimeError("orc_call_batch failed: %d" % rc)
    res = np.frombuffer(out, dtype=COL_RESULT_DTYPE, count=max(ncols, 1))[:ncols].copy()
    return res, (tm if timing else None)


def call_indels_batch(flat, conf):
    """flat: dict of numpy arrays as produced by lofreq_amd.indel.IndelColumns.flat() (the flattened indel
    fields of plp_col_t).  Returns the structured array of performed tests; conf is mutated."""
    b = IndelBatch()
    keep = []

    def ptr(a, ct):
        a = np.ascontiguousarray(a)
        keep.append(a)
        return a.ctypes.data_as(C.POINTER(ct))

    b.ncols = flat["ncols"]
    b.ref_base = ptr(flat["ref_base"].astype(np.uint8), C.c_uint8)
    for k in ("coverage_plp", "num_tails", "num_non_indels", "num_ins", "num_dels", "hrun"):
        setattr(b, k, ptr(flat[k].astype(np.int32), C.c_int32))
    n_ev = 0
    for s in (0, 1):
        b.non_fw[s] = ptr(flat["non_fw"][s].astype(np.int32), C.c_int32)
        b.non_rv[s] = ptr(flat["non_rv"][s].astype(np.int32), C.c_int32)
        b.ne_off[s] = ptr(flat["ne_off"][s].astype(np.int64), C.c_int64)
        b.ne_q[s] = ptr(flat["ne_q"][s].astype(np.int16), C.c_int16)
        b.ne_mq[s] = ptr(flat["ne_mq"][s].astype(np.int16), C.c_int16)
        b.ev_off[s] = ptr(flat["ev_off"][s].astype(np.int64), C.c_int64)
        b.key_off[s] = ptr(flat["key_off"][s].astype(np.int64), C.c_int64)
        kb = bytes(flat["key_chars"][s]) + b"\0"
        keep.append(kb)
        b.key_chars[s] = kb
        b.ev_fw[s] = ptr(flat["ev_fw"][s].astype(np.int32), C.c_int32)
        b.ev_rv[s] = ptr(flat["ev_rv"][s].astype(np.int32), C.c_int32)
        b.rd_off[s] = ptr(flat["rd_off"][s].astype(np.int64), C.c_int64)
        for k in ("rd_q", "rd_aq", "rd_mq", "rd_sq"):
            getattr(b, k)[s] = ptr(flat[k][s].astype(np.int16), C.c_int16)
        n_ev += len(flat["rd_off"][s]) - 1
    out = (IndelTest * max(n_ev, 1))()
    n = C.c_int64(0)
    rc = lib().orc_call_indels_batch(C.byref(b), C.byref(conf), out, max(n_ev, 1), C.byref(n))
    if rc != 0:
        raise RuntimeError("orc_call_indels_batch failed: %d" % rc)
    return np.frombuffer(out, dtype=INDEL_TEST_DTYPE, count=max(n_ev, 1))[: n.value].copy()


def format_indel(chrom, pos0, ref, alt, t, filter_str=None):
    """orc_format_indel on one emitted test record"""
    L = lib()
    L.orc_format_indel.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_long, C.c_char_p, C.c_char_p, C.c_int,
                                   C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_char_p]
    buf = C.create_string_buffer(1024 + len(ref) + len(alt))
    L.orc_format_indel(buf, len(buf), chrom.encode(), int(pos0), ref.encode(), alt.encode(), int(t["qual"]),
                       int(t["dp"]), C.c_float(float(t["af"])), int(t["sb"]), int(t["ref_fw"]), int(t["ref_rv"]),
                       int(t["alt_fw"]), int(t["alt_rv"]), int(t["hrun"]),
                       None if filter_str is None else filter_str.encode())
    return buf.value.decode()


def synth_fill(seed, depth, plant_period, col_begin, ncols):
    """CPU generator of the synthetic workload (include/lofreq_synth.h)."""
    spec = SynthSpec()
    lib().orc_synth_init_spec(C.byref(spec), seed, depth, plant_period)
    n = ncols * depth
    nt = np.empty(n, np.uint8)
    bq = np.empty(n, np.uint8)
    baq = np.empty(n, np.uint8)
    mq = np.empty(n, np.uint8)
    off = np.empty(ncols + 1, np.uint64)
    ref = np.empty(ncols, np.uint8)
    u8p = C.POINTER(C.c_uint8)
    lib().orc_synth_fill(C.byref(spec), col_begin, ncols, nt.ctypes.data_as(u8p), bq.ctypes.data_as(u8p),
                         baq.ctypes.data_as(u8p), mq.ctypes.data_as(u8p),
                         off.ctypes.data_as(C.POINTER(C.c_uint64)), ref.ctypes.data_as(u8p))
    return dict(nt=nt, bq=bq, baq=baq, mq=mq, col_off=off, ref_base=ref, spec=spec)


def ref_parts():
    """The reference's own fet.c/multtest.c/utils.c objects (oracle/_ref), or None if absent."""
    p = os.path.join(_HERE, "_ref", "libref_parts.so")
    if not os.path.exists(p):
        return None
    L = C.CDLL(p)
    L.kt_fisher_exact.restype = C.c_double
    L.kt_fisher_exact.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_double)] * 3
    L.fdr.restype = C.c_long
    L.fdr.argtypes = [C.POINTER(C.c_double), C.c_long, C.c_double, C.c_long,
                      C.POINTER(C.POINTER(C.c_long))]
    L.bonf_corr.argtypes = [C.POINTER(C.c_double), C.c_long, C.c_long]
    L.holm_bonf_corr.argtypes = [C.POINTER(C.c_double), C.c_long, C.c_double, C.c_long]
    L.int_median.restype = C.c_int
    L.int_median.argtypes = [C.POINTER(C.c_int), C.c_int]
    L.dbl_cmp.restype = C.c_int
    L.dbl_cmp.argtypes = [C.c_void_p, C.c_void_p]
    if hasattr(L, "kpa_ext_glocal"):
        L.kpa_ext_glocal.restype = C.c_int
        L.kpa_ext_glocal.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(KpaPar),
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


class KpaPar(C.Structure):
    _fields_ = [("d", C.c_float), ("e", C.c_float), ("bw", C.c_int)]


def kpa_glocal(ref, query, qual, d=0.00001, e=0.4, bw=7, use_reference=False):
    """The banded profile-HMM of BAQ on one read: -> (Pr, state[int32], q[uint8]).  use_reference=True runs the
    reference's own kpa_ext_glocal object (oracle/_ref/libref_parts.so) instead of the restatement."""
    ref = np.ascontiguousarray(ref, np.uint8)
    query = np.ascontiguousarray(query, np.uint8)
    qual = np.ascontiguousarray(qual, np.uint8)
    state = np.zeros(len(query), np.int32)
    q = np.zeros(len(query), np.uint8)
    if use_reference:
        R = ref_parts()
        par = KpaPar(d, e, bw)
        pr = R.kpa_ext_glocal(ref.ctypes.data, len(ref), query.ctypes.data, len(query), qual.ctypes.data, C.byref(par),
                              state.ctypes.data, q.ctypes.data, None, None)
    else:
        L = lib()
        L.orc_kpa_glocal.restype = C.c_int
        L.orc_kpa_glocal.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float,
                                     C.c_int, C.c_void_p, C.c_void_p]
        pr = L.orc_kpa_glocal(ref.ctypes.data, len(ref), query.ctypes.data, len(query), qual.ctypes.data, d, e, bw,
                              state.ctypes.data, q.ctypes.data)
    return pr, state, q


def baq_read(pos, cigar, seq, qual, ref_bytes, extended=True):
    """orc_baq_read: cigar = list of (op_char, len); seq = codes 0..4; -> lb tag bytes (BAQ + 33) or None"""
    ops = "MIDNSHP=X"
    cg = np.asarray([(l << 4) | ops.index(o) for o, l in cigar], np.uint32)
    seq = np.ascontiguousarray(seq, np.uint8)
    qual = np.ascontiguousarray(qual, np.uint8)
    out = np.zeros(len(seq), np.uint8)
    L = lib()
    L.orc_baq_read.restype = C.c_int
    L.orc_baq_read.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int64,
                               C.c_int, C.c_void_p]
    rc = L.orc_baq_read(int(pos), cg.ctypes.data, len(cg), seq.ctypes.data, qual.ctypes.data, len(seq), ref_bytes,
                        len(ref_bytes), 1 if extended else 0, out.ctypes.data)
    return out if rc else None


def baq_idaq_read(pos, cigar, seq, qual, ref_bytes, extended=True):
    """orc_baq_idaq_read -> (lb bytes, ai bytes or None, ad bytes or None)"""
    ops = "MIDNSHP=X"
    cg = np.asarray([(l << 4) | ops.index(o) for o, l in cigar], np.uint32)
    seq = np.ascontiguousarray(seq, np.uint8)
    qual = np.ascontiguousarray(qual, np.uint8)
    out = np.zeros(len(seq), np.uint8)
    iaq = np.zeros(len(seq), np.uint8)
    daq = np.zeros(len(seq), np.uint8)
    L = lib()
    L.orc_baq_idaq_read.restype = C.c_int
    L.orc_baq_idaq_read.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int64,
                                    C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.orc_baq_idaq_read(int(pos), cg.ctypes.data, len(cg), seq.ctypes.data, qual.ctypes.data, len(seq), ref_bytes,
                             len(ref_bytes), 1 if extended else 0, out.ctypes.data, iaq.ctypes.data, daq.ctypes.data)
    return out, (iaq if rc & 2 else None), (daq if rc & 4 else None)


def source_qual(pos, cigar, seq, qual, ref_bytes, nonmatch_qual=-1, min_bq=6, ign=None):
    """orc_source_qual (plp.c:427-593): cigar = list of (op_char, len); seq = codes 0..4; ign = uint8 mask over
    the reference or None -> what source_qual() returns (mplp_func stores max(., 0) in the sq tag)"""
    ops = "MIDNSHP=X"
    cg = np.asarray([(l << 4) | ops.index(o) for o, l in cigar], np.uint32)
    seq = np.ascontiguousarray(seq, np.uint8)
    qual = np.ascontiguousarray(qual, np.uint8)
    L = lib()
    L.orc_source_qual.restype = C.c_int
    L.orc_source_qual.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int64,
                                  C.c_int, C.c_int, C.c_void_p]
    if ign is not None:
        ign = np.ascontiguousarray(ign, np.uint8)
    return L.orc_source_qual(int(pos), cg.ctypes.data, len(cg), seq.ctypes.data, qual.ctypes.data, len(seq),
                             ref_bytes, len(ref_bytes), int(nonmatch_qual), int(min_bq),
                             ign.ctypes.data if ign is not None else None)


def uniq_detlim_batch(nt, bq, baq, mq, sq, col_off, ref_base, af):
    """orc_uniq_detlim_batch (uniq_snv with --use-det-lim, lofreq_uniq.c:222-333) -> (flags uint8, pvalues longdouble)"""
    col_off = np.ascontiguousarray(col_off, dtype=np.uint64)
    ncols = len(col_off) - 1
    keep, ptrs = [], []
    for a in (nt, bq, baq, mq, sq):
        if a is None:
            ptrs.append(None)
        else:
            arr = np.ascontiguousarray(a, np.uint8)
            keep.append(arr)
            ptrs.append(arr.ctypes.data)
    rb = np.ascontiguousarray(np.frombuffer(ref_base, np.uint8) if isinstance(ref_base, (bytes, bytearray)) else ref_base,
                              np.uint8)
    af = np.ascontiguousarray(af, np.float32)
    flag = np.zeros(max(ncols, 1), np.uint8)
    pv = np.zeros(max(ncols, 1), np.longdouble)
    L = lib()
    L.orc_uniq_detlim_batch.restype = C.c_int
    L.orc_uniq_detlim_batch.argtypes = [C.c_void_p] * 7 + [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.orc_uniq_detlim_batch(ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4], col_off.ctypes.data, rb.ctypes.data, ncols,
                                 af.ctypes.data, flag.ctypes.data, pv.ctypes.data)
    if rc != 0:
        raise RuntimeError("orc_uniq_detlim_batch failed: %d" % rc)
    return flag[:ncols], pv[:ncols]


def binom_cdf(n, k, pr):
    """orc_binom_cdf -> (p, cdfbin status)"""
    L = lib()
    L.orc_binom_cdf.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.c_double]
    p = C.c_double(0.0)
    st = L.orc_binom_cdf(C.byref(p), int(n), int(k), float(pr))
    return p.value, st


def ref_binom(n, k, pr):
    """the reference's own binom() (binom.c + cdflib90 compiled unmodified into oracle/_ref) -> (p, status), or None"""
    R = ref_parts()
    if R is None or not hasattr(R, "binom"):
        return None
    R.binom.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_int, C.c_double]
    p = C.c_double(0.0)
    st = R.binom(C.byref(p), None, int(n), int(k), float(pr))
    return p.value, st


def uniq_binom_batch(nt, col_off, af, alt_bases, coverage_plp=None):
    """orc_uniq_binom_batch (uniq_snv's binomial branch, lofreq_uniq.c:335-393) -> (UQ int32, p-values float64)"""
    L = lib()
    nt = np.ascontiguousarray(nt, np.uint8)
    col_off = np.ascontiguousarray(col_off, np.uint64)
    af = np.ascontiguousarray(af, np.float32)
    alt = np.frombuffer(alt_bases.encode() if isinstance(alt_bases, str) else bytes(alt_bases), np.uint8).copy()
    n = len(col_off) - 1
    uq = np.zeros(max(n, 1), np.int32)
    pv = np.zeros(max(n, 1), np.float64)
    cov = None if coverage_plp is None else np.ascontiguousarray(coverage_plp, np.int32)
    L.orc_uniq_binom_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
    rc = L.orc_uniq_binom_batch(nt.ctypes.data, col_off.ctypes.data, cov.ctypes.data if cov is not None else None, n,
                                af.ctypes.data, alt.ctypes.data, uq.ctypes.data, pv.ctypes.data)
    if rc:
        raise RuntimeError("orc_uniq_binom_batch failed: %d" % rc)
    return uq[:n], pv[:n]


def uniq_mtc(uq, mtc_type="fdr", alpha=0.001, ntests=0):
    L = lib()
    uq = np.ascontiguousarray(uq, np.int32)
    out = np.zeros(max(len(uq), 1), np.uint8)
    L.orc_uniq_mtc.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_double, C.c_long, C.c_void_p]
    rc = L.orc_uniq_mtc(uq.ctypes.data, len(uq), {"bonf": 1, "holm": 2, "fdr": 3}[mtc_type], float(alpha), int(ntests),
                        out.ctypes.data)
    if rc:
        raise RuntimeError("orc_uniq_mtc failed")
    return out[: len(uq)].astype(bool)


# ---- the column builder (oracle/orc_pileup.c: compile_plp_col over htslib's pileup entries) -------------------------

class Reads(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("pos", C.c_void_p), ("cigar_off", C.c_void_p), ("cigar", C.c_void_p),
                ("seq_off", C.c_void_p), ("seq", C.c_void_p), ("qual", C.c_void_p), ("lb", C.c_void_p),
                ("mapq", C.c_void_p), ("reverse", C.c_void_p), ("bi", C.c_void_p), ("bd", C.c_void_p),
                ("ai", C.c_void_p), ("ad", C.c_void_p), ("tag_flags", C.c_void_p), ("sq", C.c_void_p),
                ("ref", C.c_char_p), ("ref_len", C.c_int64)]


class PlpOut(C.Structure):
    _fields_ = [("ncols", C.c_int64), ("col_pos", C.c_void_p), ("nt", C.c_void_p), ("bq", C.c_void_p),
                ("baq", C.c_void_p), ("mq", C.c_void_p), ("sq", C.c_void_p), ("col_off", C.c_void_p),
                ("coverage_plp", C.c_void_p), ("num_bases", C.c_void_p), ("cons_indel", C.c_void_p),
                ("indel", IndelBatch)]


def pack_reads(reads, ref):
    """reads: dicts {pos0, cigar [(op, len)], seq (codes 0..4), qual, mapq, reverse[, lb, bi, bd, ai, ad (tag bytes),
    sq]} sorted by pos0 -> the flat arrays orc_pileup_region takes (same layout as the product's lfq_pileup_reads)"""
    ops = "MIDNSHP=X"
    n = len(reads)
    P = {"n": n, "ref": bytes(ref)}
    P["pos"] = np.asarray([r["pos0"] for r in reads] or [0], np.int32)
    cig_off = np.zeros(n + 1, np.int64)
    seq_off = np.zeros(n + 1, np.int64)
    cig = []
    for i, r in enumerate(reads):
        cig.extend((l << 4) | ops.index(o) for o, l in r["cigar"])
        cig_off[i + 1] = len(cig)
        seq_off[i + 1] = seq_off[i] + len(r["seq"])
    P["cig_off"], P["seq_off"] = cig_off, seq_off
    P["cig"] = np.asarray(cig or [0], np.uint32)
    cat = lambda k: np.concatenate([np.asarray(r[k], np.uint8) for r in reads]) if n else np.zeros(1, np.uint8)
    P["seq"], P["qual"] = cat("seq"), cat("qual")
    P["mapq"] = np.asarray([r["mapq"] for r in reads] or [0], np.uint8)
    P["rev"] = np.asarray([1 if r["reverse"] else 0 for r in reads] or [0], np.uint8)
    nb = int(seq_off[-1])
    flags = np.zeros(max(n, 1), np.uint8)
    for bit, name in enumerate(("bi", "bd", "ai", "ad")):
        P[name] = None
        if any(r.get(name) is not None for r in reads):
            arr = np.full(max(nb, 1), 33, np.uint8)
            for i, r in enumerate(reads):
                if r.get(name) is not None:
                    arr[seq_off[i]:seq_off[i + 1]] = np.asarray(r[name], np.uint8)
                    flags[i] |= 1 << bit
            P[name] = arr
    P["flags"] = flags
    P["lb"] = cat("lb") if n and all(r.get("lb") is not None for r in reads) else None
    P["sq"] = np.asarray([(-1 if r.get("sq") is None else r["sq"]) for r in reads], np.int32) \
        if any(r.get("sq") is not None for r in reads) else None
    return P


def _reads_struct(P):
    R = Reads()
    keep = []

    def ptr(a, dt):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dt)
        keep.append(a)
        return a.ctypes.data

    R.n_reads = P["n"]
    R.pos, R.cigar_off, R.cigar = ptr(P["pos"], np.int32), ptr(P["cig_off"], np.int64), ptr(P["cig"], np.uint32)
    R.seq_off, R.seq, R.qual = ptr(P["seq_off"], np.int64), ptr(P["seq"], np.uint8), ptr(P["qual"], np.uint8)
    R.lb = ptr(P.get("lb"), np.uint8)
    R.mapq, R.reverse = ptr(P["mapq"], np.uint8), ptr(P["rev"], np.uint8)
    for k in ("bi", "bd", "ai", "ad"):
        setattr(R, k, ptr(P.get(k), np.uint8))
    R.tag_flags = ptr(P.get("flags"), np.uint8)
    R.sq = ptr(P.get("sq"), np.int32)
    R.ref, R.ref_len = P["ref"], len(P["ref"])
    keep.append(P["ref"])
    return R, keep


def set_baq_hmm_params(gap_open=1e-5, gap_ext=0.4):
    """orc_set_baq_hmm_params (this process)"""
    L = lib()
    L.orc_set_baq_hmm_params.argtypes = [C.c_float, C.c_float]
    L.orc_set_baq_hmm_params.restype = None
    L.orc_set_baq_hmm_params(gap_open, gap_ext)


def _baq_range(args):
    P, r0, r1, extended, idaq = args[:5]
    L = lib()
    if len(args) > 5 and args[5] is not None:
        set_baq_hmm_params(*args[5])
    L.orc_baq_idaq_reads.restype = C.c_int
    L.orc_baq_idaq_reads.argtypes = [C.POINTER(Reads), C.c_int64, C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 4
    R, keep = _reads_struct(P)
    nb = int(P["seq_off"][-1])
    lb = np.zeros(max(nb, 1), np.uint8)
    ai = np.zeros(max(nb, 1), np.uint8) if idaq else None
    ad = np.zeros(max(nb, 1), np.uint8) if idaq else None
    fl = np.zeros(max(P["n"], 1), np.uint8)
    rc = L.orc_baq_idaq_reads(C.byref(R), int(r0), int(r1), 1 if extended else 0, 1 if idaq else 0, lb.ctypes.data,
                              ai.ctypes.data if idaq else None, ad.ctypes.data if idaq else None, fl.ctypes.data)
    assert rc == 0
    a, b = int(P["seq_off"][r0]), int(P["seq_off"][r1])
    return lb[a:b], (ai[a:b] if idaq else None), (ad[a:b] if idaq else None), fl[r0:r1]


def baq_idaq_reads(P, extended=True, idaq=True, procs=1, hmm=None):
    """the lb (and ai / ad) tags of every read of packed reads P by orc_baq_idaq_read -- what mplp_func computes on the
    fly (plp.c:667-683) -- in `procs` processes.  Sets P["lb"], P["ai"], P["ad"] and bits 2 / 3 of P["flags"]."""
    n = P["n"]
    if procs > 1 and n >= 4096:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        cuts = [n * i // procs for i in range(procs + 1)]
        with ProcessPoolExecutor(max_workers=procs, mp_context=mp.get_context("spawn")) as ex:
            parts = list(ex.map(_baq_range, [(P, cuts[i], cuts[i + 1], extended, idaq, hmm) for i in range(procs)]))
    else:
        parts = [_baq_range((P, 0, n, extended, idaq, hmm))]
        if hmm is not None:
            set_baq_hmm_params()                    # back to the defaults in this process
    P["lb"] = np.concatenate([p[0] for p in parts])
    fl = np.concatenate([p[3] for p in parts])
    flags = np.asarray(P.get("flags") if P.get("flags") is not None else np.zeros(max(n, 1), np.uint8)).copy()
    flags[:n] &= 3
    if idaq:
        P["ai"] = np.concatenate([p[1] for p in parts])
        P["ad"] = np.concatenate([p[2] for p in parts])
        flags[:n] |= (fl & 1) << 2 | (fl & 2) << 2
    P["flags"] = flags
    return P


def pileup_region(P, begin, end, min_plp_bq=3, min_plp_idq=0, use_baq=True, use_sq=False):
    """orc_pileup_region on packed reads (pack_reads, or arrays in the same layout) ->
    dict(col_pos, host = packed SNV tracks for call_batch (incl. coverage_plp / num_bases), cons_indel,
         flat = the indel fields for call_indels_batch)"""
    L = lib()
    L.orc_pileup_region.restype = C.c_void_p
    L.orc_pileup_region.argtypes = [C.POINTER(Reads), C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_plp_region_out.restype = C.POINTER(PlpOut)
    L.orc_plp_region_out.argtypes = [C.c_void_p]
    L.orc_plp_region_free.argtypes = [C.c_void_p]
    R, keep = _reads_struct(P)
    h = L.orc_pileup_region(C.byref(R), int(begin), int(end), int(min_plp_bq), int(min_plp_idq), 1 if use_baq else 0,
                            1 if use_sq else 0)
    if not h:
        raise MemoryError("orc_pileup_region")
    try:
        o = L.orc_plp_region_out(h).contents
        nc = int(o.ncols)

        def arr(p, count, dt):
            if not p or count == 0:
                return np.zeros(0, dt)
            return np.frombuffer(C.string_at(p, count * np.dtype(dt).itemsize), dt).copy()

        col_off = arr(o.col_off, nc + 1, np.uint64)
        nobs = int(col_off[-1]) if nc else 0
        pad = nobs + 32
        ib = o.indel
        host = dict(nt=arr(o.nt, pad, np.uint8), bq=arr(o.bq, pad, np.uint8),
                    baq=arr(o.baq, pad, np.uint8) if o.baq else None, mq=arr(o.mq, pad, np.uint8),
                    sq=arr(o.sq, pad, np.uint8) if o.sq else None, col_off=col_off,
                    ref_base=np.frombuffer(C.string_at(ib.ref_base, nc), np.uint8).copy() if nc else np.zeros(0, np.uint8),
                    coverage_plp=arr(o.coverage_plp, nc, np.int32), num_bases=arr(o.num_bases, nc, np.int32))
        flat = {"ncols": nc, "ref_base": host["ref_base"]}
        for k in ("coverage_plp", "num_tails", "num_non_indels", "num_ins", "num_dels", "hrun"):
            flat[k] = np.frombuffer(C.string_at(getattr(ib, k), nc * 4), np.int32).copy() if nc else np.zeros(0, np.int32)
        for k in ("non_fw", "non_rv", "ne_off", "ne_q", "ne_mq", "ev_off", "key_off", "key_chars", "ev_fw", "ev_rv", "rd_off",
                  "rd_q", "rd_aq", "rd_mq", "rd_sq"):
            flat[k] = [None, None]
        for s in (0, 1):
            i64 = lambda p, n: np.frombuffer(C.string_at(C.cast(p, C.c_void_p), n * 8), np.int64).copy()
            i32 = lambda p, n: np.frombuffer(C.string_at(C.cast(p, C.c_void_p), n * 4), np.int32).copy() if n else np.zeros(0, np.int32)
            i16 = lambda p, n: np.frombuffer(C.string_at(C.cast(p, C.c_void_p), n * 2), np.int16).copy() if n else np.zeros(0, np.int16)
            flat["non_fw"][s], flat["non_rv"][s] = i32(ib.non_fw[s], nc), i32(ib.non_rv[s], nc)
            flat["ne_off"][s] = i64(ib.ne_off[s], nc + 1)
            nne = int(flat["ne_off"][s][-1])
            flat["ne_q"][s], flat["ne_mq"][s] = i16(ib.ne_q[s], nne), i16(ib.ne_mq[s], nne)
            flat["ev_off"][s] = i64(ib.ev_off[s], nc + 1)
            nev = int(flat["ev_off"][s][-1])
            flat["key_off"][s] = i64(ib.key_off[s], nev + 1)
            flat["key_chars"][s] = C.string_at(ib.key_chars[s], int(flat["key_off"][s][-1]))
            flat["ev_fw"][s], flat["ev_rv"][s] = i32(ib.ev_fw[s], nev), i32(ib.ev_rv[s], nev)
            flat["rd_off"][s] = i64(ib.rd_off[s], nev + 1)
            nrd = int(flat["rd_off"][s][-1])
            for k in ("rd_q", "rd_aq", "rd_mq", "rd_sq"):
                flat[k][s] = i16(getattr(ib, k)[s], nrd)
        return dict(col_pos=arr(o.col_pos, nc, np.int64), host=host, cons_indel=arr(o.cons_indel, nc, np.uint8), flat=flat)
    finally:
        L.orc_plp_region_free(h)
