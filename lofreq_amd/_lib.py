"""ctypes binding of liblofreq_amd.so (the C ABI declared in include/lofreq_amd.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C lofreq_amd/csrc``.
There is no Python or CPU fallback: if the shared object is missing, importing the compute
entry points fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LFQ_AMD_LIB") or os.path.join(_HERE, "liblofreq_amd.so")   # LFQ_AMD_LIB: another build of the same library (A/B runs)

LFQ_OK = 0
LFQ_ABI_VERSION = 6      # include/lofreq_amd.h; load() refuses a library built from another header
LFQ_ERR_CAPACITY = -4
LFQ_USE_BAQ, LFQ_USE_MQ, LFQ_USE_SQ, LFQ_USE_IDAQ = 1, 2, 4, 8
LFQ_PV_NONE, LFQ_PV_LOG, LFQ_PV_LOG_FECLAMP, LFQ_PV_UNDERFLOW = 0, 1, 2, 3
LFQ_Q_MISSING = 255


class Conf(C.Structure):
    """lfq_conf == the SNV-path fields of varcall_conf_t (snpcaller.h:38-63)."""
    _fields_ = [
        ("min_bq", C.c_int32), ("min_alt_bq", C.c_int32), ("def_alt_bq", C.c_int32),
        ("min_jq", C.c_int32), ("min_alt_jq", C.c_int32), ("def_alt_jq", C.c_int32),
        ("bonf_dynamic", C.c_int32), ("min_cov", C.c_int32),
        ("bonf_subst", C.c_int64), ("sig", C.c_float), ("flag", C.c_int32),
        ("num_snv_tests", C.c_int64), ("bonf_indel", C.c_int64), ("num_indel_tests", C.c_int64),
        ("approx_threshold_n", C.c_int32), ("pad_", C.c_int32),
    ]


class Tracks(C.Structure):
    _fields_ = [
        ("nt", C.c_void_p), ("bq", C.c_void_p), ("baq", C.c_void_p), ("mq", C.c_void_p),
        ("sq", C.c_void_p), ("col_off", C.c_void_p), ("ref_base", C.c_void_p),
        ("coverage_plp", C.c_void_p), ("num_bases", C.c_void_p), ("ncols", C.c_int64),
        ("max_col_obs", C.c_int64), ("flags", C.c_int64),
    ]


LFQ_TRACKS_NT_PACKED = 1


class BatchStats(C.Structure):
    _fields_ = [("n_tested", C.c_int64), ("n_pvals", C.c_int64), ("n_obs", C.c_int64)]


class KernelTimes(C.Structure):
    _fields_ = [("ms_count", C.c_float), ("ms_scan", C.c_float), ("ms_dp", C.c_float),
                ("ms_total", C.c_float), ("ms_dp_light", C.c_float), ("ms_dp_mid", C.c_float),
                ("ms_dp_big", C.c_float), ("n_segments", C.c_int32)]


class BaqTimes(C.Structure):
    _fields_ = [("ms_kernels", C.c_float), ("n_launches", C.c_int32), ("n_reads", C.c_int64), ("n_bases", C.c_int64)]


class DpWork(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("cells", "rows", "n_light", "n_mid", "n_big", "n_light_retry",
                                         "bytes_read_count", "bytes_written_count", "n_approx_pruned")]


class BaqReads(C.Structure):
    _fields_ = [("n_reads", C.c_int64)] + [(n, C.c_void_p) for n in (
        "pos", "cigar_off", "cigar", "seq_off", "seq", "qual", "ref")] + [("ref_len", C.c_int64)]


class PileupReads(C.Structure):
    _fields_ = [("n_reads", C.c_int64)] + [(n, C.c_void_p) for n in (
        "pos", "cigar_off", "cigar", "seq_off", "seq", "qual", "baq", "mapq", "reverse", "ref")] + [("ref_len", C.c_int64),
                                                                                              ("sq", C.c_void_p)]


class PileupIndelTags(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("bi", "bd", "ai", "ad", "tag_flags", "sq")]


class IndelSide(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "non_fw", "non_rv", "ne_off", "ne_q", "ne_mq", "ev_off", "key_off", "key_chars", "ev_fw", "ev_rv",
        "rd_off", "rd_q", "rd_aq", "rd_mq", "rd_sq")]


class IndelColumnsC(C.Structure):
    _fields_ = [("ncols", C.c_int64)] + [(n, C.c_void_p) for n in (
        "ref_base", "coverage_plp", "num_tails", "num_non_indels", "num_ins", "num_dels", "hrun")] + [
        ("side", IndelSide * 2), ("cons_indel", C.c_void_p)]


INDEL_CALL_DTYPE = np.dtype([("test", "i8"), ("bonf", "i8"), ("pvalue", np.longdouble), ("qual", "i4"),
                             ("count", "i4")], align=True)
assert INDEL_CALL_DTYPE.itemsize == 48, INDEL_CALL_DTYPE.itemsize

INDEL_RECORD_DTYPE = np.dtype([
    ("col", "i8"), ("side", "i4"), ("event", "i4"), ("qual", "i4"), ("dp", "i4"), ("sb", "i4"),
    ("ref_fw", "i4"), ("ref_rv", "i4"), ("alt_fw", "i4"), ("alt_rv", "i4"), ("hrun", "i4"), ("af", "f4"),
    ("count", "i4"), ("bonf", "i8"), ("pvalue", np.longdouble)], align=True)
assert INDEL_RECORD_DTYPE.itemsize == 80, INDEL_RECORD_DTYPE.itemsize

COL_COUNTS_DTYPE = np.dtype([
    ("n_err_probs", "i4"), ("alt_counts", "i4", 3), ("alt_raw_counts", "i4", 3), ("alt_fw", "i4", 3),
    ("ref_fw", "i4"), ("ref_rv", "i4"), ("kmax", "i4"), ("tested", "u1"), ("gated", "u1"),
    ("pad_", "u1", 2), ("median_ref_bq", "i4"), ("coverage", "i4")], align=True)
assert COL_COUNTS_DTYPE.itemsize == 64

COL_PVALS_DTYPE = np.dtype([
    ("col", "i8"), ("bonf", "i8"), ("logp", "f8", 3), ("status", "u1", 3), ("ref_base", "u1"), ("pad_", "u1", 4),
    ("counts", COL_COUNTS_DTYPE), ("dp_rows", "i4"), ("pad2_", "i4"), ("reserved_", "i8")], align=True)
assert COL_PVALS_DTYPE.itemsize == 128

SNV_RECORD_DTYPE = np.dtype([
    ("col", "i8"), ("qual", "i4"), ("dp", "i4"), ("alt_raw_count", "i4"), ("sb", "i4"),
    ("ref_fw", "i4"), ("ref_rv", "i4"), ("alt_fw", "i4"), ("alt_rv", "i4"), ("hqa", "i4"),
    ("ref", "S1"), ("alt", "S1"), ("pad_", "u1", 2), ("pvalue", np.longdouble)], align=True)
assert SNV_RECORD_DTYPE.itemsize == 64, SNV_RECORD_DTYPE.itemsize

# every symbol include/lofreq_amd.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "lfq_abi_version", "lfq_strerror", "lfq_conf_init", "lfq_create", "lfq_destroy", "lfq_synchronize",
    "lfq_snv_batch_device", "lfq_batch_finish", "lfq_call_snvs_batch", "lfq_call_snvs_submit", "lfq_call_snvs_wait", "lfq_call_snvs_collect", "lfq_set_dense_strand_counts", "lfq_set_dense_counts", "lfq_set_batch_gate", "lfq_set_private_stream", "lfq_set_indel_arrays_on_host", "lfq_finalize_pvals",
    "lfq_pvalue_from_log", "lfq_format_snv_record", "lfq_format_vcf", "lfq_snvqual_thresh", "lfq_sb_phred",
    "lfq_fisher_exact", "lfq_fdr", "lfq_bonf_corr", "lfq_holm_bonf_corr", "lfq_filter_records",
    "lfq_synth_fill_device", "lfq_synth_fill_device_layout", "lfq_last_kernel_times", "lfq_last_baq_times", "lfq_last_dp_work",
    "lfq_indel_batch_device", "lfq_call_indel_tests_batch", "lfq_call_indels_batch", "lfq_format_indel_record",
    "lfq_filter_indel_records", "lfq_baq_batch", "lfq_baq_idaq_batch", "lfq_pileup_snv_tracks",
    "lfq_source_qual_batch", "lfq_pileup_indel_columns", "lfq_pileup_skip_snv_columns",
    "lfq_uniq_detlim_batch", "lfq_uniq_binom_batch", "lfq_uniq_mtc", "lfq_binom_cdf",
    "lfq_shard_exchange_counts", "lfq_shard_rebase_bonferroni", "lfq_shard_gather_records", "lfq_shard_advance_conf",
    "lfq_set_pileup_nt_packed", "lfq_set_pileup_unsorted", "lfq_set_baq_hmm_params", "lfq_pack_nt_track", "lfq_shard_allgather", "lfq_shard_set_host_allgather", "lfq_shard_gather_start", "lfq_shard_gather_wait", "lfq_shard_shm_open", "lfq_shard_shm_unlink", "lfq_shard_shm_close", "lfq_call_snvs_collect_pvals", "lfq_device_count", "lfq_pick_device", "lfq_host_alloc", "lfq_host_free",
    "lfq_readset_create", "lfq_readset_destroy", "lfq_readset_baq", "lfq_readset_source_qual",
    "lfq_readset_pileup_snv", "lfq_readset_pileup_indels", "lfq_readset_fetch_tags",
    "lfq_filter_conf_init", "lfq_filter_conf_defaults", "lfq_filter_vars", "lfq_filter_id", "lfq_filter_string",
    "lfq_filter_header_lines", "lfq_filter_var_from_snv", "lfq_filter_var_from_indel",
]

class FilterConf(C.Structure):
    """lfq_filter_conf (filter_conf_t, lofreq_filter.c:59-112)"""
    _fields_ = [("only_snvs", C.c_int32), ("only_indels", C.c_int32), ("dp_min", C.c_int32), ("dp_max", C.c_int32),
                ("af_min", C.c_float), ("af_max", C.c_float), ("sb_thresh", C.c_int32), ("sb_mtc_type", C.c_int32),
                ("sb_alpha", C.c_double), ("sb_ntests", C.c_int64), ("sb_no_compound", C.c_int32),
                ("sb_incl_indels", C.c_int32), ("snvqual_thresh", C.c_int32), ("snvqual_mtc_type", C.c_int32),
                ("snvqual_alpha", C.c_double), ("snvqual_ntests", C.c_int64), ("indelqual_thresh", C.c_int32),
                ("indelqual_mtc_type", C.c_int32), ("indelqual_alpha", C.c_double), ("indelqual_ntests", C.c_int64)]


FILTER_VAR_DTYPE = np.dtype([("is_indel", "i4"), ("qual", "i4"), ("dp", "i4"), ("sb", "i4"), ("alt_fw", "i4"),
                             ("alt_rv", "i4"), ("af", "f4"), ("pad_", "i4")], align=True)
assert FILTER_VAR_DTYPE.itemsize == 32

_lib = None


# lfq_host_allgather_fn (include/lofreq_amd.h): int (*)(void *user, int world, int rank, const void *send, void *recv, size_t bytes)
HOST_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)


def load():
    """Load the C-ABI library; raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "lofreq_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C lofreq_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    try:
        # PyTorch-ROCm bundles its own libamdhip64; load it first so that this library binds to the
        # same HIP runtime instead of bringing a second one into the process.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.lfq_abi_version.restype = C.c_int
    if L.lfq_abi_version() != LFQ_ABI_VERSION:
        raise RuntimeError("lofreq_amd: %s has ABI version %d, this package expects %d (struct layouts differ) -- rebuild it"
                           % (LIB_PATH, L.lfq_abi_version(), LFQ_ABI_VERSION))
    L.lfq_strerror.restype = C.c_char_p
    L.lfq_strerror.argtypes = [C.c_int]
    L.lfq_conf_init.argtypes = [C.POINTER(Conf)]
    L.lfq_create.argtypes = [C.POINTER(vp), C.c_int]
    L.lfq_destroy.argtypes = [vp]
    L.lfq_destroy.restype = None
    L.lfq_synchronize.argtypes = [vp]
    L.lfq_snv_batch_device.argtypes = [vp, C.POINTER(Conf), C.POINTER(Tracks), vp, vp, C.c_int64, vp]
    L.lfq_batch_finish.argtypes = [vp, C.POINTER(BatchStats)]
    L.lfq_set_dense_strand_counts.argtypes = [vp, C.c_int]
    L.lfq_set_dense_counts.argtypes = [vp, C.c_int]
    L.lfq_set_batch_gate.argtypes = [vp, C.c_int]
    L.lfq_set_private_stream.argtypes = [vp, C.c_int]
    L.lfq_last_baq_times.argtypes = [vp, C.POINTER(BaqTimes)]
    L.lfq_set_indel_arrays_on_host.argtypes = [vp, C.c_int]
    L.lfq_set_baq_hmm_params.argtypes = [vp, C.c_float, C.c_float]
    L.lfq_call_snvs_submit.argtypes = [vp, C.POINTER(Conf), C.POINTER(Tracks), C.c_int]
    L.lfq_call_snvs_wait.argtypes = [vp]
    L.lfq_call_snvs_collect.argtypes = [vp, C.POINTER(Conf), vp, C.c_int64, C.POINTER(C.c_int64), vp, C.POINTER(BatchStats)]
    L.lfq_call_snvs_batch.argtypes = [vp, C.POINTER(Conf), C.POINTER(Tracks), C.c_int, vp, C.c_int64,
                                      C.POINTER(C.c_int64), vp, C.POINTER(BatchStats)]
    L.lfq_finalize_pvals.argtypes = [C.POINTER(Conf), vp, C.c_int64, vp, vp, vp, C.c_int64,
                                     C.POINTER(C.c_int64)]
    L.lfq_pvalue_from_log.restype = C.c_longdouble
    L.lfq_pvalue_from_log.argtypes = [C.c_double, C.c_int]
    L.lfq_format_snv_record.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int64, vp, C.c_char_p]
    L.lfq_format_vcf.restype = C.c_int64
    L.lfq_format_vcf.argtypes = [vp, C.c_int64, C.c_char_p, vp, vp, C.c_int64, vp, C.c_char_p]
    L.lfq_snvqual_thresh.argtypes = [C.c_float, C.c_int64]
    L.lfq_sb_phred.argtypes = [C.c_int] * 4
    L.lfq_fisher_exact.restype = C.c_double
    L.lfq_fisher_exact.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_double)] * 3
    L.lfq_fdr.restype = C.c_int64
    L.lfq_fdr.argtypes = [C.POINTER(C.c_double), C.c_int64, C.c_double, C.c_int64, C.POINTER(C.c_int64)]
    L.lfq_bonf_corr.argtypes = [C.POINTER(C.c_double), C.c_int64, C.c_int64]
    L.lfq_bonf_corr.restype = None
    L.lfq_holm_bonf_corr.argtypes = [C.POINTER(C.c_double), C.c_int64, C.c_double, C.c_int64]
    L.lfq_holm_bonf_corr.restype = None
    L.lfq_filter_records.argtypes = [vp, C.c_int64, C.c_int, C.c_int, vp]
    L.lfq_synth_fill_device.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, C.c_int64,
                                        vp, vp, vp, vp, vp, vp, vp]
    L.lfq_synth_fill_device_layout.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, C.c_int64,
                                               vp, vp, vp, vp, vp, vp, C.c_int, vp]
    L.lfq_last_kernel_times.argtypes = [vp, C.POINTER(KernelTimes)]
    L.lfq_last_dp_work.argtypes = [vp, C.POINTER(DpWork)]
    L.lfq_indel_batch_device.argtypes = [vp, C.POINTER(Conf), C.POINTER(Tracks), vp, vp, C.c_int64, vp]
    L.lfq_call_indel_tests_batch.argtypes = [vp, C.POINTER(Conf), C.POINTER(Tracks), C.c_int, vp, C.c_int64,
                                             C.POINTER(C.c_int64), C.POINTER(BatchStats)]
    L.lfq_call_indels_batch.argtypes = [vp, C.POINTER(Conf), C.POINTER(IndelColumnsC), vp, C.c_int64,
                                        C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.lfq_filter_indel_records.argtypes = [vp, C.c_int64, C.c_int, C.c_int, vp]
    L.lfq_baq_batch.argtypes = [vp, C.POINTER(BaqReads), C.c_int, vp]
    L.lfq_baq_idaq_batch.argtypes = [vp, C.POINTER(BaqReads), C.c_int, vp, vp, vp, vp]
    L.lfq_pileup_snv_tracks.argtypes = [vp, C.POINTER(PileupReads), C.c_int64, C.c_int64, C.c_int, C.POINTER(Tracks), vp]
    L.lfq_pileup_indel_columns.argtypes = [vp, C.POINTER(PileupReads), C.POINTER(PileupIndelTags), C.c_int64, C.c_int64,
                                           C.c_int, C.POINTER(C.POINTER(IndelColumnsC)), vp]
    L.lfq_pileup_skip_snv_columns.argtypes = [vp, vp, C.c_int64]
    L.lfq_uniq_detlim_batch.argtypes = [vp, C.POINTER(Tracks), C.c_int, vp, vp, vp]
    L.lfq_uniq_binom_batch.argtypes = [vp, C.POINTER(Tracks), C.c_int, vp, vp, vp, vp]
    L.lfq_uniq_mtc.argtypes = [vp, C.c_int64, C.c_int, C.c_double, C.c_int64, vp]
    L.lfq_shard_exchange_counts.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp]
    L.lfq_shard_rebase_bonferroni.argtypes = [vp, C.c_int64, C.c_int64]
    L.lfq_shard_gather_records.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int64, C.c_int64, vp, C.c_int64,
                                           C.POINTER(C.c_int64)]
    L.lfq_shard_advance_conf.argtypes = [C.POINTER(Conf), C.c_int64]
    L.lfq_shard_gather_start.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
    L.lfq_shard_gather_wait.argtypes = [vp, vp, C.c_int64]
    L.lfq_shard_set_host_allgather.argtypes = [HOST_ALLGATHER_FN, vp]
    L.lfq_shard_shm_open.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.lfq_binom_cdf.restype = C.c_double
    L.lfq_binom_cdf.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int)]
    L.lfq_readset_create.argtypes = [vp, C.POINTER(PileupReads), C.POINTER(PileupIndelTags), C.POINTER(vp)]
    L.lfq_readset_destroy.argtypes = [vp]
    L.lfq_readset_destroy.restype = None
    L.lfq_readset_baq.argtypes = [vp, vp, C.c_int, C.c_int]
    L.lfq_readset_source_qual.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
    L.lfq_readset_pileup_snv.argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int, C.POINTER(Tracks), vp]
    L.lfq_readset_pileup_indels.argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.POINTER(IndelColumnsC)), vp]
    L.lfq_readset_fetch_tags.argtypes = [vp, vp, vp, vp, vp, vp]
    L.lfq_source_qual_batch.argtypes = [vp, C.POINTER(BaqReads), C.c_int, C.c_int, vp, vp, vp]
    L.lfq_format_indel_record.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int64, C.c_char_p, C.c_char_p,
                                          C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_char_p]
    L.lfq_filter_conf_init.argtypes = [C.POINTER(FilterConf)]
    L.lfq_filter_conf_init.restype = None
    L.lfq_filter_conf_defaults.argtypes = [C.POINTER(FilterConf)]
    L.lfq_filter_conf_defaults.restype = None
    L.lfq_filter_vars.argtypes = [C.POINTER(FilterConf), vp, C.c_int64, vp]
    L.lfq_filter_id.argtypes = [C.POINTER(FilterConf), C.c_uint32, C.c_char_p, C.c_int]
    L.lfq_filter_string.argtypes = [C.POINTER(FilterConf), C.c_uint32, C.c_char_p, C.c_int]
    L.lfq_filter_header_lines.argtypes = [C.POINTER(FilterConf), C.c_char_p, C.c_int]
    L.lfq_filter_var_from_snv.argtypes = [vp, vp]
    L.lfq_filter_var_from_snv.restype = None
    L.lfq_filter_var_from_indel.argtypes = [vp, vp]
    L.lfq_filter_var_from_indel.restype = None
    _lib = L
    return L


def check(rc, what="lofreq_amd call"):
    if rc != LFQ_OK:
        raise RuntimeError("%s failed: %s (%d)" % (what, load().lfq_strerror(rc).decode(), rc))
