"""lofreq_amd -- MI355X-native implementation of LoFreq's per-pileup-column SNV calling path.

Scope: plp_to_errprobs -> snpcaller/poissbin/pruned_calc_prob_dist ->
Bonferroni emit test, behind LoFreq's column interface.  Compute lives in hand-written HIP kernels
(lofreq_amd/csrc, gfx950) behind the C ABI of include/lofreq_amd.h; this package is the thin host
mirror of the reference interface plus region sharding across GPUs.
"""
from ._lib import (COL_COUNTS_DTYPE, COL_PVALS_DTYPE, SNV_RECORD_DTYPE, LFQ_PV_LOG, LFQ_PV_LOG_FECLAMP,
                   LFQ_PV_NONE, LFQ_PV_UNDERFLOW, LFQ_USE_BAQ, LFQ_USE_IDAQ, LFQ_USE_MQ, LFQ_USE_SQ,
                   INDEL_RECORD_DTYPE)
from .caller import (PileupBatch, SnvCaller, VarcallConf, filter_records, finalize_pvals, format_vcf,
                     format_vcf_record,
                     pvalue_from_log, snvqual_thresh, write_vcf_header, binom_cdf, uniq_mtc)
from .baq import baq_batch, encode_seq
from .pileup import DeviceTracks, ReadSet, pileup_indel_columns, pileup_snv_tracks, skip_snv_columns
from .srcq import source_qual_batch
from .indel import IndelColumns, call_indels, filter_indel_records, format_indel_record

__all__ = [
    "COL_COUNTS_DTYPE", "COL_PVALS_DTYPE", "SNV_RECORD_DTYPE", "LFQ_PV_LOG", "LFQ_PV_LOG_FECLAMP",
    "LFQ_PV_NONE", "LFQ_PV_UNDERFLOW", "LFQ_USE_BAQ", "LFQ_USE_MQ", "LFQ_USE_SQ", "PileupBatch", "SnvCaller", "VarcallConf",
    "filter_records", "finalize_pvals", "format_vcf", "format_vcf_record", "pvalue_from_log", "snvqual_thresh",
    "write_vcf_header", "LFQ_USE_IDAQ", "INDEL_RECORD_DTYPE", "IndelColumns", "call_indels", "format_indel_record", "filter_indel_records", "baq_batch", "encode_seq", "DeviceTracks", "pileup_snv_tracks",
    "source_qual_batch", "pileup_indel_columns", "skip_snv_columns", "ReadSet", "binom_cdf", "uniq_mtc",
]
