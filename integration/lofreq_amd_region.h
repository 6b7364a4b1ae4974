/*
 * lofreq_amd_region.h -- the READ-level binding: a region worker of `lofreq call` that hands the reads of a region to the
 * GPU once and gets VCF records back.
 *
 * What it replaces in the reference, per region (`lofreq call -r chr:beg-end`, or one bin of call-parallel):
 *     mplp_func's per-read work            plp.c:667-683 (BAQ / IDAQ: bam_prob_realn_core_ext), :727-735 (source quality)
 *     bam_mplp_auto + compile_plp_col      plp.c:1406-1447, 797-1288 (the pileup and the column builder)
 *     call_vars per column                 lofreq_call.c:887-935 (call_indels, call_snvs, report_var)
 * What stays with htslib on the CPU: opening the BAM, the index query, decoding records, the read-level filters of
 * mplp_func (plp.c:608-632: unmapped / secondary / QC-fail / duplicate, BED overlap).  The column callback
 * (integration/lofreq_amd_shim.c) keeps working for everything that wants plp_col_t; this path is for `lofreq call`
 * itself, whose end-to-end time was the CPU pileup + BAQ once the per-column statistics ran on the GPU.
 *
 * This file needs include/lofreq_amd.h only (no htslib, no LoFreq headers): the caller hands in the fields of a BAM
 * record as they lie in bam1_t (see INTEGRATION.md 9 for the ten lines of htslib macros that do it, and for the
 * plp.c hunk).  Records come back as VCF lines through a callback, FILTER '.', exactly what report_var writes to
 * conf->vcf_out; conf's Bonferroni factors and test counters are advanced as the per-column loop advances them, so
 * main_call's epilogue (lofreq_call.c:1506-1564) runs unchanged.
 *
 * Pipeline: lfq_region_end(k) queues region k's upload and BAQ kernels and returns; the pileups, the calls and the
 * output of region k happen inside lfq_region_end(k + 1) (or lfq_region_close), i.e. the CPU decodes the reads of
 * region k + 1 while the GPU works on region k.  Output order = region order.
 */
#ifndef LOFREQ_AMD_REGION_H
#define LOFREQ_AMD_REGION_H

#include "lofreq_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lfq_region_opts {
    int use_baq;            /* MPLP_BAQ (plp.h:44)            -- lofreq call default: on */
    int baq_extended;       /* MPLP_EXT_BAQ                   -- on */
    int use_idaq;           /* MPLP_IDAQ                      -- on with --call-indels */
    int use_sq;             /* MPLP_USE_SQ (-s)               -- off */
    int def_nm_q;           /* mplp_conf_t.def_nm_q (-T)      -- -1 */
    int call_indels;        /* --call-indels */
    int only_indels;        /* --only-indels */
    int min_mq, max_mq;     /* mplp_conf_t.min_mq / max_mq (plp.c:706-710): 0 / 255 */
    int min_plp_bq;         /* mplp_conf_t.min_plp_bq: 3 */
    int min_plp_idq;        /* mplp_conf_t.min_plp_idq: 0 */
    int no_orphan;          /* MPLP_NO_ORPHAN (plp.c:717): paired reads that are not a proper pair are dropped */
} lfq_region_opts;

void lfq_region_opts_init(lfq_region_opts *o);      /* the defaults of `lofreq call` (lofreq_call.c:1030-1066) */

typedef void (*lfq_region_emit_fn)(void *user, const char *vcf_line);
typedef struct lfq_region lfq_region;

/* ctx: the worker's context (lfq_create on lfq_pick_device()); conf: the caller's, advanced by every region */
int lfq_region_open(lfq_region **out, lfq_ctx *ctx, lfq_conf *conf, const lfq_region_opts *opts,
                    lfq_region_emit_fn emit, void *user);
/* `ref`: the contig, upper-cased (plp.c:652), valid until the NEXT lfq_region_end / lfq_region_close has returned */
int lfq_region_begin(lfq_region *r, const char *target_name, const char *ref, int64_t ref_len, int64_t beg0, int64_t end0);
/* one BAM record that passed the flag filters of plp.c:608-632, in file order (position-sorted).  The fields are
 * bam1_t's: core.pos, core.flag, core.qual, core.n_cigar + bam_get_cigar, core.l_qseq + bam_get_seq (4-bit packed,
 * two bases per byte) + bam_get_qual; bi / bd: the Z strings of the BI / BD tags (bam_aux_get(...) + 1) or NULL.
 * Applies min_mq / max_mq / no_orphan itself (plp.c:706-720).  Returns 1 if the read was taken, 0 if filtered. */
int lfq_region_add_read(lfq_region *r, int32_t pos, int flag, int mapq, int n_cigar, const uint32_t *cigar, int l_qseq,
                        const uint8_t *seq4, const uint8_t *qual, const char *bi_or_null, const char *bd_or_null);
int lfq_region_end(lfq_region *r);
/* finishes the last region; *indel_calls_wo_idaq_or_null: report_var's counter (lofreq_call.c:109-111) */
int lfq_region_close(lfq_region *r, int64_t *indel_calls_wo_idaq_or_null);

#ifdef __cplusplus
}
#endif
#endif
