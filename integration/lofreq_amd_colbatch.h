/*
 * lofreq_amd_colbatch.h -- the packing core of the column binding: pileup columns in (as the arrays plp_col_t holds,
 * plp.h:73-145), VCF lines out.  Everything integration/lofreq_amd_shim.c does after the gates of call_vars
 * (lofreq_call.c:887-935) lives here and needs include/lofreq_amd.h only -- no LoFreq header, no htslib -- so the same
 * code that runs inside `lofreq call` is driven end to end against the real library by tests/colbatch_harness.c
 * (tests/test_gpu_shim.py: golden columns -> kernels -> VCF text in ONE process).
 *
 * Contract (what call_vars guarantees, kept): a column may be freed by the caller as soon as lfq_colbatch_add returns
 * (plp.c:1440-1445) -- everything is copied; records come out in column order, a column's indel records before its SNV
 * records (call_vars :896 before :928); conf's running Bonferroni factors and test counters end up as the per-column
 * loop leaves them (lofreq_call.c:794-801, 693-696).
 *
 * Two batches: while the kernels of one run, the caller's thread -- the only thread of `lofreq call` -- goes on filling
 * the other.  A batch is collected, and its lines are emitted, when the next one is full or at lfq_colbatch_flush.
 */
#ifndef LOFREQ_AMD_COLBATCH_H
#define LOFREQ_AMD_COLBATCH_H

#include <stddef.h>

#include "lofreq_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the observations of one nucleotide of a column: base_quals[i] / baq_quals[i] / map_quals[i] / source_quals[i]
 * (int_varray_t.data / .n, plp.h:88-91) and fw_counts[i] (:96).  n_baq / n_sq: 0 = the track is absent. */
typedef struct lfq_col_nt {
    const int *bq, *baq, *mq, *sq;
    size_t n, n_baq, n_sq;
    long fw;
} lfq_col_nt;

/* one indel event of a column: ins_event / del_event (plp.h:51-71), in uthash iteration order */
typedef struct lfq_col_event {
    const char *key;
    long fw, rv;                        /* fw_rv[2] */
    const int *q, *aq, *mq, *sq;        /* *_quals, *_aln_quals, *_map_quals, *_source_quals of the event's reads */
    size_t n, n_aq, n_sq;
} lfq_col_event;

typedef struct lfq_col_view {
    const char *target;
    int pos;                            /* 0-based, plp_col_t.pos */
    char ref_base;
    int coverage_plp, num_bases;
    int take_snvs;                      /* the caller's side of the gates: !only_indels and the consensus is no indel (:928-929) */
    int take_indels;                    /* !no_indels (:896) */
    lfq_col_nt nt[5];                   /* A, C, G, T, N */
    /* indel fields (plp.h:113-130); looked at only if take_indels and num_ins + num_dels > 0 */
    int num_tails, num_non_indels, num_ins, num_dels, hrun, has_indel_aqs;
    long non_ins_fw_rv[2], non_del_fw_rv[2];
    const int *ins_quals, *ins_map_quals, *del_quals, *del_map_quals;
    size_t n_ins_quals, n_del_quals;
    const lfq_col_event *ins_events, *del_events;
    int n_ins_events, n_del_events;
} lfq_col_view;

typedef void (*lfq_colbatch_emit_fn)(void *user, const char *vcf_line);
typedef struct lfq_colbatch lfq_colbatch;

/* the context is created on first use (lfq_pick_device: LFQ_DEVICE, LOCAL_RANK, or the first free worker slot);
 * batch_cols <= 0: the default of 2^20 columns (or LFQ_SHIM_BATCH_COLS from the environment) */
int lfq_colbatch_open(lfq_colbatch **out, lfq_colbatch_emit_fn emit, void *user, long batch_cols);
/* conf: thresholds + the running factors; bonf_subst / num_snv_tests / bonf_indel / num_indel_tests are advanced when a
 * batch is collected (SNVs) or flushed (indels: synchronous, few tests) */
int lfq_colbatch_add(lfq_colbatch *b, lfq_conf *conf, const lfq_col_view *col);
/* after the last column: queues what is left and finishes everything */
int lfq_colbatch_flush(lfq_colbatch *b, lfq_conf *conf);
/* report_var's counter of indel calls without alignment qualities (lofreq_call.c:109-111), so far */
long lfq_colbatch_indel_calls_wo_idaq(const lfq_colbatch *b);
void lfq_colbatch_close(lfq_colbatch *b);

#ifdef __cplusplus
}
#endif
#endif
