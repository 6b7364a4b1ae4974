/*
 * lofreq_amd_uniqbatch.h -- the packing core of the `lofreq uniq` binding: the pileup columns at the variants' positions
 * in (as the arrays plp_col_t holds, plp.h:73-145), one UNIQ flag or UQ value per variant out.  What
 * integration/lofreq_amd_uniq.c does after the gates of uniq_snv (lofreq_uniq.c:222-393) lives here and needs
 * include/lofreq_amd.h only -- no LoFreq header, no htslib -- so the code that runs inside `lofreq uniq` is driven
 * against the real library by tests/uniqbatch_harness.c (tests/test_gpu_uniq_binding.py) on the golden fixtures.
 *
 * The reference runs one mpileup per variant and tests inside the callback (lofreq_uniq.c:690-730).  Here the callback
 * only copies the column (it may be freed as soon as lfq_uniqbatch_add_* returns, plp.c:1440-1445); the tests of all
 * variants run as ONE lfq_uniq_detlim_batch or lfq_uniq_binom_batch call at lfq_uniqbatch_flush, which hands every
 * variant's result to the caller in the order the variants were added.
 */
#ifndef LOFREQ_AMD_UNIQBATCH_H
#define LOFREQ_AMD_UNIQBATCH_H

#include <stddef.h>

#include "lofreq_amd_colbatch.h"        /* lfq_col_nt: the observations of one nucleotide of a column */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lfq_uniqbatch lfq_uniqbatch;

/* one variant's column.  nt[5]: A, C, G, T, N as plp_col_t keeps them; coverage = what uniq_snv tests with
 * (coverage_plp, minus num_tails for an indel variant, lofreq_uniq.c:248-251; the caller has already returned for
 * coverage < 1, :252-254) */
typedef struct lfq_uniq_col {
    char ref_base;
    int coverage;
    lfq_col_nt nt[5];
} lfq_uniq_col;

/* result of one variant, in the order of the add calls:
 *   det-lim mode (use_det_lim != 0):  value = 1 if uniq_snv would add the UNIQ flag (lofreq_uniq.c:316-327), else 0
 *   binomial mode:                    value = the number of the UQ= tag (:384), or -1 where the reference adds none
 *                                     (binom() failed, :379-382) */
typedef void (*lfq_uniq_result_fn)(void *user, int value);

int lfq_uniqbatch_open(lfq_uniqbatch **out, int use_det_lim);
/* a variant whose alt count comes from the column's bases: an SNV in binomial mode (alt_base = var->alt[0], base_count,
 * lofreq_uniq.c:373), ANY variant in det-lim mode (plp_to_errprobs over the column whatever the variant is, :291-301;
 * alt_base is ignored there).  af: as uniq_snv has it after its own range check (:256-273) */
int lfq_uniqbatch_add_column(lfq_uniqbatch *b, const lfq_uniq_col *col, float af, char alt_base, void *user);
/* binomial mode, indel variant: the count comes from the column's event table on the host (lofreq_uniq.c:342-370), so
 * the test is lfq_binom_cdf at flush time and no column data is kept */
int lfq_uniqbatch_add_count(lfq_uniqbatch *b, int coverage, int alt_count, float af, void *user);
/* runs everything added so far and reports it, in add order; the batch is empty afterwards */
int lfq_uniqbatch_flush(lfq_uniqbatch *b, lfq_uniq_result_fn fn);
void lfq_uniqbatch_close(lfq_uniqbatch *b);

#ifdef __cplusplus
}
#endif
#endif
