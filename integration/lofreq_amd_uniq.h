/*
 * lofreq_amd_uniq.h -- what main_uniq sees of the `lofreq uniq` binding (integration/lofreq_amd_uniq.c).
 * Included from lofreq_uniq.c after vcf.h / plp.h.
 */
#ifndef LOFREQ_AMD_UNIQ_H
#define LOFREQ_AMD_UNIQ_H

#include "plp.h"
#include "vcf.h"

/* the three fields uniq_snv reads of uniq_conf_t (which is private to lofreq_uniq.c, :97-106) */
typedef struct lfq_uniq_binding {
    float uni_freq;         /* uniq_conf_t.uni_freq: > 0 replaces every variant's AF (-f) */
    int use_det_lim;        /* uniq_conf_t.use_det_lim */
    var_t *var;             /* uniq_conf_t.var: the variant of the current mpileup call */
} lfq_uniq_binding;

void lfq_uniq_snv(const plp_col_t *p, void *confp);     /* plp_proc_func; confp = lfq_uniq_binding * */
void lfq_uniq_flush(lfq_uniq_binding *conf);            /* after the loop over the variants: tests + INFO tags */
void lfq_uniq_shutdown(void);

#endif
