/*
 * lofreq_amd_uniq.c -- the binding a LoFreq maintainer adds to src/lofreq/ to route `lofreq uniq` -- the second
 * plp_proc_func of the tree, uniq_snv (lofreq_uniq.c:222-393) -- through liblofreq_amd.so.  Compiled inside the LoFreq tree
 * (it needs LoFreq's own plp.h / vcf.h / log.h and therefore htslib).  In this repository it is exercised by
 * tests/test_uniq_binding.py (where the reference tree is mounted): compiled against the reference's own headers and
 * driven by a mock mpileup (tests/uniq_harness.c) that rebuilds plp_col_t columns and var_t variants from the golden
 * fixtures with the reference's own int_varray helpers (utils.c) and frees every column right after the callback.
 *
 * The reference's loop (main_uniq, lofreq_uniq.c:690-730) runs one mpileup per variant and tests inside the callback.  The
 * binding keeps the loop and the callback signature and moves the TEST out of the callback: lfq_uniq_snv only checks and
 * copies the column, lfq_uniq_flush runs all variants' tests as one batch on the GPU and writes UNIQ / UQ=<n> into the
 * variants' INFO exactly where uniq_snv would have.  What changes in main_uniq:
 *
 *   lofreq_uniq.c:690   plp_proc_func = &uniq_snv;                ->   plp_proc_func = &lfq_uniq_snv;
 *   (new, before :692)  lfq_uniq_binding ub = { uniq_conf.uni_freq, uniq_conf.use_det_lim, NULL };
 *   lofreq_uniq.c:697   uniq_conf.var = vars[i];                  ->   ub.var = vars[i];
 *   lofreq_uniq.c:716   mpileup(&mplp_conf, plp_proc_func, (void*)&uniq_conf, 1, ...)
 *                                                                 ->   mpileup(&mplp_conf, plp_proc_func, (void*)&ub, 1, ...)
 *   lofreq_uniq.c:719-721  if (thresh) apply_uniq_threshold(var, &filter);   moves out of the loop:
 *   (new, after :726)   lfq_uniq_flush(&ub);  lfq_uniq_shutdown();
 *                       if (uniq_conf.uniq_filter.thresh) for (i = 0; i < num_vars; i++) apply_uniq_threshold(vars[i], &uniq_conf.uniq_filter);
 *   src/lofreq/Makefile.am   lofreq_SOURCES += lofreq_amd_uniq.c lofreq_amd_uniqbatch.c;  lofreq_LDADD += -llofreq_amd
 *
 * (uniq_conf_t is private to lofreq_uniq.c, :97-106, hence the three-field view of it.)  Everything after the loop --
 * det-lim output (:734-741), apply_uniq_filter_mtc (:745-750), the PASS filter (:752-757) -- reads the INFO tags and is
 * unchanged.  This file is the part that needs LoFreq's headers; the packing and the calls into the library are
 * integration/lofreq_amd_uniqbatch.c, which needs include/lofreq_amd.h only.
 *
 * Behavioural contract (same observable behaviour as uniq_snv):
 *   - a column may be freed by mpileup right after the callback returns (plp.c:1440-1445): everything needed is copied;
 *   - the log lines of uniq_snv that depend on the input alone (wrong pileup, missing / out-of-range AF) are written by
 *     the callback, at the same point and with the same text;
 *   - INFO gets "UNIQ" (det-lim mode, :327) or "UQ=<n>" (binomial mode, :384-385) for exactly the variants and with
 *     exactly the values uniq_snv would have written; variants without coverage get nothing (:252-254).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "lofreq_amd_uniq.h"
#include "lofreq_amd_uniqbatch.h"  /* this repository: integration/, include/lofreq_amd.h */
#include "log.h"
#include "plp.h"
#include "vcf.h"

#include "uthash.h"
#include "utils.h"

static lfq_uniqbatch *g_ub;

static void check(int rc)
{
    if (rc != LFQ_OK) {     /* errors are fatal like everywhere else in LoFreq: no CPU fallback */
        LOG_FATAL("lofreq_amd: %s\n", rc == LFQ_ERR_NO_DEVICE ? "no usable MI355X / HIP device" : lfq_strerror(rc));
        exit(1);
    }
}

/* the drop-in plp_proc_func (plp.h:159-163) for `lofreq uniq`; confp: an lfq_uniq_binding */
void lfq_uniq_snv(const plp_col_t *p, void *confp)
{
    lfq_uniq_binding *conf = (lfq_uniq_binding *)confp;
    char *af_char = NULL;
    float af;
    int is_indel, coverage, i;

    is_indel = vcf_var_is_indel(conf->var);                                   /* lofreq_uniq.c:233 */
    if (0 != strcmp(p->target, conf->var->chrom) || p->pos != conf->var->pos) {      /* :244-248 */
        LOG_ERROR("wrong pileup for var. pileup for %s %d. var for %s %d\n",
                  p->target, p->pos + 1, conf->var->chrom, conf->var->pos + 1);
        return;
    }
    coverage = p->coverage_plp;                                               /* :250-256 */
    if (is_indel) {
        coverage -= p->num_tails;
    }
    if (1 > coverage) {
        return;
    }
    if (conf->uni_freq <= 0.0) {                                              /* :258-277 */
        if (!vcf_var_has_info_key(&af_char, conf->var, "AF")) {
            LOG_FATAL("%s\n", "Couldn't parse AF (key not found) from variant");
            exit(1);
        }
        af = strtof(af_char, (char **)NULL);
        free(af_char);
        if (af < 0.0 || af > 1.0) {
            float new_af;
            new_af = af < 0.0 ? 0.01 : 1.0;
            LOG_FATAL("Invalid (value out of bound) AF %f in variant. Resetting to %f\n", af, new_af);
            af = new_af;
        }
    } else {
        af = conf->uni_freq;
    }

    if (!g_ub) {
        check(lfq_uniqbatch_open(&g_ub, conf->use_det_lim));
    }
    if (!conf->use_det_lim && is_indel) {
        /* the count of an indel variant comes from the column's event table (:342-370), the test itself is scalar */
        int alt_count = 0;
        const int ref_len = (int)strlen(conf->var->ref), alt_len = (int)strlen(conf->var->alt);
        if (ref_len > alt_len) {
            del_event *it = find_del_sequence(&p->del_event_counts, conf->var->ref + 1);
            alt_count = it ? it->count : 0;
        } else {
            ins_event *it = find_ins_sequence(&p->ins_event_counts, conf->var->alt + 1);
            alt_count = it ? it->count : 0;
        }
        check(lfq_uniqbatch_add_count(g_ub, coverage, alt_count, af, conf->var));
        return;
    }
    {
        lfq_uniq_col c;
        memset(&c, 0, sizeof(c));
        c.ref_base = p->ref_base;
        c.coverage = coverage;
        for (i = 0; i < NUM_NT4; i++) {
            lfq_col_nt *o = &c.nt[i];
            o->bq = p->base_quals[i].data;     o->n = p->base_quals[i].n;
            o->baq = p->baq_quals[i].data;     o->n_baq = p->baq_quals[i].n;
            o->mq = p->map_quals[i].data;
            o->sq = p->source_quals[i].data;   o->n_sq = p->source_quals[i].n;
            o->fw = p->fw_counts[i];
        }
        /* binomial mode: base_count(p, var->alt[0]) (:373) = the column's bases of that nucleotide, counted on the device */
        check(lfq_uniqbatch_add_column(g_ub, &c, af, conf->var->alt[0], conf->var));   /* copies: the column may be freed now */
    }
}

static int g_det_lim;

static void result_to_var(void *user, int value)
{
    var_t *var = (var_t *)user;
    if (g_det_lim) {
        if (value) {
            vcf_var_add_to_info(var, "UNIQ");                                 /* uniq_flag, :88, :327 */
        }
    } else if (value >= 0) {
        char info_str[128];
        snprintf(info_str, 128, "%s=%d", "UQ", value);                        /* uniq_phred_tag, :90, :384-385 */
        vcf_var_add_to_info(var, info_str);
    }
}

/* call after the per-variant mpileup loop: runs the tests of all variants, writes their INFO tags */
void lfq_uniq_flush(lfq_uniq_binding *conf)
{
    if (!g_ub) return;
    g_det_lim = conf->use_det_lim;
    check(lfq_uniqbatch_flush(g_ub, result_to_var));
}

void lfq_uniq_shutdown(void)
{
    lfq_uniqbatch_close(g_ub);
    g_ub = NULL;
}
