/*
 * lofreq_amd_shim.c -- the binding a LoFreq maintainer adds to src/lofreq/ to route `lofreq call`'s SNV
 * path through liblofreq_amd.so.  Compiled inside the LoFreq tree (it needs LoFreq's own plp.h /
 * snpcaller.h / vcf.h and therefore htslib).  In this repository it is exercised by tests/test_shim.py (where the
 * reference tree is mounted): compiled against the reference's own headers, driven by a mock mpileup that rebuilds
 * plp_col_t columns from the golden fixtures with the reference's own int_varray / uthash helpers (utils.c), and
 * checked against the packed batches tests/golden_util.py builds from the same fixtures.
 *
 *   lofreq_call.c:1474     plp_proc_func = &call_vars;     ->   plp_proc_func = &lfq_call_vars;
 *   lofreq_call.c:1477     rc = mpileup(&mplp_conf, plp_proc_func, (void*)&varcall_conf, 1, &bam);
 *   (new, right after)     lfq_call_flush(&varcall_conf);   lfq_call_shutdown();
 *   src/lofreq/Makefile.am lofreq_SOURCES += lofreq_amd_shim.c lofreq_amd_colbatch.c;  lofreq_LDADD += -llofreq_amd
 *
 * This file is the part that needs LoFreq's headers: the gates of call_vars and the view of a plp_col_t as plain
 * arrays.  The batching, the calls into the library and the record text are integration/lofreq_amd_colbatch.c, which
 * needs include/lofreq_amd.h only and is driven against the real library by tests/colbatch_harness.c on a GPU.
 *
 * Behavioural contract (same observable behaviour as call_vars, lofreq_call.c:887-935):
 *   - columns may be freed by mpileup right after the callback returns (plp.c:1440-1445): everything
 *     needed is copied into the packed batch inside the callback;
 *   - VCF records reach conf->vcf_out in column order (flush order = arrival order);
 *   - conf->bonf_subst and the global num_snv_tests end up exactly as the per-column loop leaves them
 *     (lofreq_call.c:794-801), so main_call's epilogue (:1506-1564) is unchanged;
 *   - indels (call_indels, :896): the indel fields of each column are flattened into an lfq_indel_columns
 *     batch and go through lfq_call_indels_batch at the same flush; indel records of a column are printed
 *     before its SNV records, as call_vars does (:896 before :928); conf->bonf_indel, num_indel_tests and
 *     indel_calls_wo_idaq end up as the per-column loop leaves them.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "lofreq_amd_colbatch.h"   /* this repository: integration/, include/lofreq_amd.h */
#include "log.h"
#include "plp.h"
#include "snpcaller.h"
#include "vcf.h"

#include "uthash.h"
#include "utils.h"

extern long long int num_snv_tests;                       /* lofreq_call.c:84 */
extern long long int num_indel_tests;                     /* lofreq_call.c:85 */
extern long int indel_calls_wo_idaq;                      /* lofreq_call.c:88 */

static lfq_colbatch *g_cb;
static long g_wo_idaq_seen;

static void emit_line(void *user, const char *line)       /* vcf_write_var's sink: conf->vcf_out, in column order */
{
    vcf_printf(&((varcall_conf_t *)user)->vcf_out, "%s", line);
}

static void conf_to_lfq(const varcall_conf_t *c, lfq_conf *o)
{
    lfq_conf_init(o);
    o->min_bq = c->min_bq;       o->min_alt_bq = c->min_alt_bq;   o->def_alt_bq = c->def_alt_bq;
    o->min_jq = c->min_jq;       o->min_alt_jq = c->min_alt_jq;   o->def_alt_jq = c->def_alt_jq;
    o->bonf_dynamic = c->bonf_dynamic;  o->min_cov = c->min_cov;  o->bonf_subst = c->bonf_subst;
    o->sig = c->sig;             o->flag = c->flag & (LFQ_USE_BAQ | LFQ_USE_MQ | LFQ_USE_SQ | LFQ_USE_IDAQ);
    o->num_snv_tests = num_snv_tests;
    o->bonf_indel = c->bonf_indel;      o->num_indel_tests = num_indel_tests;
    o->approx_threshold_n = c->approx_threshold_n;          /* -t (lofreq_call.c:1283) */
}

/* the running factors and counters back where main_call's epilogue reads them (lofreq_call.c:1484, 1523-1534, 1562-1563) */
static void lfq_to_conf(const lfq_conf *o, varcall_conf_t *c)
{
    c->bonf_subst = o->bonf_subst;           /* lofreq_call.c:794-800 */
    num_snv_tests = o->num_snv_tests;        /* :801 */
    c->bonf_indel = o->bonf_indel;           /* :693-695 */
    num_indel_tests = o->num_indel_tests;    /* :696 */
    indel_calls_wo_idaq += lfq_colbatch_indel_calls_wo_idaq(g_cb) - g_wo_idaq_seen;   /* report_var, :109-111 */
    g_wo_idaq_seen = lfq_colbatch_indel_calls_wo_idaq(g_cb);
}

static void check(int rc)
{
    if (rc != LFQ_OK) {     /* errors are fatal like everywhere else in LoFreq: no CPU fallback */
        LOG_FATAL("lofreq_amd: %s\n", rc == LFQ_ERR_NO_DEVICE ? "no usable MI355X / HIP device" : lfq_strerror(rc));
        exit(1);
    }
}

static void nt_view(lfq_col_nt *o, const plp_col_t *p, int i)
{
    o->bq = p->base_quals[i].data;     o->n = p->base_quals[i].n;
    o->baq = p->baq_quals[i].data;     o->n_baq = p->baq_quals[i].n;
    o->mq = p->map_quals[i].data;
    o->sq = p->source_quals[i].data;   o->n_sq = p->source_quals[i].n;
    o->fw = p->fw_counts[i];
}

/* the drop-in plp_proc_func (plp.h:159-163) */
void lfq_call_vars(const plp_col_t *p, void *confp)
{
    varcall_conf_t *conf = (varcall_conf_t *)confp;
    lfq_col_view v;
    lfq_col_event *ev = NULL;
    lfq_conf lc;
    int i;

    if (p->ref_base == 'N') return;                                   /* lofreq_call.c:892 */
    if (!g_cb) {
        check(lfq_colbatch_open(&g_cb, emit_line, conf, 0));
    }
    memset(&v, 0, sizeof(v));
    v.target = p->target;  v.pos = p->pos;  v.ref_base = p->ref_base;
    v.coverage_plp = p->coverage_plp;  v.num_bases = p->num_bases;
    v.take_indels = !conf->no_indels;                                 /* :896 */
    v.take_snvs = !conf->only_indels                                  /* :928 */
                  && !(p->cons_base[0] == '+' || p->cons_base[0] == '-');     /* :929 */
    for (i = 0; i < NUM_NT4; i++) nt_view(&v.nt[i], p, i);
    if (v.take_indels && (p->num_ins || p->num_dels)) {
        ins_event *ie, *ie_tmp;
        del_event *de, *de_tmp;
        const unsigned n_ie = HASH_CNT(hh_ins, p->ins_event_counts), n_de = HASH_CNT(hh_del, p->del_event_counts);
        int k = 0;
        ev = (lfq_col_event *)calloc((size_t)n_ie + n_de + 1, sizeof(*ev));
        if (!ev) check(LFQ_ERR_NOMEM);
        v.num_tails = p->num_tails;  v.num_non_indels = p->num_non_indels;
        v.num_ins = p->num_ins;  v.num_dels = p->num_dels;  v.hrun = p->hrun;  v.has_indel_aqs = p->has_indel_aqs;
        v.non_ins_fw_rv[0] = p->non_ins_fw_rv[0];  v.non_ins_fw_rv[1] = p->non_ins_fw_rv[1];
        v.non_del_fw_rv[0] = p->non_del_fw_rv[0];  v.non_del_fw_rv[1] = p->non_del_fw_rv[1];
        v.ins_quals = p->ins_quals.data;  v.ins_map_quals = p->ins_map_quals.data;  v.n_ins_quals = p->ins_quals.n;
        v.del_quals = p->del_quals.data;  v.del_map_quals = p->del_map_quals.data;  v.n_del_quals = p->del_quals.n;
        v.ins_events = ev;
        HASH_ITER(hh_ins, p->ins_event_counts, ie, ie_tmp) {    /* uthash insertion order = reference order */
            lfq_col_event *e = &ev[k++];
            e->key = ie->key;  e->fw = ie->fw_rv[0];  e->rv = ie->fw_rv[1];
            e->q = ie->ins_quals.data;  e->n = ie->ins_quals.n;
            e->aq = ie->ins_aln_quals.data;  e->n_aq = ie->ins_aln_quals.n;
            e->mq = ie->ins_map_quals.data;
            e->sq = ie->ins_source_quals.data;  e->n_sq = ie->ins_source_quals.n;
        }
        v.n_ins_events = k;
        v.del_events = ev + k;
        HASH_ITER(hh_del, p->del_event_counts, de, de_tmp) {
            lfq_col_event *e = &ev[k++];
            e->key = de->key;  e->fw = de->fw_rv[0];  e->rv = de->fw_rv[1];
            e->q = de->del_quals.data;  e->n = de->del_quals.n;
            e->aq = de->del_aln_quals.data;  e->n_aq = de->del_aln_quals.n;
            e->mq = de->del_map_quals.data;
            e->sq = de->del_source_quals.data;  e->n_sq = de->del_source_quals.n;
        }
        v.n_del_events = k - v.n_ins_events;
    }
    conf_to_lfq(conf, &lc);
    check(lfq_colbatch_add(g_cb, &lc, &v));       /* copies what it needs: the column may be freed now (plp.c:1440-1445) */
    lfq_to_conf(&lc, conf);
    free(ev);
}

/* call after mpileup() returns: queues what is left and finishes everything */
void lfq_call_flush(varcall_conf_t *conf)
{
    lfq_conf lc;
    if (!g_cb) return;
    conf_to_lfq(conf, &lc);
    check(lfq_colbatch_flush(g_cb, &lc));
    lfq_to_conf(&lc, conf);
}

void lfq_call_shutdown(void)
{
    lfq_colbatch_close(g_cb);
    g_cb = NULL;
    g_wo_idaq_seen = 0;
}
