/*
 * lofreq_amd_shim.c -- the binding a LoFreq maintainer adds to src/lofreq/ to route `lofreq call`'s SNV
 * path through liblofreq_amd.so.  Compiled inside the LoFreq tree (it needs LoFreq's own plp.h /
 * snpcaller.h / vcf.h and therefore htslib); NOT built in this repository.
 *
 *   lofreq_call.c:1474     plp_proc_func = &call_vars;     ->   plp_proc_func = &lfq_call_vars;
 *   lofreq_call.c:1477     rc = mpileup(&mplp_conf, plp_proc_func, (void*)&varcall_conf, 1, &bam);
 *   (new, right after)     lfq_call_flush(&varcall_conf);   lfq_call_shutdown();
 *   src/lofreq/Makefile.am lofreq_LDADD += -llofreq_amd
 *
 * Behavioural contract (same observable behaviour as call_vars, lofreq_call.c:887-935):
 *   - columns may be freed by mpileup right after the callback returns (plp.c:1440-1445): everything
 *     needed is copied into the packed batch inside the callback;
 *   - VCF records reach conf->vcf_out in column order (flush order = arrival order);
 *   - conf->bonf_subst and the global num_snv_tests end up exactly as the per-column loop leaves them
 *     (lofreq_call.c:794-801), so main_call's epilogue (:1506-1564) is unchanged;
 *   - indels (call_indels, :896) are still called on the CPU, per column, as before.
 */
#include <stdlib.h>
#include <string.h>

#include "lofreq_amd.h"   /* this repository: include/lofreq_amd.h */
#include "log.h"
#include "plp.h"
#include "snpcaller.h"
#include "vcf.h"

extern long long int num_snv_tests;                       /* lofreq_call.c:84 */
extern void call_indels(const plp_col_t *p, varcall_conf_t *conf);   /* lofreq_call.c:619 */

#define LFQ_BATCH_COLS (1 << 20)        /* flush every 2^20 columns (or at the end) */

typedef struct {
    lfq_ctx *ctx;
    /* packed tracks (host), grown on demand */
    uint8_t *nt, *bq, *baq, *mq, *sq;
    uint64_t *col_off;
    uint8_t *ref_base;
    int32_t *cov, *nbases;
    int64_t ncols, nobs, cap_cols, cap_obs;
    int64_t max_depth;
    int use_sq, use_baq;
    /* per-column metadata needed to print records after the flush */
    char **target;
    int *pos;
} lfq_batch;

static lfq_batch B;

static void grow_obs(int64_t need)
{
    if (need <= B.cap_obs) return;
    while (B.cap_obs < need) B.cap_obs = B.cap_obs ? 2 * B.cap_obs : (1 << 24);
    B.nt = realloc(B.nt, B.cap_obs);  B.bq = realloc(B.bq, B.cap_obs);
    B.baq = realloc(B.baq, B.cap_obs); B.mq = realloc(B.mq, B.cap_obs);
    B.sq = realloc(B.sq, B.cap_obs);
}

static void grow_cols(int64_t need)
{
    if (need <= B.cap_cols) return;
    while (B.cap_cols < need) B.cap_cols = B.cap_cols ? 2 * B.cap_cols : (1 << 16);
    B.col_off = realloc(B.col_off, (B.cap_cols + 1) * sizeof(uint64_t));
    B.ref_base = realloc(B.ref_base, B.cap_cols);
    B.cov = realloc(B.cov, B.cap_cols * sizeof(int32_t));
    B.nbases = realloc(B.nbases, B.cap_cols * sizeof(int32_t));
    B.target = realloc(B.target, B.cap_cols * sizeof(char *));
    B.pos = realloc(B.pos, B.cap_cols * sizeof(int));
}

static void conf_to_lfq(const varcall_conf_t *c, lfq_conf *o)
{
    lfq_conf_init(o);
    o->min_bq = c->min_bq;       o->min_alt_bq = c->min_alt_bq;   o->def_alt_bq = c->def_alt_bq;
    o->min_jq = c->min_jq;       o->min_alt_jq = c->min_alt_jq;   o->def_alt_jq = c->def_alt_jq;
    o->bonf_dynamic = c->bonf_dynamic;  o->min_cov = c->min_cov;  o->bonf_subst = c->bonf_subst;
    o->sig = c->sig;             o->flag = c->flag & (LFQ_USE_BAQ | LFQ_USE_MQ | LFQ_USE_SQ);
    o->num_snv_tests = num_snv_tests;
}

/* call after mpileup() returns, and whenever the batch is full */
void lfq_call_flush(varcall_conf_t *conf)
{
    lfq_conf lc;
    lfq_tracks t;
    lfq_snv_record *rec;
    int64_t n_rec = 0, i;
    int rc;

    if (B.ncols == 0) return;
    if (!B.ctx && lfq_create(&B.ctx, 0) != LFQ_OK) {
        LOG_FATAL("%s\n", "lofreq_amd: no usable MI355X / HIP device");
        exit(1);
    }
    B.col_off[B.ncols] = (uint64_t)B.nobs;
    conf_to_lfq(conf, &lc);
    memset(&t, 0, sizeof(t));
    t.nt = B.nt; t.bq = B.bq; t.mq = B.mq;
    t.baq = B.use_baq ? B.baq : NULL;
    t.sq = B.use_sq ? B.sq : NULL;
    t.col_off = B.col_off; t.ref_base = B.ref_base;
    t.coverage_plp = B.cov; t.num_bases = B.nbases;
    t.ncols = B.ncols; t.max_col_obs = B.max_depth;

    rec = malloc(sizeof(lfq_snv_record) * (size_t)(3 * B.ncols));
    rc = lfq_call_snvs_batch(B.ctx, &lc, &t, /*tracks_on_device=*/0, rec, 3 * B.ncols, &n_rec, NULL, NULL);
    if (rc != LFQ_OK) {
        LOG_FATAL("lofreq_amd: %s\n", lfq_strerror(rc));
        exit(1);
    }
    for (i = 0; i < n_rec; i++) {           /* vcf_write_var (vcf.c:469-497), FILTER '.' like report_var */
        char line[512];
        lfq_format_snv_record(line, sizeof(line), B.target[rec[i].col], B.pos[rec[i].col], &rec[i], NULL);
        vcf_printf(&conf->vcf_out, "%s", line);
    }
    free(rec);
    conf->bonf_subst = lc.bonf_subst;        /* lofreq_call.c:794-800 */
    num_snv_tests = lc.num_snv_tests;        /* lofreq_call.c:801 */
    for (i = 0; i < B.ncols; i++) free(B.target[i]);
    B.ncols = 0; B.nobs = 0; B.max_depth = 0;
}

/* the drop-in plp_proc_func (plp.h:159-163) */
void lfq_call_vars(const plp_col_t *p, void *confp)
{
    varcall_conf_t *conf = (varcall_conf_t *)confp;
    int i;
    unsigned long j;
    int64_t depth = 0, c;

    if (p->ref_base == 'N') return;                                   /* lofreq_call.c:892 */
    if (!conf->no_indels) call_indels(p, conf);                       /* :896, unchanged, CPU */
    if (conf->only_indels) return;                                    /* :928 */
    if (p->cons_base[0] == '+' || p->cons_base[0] == '-') return;     /* :929 */
    /* the remaining gates (:930 num_bases*2 < coverage_plp, :747 min_cov, :754) run on the device */

    for (i = 0; i < NUM_NT4; i++) depth += p->base_quals[i].n;
    grow_cols(B.ncols + 1);
    grow_obs(B.nobs + depth);
    c = B.ncols;
    B.col_off[c] = (uint64_t)B.nobs;
    B.ref_base[c] = (uint8_t)p->ref_base;
    B.cov[c] = p->coverage_plp;
    B.nbases[c] = p->num_bases;
    B.target[c] = strdup(p->target);
    B.pos[c] = p->pos;
    for (i = 0; i < NUM_NT4; i++) {            /* plp_col_t keeps one int array per nucleotide (plp.h:88-91) */
        const long fw = p->fw_counts[i];       /* strand only matters as a count: forward reads first */
        for (j = 0; j < p->base_quals[i].n; j++) {
            const int64_t o = B.nobs++;
            int q;
            B.nt[o] = (uint8_t)(i | (((long)j >= fw) ? 8 : 0));
            B.bq[o] = (uint8_t)p->base_quals[i].data[j];
            q = p->baq_quals[i].n ? p->baq_quals[i].data[j] : -1;
            B.baq[o] = (uint8_t)(q < 0 ? LFQ_Q_MISSING : q);
            B.mq[o] = (uint8_t)p->map_quals[i].data[j];
            q = p->source_quals[i].n ? p->source_quals[i].data[j] : -1;
            B.sq[o] = (uint8_t)(q < 0 || q > 254 ? (q < 0 ? LFQ_Q_MISSING : 254) : q);
            if (p->baq_quals[i].n) B.use_baq = 1;
            if (p->source_quals[i].n) B.use_sq = 1;
        }
    }
    if (depth > B.max_depth) B.max_depth = depth;
    B.ncols++;
    if (B.ncols >= LFQ_BATCH_COLS) lfq_call_flush(conf);
}

void lfq_call_shutdown(void)
{
    if (B.ctx) lfq_destroy(B.ctx);
    memset(&B, 0, sizeof(B));
}
