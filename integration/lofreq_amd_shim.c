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
 *   src/lofreq/Makefile.am lofreq_LDADD += -llofreq_amd
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

#include "lofreq_amd.h"   /* this repository: include/lofreq_amd.h */
#include "log.h"
#include "plp.h"
#include "snpcaller.h"
#include "vcf.h"

#include "uthash.h"
#include "utils.h"

extern long long int num_snv_tests;                       /* lofreq_call.c:84 */
extern long long int num_indel_tests;                     /* lofreq_call.c:85 */
extern long int indel_calls_wo_idaq;                      /* lofreq_call.c:88 */

#define LFQ_BATCH_COLS (1 << 20)        /* flush every 2^20 columns (or at the end) */
#define LFQ_BATCH_OBS ((int64_t)1 << 28) /* ... or when a track holds 256 Mi observations: 26 000 columns at 10 000x (1.1 GiB of
                                          * pinned host tracks per batch, two batches, one staging allocation on the device) */
#define LFQ_BATCH_INDEL_READS (1 << 28) /* ... or when the flattened indel columns hold this many reads */

/* allocation results are checked: running out of host memory is a LOG_FATAL like everywhere else in LoFreq */
static void *lfq_xrealloc(void *p, size_t n)
{
    void *q = realloc(p, n ? n : 1);
    if (!q) {
        LOG_FATAL("lofreq_amd: out of memory (%lu bytes)\n", (unsigned long)n);
        exit(1);
    }
    return q;
}
static char *lfq_xstrdup(const char *s)
{
    char *q = strdup(s);
    if (!q) {
        LOG_FATAL("%s\n", "lofreq_amd: out of memory");
        exit(1);
    }
    return q;
}

typedef struct {
    /* packed tracks (host, pinned: lfq_host_alloc), grown on demand */
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
    int64_t *seq;                       /* arrival number of the column (merge key with the indel batch) */
} lfq_batch;

/* Two batches: while the kernels of one run (lfq_call_snvs_submit returns when its copies and kernels are queued) mpileup's
 * thread -- the only thread of `lofreq call` -- goes on filling the other.  A batch is collected, and its records are
 * printed, when the next one is full or at the final flush: output order = column order. */
static lfq_batch BB[2];
static int g_cur;
#define B BB[g_cur]
static lfq_ctx *g_ctx;
static int64_t g_seq;
static struct {
    int active;                         /* a submitted batch waits for its collect */
    int which;                          /* its buffer set */
    lfq_conf lc;                        /* the conf it was submitted with */
    char **iline;                       /* formatted indel records of the same columns, with their arrival numbers */
    int64_t *iseq;
    int64_t n_iline;
} P;

/* ---- indel fields of the columns that carry indel events (lfq_indel_columns, flattened) ------------- */
typedef struct { void *p; int64_t n, cap; size_t elt; } vec;
#define VEC(T) {NULL, 0, 0, sizeof(T)}
static void *vpush(vec *v, int64_t k)
{
    if (v->n + k > v->cap) {
        while (v->n + k > v->cap) v->cap = v->cap ? 2 * v->cap : 1024;
        v->p = lfq_xrealloc(v->p, (size_t)v->cap * v->elt);
    }
    v->n += k;
    return (char *)v->p + (size_t)(v->n - k) * v->elt;
}
typedef struct {
    vec non_fw, non_rv, ne_off, ne_q, ne_mq, ev_off, key_off, key_chars, ev_fw, ev_rv, rd_off, rd_q, rd_aq, rd_mq, rd_sq;
} side_vecs;
static struct {
    vec ref_base, cov, tails, non_indels, num_ins, num_dels, hrun, seq, pos, has_aq;
    vec target;
    side_vecs sd[2];
    int64_t ncols;
    int init;
} I;

static void indel_init(void)
{
    int s;
    vec i32 = VEC(int32_t), i64 = VEC(int64_t), i16 = VEC(int16_t), ch = VEC(char), u8 = VEC(uint8_t), ptr = VEC(char *);
    I.ref_base = u8; I.cov = I.tails = I.non_indels = I.num_ins = I.num_dels = I.hrun = I.pos = I.has_aq = i32;
    I.seq = i64; I.target = ptr;
    for (s = 0; s < 2; s++) {
        side_vecs *v = &I.sd[s];
        v->non_fw = v->non_rv = v->ev_fw = v->ev_rv = i32;
        v->ne_off = v->ev_off = v->key_off = v->rd_off = i64;
        v->ne_q = v->ne_mq = v->rd_q = v->rd_aq = v->rd_mq = v->rd_sq = i16;
        v->key_chars = ch;
        *(int64_t *)vpush(&v->ne_off, 1) = 0;
        *(int64_t *)vpush(&v->ev_off, 1) = 0;
        *(int64_t *)vpush(&v->key_off, 1) = 0;
        *(int64_t *)vpush(&v->rd_off, 1) = 0;
    }
    I.ncols = 0;
    I.init = 1;
}

static void push_quals(vec *v, const int_varray_t *a)
{
    unsigned long j;
    int16_t *d = vpush(v, (int64_t)a->n);
    for (j = 0; j < a->n; j++) d[j] = (int16_t)a->data[j];
}

static void push_event(side_vecs *v, const char *key, const long fw_rv[2], const int_varray_t *q,
                       const int_varray_t *aq, const int_varray_t *mq, const int_varray_t *sq)
{
    unsigned long j;
    size_t kl = strlen(key);
    int16_t *d;
    memcpy(vpush(&v->key_chars, (int64_t)kl), key, kl);
    *(int64_t *)vpush(&v->key_off, 1) = v->key_chars.n;
    *(int32_t *)vpush(&v->ev_fw, 1) = (int32_t)fw_rv[0];
    *(int32_t *)vpush(&v->ev_rv, 1) = (int32_t)fw_rv[1];
    push_quals(&v->rd_q, q);
    push_quals(&v->rd_mq, mq);
    d = vpush(&v->rd_aq, (int64_t)q->n);                    /* -1 where the BAM carried no ai/ad tag */
    for (j = 0; j < q->n; j++) d[j] = (int16_t)(j < aq->n ? aq->data[j] : -1);
    d = vpush(&v->rd_sq, (int64_t)q->n);
    for (j = 0; j < q->n; j++) d[j] = (int16_t)(j < sq->n ? sq->data[j] : -1);
    *(int64_t *)vpush(&v->rd_off, 1) = v->rd_q.n;
}

/* copy the indel fields of one column (plp.h:113-130) */
static void indel_add_column(const plp_col_t *p, int64_t seq)
{
    ins_event *ie, *ie_tmp;
    del_event *de, *de_tmp;
    if (!I.init) indel_init();
    if (p->num_ins == 0 && p->num_dels == 0) return;        /* no event, no test (lofreq_call.c:684, :706) */
    *(uint8_t *)vpush(&I.ref_base, 1) = (uint8_t)p->ref_base;
    *(int32_t *)vpush(&I.cov, 1) = p->coverage_plp;
    *(int32_t *)vpush(&I.tails, 1) = p->num_tails;
    *(int32_t *)vpush(&I.non_indels, 1) = p->num_non_indels;
    *(int32_t *)vpush(&I.num_ins, 1) = p->num_ins;
    *(int32_t *)vpush(&I.num_dels, 1) = p->num_dels;
    *(int32_t *)vpush(&I.hrun, 1) = p->hrun;
    *(int32_t *)vpush(&I.pos, 1) = p->pos;
    *(int32_t *)vpush(&I.has_aq, 1) = p->has_indel_aqs;
    *(int64_t *)vpush(&I.seq, 1) = seq;
    *(char **)vpush(&I.target, 1) = lfq_xstrdup(p->target);
    *(int32_t *)vpush(&I.sd[0].non_fw, 1) = (int32_t)p->non_ins_fw_rv[0];
    *(int32_t *)vpush(&I.sd[0].non_rv, 1) = (int32_t)p->non_ins_fw_rv[1];
    *(int32_t *)vpush(&I.sd[1].non_fw, 1) = (int32_t)p->non_del_fw_rv[0];
    *(int32_t *)vpush(&I.sd[1].non_rv, 1) = (int32_t)p->non_del_fw_rv[1];
    push_quals(&I.sd[0].ne_q, &p->ins_quals);
    push_quals(&I.sd[0].ne_mq, &p->ins_map_quals);
    push_quals(&I.sd[1].ne_q, &p->del_quals);
    push_quals(&I.sd[1].ne_mq, &p->del_map_quals);
    *(int64_t *)vpush(&I.sd[0].ne_off, 1) = I.sd[0].ne_q.n;
    *(int64_t *)vpush(&I.sd[1].ne_off, 1) = I.sd[1].ne_q.n;
    HASH_ITER(hh_ins, p->ins_event_counts, ie, ie_tmp) {    /* uthash insertion order = reference order */
        push_event(&I.sd[0], ie->key, ie->fw_rv, &ie->ins_quals, &ie->ins_aln_quals, &ie->ins_map_quals,
                   &ie->ins_source_quals);
    }
    HASH_ITER(hh_del, p->del_event_counts, de, de_tmp) {
        push_event(&I.sd[1], de->key, de->fw_rv, &de->del_quals, &de->del_aln_quals, &de->del_map_quals,
                   &de->del_source_quals);
    }
    *(int64_t *)vpush(&I.sd[0].ev_off, 1) = I.sd[0].ev_fw.n;
    *(int64_t *)vpush(&I.sd[1].ev_off, 1) = I.sd[1].ev_fw.n;
    I.ncols++;
}

/* run the flattened indel columns; returns malloc'ed records */
static lfq_indel_record *indel_flush(varcall_conf_t *conf, lfq_conf *lc, int64_t *n_rec)
{
    lfq_indel_columns c;
    lfq_indel_record *rec;
    int64_t nev, ntests = 0;
    int s, rc;
    *n_rec = 0;
    if (!I.init || I.ncols == 0) return NULL;
    memset(&c, 0, sizeof(c));
    c.ncols = I.ncols;
    c.ref_base = I.ref_base.p;  c.coverage_plp = I.cov.p;  c.num_tails = I.tails.p;
    c.num_non_indels = I.non_indels.p;  c.num_ins = I.num_ins.p;  c.num_dels = I.num_dels.p;  c.hrun = I.hrun.p;
    for (s = 0; s < 2; s++) {
        side_vecs *v = &I.sd[s];
        lfq_indel_side *o = &c.side[s];
        o->non_fw = v->non_fw.p;  o->non_rv = v->non_rv.p;  o->ne_off = v->ne_off.p;  o->ne_q = v->ne_q.p;
        o->ne_mq = v->ne_mq.p;  o->ev_off = v->ev_off.p;  o->key_off = v->key_off.p;  o->key_chars = v->key_chars.p;
        o->ev_fw = v->ev_fw.p;  o->ev_rv = v->ev_rv.p;  o->rd_off = v->rd_off.p;  o->rd_q = v->rd_q.p;
        o->rd_aq = v->rd_aq.p;  o->rd_mq = v->rd_mq.p;  o->rd_sq = v->rd_sq.p;
    }
    nev = I.sd[0].ev_fw.n + I.sd[1].ev_fw.n;
    rec = lfq_xrealloc(NULL, sizeof(lfq_indel_record) * (size_t)(nev + 1));
    rc = lfq_call_indels_batch(g_ctx, lc, &c, rec, nev, n_rec, &ntests);
    if (rc != LFQ_OK) {
        LOG_FATAL("lofreq_amd: %s\n", lfq_strerror(rc));
        exit(1);
    }
    conf->bonf_indel = lc->bonf_indel;       /* lofreq_call.c:693-695 */
    num_indel_tests = lc->num_indel_tests;   /* :696 */
    return rec;
}

static char *indel_line(const lfq_indel_record *r)
{
    const side_vecs *v = &I.sd[r->side];
    const int64_t *koff = v->key_off.p;
    const int64_t kl = koff[r->event + 1] - koff[r->event];
    const char rb = (char)((uint8_t *)I.ref_base.p)[r->col];
    char *ref = lfq_xrealloc(NULL, (size_t)kl + 2), *alt = lfq_xrealloc(NULL, (size_t)kl + 2);
    char *line = lfq_xrealloc(NULL, (size_t)kl * 2 + 1024);
    ref[0] = alt[0] = rb;                                    /* ins_to_str / del_to_str (lofreq_call.c:255-303) */
    memcpy((r->side == 0 ? alt : ref) + 1, (const char *)v->key_chars.p + koff[r->event], (size_t)kl);
    (r->side == 0 ? alt : ref)[kl + 1] = 0;
    (r->side == 0 ? ref : alt)[1] = 0;
    lfq_format_indel_record(line, (int)(kl * 2 + 1024), ((char **)I.target.p)[r->col], ((int32_t *)I.pos.p)[r->col],
                            ref, alt, r->qual, r->dp, r->af, r->sb, r->ref_fw, r->ref_rv, r->alt_fw, r->alt_rv,
                            r->hrun, NULL);
    if (!((int32_t *)I.has_aq.p)[r->col]) indel_calls_wo_idaq += 1;   /* report_var, lofreq_call.c:109-111 */
    free(ref); free(alt);
    return line;
}

static void indel_reset(void)
{
    int64_t i;
    int s;
    for (i = 0; i < I.ncols; i++) free(((char **)I.target.p)[i]);
    I.ref_base.n = I.cov.n = I.tails.n = I.non_indels.n = I.num_ins.n = I.num_dels.n = I.hrun.n = 0;
    I.seq.n = I.pos.n = I.has_aq.n = I.target.n = 0;
    for (s = 0; s < 2; s++) {
        side_vecs *v = &I.sd[s];
        v->non_fw.n = v->non_rv.n = v->ne_q.n = v->ne_mq.n = v->key_chars.n = v->ev_fw.n = v->ev_rv.n = 0;
        v->rd_q.n = v->rd_aq.n = v->rd_mq.n = v->rd_sq.n = 0;
        v->ne_off.n = v->ev_off.n = v->key_off.n = v->rd_off.n = 1;     /* keep the leading 0 */
    }
    I.ncols = 0;
}

/* the five observation tracks live in pinned memory: their upload is then a DMA that lfq_call_snvs_submit only queues */
static uint8_t *pinned_grow(uint8_t *p, int64_t used, int64_t cap)
{
    uint8_t *q = (uint8_t *)lfq_host_alloc((size_t)cap);
    if (!q) {
        LOG_FATAL("lofreq_amd: no pinned host memory (%lu bytes): is there a HIP device?\n", (unsigned long)cap);
        exit(1);
    }
    if (p && used > 0) memcpy(q, p, (size_t)used);
    lfq_host_free(p);
    return q;
}

static void grow_obs(int64_t need)
{
    const int64_t used = B.nobs;
    if (need <= B.cap_obs) return;
    while (B.cap_obs < need) B.cap_obs = B.cap_obs ? 2 * B.cap_obs : (1 << 24);
    B.nt = pinned_grow(B.nt, (used + 1) / 2 + 4, B.cap_obs);  B.bq = pinned_grow(B.bq, used, B.cap_obs);
    B.baq = pinned_grow(B.baq, used, B.cap_obs); B.mq = pinned_grow(B.mq, used, B.cap_obs);
    B.sq = pinned_grow(B.sq, used, B.cap_obs);
}

static void grow_cols(int64_t need)
{
    if (need <= B.cap_cols) return;
    while (B.cap_cols < need) B.cap_cols = B.cap_cols ? 2 * B.cap_cols : (1 << 16);
    B.col_off = lfq_xrealloc(B.col_off, (B.cap_cols + 1) * sizeof(uint64_t));
    B.ref_base = lfq_xrealloc(B.ref_base, B.cap_cols);
    B.cov = lfq_xrealloc(B.cov, B.cap_cols * sizeof(int32_t));
    B.nbases = lfq_xrealloc(B.nbases, B.cap_cols * sizeof(int32_t));
    B.target = lfq_xrealloc(B.target, B.cap_cols * sizeof(char *));
    B.pos = lfq_xrealloc(B.pos, B.cap_cols * sizeof(int));
    B.seq = lfq_xrealloc(B.seq, B.cap_cols * sizeof(int64_t));
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

static void ensure_ctx(void)
{
    if (!g_ctx) {
        /* one `lofreq call -r <bin>` per worker of the parallel wrapper (lofreq2_call_pparallel.py:640-667): each
         * process takes a GPU of its own -- LFQ_DEVICE, LOCAL_RANK, or the first free worker slot of the node */
        const int dev = lfq_pick_device(0, NULL);
        if (lfq_abi_version() != LFQ_ABI_VERSION) {     /* struct layouts (lfq_conf, lfq_dp_work) belong to the version */
            LOG_FATAL("lofreq_amd: library ABI %d, shim compiled against %d\n", lfq_abi_version(), LFQ_ABI_VERSION);
            exit(1);
        }
        if (dev < 0 || lfq_create(&g_ctx, dev) != LFQ_OK) {
            LOG_FATAL("%s\n", "lofreq_amd: no usable MI355X / HIP device");
            exit(1);
        }
    }
}

/* wait for the batch submitted at the previous flush, finish it on the host, print its columns' records */
static void collect_pending(varcall_conf_t *conf)
{
    lfq_batch *b = &BB[P.which];
    lfq_snv_record *rec;
    int64_t n_rec = 0, i, k;
    int rc;
    if (!P.active) return;
    rec = lfq_xrealloc(NULL, sizeof(lfq_snv_record) * (size_t)(3 * b->ncols + 1));
    rc = b->ncols ? lfq_call_snvs_collect(g_ctx, &P.lc, rec, 3 * b->ncols, &n_rec, NULL, NULL) : LFQ_OK;
    if (rc != LFQ_OK) {
        LOG_FATAL("lofreq_amd: %s\n", lfq_strerror(rc));
        exit(1);
    }
    if (b->ncols) {
        conf->bonf_subst = P.lc.bonf_subst;      /* lofreq_call.c:794-800 */
        num_snv_tests = P.lc.num_snv_tests;      /* lofreq_call.c:801 */
    }
    /* merge by arrival number; a column's indel records precede its SNV records (call_vars :896 / :928) */
    for (i = 0, k = 0; i < n_rec || k < P.n_iline;) {
        const int64_t s_snv = i < n_rec ? b->seq[rec[i].col] : INT64_MAX;
        const int64_t s_ind = k < P.n_iline ? P.iseq[k] : INT64_MAX;
        if (s_ind <= s_snv) {
            vcf_printf(&conf->vcf_out, "%s", P.iline[k]);
            free(P.iline[k++]);
        } else {                            /* vcf_write_var (vcf.c:469-497), FILTER '.' like report_var */
            char line[512];
            lfq_format_snv_record(line, sizeof(line), b->target[rec[i].col], b->pos[rec[i].col], &rec[i], NULL);
            vcf_printf(&conf->vcf_out, "%s", line);
            i++;
        }
    }
    free(rec);
    free(P.iline); free(P.iseq);
    P.iline = NULL; P.iseq = NULL; P.n_iline = 0;
    for (i = 0; i < b->ncols; i++) free(b->target[i]);
    b->ncols = 0; b->nobs = 0; b->max_depth = 0;
    P.active = 0;
}

/* the batch is full: finish the previous one (its kernels ran while this one was filled), queue this one, go on */
static void flush_async(varcall_conf_t *conf)
{
    lfq_conf lc;
    lfq_tracks t;
    lfq_indel_record *irec;
    int64_t n_irec = 0, k;
    int rc;

    if (B.ncols == 0 && (!I.init || I.ncols == 0)) return;
    ensure_ctx();
    collect_pending(conf);                   /* conf now carries the running factors up to this batch's first column */
    conf_to_lfq(conf, &lc);
    irec = indel_flush(conf, &lc, &n_irec);  /* call_indels of this batch's columns: synchronous, few tests */
    P.iline = lfq_xrealloc(NULL, sizeof(char *) * (size_t)(n_irec + 1));
    P.iseq = lfq_xrealloc(NULL, sizeof(int64_t) * (size_t)(n_irec + 1));
    for (k = 0; k < n_irec; k++) {
        P.iline[k] = indel_line(&irec[k]);
        P.iseq[k] = ((int64_t *)I.seq.p)[irec[k].col];
    }
    P.n_iline = n_irec;
    free(irec);
    if (I.init) indel_reset();
    P.which = g_cur;
    P.active = 1;
    if (B.ncols > 0) {
        B.col_off[B.ncols] = (uint64_t)B.nobs;
        memset(&t, 0, sizeof(t));
        t.nt = B.nt; t.bq = B.bq; t.mq = B.mq;
        t.baq = B.use_baq ? B.baq : NULL;
        t.sq = B.use_sq ? B.sq : NULL;
        t.col_off = B.col_off; t.ref_base = B.ref_base;
        t.coverage_plp = B.cov; t.num_bases = B.nbases;
        t.ncols = B.ncols; t.max_col_obs = B.max_depth;
        t.flags = LFQ_TRACKS_NT_PACKED;      /* half the nt bytes over PCIe, the 1.5-bytes-per-observation count kernel */
        P.lc = lc;
        rc = lfq_call_snvs_submit(g_ctx, &P.lc, &t, /*tracks_on_device=*/0);    /* copies + kernels queued; returns */
        if (rc != LFQ_OK) {
            LOG_FATAL("lofreq_amd: %s\n", lfq_strerror(rc));
            exit(1);
        }
    }
    g_cur ^= 1;                              /* the other buffer set was collected above: it is empty */
}

/* call after mpileup() returns: queues what is left and finishes everything */
void lfq_call_flush(varcall_conf_t *conf)
{
    flush_async(conf);
    if (P.active) {
        ensure_ctx();
        collect_pending(conf);
    }
}

static int64_t batch_cols(void)         /* LFQ_BATCH_COLS, or LFQ_SHIM_BATCH_COLS from the environment (tuning, tests) */
{
    static int64_t n;
    if (!n) {
        const char *e = getenv("LFQ_SHIM_BATCH_COLS");
        n = (e && atoll(e) > 0) ? atoll(e) : LFQ_BATCH_COLS;
    }
    return n;
}

/* the drop-in plp_proc_func (plp.h:159-163) */
void lfq_call_vars(const plp_col_t *p, void *confp)
{
    varcall_conf_t *conf = (varcall_conf_t *)confp;
    int i;
    unsigned long j;
    int64_t depth = 0, c;

    if (p->ref_base == 'N') return;                                   /* lofreq_call.c:892 */
    g_seq++;
    if (!conf->no_indels) indel_add_column(p, g_seq);                 /* :896 */
    if (conf->only_indels) goto maybe_flush;                          /* :928 */
    if (p->cons_base[0] == '+' || p->cons_base[0] == '-') goto maybe_flush;   /* :929 */
    /* the remaining gates (:930 num_bases*2 < coverage_plp, :747 min_cov, :754) run on the device */

    for (i = 0; i < NUM_NT4; i++) depth += p->base_quals[i].n;
    grow_cols(B.ncols + 1);
    grow_obs(B.nobs + depth);
    c = B.ncols;
    B.col_off[c] = (uint64_t)B.nobs;
    B.ref_base[c] = (uint8_t)p->ref_base;
    B.cov[c] = p->coverage_plp;
    B.nbases[c] = p->num_bases;
    B.target[c] = lfq_xstrdup(p->target);
    B.pos[c] = p->pos;
    B.seq[c] = g_seq;
    for (i = 0; i < NUM_NT4; i++) {            /* plp_col_t keeps one int array per nucleotide (plp.h:88-91) */
        const long fw = p->fw_counts[i];       /* strand only matters as a count: forward reads first */
        for (j = 0; j < p->base_quals[i].n; j++) {
            const int64_t o = B.nobs++;
            int q;
            /* LFQ_TRACKS_NT_PACKED: observation o sits in byte (o >> 3) * 4 + (o & 3), low nibble for o & 7 < 4 */
            {
                uint8_t *b = &B.nt[((o >> 3) << 2) + (o & 3)];
                const uint8_t v = (uint8_t)(i | (((long)j >= fw) ? 8 : 0));
                *b = (o & 4) ? (uint8_t)((*b & 0x0F) | (v << 4)) : v;
            }
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
maybe_flush:
    if (B.ncols >= batch_cols() || B.nobs >= LFQ_BATCH_OBS
        || (I.init && I.sd[0].ne_q.n + I.sd[1].ne_q.n >= LFQ_BATCH_INDEL_READS)) {
        flush_async(conf);
    }
}

void lfq_call_shutdown(void)
{
    int s;
    vec *iv[] = {&I.ref_base, &I.cov, &I.tails, &I.non_indels, &I.num_ins, &I.num_dels, &I.hrun, &I.seq, &I.pos,
                 &I.has_aq, &I.target};
    size_t k;
    for (g_cur = 0; g_cur < 2; g_cur++) {
        lfq_host_free(B.nt); lfq_host_free(B.bq); lfq_host_free(B.baq); lfq_host_free(B.mq); lfq_host_free(B.sq);
        free(B.col_off); free(B.ref_base); free(B.cov); free(B.nbases); free(B.target); free(B.pos); free(B.seq);
        memset(&B, 0, sizeof(B));
    }
    g_cur = 0;
    if (g_ctx) lfq_destroy(g_ctx);
    g_ctx = NULL;
    memset(&P, 0, sizeof(P));
    if (I.init) {
        for (k = 0; k < sizeof(iv) / sizeof(iv[0]); k++) free(iv[k]->p);
        for (s = 0; s < 2; s++) {
            side_vecs *v = &I.sd[s];
            vec *sv[] = {&v->non_fw, &v->non_rv, &v->ne_off, &v->ne_q, &v->ne_mq, &v->ev_off, &v->key_off, &v->key_chars,
                         &v->ev_fw, &v->ev_rv, &v->rd_off, &v->rd_q, &v->rd_aq, &v->rd_mq, &v->rd_sq};
            for (k = 0; k < sizeof(sv) / sizeof(sv[0]); k++) free(sv[k]->p);
        }
        memset(&I, 0, sizeof(I));
    }
}
