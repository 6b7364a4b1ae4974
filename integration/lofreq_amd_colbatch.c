/*
 * lofreq_amd_colbatch.c -- see lofreq_amd_colbatch.h.  Plain C against include/lofreq_amd.h.
 *
 * What it replaces in the reference, per column: the body of call_vars after its gates -- call_indels
 * (lofreq_call.c:619-726) and call_snvs (:735-879) with report_var (:93-137) -- by copying the column into packed
 * batches and running them through lfq_call_indels_batch / lfq_call_snvs_submit + _collect.
 */
#include "lofreq_amd_colbatch.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CB_BATCH_COLS (1 << 20)         /* flush every 2^20 columns (or at the end) */
#define CB_BATCH_OBS ((int64_t)1 << 28) /* ... or when a track holds 256 Mi observations: 26 000 columns at 10 000x (1.1 GiB of
                                         * pinned host tracks per batch, two batches, one staging allocation on the device) */
#define CB_BATCH_INDEL_READS (1 << 28)  /* ... or when the flattened indel columns hold this many reads */

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
} cb_batch;

typedef struct { void *p; int64_t n, cap; size_t elt; } vec;
typedef struct {
    vec non_fw, non_rv, ne_off, ne_q, ne_mq, ev_off, key_off, key_chars, ev_fw, ev_rv, rd_off, rd_q, rd_aq, rd_mq, rd_sq;
} side_vecs;

struct lfq_colbatch {
    cb_batch bb[2];
    int cur;
    lfq_ctx *ctx;
    int64_t seq;
    int64_t batch_cols;
    lfq_colbatch_emit_fn emit;
    void *user;
    long wo_idaq;
    int oom;                            /* an allocation failed: every later call returns LFQ_ERR_NOMEM */
    struct {
        int active;                     /* a submitted batch waits for its collect */
        int which;                      /* its buffer set */
        lfq_conf lc;                    /* the conf it was submitted with */
        char **iline;                   /* formatted indel records of the same columns, with their arrival numbers */
        int64_t *iseq;
        int64_t n_iline;
    } pend;
    /* indel fields of the columns that carry indel events (lfq_indel_columns, flattened) */
    struct {
        vec ref_base, cov, tails, non_indels, num_ins, num_dels, hrun, seq, pos, has_aq;
        vec target;
        side_vecs sd[2];
        int64_t ncols;
    } I;
};

static void *cb_realloc(lfq_colbatch *b, void *p, size_t n)
{
    void *q = realloc(p, n ? n : 1);
    if (!q) {
        b->oom = 1;
        return p;
    }
    return q;
}

static void *vpush(lfq_colbatch *b, vec *v, int64_t k)
{
    if (v->n + k > v->cap) {
        int64_t cap = v->cap;
        void *q;
        while (v->n + k > cap) cap = cap ? 2 * cap : 1024;
        q = realloc(v->p, (size_t)cap * v->elt);
        if (!q) {
            static int64_t sink[4];
            b->oom = 1;
            return sink;                 /* the caller's few writes land here; the batch is abandoned (oom is sticky) */
        }
        v->p = q;
        v->cap = cap;
    }
    v->n += k;
    return (char *)v->p + (size_t)(v->n - k) * v->elt;
}

static void indel_init(lfq_colbatch *b)
{
    int s;
#define V(T) {NULL, 0, 0, sizeof(T)}
    const vec i32 = V(int32_t), i64 = V(int64_t), i16 = V(int16_t), ch = V(char), u8 = V(uint8_t), ptr = V(char *);
#undef V
    b->I.ref_base = u8;
    b->I.cov = b->I.tails = b->I.non_indels = b->I.num_ins = b->I.num_dels = b->I.hrun = b->I.pos = b->I.has_aq = i32;
    b->I.seq = i64;
    b->I.target = ptr;
    for (s = 0; s < 2; s++) {
        side_vecs *v = &b->I.sd[s];
        v->non_fw = v->non_rv = v->ev_fw = v->ev_rv = i32;
        v->ne_off = v->ev_off = v->key_off = v->rd_off = i64;
        v->ne_q = v->ne_mq = v->rd_q = v->rd_aq = v->rd_mq = v->rd_sq = i16;
        v->key_chars = ch;
        *(int64_t *)vpush(b, &v->ne_off, 1) = 0;
        *(int64_t *)vpush(b, &v->ev_off, 1) = 0;
        *(int64_t *)vpush(b, &v->key_off, 1) = 0;
        *(int64_t *)vpush(b, &v->rd_off, 1) = 0;
    }
    b->I.ncols = 0;
}

static void push_quals(lfq_colbatch *b, vec *v, const int *a, size_t n)
{
    size_t j;
    int16_t *d = (int16_t *)vpush(b, v, (int64_t)n);
    if (b->oom) return;
    for (j = 0; j < n; j++) d[j] = (int16_t)a[j];
}

static void push_event(lfq_colbatch *b, side_vecs *v, const lfq_col_event *e)
{
    size_t j;
    const size_t kl = strlen(e->key);
    int16_t *d;
    char *k = (char *)vpush(b, &v->key_chars, (int64_t)kl);
    if (b->oom) return;
    memcpy(k, e->key, kl);
    *(int64_t *)vpush(b, &v->key_off, 1) = v->key_chars.n;
    *(int32_t *)vpush(b, &v->ev_fw, 1) = (int32_t)e->fw;
    *(int32_t *)vpush(b, &v->ev_rv, 1) = (int32_t)e->rv;
    push_quals(b, &v->rd_q, e->q, e->n);
    push_quals(b, &v->rd_mq, e->mq, e->n);
    d = (int16_t *)vpush(b, &v->rd_aq, (int64_t)e->n);          /* -1 where the BAM carried no ai/ad tag */
    if (b->oom) return;
    for (j = 0; j < e->n; j++) d[j] = (int16_t)(j < e->n_aq ? e->aq[j] : -1);
    d = (int16_t *)vpush(b, &v->rd_sq, (int64_t)e->n);
    if (b->oom) return;
    for (j = 0; j < e->n; j++) d[j] = (int16_t)(j < e->n_sq ? e->sq[j] : -1);
    *(int64_t *)vpush(b, &v->rd_off, 1) = v->rd_q.n;
}

/* copy the indel fields of one column (plp.h:113-130) */
static void indel_add_column(lfq_colbatch *b, const lfq_col_view *p, int64_t seq)
{
    int e;
    char *t;
    if (p->num_ins == 0 && p->num_dels == 0) return;        /* no event, no test (lofreq_call.c:684, :706) */
    t = strdup(p->target);
    if (!t) {
        b->oom = 1;
        return;
    }
    *(uint8_t *)vpush(b, &b->I.ref_base, 1) = (uint8_t)p->ref_base;
    *(int32_t *)vpush(b, &b->I.cov, 1) = p->coverage_plp;
    *(int32_t *)vpush(b, &b->I.tails, 1) = p->num_tails;
    *(int32_t *)vpush(b, &b->I.non_indels, 1) = p->num_non_indels;
    *(int32_t *)vpush(b, &b->I.num_ins, 1) = p->num_ins;
    *(int32_t *)vpush(b, &b->I.num_dels, 1) = p->num_dels;
    *(int32_t *)vpush(b, &b->I.hrun, 1) = p->hrun;
    *(int32_t *)vpush(b, &b->I.pos, 1) = p->pos;
    *(int32_t *)vpush(b, &b->I.has_aq, 1) = p->has_indel_aqs;
    *(int64_t *)vpush(b, &b->I.seq, 1) = seq;
    *(char **)vpush(b, &b->I.target, 1) = t;
    *(int32_t *)vpush(b, &b->I.sd[0].non_fw, 1) = (int32_t)p->non_ins_fw_rv[0];
    *(int32_t *)vpush(b, &b->I.sd[0].non_rv, 1) = (int32_t)p->non_ins_fw_rv[1];
    *(int32_t *)vpush(b, &b->I.sd[1].non_fw, 1) = (int32_t)p->non_del_fw_rv[0];
    *(int32_t *)vpush(b, &b->I.sd[1].non_rv, 1) = (int32_t)p->non_del_fw_rv[1];
    push_quals(b, &b->I.sd[0].ne_q, p->ins_quals, p->n_ins_quals);
    push_quals(b, &b->I.sd[0].ne_mq, p->ins_map_quals, p->n_ins_quals);
    push_quals(b, &b->I.sd[1].ne_q, p->del_quals, p->n_del_quals);
    push_quals(b, &b->I.sd[1].ne_mq, p->del_map_quals, p->n_del_quals);
    *(int64_t *)vpush(b, &b->I.sd[0].ne_off, 1) = b->I.sd[0].ne_q.n;
    *(int64_t *)vpush(b, &b->I.sd[1].ne_off, 1) = b->I.sd[1].ne_q.n;
    for (e = 0; e < p->n_ins_events; e++) {                 /* the caller's order = uthash insertion order = reference order */
        push_event(b, &b->I.sd[0], &p->ins_events[e]);
    }
    for (e = 0; e < p->n_del_events; e++) {
        push_event(b, &b->I.sd[1], &p->del_events[e]);
    }
    *(int64_t *)vpush(b, &b->I.sd[0].ev_off, 1) = b->I.sd[0].ev_fw.n;
    *(int64_t *)vpush(b, &b->I.sd[1].ev_off, 1) = b->I.sd[1].ev_fw.n;
    b->I.ncols++;
}

/* run the flattened indel columns; returns malloc'ed records */
static int indel_flush(lfq_colbatch *b, lfq_conf *lc, lfq_indel_record **out, int64_t *n_rec)
{
    lfq_indel_columns c;
    lfq_indel_record *rec;
    int64_t nev, ntests = 0;
    int s, rc;
    *n_rec = 0;
    *out = NULL;
    if (b->I.ncols == 0) return LFQ_OK;
    memset(&c, 0, sizeof(c));
    c.ncols = b->I.ncols;
    c.ref_base = b->I.ref_base.p;  c.coverage_plp = b->I.cov.p;  c.num_tails = b->I.tails.p;
    c.num_non_indels = b->I.non_indels.p;  c.num_ins = b->I.num_ins.p;  c.num_dels = b->I.num_dels.p;  c.hrun = b->I.hrun.p;
    for (s = 0; s < 2; s++) {
        side_vecs *v = &b->I.sd[s];
        lfq_indel_side *o = &c.side[s];
        o->non_fw = v->non_fw.p;  o->non_rv = v->non_rv.p;  o->ne_off = v->ne_off.p;  o->ne_q = v->ne_q.p;
        o->ne_mq = v->ne_mq.p;  o->ev_off = v->ev_off.p;  o->key_off = v->key_off.p;  o->key_chars = v->key_chars.p;
        o->ev_fw = v->ev_fw.p;  o->ev_rv = v->ev_rv.p;  o->rd_off = v->rd_off.p;  o->rd_q = v->rd_q.p;
        o->rd_aq = v->rd_aq.p;  o->rd_mq = v->rd_mq.p;  o->rd_sq = v->rd_sq.p;
    }
    nev = b->I.sd[0].ev_fw.n + b->I.sd[1].ev_fw.n;
    rec = (lfq_indel_record *)malloc(sizeof(lfq_indel_record) * (size_t)(nev + 1));
    if (!rec) return LFQ_ERR_NOMEM;
    rc = lfq_call_indels_batch(b->ctx, lc, &c, rec, nev, n_rec, &ntests);     /* advances lc->bonf_indel / num_indel_tests (:693-696) */
    if (rc != LFQ_OK) {
        free(rec);
        return rc;
    }
    *out = rec;
    return LFQ_OK;
}

static char *indel_line(lfq_colbatch *b, const lfq_indel_record *r)
{
    const side_vecs *v = &b->I.sd[r->side];
    const int64_t *koff = v->key_off.p;
    const int64_t kl = koff[r->event + 1] - koff[r->event];
    const char rb = (char)((uint8_t *)b->I.ref_base.p)[r->col];
    char *ref = (char *)malloc((size_t)kl + 2), *alt = (char *)malloc((size_t)kl + 2);
    char *line = (char *)malloc((size_t)kl * 2 + 1024);
    if (!ref || !alt || !line) {
        free(ref); free(alt); free(line);
        b->oom = 1;
        return NULL;
    }
    ref[0] = alt[0] = rb;                                    /* ins_to_str / del_to_str (lofreq_call.c:255-303) */
    memcpy((r->side == 0 ? alt : ref) + 1, (const char *)v->key_chars.p + koff[r->event], (size_t)kl);
    (r->side == 0 ? alt : ref)[kl + 1] = 0;
    (r->side == 0 ? ref : alt)[1] = 0;
    lfq_format_indel_record(line, (int)(kl * 2 + 1024), ((char **)b->I.target.p)[r->col], ((int32_t *)b->I.pos.p)[r->col],
                            ref, alt, r->qual, r->dp, r->af, r->sb, r->ref_fw, r->ref_rv, r->alt_fw, r->alt_rv,
                            r->hrun, NULL);
    if (!((int32_t *)b->I.has_aq.p)[r->col]) b->wo_idaq += 1;   /* report_var, lofreq_call.c:109-111 */
    free(ref); free(alt);
    return line;
}

static void indel_reset(lfq_colbatch *b)
{
    int64_t i;
    int s;
    for (i = 0; i < b->I.target.n; i++) free(((char **)b->I.target.p)[i]);
    b->I.ref_base.n = b->I.cov.n = b->I.tails.n = b->I.non_indels.n = b->I.num_ins.n = b->I.num_dels.n = b->I.hrun.n = 0;
    b->I.seq.n = b->I.pos.n = b->I.has_aq.n = b->I.target.n = 0;
    for (s = 0; s < 2; s++) {
        side_vecs *v = &b->I.sd[s];
        v->non_fw.n = v->non_rv.n = v->ne_q.n = v->ne_mq.n = v->key_chars.n = v->ev_fw.n = v->ev_rv.n = 0;
        v->rd_q.n = v->rd_aq.n = v->rd_mq.n = v->rd_sq.n = 0;
        v->ne_off.n = v->ev_off.n = v->key_off.n = v->rd_off.n = 1;     /* keep the leading 0 */
    }
    b->I.ncols = 0;
}

/* the five observation tracks live in pinned memory: their upload is then a DMA that lfq_call_snvs_submit only queues */
static int pinned_grow(uint8_t **p, int64_t used, int64_t cap)
{
    uint8_t *q = (uint8_t *)lfq_host_alloc((size_t)cap);
    if (!q) return LFQ_ERR_NOMEM;                           /* (no pinned memory: is there a HIP device?) */
    if (*p && used > 0) memcpy(q, *p, (size_t)used);
    lfq_host_free(*p);
    *p = q;
    return LFQ_OK;
}

static int grow_obs(lfq_colbatch *b, int64_t need)
{
    cb_batch *B = &b->bb[b->cur];
    const int64_t used = B->nobs;
    int64_t cap = B->cap_obs;
    if (need <= cap) return LFQ_OK;
    while (cap < need) cap = cap ? 2 * cap : (1 << 24);
    if (pinned_grow(&B->nt, (used + 1) / 2 + 4, cap) || pinned_grow(&B->bq, used, cap) || pinned_grow(&B->baq, used, cap)
        || pinned_grow(&B->mq, used, cap) || pinned_grow(&B->sq, used, cap)) {
        b->oom = 1;
        return LFQ_ERR_NOMEM;
    }
    B->cap_obs = cap;
    return LFQ_OK;
}

static int grow_cols(lfq_colbatch *b, int64_t need)
{
    cb_batch *B = &b->bb[b->cur];
    int64_t cap = B->cap_cols;
    if (need <= cap) return LFQ_OK;
    while (cap < need) cap = cap ? 2 * cap : (1 << 16);
    B->col_off = (uint64_t *)cb_realloc(b, B->col_off, (size_t)(cap + 1) * sizeof(uint64_t));
    B->ref_base = (uint8_t *)cb_realloc(b, B->ref_base, (size_t)cap);
    B->cov = (int32_t *)cb_realloc(b, B->cov, (size_t)cap * sizeof(int32_t));
    B->nbases = (int32_t *)cb_realloc(b, B->nbases, (size_t)cap * sizeof(int32_t));
    B->target = (char **)cb_realloc(b, B->target, (size_t)cap * sizeof(char *));
    B->pos = (int *)cb_realloc(b, B->pos, (size_t)cap * sizeof(int));
    B->seq = (int64_t *)cb_realloc(b, B->seq, (size_t)cap * sizeof(int64_t));
    if (b->oom) return LFQ_ERR_NOMEM;
    B->cap_cols = cap;
    return LFQ_OK;
}

static int ensure_ctx(lfq_colbatch *b)
{
    if (!b->ctx) {
        /* one `lofreq call -r <bin>` per worker of the parallel wrapper (lofreq2_call_pparallel.py:640-667): each
         * process takes a GPU of its own -- LFQ_DEVICE, LOCAL_RANK, or the first free worker slot of the node */
        const int dev = lfq_pick_device(0, NULL);
        int rc;
        if (dev < 0) return LFQ_ERR_NO_DEVICE;
        rc = lfq_create(&b->ctx, dev);
        if (rc != LFQ_OK) return rc;
        /* only the records are read here, never the dense per-column entries: strand counts where a record comes out
         * (DP4 / SB, lofreq_call.c:117-129) and no entries for untested columns */
        rc = lfq_set_dense_strand_counts(b->ctx, 0);
        if (rc == LFQ_OK) rc = lfq_set_dense_counts(b->ctx, 0);
        return rc;
    }
    return LFQ_OK;
}

/* wait for the batch submitted at the previous flush, finish it on the host, emit its columns' records */
static int collect_pending(lfq_colbatch *b, lfq_conf *conf)
{
    cb_batch *B = &b->bb[b->pend.which];
    lfq_snv_record *rec;
    int64_t n_rec = 0, i, k;
    int rc;
    if (!b->pend.active) return LFQ_OK;
    rec = (lfq_snv_record *)malloc(sizeof(lfq_snv_record) * (size_t)(3 * B->ncols + 1));
    if (!rec) return LFQ_ERR_NOMEM;
    rc = B->ncols ? lfq_call_snvs_collect(b->ctx, &b->pend.lc, rec, 3 * B->ncols, &n_rec, NULL, NULL) : LFQ_OK;
    if (rc != LFQ_OK) {
        free(rec);
        return rc;
    }
    if (B->ncols) {
        conf->bonf_subst = b->pend.lc.bonf_subst;           /* lofreq_call.c:794-800 */
        conf->num_snv_tests = b->pend.lc.num_snv_tests;     /* lofreq_call.c:801 */
    }
    /* merge by arrival number; a column's indel records precede its SNV records (call_vars :896 / :928) */
    for (i = 0, k = 0; i < n_rec || k < b->pend.n_iline;) {
        const int64_t s_snv = i < n_rec ? B->seq[rec[i].col] : INT64_MAX;
        const int64_t s_ind = k < b->pend.n_iline ? b->pend.iseq[k] : INT64_MAX;
        if (s_ind <= s_snv) {
            b->emit(b->user, b->pend.iline[k]);
            free(b->pend.iline[k++]);
        } else {                            /* vcf_write_var (vcf.c:469-497), FILTER '.' like report_var */
            char line[512];
            lfq_format_snv_record(line, sizeof(line), B->target[rec[i].col], B->pos[rec[i].col], &rec[i], NULL);
            b->emit(b->user, line);
            i++;
        }
    }
    free(rec);
    free(b->pend.iline); free(b->pend.iseq);
    b->pend.iline = NULL; b->pend.iseq = NULL; b->pend.n_iline = 0;
    for (i = 0; i < B->ncols; i++) free(B->target[i]);
    B->ncols = 0; B->nobs = 0; B->max_depth = 0;
    b->pend.active = 0;
    return LFQ_OK;
}

/* the batch is full: finish the previous one (its kernels ran while this one was filled), queue this one, go on */
static int flush_async(lfq_colbatch *b, lfq_conf *conf)
{
    cb_batch *B = &b->bb[b->cur];
    lfq_conf lc;
    lfq_tracks t;
    lfq_indel_record *irec = NULL;
    int64_t n_irec = 0, k;
    int rc;

    if (B->ncols == 0 && b->I.ncols == 0) return LFQ_OK;
    if ((rc = ensure_ctx(b)) != LFQ_OK) return rc;
    if ((rc = collect_pending(b, conf)) != LFQ_OK) return rc;      /* conf now carries the running factors up to this batch's first column */
    lc = *conf;
    if ((rc = indel_flush(b, &lc, &irec, &n_irec)) != LFQ_OK) return rc;     /* call_indels of this batch's columns: synchronous, few tests */
    conf->bonf_indel = lc.bonf_indel;                       /* lofreq_call.c:693-695 */
    conf->num_indel_tests = lc.num_indel_tests;             /* :696 */
    b->pend.iline = (char **)malloc(sizeof(char *) * (size_t)(n_irec + 1));
    b->pend.iseq = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_irec + 1));
    if (!b->pend.iline || !b->pend.iseq) {
        free(irec);
        return LFQ_ERR_NOMEM;
    }
    for (k = 0; k < n_irec; k++) {
        b->pend.iline[k] = indel_line(b, &irec[k]);
        b->pend.iseq[k] = ((int64_t *)b->I.seq.p)[irec[k].col];
        if (!b->pend.iline[k]) {
            free(irec);
            b->pend.n_iline = k;
            return LFQ_ERR_NOMEM;
        }
    }
    b->pend.n_iline = n_irec;
    free(irec);
    indel_reset(b);
    b->pend.which = b->cur;
    b->pend.active = 1;
    if (B->ncols > 0) {
        B->col_off[B->ncols] = (uint64_t)B->nobs;
        memset(&t, 0, sizeof(t));
        t.nt = B->nt; t.bq = B->bq; t.mq = B->mq;
        t.baq = B->use_baq ? B->baq : NULL;
        t.sq = B->use_sq ? B->sq : NULL;
        t.col_off = B->col_off; t.ref_base = B->ref_base;
        t.coverage_plp = B->cov; t.num_bases = B->nbases;
        t.ncols = B->ncols; t.max_col_obs = B->max_depth;
        t.flags = LFQ_TRACKS_NT_PACKED;      /* half the nt bytes over PCIe, the 1.5-bytes-per-observation count kernel */
        b->pend.lc = lc;
        rc = lfq_call_snvs_submit(b->ctx, &b->pend.lc, &t, /*tracks_on_device=*/0);    /* copies + kernels queued; returns */
        if (rc != LFQ_OK) return rc;
    }
    b->cur ^= 1;                             /* the other buffer set was collected above: it is empty */
    return LFQ_OK;
}

int lfq_colbatch_open(lfq_colbatch **out, lfq_colbatch_emit_fn emit, void *user, long batch_cols)
{
    lfq_colbatch *b;
    const char *e = getenv("LFQ_SHIM_BATCH_COLS");          /* tuning, tests */
    if (!out || !emit) return LFQ_ERR_INVALID;
    *out = NULL;
    if (lfq_abi_version() != LFQ_ABI_VERSION) return LFQ_ERR_UNSUPPORTED;    /* struct layouts belong to the version */
    b = (lfq_colbatch *)calloc(1, sizeof(*b));
    if (!b) return LFQ_ERR_NOMEM;
    b->emit = emit;
    b->user = user;
    b->batch_cols = batch_cols > 0 ? batch_cols : ((e && atoll(e) > 0) ? atoll(e) : CB_BATCH_COLS);
    indel_init(b);
    if (b->oom) {
        lfq_colbatch_close(b);
        return LFQ_ERR_NOMEM;
    }
    *out = b;
    return LFQ_OK;
}

int lfq_colbatch_add(lfq_colbatch *b, lfq_conf *conf, const lfq_col_view *p)
{
    cb_batch *B;
    int i;
    size_t j;
    int64_t depth = 0, c;

    if (!b || !conf || !p) return LFQ_ERR_INVALID;
    if (b->oom) return LFQ_ERR_NOMEM;
    B = &b->bb[b->cur];
    b->seq++;
    if (p->take_indels) indel_add_column(b, p, b->seq);                /* lofreq_call.c:896 */
    if (p->take_snvs) {
        /* the remaining gates (:930 num_bases*2 < coverage_plp, :747 min_cov, :754) run on the device */
        for (i = 0; i < 5; i++) depth += (int64_t)p->nt[i].n;
        if (grow_cols(b, B->ncols + 1) != LFQ_OK || grow_obs(b, B->nobs + depth) != LFQ_OK) return LFQ_ERR_NOMEM;
        c = B->ncols;
        B->target[c] = strdup(p->target);
        if (!B->target[c]) {
            b->oom = 1;
            return LFQ_ERR_NOMEM;
        }
        B->col_off[c] = (uint64_t)B->nobs;
        B->ref_base[c] = (uint8_t)p->ref_base;
        B->cov[c] = p->coverage_plp;
        B->nbases[c] = p->num_bases;
        B->pos[c] = p->pos;
        B->seq[c] = b->seq;
        for (i = 0; i < 5; i++) {                  /* plp_col_t keeps one int array per nucleotide (plp.h:88-91) */
            const lfq_col_nt *n = &p->nt[i];
            const long fw = n->fw;                 /* strand only matters as a count: forward reads first */
            for (j = 0; j < n->n; j++) {
                const int64_t o = B->nobs++;
                int q;
                /* LFQ_TRACKS_NT_PACKED: observation o sits in byte (o >> 3) * 4 + (o & 3), low nibble for o & 7 < 4 */
                {
                    uint8_t *d = &B->nt[((o >> 3) << 2) + (o & 3)];
                    const uint8_t v = (uint8_t)(i | (((long)j >= fw) ? 8 : 0));
                    *d = (o & 4) ? (uint8_t)((*d & 0x0F) | (v << 4)) : v;
                }
                B->bq[o] = (uint8_t)n->bq[j];
                q = n->n_baq ? n->baq[j] : -1;
                B->baq[o] = (uint8_t)(q < 0 ? LFQ_Q_MISSING : q);
                B->mq[o] = (uint8_t)n->mq[j];
                q = n->n_sq ? n->sq[j] : -1;
                B->sq[o] = (uint8_t)(q < 0 || q > 254 ? (q < 0 ? LFQ_Q_MISSING : 254) : q);
            }
            if (n->n && n->n_baq) B->use_baq = 1;
            if (n->n && n->n_sq) B->use_sq = 1;
        }
        if (depth > B->max_depth) B->max_depth = depth;
        B->ncols++;
    }
    if (b->oom) return LFQ_ERR_NOMEM;
    if (B->ncols >= b->batch_cols || B->nobs >= CB_BATCH_OBS
        || b->I.sd[0].ne_q.n + b->I.sd[1].ne_q.n >= CB_BATCH_INDEL_READS) {
        return flush_async(b, conf);
    }
    return LFQ_OK;
}

int lfq_colbatch_flush(lfq_colbatch *b, lfq_conf *conf)
{
    int rc;
    if (!b || !conf) return LFQ_ERR_INVALID;
    if (b->oom) return LFQ_ERR_NOMEM;
    if ((rc = flush_async(b, conf)) != LFQ_OK) return rc;
    if (b->pend.active) {
        if ((rc = ensure_ctx(b)) != LFQ_OK) return rc;
        return collect_pending(b, conf);
    }
    return LFQ_OK;
}

long lfq_colbatch_indel_calls_wo_idaq(const lfq_colbatch *b)
{
    return b ? b->wo_idaq : 0;
}

void lfq_colbatch_close(lfq_colbatch *b)
{
    int s, w;
    int64_t i;
    size_t k;
    if (!b) return;
    for (w = 0; w < 2; w++) {
        cb_batch *B = &b->bb[w];
        for (i = 0; i < B->ncols; i++) free(B->target[i]);
        lfq_host_free(B->nt); lfq_host_free(B->bq); lfq_host_free(B->baq); lfq_host_free(B->mq); lfq_host_free(B->sq);
        free(B->col_off); free(B->ref_base); free(B->cov); free(B->nbases); free(B->target); free(B->pos); free(B->seq);
    }
    for (i = 0; i < b->pend.n_iline; i++) free(b->pend.iline[i]);
    free(b->pend.iline); free(b->pend.iseq);
    if (b->ctx) lfq_destroy(b->ctx);
    for (i = 0; i < b->I.target.n; i++) free(((char **)b->I.target.p)[i]);
    {
        vec *iv[] = {&b->I.ref_base, &b->I.cov, &b->I.tails, &b->I.non_indels, &b->I.num_ins, &b->I.num_dels, &b->I.hrun,
                     &b->I.seq, &b->I.pos, &b->I.has_aq, &b->I.target};
        for (k = 0; k < sizeof(iv) / sizeof(iv[0]); k++) free(iv[k]->p);
    }
    for (s = 0; s < 2; s++) {
        side_vecs *v = &b->I.sd[s];
        vec *sv[] = {&v->non_fw, &v->non_rv, &v->ne_off, &v->ne_q, &v->ne_mq, &v->ev_off, &v->key_off, &v->key_chars,
                     &v->ev_fw, &v->ev_rv, &v->rd_off, &v->rd_q, &v->rd_aq, &v->rd_mq, &v->rd_sq};
        for (k = 0; k < sizeof(sv) / sizeof(sv[0]); k++) free(sv[k]->p);
    }
    free(b);
}
