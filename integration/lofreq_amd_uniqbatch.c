/*
 * lofreq_amd_uniqbatch.c -- see lofreq_amd_uniqbatch.h.  Needs include/lofreq_amd.h only.
 *
 * Packing = what lofreq_amd_colbatch.c does for `lofreq call`: one byte per observation and track, the nucleotides of a
 * column in A, C, G, T, N order, forward reads first (strand only matters as a count), nt nibble-packed
 * (LFQ_TRACKS_NT_PACKED), BAQ / SQ -1 -> 255.  uniq's own mpileup carries neither BAQ nor source qualities
 * (lofreq_uniq.c:461-465: flag = MPLP_NO_ORPHAN), so those tracks are normally absent and stay NULL.
 */
#include "lofreq_amd_uniqbatch.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct ub_job {
    void *user;
    int64_t col;                /* column index in the tracks, or -1: a host-side count (add_count) */
    int coverage, alt_count;    /* add_count only */
    float af;
} ub_job;

struct lfq_uniqbatch {
    lfq_ctx *ctx;
    int use_det_lim;
    int oom;
    /* tracks (plain host memory: a few hundred columns per run, staged by the library) */
    uint8_t *nt, *bq, *baq, *mq, *sq;
    int64_t nobs, cap_obs;
    int use_baq, use_sq;
    uint64_t *col_off;
    uint8_t *ref_base;
    int32_t *cov;
    float *af;
    char *alt;
    int64_t ncols, cap_cols, max_depth;
    ub_job *jobs;
    int64_t njobs, cap_jobs;
};

static void *ub_realloc(lfq_uniqbatch *b, void *p, size_t bytes)
{
    void *q = realloc(p, bytes ? bytes : 1);
    if (!q) {
        b->oom = 1;
        return p;
    }
    return q;
}

/* observation tracks: 16-byte aligned, readable 16 bytes past the end (the contract of lfq_tracks) */
static int grow_obs(lfq_uniqbatch *b, int64_t need)
{
    int64_t cap = b->cap_obs;
    uint8_t **t[5];
    int i;
    if (need <= cap) return LFQ_OK;
    while (cap < need) cap = cap ? 2 * cap : (1 << 16);
    t[0] = &b->nt; t[1] = &b->bq; t[2] = &b->baq; t[3] = &b->mq; t[4] = &b->sq;
    for (i = 0; i < 5; i++) {
        void *q = NULL;
        if (posix_memalign(&q, 64, (size_t)cap + 64) != 0) {
            b->oom = 1;
            return LFQ_ERR_NOMEM;
        }
        memset(q, 0, (size_t)cap + 64);
        if (*t[i]) memcpy(q, *t[i], (size_t)b->cap_obs);
        free(*t[i]);
        *t[i] = (uint8_t *)q;
    }
    b->cap_obs = cap;
    return LFQ_OK;
}

static int grow_cols(lfq_uniqbatch *b, int64_t need)
{
    int64_t cap = b->cap_cols;
    if (need <= cap) return LFQ_OK;
    while (cap < need) cap = cap ? 2 * cap : 1024;
    b->col_off = (uint64_t *)ub_realloc(b, b->col_off, (size_t)(cap + 1) * sizeof(uint64_t));
    b->ref_base = (uint8_t *)ub_realloc(b, b->ref_base, (size_t)cap + 16);
    b->cov = (int32_t *)ub_realloc(b, b->cov, (size_t)cap * sizeof(int32_t));
    b->af = (float *)ub_realloc(b, b->af, (size_t)cap * sizeof(float));
    b->alt = (char *)ub_realloc(b, b->alt, (size_t)cap + 1);
    if (b->oom) return LFQ_ERR_NOMEM;
    b->cap_cols = cap;
    return LFQ_OK;
}

static ub_job *new_job(lfq_uniqbatch *b)
{
    if (b->njobs == b->cap_jobs) {
        const int64_t cap = b->cap_jobs ? 2 * b->cap_jobs : 1024;
        b->jobs = (ub_job *)ub_realloc(b, b->jobs, (size_t)cap * sizeof(ub_job));
        if (b->oom) return NULL;
        b->cap_jobs = cap;
    }
    return &b->jobs[b->njobs++];
}

int lfq_uniqbatch_open(lfq_uniqbatch **out, int use_det_lim)
{
    lfq_uniqbatch *b;
    if (!out) return LFQ_ERR_INVALID;
    *out = NULL;
    if (lfq_abi_version() != LFQ_ABI_VERSION) return LFQ_ERR_UNSUPPORTED;    /* struct layouts belong to the version */
    b = (lfq_uniqbatch *)calloc(1, sizeof(*b));
    if (!b) return LFQ_ERR_NOMEM;
    b->use_det_lim = use_det_lim ? 1 : 0;
    *out = b;
    return LFQ_OK;
}

int lfq_uniqbatch_add_column(lfq_uniqbatch *b, const lfq_uniq_col *p, float af, char alt_base, void *user)
{
    int64_t depth = 0, c;
    ub_job *j;
    int i;
    size_t k;
    if (!b || !p) return LFQ_ERR_INVALID;
    if (b->oom) return LFQ_ERR_NOMEM;
    for (i = 0; i < 5; i++) depth += (int64_t)p->nt[i].n;
    if (grow_cols(b, b->ncols + 1) != LFQ_OK || grow_obs(b, b->nobs + depth + 8) != LFQ_OK) return LFQ_ERR_NOMEM;
    j = new_job(b);
    if (!j) return LFQ_ERR_NOMEM;
    c = b->ncols++;
    j->user = user;
    j->col = c;
    j->coverage = p->coverage;
    j->alt_count = 0;
    j->af = af;
    b->col_off[c] = (uint64_t)b->nobs;
    b->ref_base[c] = (uint8_t)p->ref_base;
    b->cov[c] = p->coverage;
    b->af[c] = af;
    b->alt[c] = alt_base;
    for (i = 0; i < 5; i++) {                      /* plp_col_t keeps one int array per nucleotide (plp.h:88-91) */
        const lfq_col_nt *n = &p->nt[i];
        const long fw = n->fw;                     /* strand only matters as a count: forward reads first */
        for (k = 0; k < n->n; k++) {
            const int64_t o = b->nobs++;
            int q;
            /* LFQ_TRACKS_NT_PACKED: observation o sits in byte (o >> 3) * 4 + (o & 3), low nibble for o & 7 < 4 */
            uint8_t *d = &b->nt[((o >> 3) << 2) + (o & 3)];
            const uint8_t v = (uint8_t)(i | (((long)k >= fw) ? 8 : 0));
            *d = (o & 4) ? (uint8_t)((*d & 0x0F) | (v << 4)) : v;
            b->bq[o] = (uint8_t)n->bq[k];
            q = n->n_baq ? n->baq[k] : -1;
            b->baq[o] = (uint8_t)(q < 0 ? LFQ_Q_MISSING : q);
            b->mq[o] = (uint8_t)n->mq[k];
            q = n->n_sq ? n->sq[k] : -1;
            b->sq[o] = (uint8_t)(q < 0 || q > 254 ? (q < 0 ? LFQ_Q_MISSING : 254) : q);
        }
        if (n->n && n->n_baq) b->use_baq = 1;
        if (n->n && n->n_sq) b->use_sq = 1;
    }
    if (depth > b->max_depth) b->max_depth = depth;
    return LFQ_OK;
}

int lfq_uniqbatch_add_count(lfq_uniqbatch *b, int coverage, int alt_count, float af, void *user)
{
    ub_job *j;
    if (!b || b->use_det_lim) return LFQ_ERR_INVALID;       /* det-lim tests always run on the column itself */
    if (b->oom) return LFQ_ERR_NOMEM;
    j = new_job(b);
    if (!j) return LFQ_ERR_NOMEM;
    j->user = user;
    j->col = -1;
    j->coverage = coverage;
    j->alt_count = alt_count;
    j->af = af;
    return LFQ_OK;
}

int lfq_uniqbatch_flush(lfq_uniqbatch *b, lfq_uniq_result_fn fn)
{
    int rc = LFQ_OK;
    int64_t i;
    uint8_t *det = NULL;
    int32_t *uq = NULL;
    if (!b || !fn) return LFQ_ERR_INVALID;
    if (b->oom) return LFQ_ERR_NOMEM;
    if (b->ncols > 0) {
        lfq_tracks t;
        if (!b->ctx) {
            /* one `lofreq uniq` per worker of a wrapper script: LFQ_DEVICE, LOCAL_RANK, or the first free slot of the node */
            const int dev = lfq_pick_device(0, NULL);
            if (dev < 0) return LFQ_ERR_NO_DEVICE;
            rc = lfq_create(&b->ctx, dev);
            if (rc != LFQ_OK) return rc;
        }
        memset(&t, 0, sizeof(t));
        b->col_off[b->ncols] = (uint64_t)b->nobs;
        t.nt = b->nt;  t.bq = b->bq;  t.mq = b->mq;
        t.baq = b->use_baq ? b->baq : NULL;
        t.sq = b->use_sq ? b->sq : NULL;
        t.col_off = b->col_off;
        t.ref_base = b->ref_base;
        t.coverage_plp = b->cov;
        t.ncols = b->ncols;
        t.max_col_obs = b->max_depth;
        t.flags = LFQ_TRACKS_NT_PACKED;
        if (b->use_det_lim) {
            det = (uint8_t *)malloc((size_t)b->ncols);
            if (!det) return LFQ_ERR_NOMEM;
            /* (plp_to_errprobs does not look at coverage_plp; the only gate of uniq_snv, coverage >= 1, was the caller's) */
            t.coverage_plp = NULL;
            rc = lfq_uniq_detlim_batch(b->ctx, &t, /*tracks_on_device=*/0, b->af, det, NULL);
        } else {
            uq = (int32_t *)malloc((size_t)b->ncols * sizeof(int32_t));
            if (!uq) return LFQ_ERR_NOMEM;
            rc = lfq_uniq_binom_batch(b->ctx, &t, /*tracks_on_device=*/0, b->af, b->alt, uq, NULL);
        }
    }
    for (i = 0; rc == LFQ_OK && i < b->njobs; i++) {
        const ub_job *j = &b->jobs[i];
        if (j->col >= 0) {
            fn(j->user, b->use_det_lim ? (int)det[j->col] : (int)uq[j->col]);
        } else {
            /* uniq_snv's binomial branch with a count from the event table (lofreq_uniq.c:342-370, 379-384): the AF reset
             * of :262-268 is the caller's (it logs it); PROB_TO_PHREDQUAL_SAFE as lfq_uniq_binom_batch applies it */
            int32_t one = -1;
            int st = 0;
            const double pv = lfq_binom_cdf(j->coverage, j->alt_count, (double)j->af, &st);
            if (st == 0) {
                one = (pv <= 0.0) ? INT_MAX : (int32_t)(-10.0 * log10l(pv));      /* PROB_TO_PHREDQUAL_SAFE, utils.h:46 */
            }
            fn(j->user, (int)one);
        }
    }
    free(det);
    free(uq);
    b->ncols = b->nobs = b->njobs = 0;
    b->max_depth = 0;
    b->use_baq = b->use_sq = 0;
    return rc;
}

void lfq_uniqbatch_close(lfq_uniqbatch *b)
{
    if (!b) return;
    free(b->nt); free(b->bq); free(b->baq); free(b->mq); free(b->sq);
    free(b->col_off); free(b->ref_base); free(b->cov); free(b->af); free(b->alt); free(b->jobs);
    if (b->ctx) lfq_destroy(b->ctx);
    free(b);
}
