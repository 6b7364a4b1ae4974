/*
 * lofreq_amd_region.c -- see lofreq_amd_region.h.  Plain C against include/lofreq_amd.h.
 */
#include "lofreq_amd_region.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* The per-base arrays of a region (bases, qualities, BI, BD: 150 bytes per read each) live in pinned memory
 * (lfq_host_alloc): lfq_readset_create then queues their copies as DMA transfers and returns, the BAQ kernels wait for the
 * chunks they need on the device, and this thread goes on decoding the next region instead of sitting out the upload
 * (grow-only, so the cost of pinning is paid in the first regions). */
typedef struct { void *p; int64_t n, cap; size_t elt; int pinned; } rvec;

static int rv_reserve(rvec *v, int64_t more)
{
    if (v->n + more > v->cap) {
        int64_t c = v->cap ? v->cap : (v->pinned ? (int64_t)1 << 16 : 4096);
        void *q;
        while (v->n + more > c) {
            c *= 2;
        }
        if (v->pinned) {
            q = lfq_host_alloc((size_t)c * v->elt);
            if (q && v->n > 0) {
                memcpy(q, v->p, (size_t)v->n * v->elt);
            }
            if (q) {
                lfq_host_free(v->p);
            }
        } else {
            q = realloc(v->p, (size_t)c * v->elt);
        }
        if (!q) {
            return LFQ_ERR_NOMEM;
        }
        v->p = q;
        v->cap = c;
    }
    return LFQ_OK;
}

/* the reads of one region as the flat arrays lfq_readset_create takes; two of them, because the host arrays of region k
 * must stay as they are until region k is finished -- which happens while region k + 1 is being filled */
typedef struct {
    rvec pos, cig_off, cig, seq_off, seq, qual, mapq, rev, bi, bd, flags;
    int64_t n, nb;
    int any_bi, any_bd;
    char *target;
    const char *ref;
    int64_t ref_len, beg, end;
    lfq_readset *rs;
    int started;
    /* the SNV tracks of the region: their first pass runs beside the BAQ kernels, the scatter pass is queued behind them
     * (lfq_readset_pileup_snv returns at once) */
    lfq_tracks t;
    int have_tracks;
    int64_t *col_pos_s, pos_cap;
} reg_buf;

struct lfq_region {
    lfq_ctx *ctx;
    lfq_conf *conf;
    lfq_region_opts o;
    lfq_region_emit_fn emit;
    void *user;
    reg_buf buf[2];
    int cur;                        /* buffer being filled */
    int open;                       /* lfq_region_begin called, lfq_region_end not yet */
    int64_t wo_idaq;
    /* outputs, grown on demand */
    int64_t *col_pos_i, pos_cap;
    lfq_snv_record *srec;
    int64_t srec_cap;
    lfq_indel_record *irec;
    int64_t irec_cap;
};

void lfq_region_opts_init(lfq_region_opts *o)
{
    memset(o, 0, sizeof(*o));
    o->use_baq = 1;
    o->baq_extended = 1;
    o->use_idaq = 0;
    o->def_nm_q = -1;
    o->max_mq = 255;
    o->min_plp_bq = 3;
}

static void buf_init(reg_buf *b)
{
    memset(b, 0, sizeof(*b));
    b->pos.elt = sizeof(int32_t);
    b->cig_off.elt = b->seq_off.elt = sizeof(int64_t);
    b->cig.elt = sizeof(uint32_t);
    b->seq.elt = b->qual.elt = b->mapq.elt = b->rev.elt = b->bi.elt = b->bd.elt = b->flags.elt = 1;
    /* (the small per-read arrays too: a pageable source is staged by the runtime with copy kernels, 7 ms for the 50 MB of a
     * 2 M-read region, in front of everything else on the upload stream) */
    b->seq.pinned = b->qual.pinned = b->bi.pinned = b->bd.pinned = 1;
    b->pos.pinned = b->cig_off.pinned = b->cig.pinned = b->seq_off.pinned = b->mapq.pinned = b->rev.pinned = b->flags.pinned = 1;
}

static void buf_free(reg_buf *b)
{
    rvec *all[] = {&b->pos, &b->cig_off, &b->cig, &b->seq_off, &b->seq, &b->qual, &b->mapq, &b->rev, &b->bi, &b->bd, &b->flags};
    size_t i;
    for (i = 0; i < sizeof(all) / sizeof(all[0]); i++) {
        if (all[i]->pinned) {
            lfq_host_free(all[i]->p);
        } else {
            free(all[i]->p);
        }
    }
    free(b->target);
    free(b->col_pos_s);
}

int lfq_region_open(lfq_region **out, lfq_ctx *ctx, lfq_conf *conf, const lfq_region_opts *opts,
                    lfq_region_emit_fn emit, void *user)
{
    lfq_region *r;
    if (!out || !ctx || !conf || !opts || !emit) {
        return LFQ_ERR_INVALID;
    }
    if (lfq_abi_version() != LFQ_ABI_VERSION) {       /* lfq_conf and the read / column structs belong to the version */
        return LFQ_ERR_UNSUPPORTED;
    }
    r = (lfq_region *)calloc(1, sizeof(*r));
    if (!r) {
        return LFQ_ERR_NOMEM;
    }
    r->ctx = ctx;
    r->conf = conf;
    r->o = *opts;
    r->emit = emit;
    r->user = user;
    buf_init(&r->buf[0]);
    buf_init(&r->buf[1]);
    /* the bulk of the indel columns (the quality arrays of the reads WITHOUT an event) stays on the device */
    lfq_set_indel_arrays_on_host(ctx, 0);
    *out = r;
    return LFQ_OK;
}

int lfq_region_begin(lfq_region *r, const char *target_name, const char *ref, int64_t ref_len, int64_t beg0, int64_t end0)
{
    reg_buf *b;
    if (!r || r->open || !target_name || !ref || ref_len <= 0 || beg0 < 0 || end0 < beg0) {
        return LFQ_ERR_INVALID;
    }
    b = &r->buf[r->cur];
    if (b->started) {
        return LFQ_ERR_INVALID;                 /* (cannot happen: lfq_region_end finishes the older region first) */
    }
    b->pos.n = b->cig.n = b->seq.n = b->qual.n = b->mapq.n = b->rev.n = b->bi.n = b->bd.n = b->flags.n = 0;
    b->cig_off.n = b->seq_off.n = 0;
    b->n = b->nb = 0;
    b->any_bi = b->any_bd = 0;
    free(b->target);
    b->target = strdup(target_name);
    if (!b->target) {
        return LFQ_ERR_NOMEM;
    }
    b->ref = ref;
    b->ref_len = ref_len;
    b->beg = beg0;
    b->end = end0 > ref_len ? ref_len : end0;
    if (rv_reserve(&b->cig_off, 1) || rv_reserve(&b->seq_off, 1)) {
        return LFQ_ERR_NOMEM;
    }
    ((int64_t *)b->cig_off.p)[0] = 0;
    ((int64_t *)b->seq_off.p)[0] = 0;
    b->cig_off.n = b->seq_off.n = 1;
    r->open = 1;
    return LFQ_OK;
}

int lfq_region_add_read(lfq_region *r, int32_t pos, int flag, int mapq, int n_cigar, const uint32_t *cigar, int l_qseq,
                        const uint8_t *seq4, const uint8_t *qual, const char *bi, const char *bd)
{
    /* the 4-bit BAM base "=ACMGRSVTWYHKDBN" -> the library's base code: 0..3 = A, C, G, T and 4 = N as seq_nt16_int (htslib
     * hts.c) has them, 5..15 = the other letters in the order "=MRSVWYHKDB" (include/lofreq_amd.h: they behave like N except
     * where the reference compares or prints the LETTER of a read base) */
    static const uint8_t nt16_int[16] = {5, 0, 1, 6, 2, 7, 8, 9, 3, 10, 11, 12, 13, 14, 15, 4};
    reg_buf *b;
    int i;
    if (!r || !r->open || n_cigar < 0 || l_qseq < 0 || (n_cigar > 0 && !cigar) || (l_qseq > 0 && (!seq4 || !qual))) {
        return LFQ_ERR_INVALID;
    }
    /* plp.c:706-720 */
    if (mapq > r->o.max_mq) {
        mapq = r->o.max_mq;
    } else if (mapq < r->o.min_mq) {
        return 0;
    } else if (r->o.no_orphan && (flag & 1) && !(flag & 2)) {      /* BAM_FPAIRED without BAM_FPROPER_PAIR */
        return 0;
    }
    if (n_cigar == 0 || l_qseq == 0) {
        return 0;                               /* nothing the pileup could place */
    }
    b = &r->buf[r->cur];
    if (rv_reserve(&b->pos, 1) || rv_reserve(&b->cig_off, 1) || rv_reserve(&b->seq_off, 1) || rv_reserve(&b->cig, n_cigar)
        || rv_reserve(&b->seq, l_qseq + 16) || rv_reserve(&b->qual, l_qseq + 16) || rv_reserve(&b->bi, l_qseq + 16)
        || rv_reserve(&b->bd, l_qseq + 16) || rv_reserve(&b->mapq, 1) || rv_reserve(&b->rev, 1) || rv_reserve(&b->flags, 1)) {
        return LFQ_ERR_NOMEM;
    }
    ((int32_t *)b->pos.p)[b->n] = pos;
    memcpy((uint32_t *)b->cig.p + b->cig.n, cigar, (size_t)n_cigar * 4);
    b->cig.n += n_cigar;
    ((int64_t *)b->cig_off.p)[b->n + 1] = b->cig.n;
    for (i = 0; i < l_qseq; i++) {
        ((uint8_t *)b->seq.p)[b->nb + i] = nt16_int[(seq4[i >> 1] >> ((~i & 1) << 2)) & 0xf];     /* bam_seqi */
    }
    memcpy((uint8_t *)b->qual.p + b->nb, qual, (size_t)l_qseq);
    /* BI / BD: the tag bytes as they are (quality + 33); a read without the tag counts as quality 0 (plp.c:1024-1060) */
    if (bi && (int)strlen(bi) >= l_qseq) {
        memcpy((uint8_t *)b->bi.p + b->nb, bi, (size_t)l_qseq);
        b->any_bi = 1;
    } else {
        memset((uint8_t *)b->bi.p + b->nb, 33, (size_t)l_qseq);
        bi = NULL;
    }
    if (bd && (int)strlen(bd) >= l_qseq) {
        memcpy((uint8_t *)b->bd.p + b->nb, bd, (size_t)l_qseq);
        b->any_bd = 1;
    } else {
        memset((uint8_t *)b->bd.p + b->nb, 33, (size_t)l_qseq);
        bd = NULL;
    }
    ((uint8_t *)b->flags.p)[b->n] = (uint8_t)((bi ? 1 : 0) | (bd ? 2 : 0));
    ((uint8_t *)b->mapq.p)[b->n] = (uint8_t)mapq;
    ((uint8_t *)b->rev.p)[b->n] = (flag & 16) ? 1 : 0;                  /* bam_is_rev */
    b->nb += l_qseq;
    b->seq.n = b->qual.n = b->bi.n = b->bd.n = b->nb;
    ((int64_t *)b->seq_off.p)[b->n + 1] = b->nb;
    b->n += 1;
    b->pos.n = b->mapq.n = b->rev.n = b->flags.n = b->n;
    b->cig_off.n = b->seq_off.n = b->n + 1;
    return 1;
}

/* region k: upload (asynchronous) + BAQ / IDAQ kernels (queued) + source quality */
static int region_start(lfq_region *r, reg_buf *b)
{
    lfq_pileup_reads rd;
    lfq_pileup_indel_tags tg;
    int rc;
    if (b->n == 0) {
        b->started = 1;
        return LFQ_OK;
    }
    memset(&rd, 0, sizeof(rd));
    memset(&tg, 0, sizeof(tg));
    rd.n_reads = b->n;
    rd.pos = (const int32_t *)b->pos.p;
    rd.cigar_off = (const int64_t *)b->cig_off.p;
    rd.cigar = (const uint32_t *)b->cig.p;
    rd.seq_off = (const int64_t *)b->seq_off.p;
    rd.seq = (const uint8_t *)b->seq.p;
    rd.qual = (const uint8_t *)b->qual.p;
    rd.mapq = (const uint8_t *)b->mapq.p;
    rd.reverse = (const uint8_t *)b->rev.p;
    rd.ref = b->ref;
    rd.ref_len = b->ref_len;
    tg.bi = b->any_bi ? (const uint8_t *)b->bi.p : NULL;
    tg.bd = b->any_bd ? (const uint8_t *)b->bd.p : NULL;
    tg.tag_flags = (const uint8_t *)b->flags.p;
    rc = lfq_readset_create(r->ctx, &rd, &tg, &b->rs);
    if (rc != LFQ_OK) {
        return rc;
    }
    if (r->o.use_baq || r->o.use_idaq) {                                /* plp.c:667-683 */
        rc = lfq_readset_baq(r->ctx, b->rs, r->o.baq_extended, r->o.use_idaq ? 1 : 0);
        if (rc != LFQ_OK) {
            /* the read set waits for its queued copies before it goes: the caller may refill this buffer's arrays */
            lfq_readset_destroy(b->rs);
            b->rs = NULL;
            return rc;
        }
    }
    if (r->o.use_sq) {                                                  /* plp.c:727-735; DEFAULT_MIN_BQ = 6 */
        rc = lfq_readset_source_qual(r->ctx, b->rs, r->o.def_nm_q, 6, NULL, NULL);
        if (rc != LFQ_OK) {
            lfq_readset_destroy(b->rs);
            b->rs = NULL;
            return rc;
        }
    }
    b->have_tracks = 0;
    b->started = 1;
    return LFQ_OK;
}

/* the SNV tracks of the region; returns when the scatter pass is queued */
static int region_snv_tracks(lfq_region *r, reg_buf *b)
{
    const int64_t width = b->end - b->beg + 1;
    int rc;
    if (width > b->pos_cap) {
        int64_t *a = (int64_t *)realloc(b->col_pos_s, sizeof(int64_t) * (size_t)width);
        if (!a) {
            return LFQ_ERR_NOMEM;
        }
        b->col_pos_s = a;
        b->pos_cap = width;
    }
    rc = lfq_readset_pileup_snv(r->ctx, b->rs, b->beg, b->end, r->o.min_plp_bq, &b->t, b->col_pos_s);
    b->have_tracks = rc == LFQ_OK;
    return rc;
}

static int grow_out(lfq_region *r, int64_t width)
{
    if (width > r->pos_cap) {
        int64_t *c = (int64_t *)realloc(r->col_pos_i, sizeof(int64_t) * (size_t)width);
        if (!c) {
            return LFQ_ERR_NOMEM;
        }
        r->col_pos_i = c;
        r->pos_cap = width;
    }
    return LFQ_OK;
}

/* region k: both pileups, call_indels + call_snvs, output in column order, indels before SNVs within a column
 * (call_vars, lofreq_call.c:896 before :928) */
static int region_finish(lfq_region *r, reg_buf *b)
{
    const lfq_indel_columns *cols = NULL;
    lfq_tracks t;
    int64_t n_irec = 0, n_srec = 0, n_tests = 0, i = 0, k = 0;
    int rc = LFQ_OK;
    char line[768];

    if (!b->started) {
        return LFQ_OK;
    }
    b->started = 0;
    if (b->n == 0 || !b->rs) {
        return LFQ_OK;
    }
    rc = grow_out(r, b->end - b->beg + 1);
    /* the consensus-indel gate of call_vars (:928-931) needs the indel fields even when no indel is called -- but
     * without BI / BD no event can win the consensus (its quality sum is 0, plp.c:1236-1270) */
    if (rc == LFQ_OK && (r->o.call_indels || b->any_bi || b->any_bd)) {
        rc = lfq_readset_pileup_indels(r->ctx, b->rs, b->beg, b->end, r->o.min_plp_idq, &cols, r->col_pos_i);
        if (rc == LFQ_OK && !r->o.only_indels) {
            rc = region_snv_tracks(r, b);       /* its scatter pass runs under the host part of the indel tests below */
        }
        if (rc == LFQ_OK && r->o.call_indels && cols && cols->ncols > 0) {
            const int64_t nev = cols->side[0].ev_off[cols->ncols] + cols->side[1].ev_off[cols->ncols];
            if (nev + 16 > r->irec_cap) {
                lfq_indel_record *q = (lfq_indel_record *)realloc(r->irec, sizeof(lfq_indel_record) * (size_t)(nev + 16));
                if (!q) {
                    rc = LFQ_ERR_NOMEM;
                } else {
                    r->irec = q;
                    r->irec_cap = nev + 16;
                }
            }
            if (rc == LFQ_OK) {
                rc = lfq_call_indels_batch(r->ctx, r->conf, cols, r->irec, r->irec_cap, &n_irec, &n_tests);
            }
        }
    }
    if (rc == LFQ_OK && !r->o.only_indels && !b->have_tracks) {
        rc = region_snv_tracks(r, b);           /* (no indel pileup ran) */
    }
    if (rc == LFQ_OK && b->have_tracks) {
        t = b->t;
        if (cols && cols->cons_indel && t.ncols > 0) {
            /* both pileups cover [beg, end]: another column count is an internal error, and going on would emit SNVs
             * at columns whose consensus is an indel (lofreq_call.c:928-931) */
            rc = cols->ncols == t.ncols ? lfq_pileup_skip_snv_columns(r->ctx, cols->cons_indel, cols->ncols) : LFQ_ERR_INVALID;
        }
        if (rc == LFQ_OK && t.ncols > 0) {
            if (3 * t.ncols > r->srec_cap) {
                lfq_snv_record *q = (lfq_snv_record *)realloc(r->srec, sizeof(lfq_snv_record) * (size_t)(3 * t.ncols));
                if (!q) {
                    rc = LFQ_ERR_NOMEM;
                } else {
                    r->srec = q;
                    r->srec_cap = 3 * t.ncols;
                }
            }
            if (rc == LFQ_OK) {
                rc = lfq_call_snvs_batch(r->ctx, r->conf, &t, /*tracks_on_device=*/1, r->srec, r->srec_cap, &n_srec, NULL, NULL);
            }
        }
    }
    while (rc == LFQ_OK && (i < n_srec || k < n_irec)) {
        const int64_t p_snv = i < n_srec ? b->col_pos_s[r->srec[i].col] : INT64_MAX;
        const int64_t p_ind = k < n_irec ? r->col_pos_i[r->irec[k].col] : INT64_MAX;
        if (p_ind <= p_snv) {
            /* ins_to_str / del_to_str (lofreq_call.c:255-303): REF / ALT of an indel event */
            const lfq_indel_record *e = &r->irec[k++];
            const lfq_indel_side *sd = &cols->side[e->side];
            const int64_t kl = sd->key_off[e->event + 1] - sd->key_off[e->event];
            const int64_t r0 = sd->rd_off[e->event], r1 = sd->rd_off[e->event + 1];
            char *ref = (char *)malloc((size_t)kl + 2), *alt = (char *)malloc((size_t)kl + 2);
            char *big = (char *)malloc((size_t)kl * 2 + 1024);
            int64_t q, has_aq = 0, c0, c1;
            int s;
            if (!ref || !alt || !big) {
                free(ref); free(alt); free(big);
                rc = LFQ_ERR_NOMEM;
                break;
            }
            ref[0] = alt[0] = (char)cols->ref_base[e->col];
            memcpy((e->side == 0 ? alt : ref) + 1, sd->key_chars + sd->key_off[e->event], (size_t)kl);
            (e->side == 0 ? alt : ref)[kl + 1] = 0;
            (e->side == 0 ? ref : alt)[1] = 0;
            lfq_format_indel_record(big, (int)(kl * 2 + 1024), b->target, p_ind, ref, alt, e->qual, e->dp, e->af, e->sb,
                                    e->ref_fw, e->ref_rv, e->alt_fw, e->alt_rv, e->hrun, NULL);
            r->emit(r->user, big);
            /* report_var: an indel call in a column where no read carried an alignment quality (has_indel_aqs, plp.c:1076, 1121) */
            for (s = 0; s < 2 && !has_aq; s++) {
                c0 = cols->side[s].rd_off[cols->side[s].ev_off[e->col]];
                c1 = cols->side[s].rd_off[cols->side[s].ev_off[e->col + 1]];
                for (q = c0; q < c1; q++) {
                    if (cols->side[s].rd_aq[q] >= 0) {
                        has_aq = 1;
                        break;
                    }
                }
            }
            (void)r0; (void)r1;
            if (!has_aq) {
                r->wo_idaq += 1;
            }
            free(ref); free(alt); free(big);
        } else {
            lfq_format_snv_record(line, (int)sizeof(line), b->target, p_snv, &r->srec[i], NULL);
            r->emit(r->user, line);
            i++;
        }
    }
    lfq_readset_destroy(b->rs);
    b->rs = NULL;
    return rc;
}

int lfq_region_end(lfq_region *r)
{
    reg_buf *mine, *prev;
    int rc;
    if (!r || !r->open) {
        return LFQ_ERR_INVALID;
    }
    r->open = 0;
    mine = &r->buf[r->cur];
    prev = &r->buf[r->cur ^ 1];
    /* the region before this one: its BAQ kernels ran while this one's reads were decoded */
    rc = region_finish(r, prev);
    if (rc != LFQ_OK) {
        return rc;
    }
    rc = region_start(r, mine);
    r->cur ^= 1;
    return rc;
}

int lfq_region_close(lfq_region *r, int64_t *wo_idaq)
{
    int rc;
    if (!r) {
        return LFQ_ERR_INVALID;
    }
    rc = region_finish(r, &r->buf[r->cur]);             /* the older one, if it is still pending */
    if (rc == LFQ_OK) {
        rc = region_finish(r, &r->buf[r->cur ^ 1]);
    }
    if (wo_idaq) {
        *wo_idaq = r->wo_idaq;
    }
    if (r->buf[0].rs) lfq_readset_destroy(r->buf[0].rs);
    if (r->buf[1].rs) lfq_readset_destroy(r->buf[1].rs);
    lfq_set_indel_arrays_on_host(r->ctx, 1);
    buf_free(&r->buf[0]);
    buf_free(&r->buf[1]);
    free(r->col_pos_i);
    free(r->srec);
    free(r->irec);
    free(r);
    return rc;
}
