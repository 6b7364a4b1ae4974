/*
 * orc_pileup.c -- CPU restatement of the column builder of `lofreq call`: compile_plp_col (plp.c:797-1288) over the
 * pileup entries htslib's bam_plp / resolve_cigar2 hand it, for all columns of a region at once.
 *
 * TEST INFRASTRUCTURE ONLY (the checker of the reads -> columns -> calls parity tests); nothing in the product path
 * may include, link or call this.
 *
 * Pinning: the product's device pileup is pinned on `lofreq plpsummary` dumps of the reference's 2.1.4 binary
 * (tests/golden/pileup_indels.json, plpindel_*.json); tests/test_orc_pileup.py holds THIS restatement against the same
 * dumps and the binary's VCFs, so that it can stand in for the binary at sizes no fixture covers.
 *
 * htslib itself is absent from the image.  What is restated of it is the published behaviour of its pileup iterator
 * (sam.c resolve_cigar2 / bam_plp_next): every reference position of an M / = / X / D / N operation of a read is one
 * pileup entry; inside D / N the entry is a deletion (is_del; is_refskip for N) whose qpos is the query position of the
 * next base; the entry at the LAST position of an operation carries the indel that follows (I: +len, D: -len, P then I:
 * the insertions up to the next reference-consuming operation); is_tail = last reference position of the read; entries
 * of a column come in the order of the reads in the (position-sorted) input.
 */
#include <ctype.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "lofreq_oracle.h"

typedef struct { void *p; int64_t n, cap; size_t elt; } ovec;

static void *ov_push(ovec *v, int64_t k)
{
    if (v->n + k > v->cap) {
        int64_t c = v->cap ? v->cap : 1024;
        while (v->n + k > c) {
            c *= 2;
        }
        void *q = realloc(v->p, (size_t)c * v->elt);
        if (!q) {
            abort();                    /* test infrastructure: out of memory is fatal */
        }
        v->p = q;
        v->cap = c;
    }
    v->n += k;
    return (char *)v->p + (size_t)(v->n - k) * v->elt;
}
#define OV(T) {NULL, 0, 0, sizeof(T)}
#define PUSH(v, T, x) (*(T *)ov_push(&(v), 1) = (T)(x))

/* the cursor of one active read: the reference-consuming operation that covers the current position */
typedef struct {
    int64_t r;          /* read index */
    int k;              /* cigar operation index */
    int64_t x0;         /* reference position where operation k starts */
    int64_t y0;         /* query position where operation k starts */
    int64_t end;        /* last reference position of the read */
} orc_cursor;

/* an event (distinct inserted / deleted sequence) of the current column, in order of first appearance: uthash iterates
 * in insertion order (utils.h:101-135, add_ins_sequence / add_del_sequence utils.c:559-650) */
typedef struct {
    char key[256];
    int klen;
    int fw, rv, cons_quals;
    ovec q, aq, mq, sq;
} orc_event;

struct orc_plp_region {
    orc_plp_out o;
    ovec col_pos, nt, bq, baq, mq, sq, col_off, ref_base, cov, nbases, cons_indel;
    ovec tails, non_indels, n_ins, n_dels, hrun;
    ovec non_fw[2], non_rv[2], ne_off[2], ne_q[2], ne_mq[2], ev_off[2], key_off[2], key_chars[2], ev_fw[2], ev_rv[2],
         rd_off[2], rd_q[2], rd_aq[2], rd_mq[2], rd_sq[2];
};

static int op_consumes_ref(int op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }     /* M D N = X */
static int op_consumes_query(int op) { return op == 0 || op == 1 || op == 4 || op == 7 || op == 8; }   /* M I S = X */

static int orc_get_hrun(int64_t pos, const char *ref, int64_t ref_len)     /* plp.c:751-787 */
{
    int hrun = 1;
    int64_t i = pos + 1;
    char c;
    if (i >= ref_len) {
        return hrun;
    }
    c = (char)toupper((unsigned char)ref[i]);
    for (i = i + 1; i < ref_len; i++) {
        if (toupper((unsigned char)ref[i]) == c) {
            hrun += 1;
        } else {
            break;
        }
    }
    for (i = pos; i >= 0; i--) {
        if (toupper((unsigned char)ref[i]) == c) {
            hrun += 1;
        } else {
            break;
        }
    }
    return hrun;
}

static orc_event *find_event(ovec *evs, const char *key, int klen)
{
    int64_t i;
    orc_event *e = (orc_event *)evs->p;
    for (i = 0; i < evs->n; i++) {
        if (e[i].klen == klen && memcmp(e[i].key, key, (size_t)klen) == 0) {
            return &e[i];
        }
    }
    e = (orc_event *)ov_push(evs, 1);
    memset(e, 0, sizeof(*e));
    memcpy(e->key, key, (size_t)klen);
    e->klen = klen;
    e->q.elt = e->aq.elt = e->mq.elt = e->sq.elt = sizeof(int16_t);
    return e;
}

static int16_t clamp16(int v) { return (int16_t)(v > 32767 ? 32767 : v); }

orc_plp_region *orc_pileup_region(const orc_reads *rd, int64_t begin, int64_t end, int min_plp_bq, int min_plp_idq,
                                  int use_baq, int use_sq)
{
    orc_plp_region *R = (orc_plp_region *)calloc(1, sizeof(*R));
    ovec act = OV(orc_cursor), evs[2] = {OV(orc_event), OV(orc_event)};
    int64_t next = 0, pos, s;
    if (!R) {
        return NULL;
    }
#define INIT(v, T) R->v.elt = sizeof(T)
    INIT(col_pos, int64_t); INIT(nt, uint8_t); INIT(bq, uint8_t); INIT(baq, uint8_t); INIT(mq, uint8_t); INIT(sq, uint8_t);
    INIT(col_off, uint64_t); INIT(ref_base, uint8_t); INIT(cov, int32_t); INIT(nbases, int32_t); INIT(cons_indel, uint8_t);
    INIT(tails, int32_t); INIT(non_indels, int32_t); INIT(n_ins, int32_t); INIT(n_dels, int32_t); INIT(hrun, int32_t);
    for (s = 0; s < 2; s++) {
        INIT(non_fw[s], int32_t); INIT(non_rv[s], int32_t); INIT(ne_off[s], int64_t); INIT(ne_q[s], int16_t);
        INIT(ne_mq[s], int16_t); INIT(ev_off[s], int64_t); INIT(key_off[s], int64_t); INIT(key_chars[s], char);
        INIT(ev_fw[s], int32_t); INIT(ev_rv[s], int32_t); INIT(rd_off[s], int64_t); INIT(rd_q[s], int16_t);
        INIT(rd_aq[s], int16_t); INIT(rd_mq[s], int16_t); INIT(rd_sq[s], int16_t);
        PUSH(R->ne_off[s], int64_t, 0);
        PUSH(R->ev_off[s], int64_t, 0);
        PUSH(R->key_off[s], int64_t, 0);
        PUSH(R->rd_off[s], int64_t, 0);
    }
    PUSH(R->col_off, uint64_t, 0);

    for (pos = begin; pos < end; pos++) {
        int64_t a, n_plp = 0, w = 0;
        int32_t num_bases = 0, num_tails = 0, num_non_indels = 0, num_ins = 0, num_dels = 0;
        int32_t non_fw[2] = {0, 0}, non_rv[2] = {0, 0};
        int ins_nonevent_qual = 0, del_nonevent_qual = 0;
        ovec ne_q[2] = {OV(int16_t), OV(int16_t)}, ne_mq[2] = {OV(int16_t), OV(int16_t)};
        char ref_base;
        orc_cursor *A;

        /* reads that start at or before this position enter the pileup (bam_plp_push: input sorted by position) */
        while (next < rd->n_reads && rd->pos[next] <= pos) {
            const int64_t c0 = rd->cigar_off[next], c1 = rd->cigar_off[next + 1];
            int64_t x = rd->pos[next], y = 0, k, rlen = 0;
            orc_cursor cu;
            for (k = c0; k < c1; k++) {
                if (op_consumes_ref((int)(rd->cigar[k] & 15))) {
                    rlen += rd->cigar[k] >> 4;
                }
            }
            if (rlen > 0 && x + rlen > pos) {
                /* position the cursor on the first reference-consuming operation */
                for (k = c0; k < c1 && !op_consumes_ref((int)(rd->cigar[k] & 15)); k++) {
                    if (op_consumes_query((int)(rd->cigar[k] & 15))) {
                        y += rd->cigar[k] >> 4;
                    }
                }
                cu.r = next; cu.k = (int)(k - c0); cu.x0 = x; cu.y0 = y; cu.end = x + rlen - 1;
                *(orc_cursor *)ov_push(&act, 1) = cu;
            }
            next++;
        }
        /* drop the reads that ended, advance the others to the operation covering `pos` */
        A = (orc_cursor *)act.p;
        for (a = 0; a < act.n; a++) {
            orc_cursor cu = A[a];
            const int64_t c0 = rd->cigar_off[cu.r], c1 = rd->cigar_off[cu.r + 1];
            if (cu.end < pos) {
                continue;
            }
            while (pos >= cu.x0 + (int64_t)(rd->cigar[c0 + cu.k] >> 4)) {
                const int op = (int)(rd->cigar[c0 + cu.k] & 15);
                const int64_t l = rd->cigar[c0 + cu.k] >> 4;
                cu.x0 += l;                                     /* (a reference-consuming operation) */
                if (op_consumes_query(op)) {
                    cu.y0 += l;
                }
                cu.k++;
                while (c0 + cu.k < c1 && !op_consumes_ref((int)(rd->cigar[c0 + cu.k] & 15))) {
                    if (op_consumes_query((int)(rd->cigar[c0 + cu.k] & 15))) {
                        cu.y0 += rd->cigar[c0 + cu.k] >> 4;
                    }
                    cu.k++;
                }
            }
            A[w++] = cu;
        }
        act.n = w;
        n_plp = act.n;
        if (n_plp == 0) {
            continue;                   /* mpileup yields no column for a position without alignments */
        }

        ref_base = (rd->ref && pos < rd->ref_len) ? rd->ref[pos] : 'N';        /* plp.c:818-823 */
        if (!(ref_base == 'A' || ref_base == 'C' || ref_base == 'T' || ref_base == 'G' || ref_base == 'N')) {
            ref_base = 'N';
        }
        evs[0].n = evs[1].n = 0;

        for (a = 0; a < n_plp; a++) {                                           /* plp.c:839-1192 */
            const orc_cursor cu = A[a];
            const int64_t r = cu.r, c0 = rd->cigar_off[r], c1 = rd->cigar_off[r + 1];
            const int64_t so = rd->seq_off[r], lq = rd->seq_off[r + 1] - so;
            const int op = (int)(rd->cigar[c0 + cu.k] & 15);
            const int64_t l = rd->cigar[c0 + cu.k] >> 4;
            const int is_del = (op == 2 || op == 3), is_refskip = (op == 3);
            const int is_tail = (pos == cu.end);
            const int rev = rd->reverse[r] ? 1 : 0;
            const int fl = rd->tag_flags ? rd->tag_flags[r] : 15;
            int64_t qpos = is_del ? cu.y0 : cu.y0 + (pos - cu.x0);
            int indel = 0, base_skip = 0, iq = 0, dq = 0, iaq = -1, daq = -1;
            const int mq = rd->mapq[r];
            const int sq = rd->sq ? rd->sq[r] : -1;         /* the caller hands in sq only when MPLP_USE_SQ is on (plp.c:876-878) */

            if (qpos > lq - 1) {
                qpos = lq - 1;          /* a deletion at the very end of the query (resolve_cigar2 keeps qpos inside) */
            }
            if (pos == cu.x0 + l - 1 && c0 + cu.k + 1 < c1) {                   /* the indel that follows this operation */
                const int o2 = (int)(rd->cigar[c0 + cu.k + 1] & 15);
                const int64_t l2 = rd->cigar[c0 + cu.k + 1] >> 4;
                if (o2 == 2) {
                    indel = -(int)l2;
                } else if (o2 == 1) {
                    indel = (int)l2;
                } else if (o2 == 6 && c0 + cu.k + 2 < c1) {                     /* pad: insertions behind it */
                    int64_t k3, l3 = 0;
                    for (k3 = c0 + cu.k + 2; k3 < c1; k3++) {
                        const int o3 = (int)(rd->cigar[k3] & 15);
                        if (o3 == 1) {
                            l3 += rd->cigar[k3] >> 4;
                        } else if (o3 == 2 || o3 == 0 || o3 == 3 || o3 == 7 || o3 == 8) {
                            break;
                        }
                    }
                    indel = (int)l3;
                }
            }

            if (!is_del) {                                                      /* plp.c:915-1015 */
                const int nt4 = rd->seq[so + qpos] > 4 ? 4 : rd->seq[so + qpos];
                int bq = rd->qual[so + qpos];
                if (is_tail) {
                    num_tails += 1;
                }
                if (bq < min_plp_bq) {
                    base_skip = 1;
                } else {
                    if (bq > 93) {
                        bq = 93;                                                /* SANGER_PHRED_MAX */
                    }
                    PUSH(R->nt, uint8_t, nt4 | (rev << 3));
                    PUSH(R->bq, uint8_t, bq);
                    if (use_baq) {
                        int baq = (rd->lb) ? (int)rd->lb[so + qpos] - 33 : -1;  /* :956-962 */
                        PUSH(R->baq, uint8_t, baq < 0 ? 255 : (baq > 254 ? 254 : baq));
                    }
                    PUSH(R->mq, uint8_t, mq);
                    if (use_sq) {
                        PUSH(R->sq, uint8_t, sq < 0 ? 255 : (sq > 254 ? 254 : sq));
                    }
                }
            }
            if (!(is_del || is_refskip || base_skip)) {                         /* :1019-1022 */
                num_bases += 1;
            }
            if (rd->bi && (fl & 1)) {
                iq = (int)rd->bi[so + qpos] - 33;                               /* :1024-1037 */
            }
            if (rd->bd && (fl & 2)) {
                dq = (int)rd->bd[so + qpos] - 33;                               /* :1039-1060 */
            }
            if (iq < min_plp_idq || dq < min_plp_idq) {                         /* :1062 */
                continue;                                                       /* (num_ign_indels is not read downstream) */
            }
            if (indel > 0) {                                                    /* :1071-1111 */
                char key[256];
                int j, klen = indel > 255 ? 255 : indel;
                orc_event *e;
                if (rd->ai && (fl & 4)) {
                    iaq = (int)rd->ai[so + qpos] - 33;
                }
                for (j = 1; j <= klen; j++) {
                    const int64_t q = qpos + j;
                    key[j - 1] = (q < lq) ? ORC_SEQ_LETTER(rd->seq[so + q]) : 'N';     /* seq_nt16_str letter, plp.c:1092-1093 */
                }
                num_ins += 1;
                e = find_event(&evs[0], key, klen);
                e->cons_quals += iq;
                if (rev) e->rv++; else e->fw++;
                PUSH(e->q, int16_t, clamp16(iq)); PUSH(e->aq, int16_t, clamp16(iaq));
                PUSH(e->mq, int16_t, clamp16(mq)); PUSH(e->sq, int16_t, clamp16(sq));
                PUSH(ne_q[1], int16_t, clamp16(dq)); PUSH(ne_mq[1], int16_t, clamp16(mq));
                del_nonevent_qual += dq;
                if (rev) non_rv[1]++; else non_fw[1]++;
            } else if (indel < 0) {                                             /* :1115-1170 */
                char key[256];
                int j, klen = -indel > 255 ? 255 : -indel;
                orc_event *e;
                if (rd->ad && (fl & 8)) {
                    daq = (int)rd->ad[so + qpos] - 33;
                }
                for (j = 1; j <= klen; j++) {
                    const int c = (rd->ref && pos + j < rd->ref_len) ? rd->ref[pos + j] : 'N';
                    key[j - 1] = (char)toupper((unsigned char)c);
                }
                num_dels += 1;
                e = find_event(&evs[1], key, klen);
                e->cons_quals += dq;
                if (rev) e->rv++; else e->fw++;
                PUSH(e->q, int16_t, clamp16(dq)); PUSH(e->aq, int16_t, clamp16(daq));
                PUSH(e->mq, int16_t, clamp16(mq)); PUSH(e->sq, int16_t, clamp16(sq));
                PUSH(ne_q[0], int16_t, clamp16(iq)); PUSH(ne_mq[0], int16_t, clamp16(mq));
                ins_nonevent_qual += iq;
                if (rev) non_rv[0]++; else non_fw[0]++;
            } else {                                                            /* :1172-1190 */
                num_non_indels += 1;
                PUSH(ne_q[0], int16_t, clamp16(iq)); PUSH(ne_mq[0], int16_t, clamp16(mq));
                ins_nonevent_qual += iq;
                if (rev) non_rv[0]++; else non_fw[0]++;
                PUSH(ne_q[1], int16_t, clamp16(dq)); PUSH(ne_mq[1], int16_t, clamp16(mq));
                del_nonevent_qual += dq;
                if (rev) non_rv[1]++; else non_fw[1]++;
            }
        }

        /* the column */
        PUSH(R->col_pos, int64_t, pos);
        PUSH(R->col_off, uint64_t, R->nt.n);
        PUSH(R->ref_base, uint8_t, ref_base);
        PUSH(R->cov, int32_t, n_plp);
        PUSH(R->nbases, int32_t, num_bases);
        PUSH(R->tails, int32_t, num_tails);
        PUSH(R->non_indels, int32_t, num_non_indels);
        PUSH(R->n_ins, int32_t, num_ins);
        PUSH(R->n_dels, int32_t, num_dels);
        PUSH(R->hrun, int32_t, rd->ref ? orc_get_hrun(pos, rd->ref, rd->ref_len) : -1);
        {
            int maxq[2] = {0, 0};                                               /* consensus: plp.c:1236-1270 */
            for (s = 0; s < 2; s++) {
                int64_t i;
                orc_event *e = (orc_event *)evs[s].p;
                for (i = 0; i < evs[s].n; i++) {
                    if (e[i].cons_quals > maxq[s]) {
                        maxq[s] = e[i].cons_quals;
                    }
                }
            }
            PUSH(R->cons_indel, uint8_t, (maxq[0] > ins_nonevent_qual || maxq[1] > del_nonevent_qual) ? 1 : 0);
        }
        for (s = 0; s < 2; s++) {
            int64_t i;
            orc_event *e = (orc_event *)evs[s].p;
            const int has_event = evs[0].n > 0 || evs[1].n > 0;
            PUSH(R->non_fw[s], int32_t, non_fw[s]);
            PUSH(R->non_rv[s], int32_t, non_rv[s]);
            if (has_event && ne_q[s].n > 0) {       /* the quality arrays of the reads without an event: only where call_indels reads them */
                memcpy(ov_push(&R->ne_q[s], ne_q[s].n), ne_q[s].p, (size_t)ne_q[s].n * 2);
                memcpy(ov_push(&R->ne_mq[s], ne_mq[s].n), ne_mq[s].p, (size_t)ne_mq[s].n * 2);
            }
            PUSH(R->ne_off[s], int64_t, R->ne_q[s].n);
            for (i = 0; i < evs[s].n; i++) {
                memcpy(ov_push(&R->key_chars[s], e[i].klen), e[i].key, (size_t)e[i].klen);
                PUSH(R->key_off[s], int64_t, R->key_chars[s].n);
                PUSH(R->ev_fw[s], int32_t, e[i].fw);
                PUSH(R->ev_rv[s], int32_t, e[i].rv);
                memcpy(ov_push(&R->rd_q[s], e[i].q.n), e[i].q.p, (size_t)e[i].q.n * 2);
                memcpy(ov_push(&R->rd_aq[s], e[i].q.n), e[i].aq.p, (size_t)e[i].q.n * 2);
                memcpy(ov_push(&R->rd_mq[s], e[i].q.n), e[i].mq.p, (size_t)e[i].q.n * 2);
                memcpy(ov_push(&R->rd_sq[s], e[i].q.n), e[i].sq.p, (size_t)e[i].q.n * 2);
                PUSH(R->rd_off[s], int64_t, R->rd_q[s].n);
                free(e[i].q.p); free(e[i].aq.p); free(e[i].mq.p); free(e[i].sq.p);
            }
            PUSH(R->ev_off[s], int64_t, R->ev_fw[s].n);
            free(ne_q[s].p);
            free(ne_mq[s].p);
        }
    }
    free(act.p);
    free(evs[0].p);
    free(evs[1].p);
    /* padding the 16-byte contract of the packed tracks asks for */
    for (s = 0; s < 32; s++) {
        PUSH(R->nt, uint8_t, 0); PUSH(R->bq, uint8_t, 0); PUSH(R->baq, uint8_t, 0); PUSH(R->mq, uint8_t, 0); PUSH(R->sq, uint8_t, 0);
        PUSH(R->key_chars[0], char, 0); PUSH(R->key_chars[1], char, 0);
    }
    {
        orc_plp_out *o = &R->o;
        o->ncols = R->col_pos.n;
        o->col_pos = (const int64_t *)R->col_pos.p;
        o->nt = (const uint8_t *)R->nt.p; o->bq = (const uint8_t *)R->bq.p;
        o->baq = use_baq ? (const uint8_t *)R->baq.p : NULL;
        o->mq = (const uint8_t *)R->mq.p;
        o->sq = use_sq ? (const uint8_t *)R->sq.p : NULL;
        o->col_off = (const uint64_t *)R->col_off.p;
        o->coverage_plp = (const int32_t *)R->cov.p; o->num_bases = (const int32_t *)R->nbases.p;
        o->cons_indel = (const uint8_t *)R->cons_indel.p;
        o->indel.ncols = o->ncols;
        o->indel.ref_base = (const uint8_t *)R->ref_base.p;
        o->indel.coverage_plp = o->coverage_plp;
        o->indel.num_tails = (const int32_t *)R->tails.p;
        o->indel.num_non_indels = (const int32_t *)R->non_indels.p;
        o->indel.num_ins = (const int32_t *)R->n_ins.p; o->indel.num_dels = (const int32_t *)R->n_dels.p;
        o->indel.hrun = (const int32_t *)R->hrun.p;
        for (s = 0; s < 2; s++) {
            o->indel.non_fw[s] = (const int32_t *)R->non_fw[s].p; o->indel.non_rv[s] = (const int32_t *)R->non_rv[s].p;
            o->indel.ne_off[s] = (const int64_t *)R->ne_off[s].p;
            o->indel.ne_q[s] = (const int16_t *)R->ne_q[s].p; o->indel.ne_mq[s] = (const int16_t *)R->ne_mq[s].p;
            o->indel.ev_off[s] = (const int64_t *)R->ev_off[s].p; o->indel.key_off[s] = (const int64_t *)R->key_off[s].p;
            o->indel.key_chars[s] = (const char *)R->key_chars[s].p;
            o->indel.ev_fw[s] = (const int32_t *)R->ev_fw[s].p; o->indel.ev_rv[s] = (const int32_t *)R->ev_rv[s].p;
            o->indel.rd_off[s] = (const int64_t *)R->rd_off[s].p;
            o->indel.rd_q[s] = (const int16_t *)R->rd_q[s].p; o->indel.rd_aq[s] = (const int16_t *)R->rd_aq[s].p;
            o->indel.rd_mq[s] = (const int16_t *)R->rd_mq[s].p; o->indel.rd_sq[s] = (const int16_t *)R->rd_sq[s].p;
        }
    }
    return R;
}

const orc_plp_out *orc_plp_region_out(const orc_plp_region *R) { return R ? &R->o : NULL; }

void orc_plp_region_free(orc_plp_region *R)
{
    int s;
    if (!R) {
        return;
    }
    ovec *all[] = {&R->col_pos, &R->nt, &R->bq, &R->baq, &R->mq, &R->sq, &R->col_off, &R->ref_base, &R->cov, &R->nbases,
                   &R->cons_indel, &R->tails, &R->non_indels, &R->n_ins, &R->n_dels, &R->hrun};
    for (s = 0; s < (int)(sizeof(all) / sizeof(all[0])); s++) {
        free(all[s]->p);
    }
    for (s = 0; s < 2; s++) {
        ovec *sd[] = {&R->non_fw[s], &R->non_rv[s], &R->ne_off[s], &R->ne_q[s], &R->ne_mq[s], &R->ev_off[s], &R->key_off[s],
                      &R->key_chars[s], &R->ev_fw[s], &R->ev_rv[s], &R->rd_off[s], &R->rd_q[s], &R->rd_aq[s], &R->rd_mq[s],
                      &R->rd_sq[s]};
        int i;
        for (i = 0; i < (int)(sizeof(sd) / sizeof(sd[0])); i++) {
            free(sd[i]->p);
        }
    }
    free(R);
}
