/*
 * lfq_baq.hip -- base alignment quality (BAQ) on CDNA4: the per-read pre-step of `lofreq call`
 * (~80 % of the reference's default end-to-end wall time).
 *
 *   reference   bam_prob_realn_core_ext (bam_md_ext.c:260-491), BAQ half: alignment window, band width,
 *               kpa_ext_glocal (kprobaln_ext.c:80-270): banded profile HMM, scaled forward / backward in
 *               doubles, MAP state + posterior per query base; then the (extended) BAQ of every base.
 *
 * Mapping: ONE READ PER LANE.  The recurrence inside a row is sequential (the deletion state of cell k needs
 * cell k-1 of the same row, kprobaln_ext.c:166), and results have to be bit-identical to the reference's
 * doubles, so a lane walks its read's rows and cells in exactly the reference's order; 64 reads advance in
 * lock-step per wavefront.  Narrow-band reads (the default band of 7, and band 8) keep the row the recurrences work on
 * in REGISTERS and send every second forward row to HBM for the MAP step (lfq_baq_reg_kernel); everything else keeps
 * every row in HBM, interleaved per wavefront ([row][cell][lane]: every access of a wave is one coalesced 512-byte
 * line), with two rotating backward rows (lfq_baq_kernel).  The MAP step of a row is done as soon as the backward
 * row exists, so no backward matrix is stored.  See DESIGN.md 6b.
 *
 * -ffp-contract=off (Makefile): no FMA contraction, every operation rounds like the reference's SSE2 build.
 */
#include <type_traits>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "lfq_internal.h"

#define LFQ_BAQ_EI .25
#define LFQ_BAQ_EM .33333333333

/* a read's geometry record completed from the resident arrays (32 bytes per read used to cross the link for this) */
__device__ __forceinline__ LfqBaqRead lfq_baq_read_of(const LfqBaqArgs &A, int64_t rid)
{
    const LfqBaqGeom g = A.geom[rid];
    LfqBaqRead R;
    R.pos = A.pos[rid];
    R.l_qseq = (int32_t)(A.seq_off[rid + 1] - A.seq_off[rid]);
    R.xb = g.xb;
    R.l_ref = g.l_ref;
    R.bw = g.bw;
    R.cigar_off = A.cigar_off[rid];
    R.n_cigar = (int32_t)(A.cigar_off[rid + 1] - R.cigar_off);
    return R;
}

__device__ __forceinline__ int lfq_baq_u(int bw, int i, int k)      /* set_u, kprobaln_ext.c:46 */
{
    int x = i - bw;
    x = x > 0 ? x : 0;
    return (k - x + 1) * 3;
}

__device__ __forceinline__ int lfq_baq_code(int ch)                 /* seq_nt16_int[seq_nt16_table[ch]] (htslib) */
{
    switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

__device__ __forceinline__ double lfq_baq_emit(int r, int qy, double ql)
{
    return (r > 3 || qy > 3) ? 1. : (r == qy ? 1. - ql : ql * LFQ_BAQ_EM);
}

/* The general kernel: any band, every row in HBM (what reads with a band beyond 8 or a reference window beyond
 * LFQ_BAQ_LDS_MAX_LREF get; the narrow bands run lfq_baq_reg_kernel below). */
__global__ __launch_bounds__(64) void lfq_baq_kernel(LfqBaqArgs A, int64_t n_launch)
{
    const int lane = (int)threadIdx.x;
    const int64_t ridx = (int64_t)blockIdx.x * 64 + lane;
    const bool live = ridx < n_launch;
    const int64_t rid = A.order ? (int64_t)A.order[A.first_read + (live ? ridx : 0)] : A.first_read + (live ? ridx : 0);
    const LfqBaqRead R = lfq_baq_read_of(A, rid);
    const int W = A.W, rows = A.rows;
    /* this wavefront's scratch */
    double *F = A.scratch + (size_t)blockIdx.x * ((size_t)rows * W + 2 * (size_t)W + 2 * ((size_t)rows + 2)) * 64;
    double *B = F + (size_t)rows * W * 64;
    double *S = B + 2 * (size_t)W * 64;
    int32_t *expect = A.expect + (size_t)blockIdx.x * rows * 64;
    uint8_t *left = A.tmp8 + (size_t)blockIdx.x * 2 * rows * 64, *rght = left + (size_t)rows * 64;
#define FQ(i_, u_) F[((size_t)(i_) * W + (u_)) * 64 + lane]
#define BQ(r_, u_) B[((size_t)(r_) * W + (u_)) * 64 + lane]
#define SQ(i_) S[(size_t)(i_) * 64 + lane]
    if (!live || R.l_qseq <= 0 || R.l_ref <= 0) {
        return;
    }
    const int l_query = R.l_qseq, l_ref = R.l_ref;
    const int64_t s0 = A.seq_off[rid];
    const uint8_t *query = A.seq + s0 - 1, *iqual = A.qual + s0 - 1;     /* 1-based like the reference */
    const uint8_t *refw = A.ref + R.xb - 1;
    uint8_t *out = A.lb_out + s0;
#define RC(k_) lfq_baq_code(refw[k_])
    int bw = l_ref > l_query ? l_ref : l_query;                          /* kprobaln_ext.c:99-101 */
    if (bw > R.bw) bw = R.bw;
    if (bw < abs(l_ref - l_query)) bw = abs(l_ref - l_query);
    const int bw2 = bw * 2 + 1;
    const int Wr = bw2 * 3 + 6;                                          /* this read's row width (<= W) */
    const float par_d = A.par_d, par_e = A.par_e;                        /* kpa_ext_par_t: lfq_set_baq_hmm_params */
    double m[9];
    const double sM = 1. / (2 * l_query + 2), sI = sM;                   /* :127-132 */
    m[0] = (1 - par_d - par_d) * (1 - sM); m[1] = m[2] = par_d * (1 - sM);
    m[3] = (1 - par_e) * (1 - sI); m[4] = par_e * (1 - sI); m[5] = 0.;
    m[6] = 1 - par_e; m[7] = 0.; m[8] = par_e;
    const double bM = (1 - par_d) / l_ref, bI = par_d / l_ref;

    /* ---- forward (:134-190) ----
     * Rows >= 2 are stored UNSCALED; the reference's `fi[k] *= 1/sum` (:181) is applied by whoever reads the
     * cell (the same multiplication of the same two doubles: identical value), which saves one read + write
     * pass over the matrix.  SQ(i) keeps s[i], RQ(i) the reciprocal the reference multiplies with.
     * Only the cells next to the band are zeroed (the reference reads them from calloc'ed memory). */
#define RQ(i_) S[(size_t)(rows + 2 + (i_)) * 64 + lane]
    for (int u = 0; u < Wr; u++) {
        FQ(0, u) = 0.;
        FQ(1, u) = 0.;
    }
    FQ(0, lfq_baq_u(bw, 0, 0)) = 1.;
    SQ(0) = 1.;
    RQ(0) = 1.;
    RQ(1) = 1.;                                                          /* row 1 is rescaled in place (division, :154) */
    {
        double sum = 0.;
        const int end = l_ref < bw + 1 ? l_ref : bw + 1;
        const double ql = A.qual2prob[iqual[1]];
        for (int k = 1; k <= end; ++k) {
            const int u = lfq_baq_u(bw, 1, k);
            const double e = lfq_baq_emit(RC(k), query[1], ql);
            const double f0 = e * bM, f1 = LFQ_BAQ_EI * bI;
            FQ(1, u + 0) = f0;
            FQ(1, u + 1) = f1;
            sum += f0 + f1;
        }
        SQ(1) = sum;
        const int b_ = lfq_baq_u(bw, 1, 1), e_ = lfq_baq_u(bw, 1, end) + 2;
        for (int k = b_; k <= e_; ++k) {
            const double v = FQ(1, k) / sum;
            FQ(1, k) = v;
        }
    }
    /* the per-row scalars are loaded one row ahead (a global load is an exposed round trip at this occupancy);
     * the pending scale of row i-1 is the value just stored to RQ(i-1) and stays in a register */
    double rs_next = 1.;                             /* RQ(1) */
    int qy_next = l_query >= 2 ? (int)query[2] : 0;
    double ql_next = l_query >= 2 ? (double)A.qual2prob[iqual[2]] : 0.;
    for (int i = 2; i <= l_query; ++i) {
        double sum = 0.;
        const double qli = ql_next;
        const double rs = rs_next;                   /* pending scale of row i-1 */
        const int qyi = qy_next;
        if (i < l_query) {
            qy_next = query[i + 1];
            ql_next = A.qual2prob[iqual[i + 1]];
        }
        int beg = 1, end = l_ref, x;
        x = i - bw; beg = beg > x ? beg : x;
        x = i + bw; end = end < x ? end : x;
        const int b_ = lfq_baq_u(bw, i, beg), e_ = lfq_baq_u(bw, i, end) + 2;
        for (int u = (b_ >= 3 ? b_ - 3 : 0); u < b_; u++) {
            FQ(i, u) = 0.;
        }
        for (int u = e_ + 1; u < Wr && u <= e_ + 3; u++) {
            FQ(i, u) = 0.;
        }
    
        double m_prev = 0., d_prev = 0.;             /* cell k-1 of this row (unscaled, like the reference at that point) */
        for (int k = beg; k <= end; ++k) {
            const int u = lfq_baq_u(bw, i, k), v11 = lfq_baq_u(bw, i - 1, k - 1), v10 = lfq_baq_u(bw, i - 1, k);
            const double e = lfq_baq_emit(RC(k), qyi, qli);
            const double a0 = FQ(i - 1, v11 + 0) * rs, a1 = FQ(i - 1, v11 + 1) * rs, a2 = FQ(i - 1, v11 + 2) * rs;
            const double c0 = FQ(i - 1, v10 + 0) * rs, c1 = FQ(i - 1, v10 + 1) * rs;
            const double f0 = e * (m[0] * a0 + m[3] * a1 + m[6] * a2);
            const double f1 = LFQ_BAQ_EI * (m[1] * c0 + m[4] * c1);
            const double f2 = m[2] * m_prev + m[8] * d_prev;
            FQ(i, u + 0) = f0;
            FQ(i, u + 1) = f1;
            FQ(i, u + 2) = f2;
            m_prev = f0;
            d_prev = f2;
            sum += f0 + f1 + f2;
        }
        SQ(i) = sum;
        rs_next = 1. / sum;
        RQ(i) = rs_next;
    }
    {
        double sum = 0.;
        const double rs = RQ(l_query);
        for (int k = 1; k <= l_ref; ++k) {
            const int u = lfq_baq_u(bw, l_query, k);
            if (u < 3 || u >= bw2 * 3 + 3) continue;
            sum += (FQ(l_query, u + 0) * rs) * sM + (FQ(l_query, u + 1) * rs) * sI;
        }
        SQ(l_query + 1) = sum;
    }

    /* ---- expected reference offset of every matched query base (bam_md_ext.c:409-447) ---- */
    for (int i = 0; i < l_query; i++) {
        expect[(size_t)i * 64 + lane] = INT32_MIN;       /* not in a match block (the offset itself can be negative) */
    }
    {
        const uint32_t *cg = A.cigar + R.cigar_off;
        int x = R.pos, y = 0;
        for (int k = 0; k < R.n_cigar; ++k) {
            const int op = cg[k] & 0xf, l = cg[k] >> 4;
            if (op == 0 || op == 7 || op == 8) {
                for (int i = y; i < y + l; ++i) {
                    expect[(size_t)i * 64 + lane] = x - R.xb + (i - y);
                }
                x += l; y += l;
            } else if (op == 4 || op == 1) {
                y += l;
            } else if (op == 2) {
                x += l;
            }
        }
    }

    /* ---- indel table for idaq (bam_md_ext.c:95-234): which posterior cells each indel needs ---- */
    int n_tab = 0, n_ins = 0, n_del = 0;
    int32_t *itab = A.itab ? A.itab + (size_t)blockIdx.x * LFQ_BAQ_MAX_INDELS * 4 * 64 : nullptr;
    double *terms = A.terms ? A.terms + (size_t)blockIdx.x * LFQ_BAQ_MAX_TERMS * 64 : nullptr;
#define IT(e_, f_) itab[((size_t)(e_) * 4 + (f_)) * 64 + lane]
#define TM(t_) terms[(size_t)(t_) * 64 + lane]
    uint8_t *ai = A.ai_out ? A.ai_out + s0 : nullptr, *ad = A.ad_out ? A.ad_out + s0 : nullptr;
    if (itab) {
        const uint32_t *cg = A.cigar + R.cigar_off;
        const int xe = R.xb + l_ref;
        int x = R.pos, y = 0, n_terms = 0;
        for (int i = 0; i < l_query; i++) {
            ai[i] = '~';
            ad[i] = '~';
        }
        for (int k = 0; k < R.n_cigar; ++k) {
            const int op = cg[k] & 0xf, oplen = cg[k] >> 4;
            if (op == 0 || op == 7 || op == 8) {
                x += oplen; y += oplen;
            } else if (op == 2) {                   /* deletion; the reference's skips do not advance x (:112-114) */
                const int rpos = x, qpos = y;
                if (qpos == 0) continue;
                if (oplen > 16) continue;
                n_del += 1;
                x += oplen;
                int ref_i = x, rep = 0, rep_i = 0;
                while (ref_i < xe) {
                    if (A.ref[ref_i] != A.ref[rpos + rep_i]) break;
                    rep += 1; ref_i += 1; rep_i += 1;
                    if (rep_i >= oplen) rep_i = 0;
                }
                int nt = rep + 1;
                if (qpos + nt - 1 > l_query) nt = l_query - qpos + 1;          /* `if (qpos+j > l_qseq) break` */
                if (n_tab < LFQ_BAQ_MAX_INDELS && n_terms + nt <= LFQ_BAQ_MAX_TERMS) {
                    IT(n_tab, 0) = (qpos << 1) | 1;                            /* bit 0: deletion */
                    IT(n_tab, 1) = rpos - R.xb + 1;
                    IT(n_tab, 2) = nt;
                    IT(n_tab, 3) = n_terms;
                    for (int j = 0; j < nt; j++) {
                        TM(n_terms + j) = -1.;                                 /* "not added" */
                    }
                    n_terms += nt;
                    n_tab += 1;
                }
            } else if (op == 1) {                   /* insertion; the skips do not advance y (:181-183) */
                const int rpos = x, qpos = y;
                if (oplen > 16) continue;
                n_ins += 1;
                if (qpos == 0) continue;
                y += oplen;
                int ref_i = x, rep = 0, rep_i = 0;
                while (ref_i < xe) {
                    const unsigned b = query[1 + qpos + rep_i];                /* code -> seq_nt16_str letter */
                    if (A.ref[ref_i] != (uint8_t)lfq_seq_letter(b)) break;
                    rep += 1; ref_i += 1; rep_i += 1;
                    if (rep_i >= oplen) rep_i = 0;
                }
                int nt = rep + 1;
                if (qpos + nt > l_query) nt = l_query - qpos;                  /* `if (qpos+j+1 > l_qseq) break` */
                if (nt < 0) nt = 0;
                if (n_tab < LFQ_BAQ_MAX_INDELS && n_terms + nt <= LFQ_BAQ_MAX_TERMS) {
                    IT(n_tab, 0) = qpos << 1;
                    IT(n_tab, 1) = rpos - R.xb;
                    IT(n_tab, 2) = nt;
                    IT(n_tab, 3) = n_terms;
                    for (int j = 0; j < nt; j++) {
                        TM(n_terms + j) = -1.;
                    }
                    n_terms += nt;
                    n_tab += 1;
                }
            } else if (op == 4) {
                y += oplen;
            }
        }
        A.tag_flags[rid] = (uint8_t)((n_ins ? 1 : 0) | (n_del ? 2 : 0));
    }

    /* ---- backward (:206-238), with the MAP step of a row (:254-281) as soon as the row exists ---- */
    int cur = 0;
    for (int u = 0; u < Wr; u++) {
        BQ(0, u) = 0.;
    }
    {
        const double sl = SQ(l_query), sl1 = SQ(l_query + 1);
        for (int k = 1; k <= l_ref; ++k) {
            const int u = lfq_baq_u(bw, l_query, k);
            if (u < 3 || u >= bw2 * 3 + 3) continue;
            BQ(0, u + 0) = sM / sl / sl1;
            BQ(0, u + 1) = sI / sl / sl1;
        }
    }
    /* scalars of row i, loaded one row ahead: query[i+1], qual2prob[iqual[i+1]], s[i], 1/s[i], expect[i-1], iqual[i] */
    int p_qy = 0, p_ex = expect[(size_t)(l_query - 1) * 64 + lane], p_iq = iqual[l_query];
    double p_ql = 0., p_s = SQ(l_query), p_r = RQ(l_query);
    for (int i = l_query; i >= 1; --i) {
        const int c_qy = p_qy, c_ex = p_ex, c_iq = p_iq;
        const double c_ql = p_ql, c_s = p_s, c_r = p_r;
        if (i > 1) {
            p_qy = query[i];
            p_ql = A.qual2prob[c_iq];
            p_s = SQ(i - 1);
            p_r = RQ(i - 1);
            p_ex = expect[(size_t)(i - 2) * 64 + lane];
            p_iq = iqual[i - 1];
        }
        if (i < l_query) {
            const int nxt = cur ^ 1;                  /* row i goes to `nxt`, row i+1 is in `cur` */
            for (int u = 0; u < Wr; u++) {
                BQ(nxt, u) = 0.;
            }
            int beg = 1, end = l_ref, x;
            const double y = (i > 1), qli1 = c_ql;
            const int qyi1 = c_qy;
            x = i - bw; beg = beg > x ? beg : x;
            x = i + bw; end = end < x ? end : x;
            for (int k = end; k >= beg; --k) {
                const int u = lfq_baq_u(bw, i, k), v11 = lfq_baq_u(bw, i + 1, k + 1), v10 = lfq_baq_u(bw, i + 1, k),
                          v01 = lfq_baq_u(bw, i, k + 1);
                const double e = (k >= l_ref ? 0 : lfq_baq_emit(lfq_baq_code(refw[k + 1]), qyi1, qli1)) * BQ(cur, v11);
                const double d01 = BQ(nxt, v01 + 2);
                const double b0 = e * m[0] + LFQ_BAQ_EI * m[1] * BQ(cur, v10 + 1) + m[2] * d01;
                const double b1 = e * m[3] + LFQ_BAQ_EI * m[4] * BQ(cur, v10 + 1);
                const double b2 = (e * m[6] + m[8] * d01) * y;
                BQ(nxt, u + 0) = b0;
                BQ(nxt, u + 1) = b1;
                BQ(nxt, u + 2) = b2;
            }
            const int b_ = lfq_baq_u(bw, i, beg), e_ = lfq_baq_u(bw, i, end) + 2;
            const double ys = 1. / c_s;
            for (int k = b_; k <= e_; ++k) {
                BQ(nxt, k) = BQ(nxt, k) * ys;
            }
            cur = nxt;
        }
        /* MAP of row i */
        double sum = 0., max = 0.;
        const double rsi = c_r;
        int beg = 1, end = l_ref, x, max_k = -1;
        x = i - bw; beg = beg > x ? beg : x;
        x = i + bw; end = end < x ? end : x;
        for (int k = beg; k <= end; ++k) {
            const int u = lfq_baq_u(bw, i, k);
            double z;
            z = (FQ(i, u + 0) * rsi) * BQ(cur, u + 0); if (z > max) max = z, max_k = (k - 1) << 2 | 0; sum += z;
            z = (FQ(i, u + 1) * rsi) * BQ(cur, u + 1); if (z > max) max = z, max_k = (k - 1) << 2 | 1; sum += z;
        }
        for (int e = 0; e < n_tab; e++) {           /* pd cells of this row that an indel needs (:147-163, :207-224) */
            const int t0 = IT(e, 0), is_del = t0 & 1, qpos = t0 >> 1;
            const int j = is_del ? i - qpos : i - qpos - 1;
            if (j < 0 || j >= IT(e, 2)) continue;
            const int u = lfq_baq_u(bw, i, IT(e, 1) + j);
            if (u < 3 || u >= bw2 * 3 + 3) continue;                           /* u_within_limits */
            const int st = is_del ? 2 : 1;
            TM(IT(e, 3) + j) = (FQ(i, u + st) * rsi) * BQ(cur, u + st) * c_s;
        }
        max /= sum;
        int qk = (int)(-4.343 * log(1. - max) + .499);
        qk = qk > 100 ? 99 : qk;
        /* bam_md_ext.c:413-416 / :435-436: a base the HMM does not put where the CIGAR puts it gets 0 (extended
         * BAQ; the plain variant overwrites the 0 with q again -- reproduced); unaligned bases keep their BQ */
        const int ex = c_ex;
        int bq = c_iq;
        if (ex != INT32_MIN) {
            const bool off = (max_k & 3) != 0 || (max_k >> 2) != ex;
            bq = A.baq_extended ? (off ? 0 : qk) : qk;
        }
#ifdef LFQ_TRACE
        if (R.pos == 1 && i <= 20) printf("baq i %d max_k %d (k %d st %d) expect %d qk %d max %g sum %g bw %d l_ref %d xb %d\n", i, max_k, max_k >> 2, max_k & 3, ex, qk, max, sum, bw, l_ref, R.xb);
#endif
        out[i - 1] = (uint8_t)bq;
    }

    /* ---- extended BAQ: min of the running maxima from both ends of each match block (:437-446) ---- */
    if (A.baq_extended) {
        const uint32_t *cg = A.cigar + R.cigar_off;
        int y = 0;
        for (int k = 0; k < R.n_cigar; ++k) {
            const int op = cg[k] & 0xf, l = cg[k] >> 4;
            if (op == 0 || op == 7 || op == 8) {
                if (l > 0) {
                    uint8_t run = out[y];
                    left[(size_t)y * 64 + lane] = run;
                    for (int i = y + 1; i < y + l; ++i) {
                        run = out[i] > run ? out[i] : run;
                        left[(size_t)i * 64 + lane] = run;
                    }
                    run = out[y + l - 1];
                    rght[(size_t)(y + l - 1) * 64 + lane] = run;
                    for (int i = y + l - 2; i >= y; --i) {
                        run = out[i] > run ? out[i] : run;
                        rght[(size_t)i * 64 + lane] = run;
                    }
                    for (int i = y; i < y + l; ++i) {
                        const uint8_t a = left[(size_t)i * 64 + lane], b = rght[(size_t)i * 64 + lane];
                        out[i] = a < b ? a : b;
                    }
                }
                y += l;
            } else if (op == 4 || op == 1) {
                y += l;
            }
        }
    }
    for (int i = 0; i < l_query; ++i) {                                  /* :456-462 */
        const int v = out[i] > 93 ? 93 : out[i];
        out[i] = (uint8_t)(v + 33);
    }
    /* ---- idaq: sum each indel's terms in the reference's order (j ascending), 1 - sum -> phred char ---- */
    for (int e = 0; e < n_tab; e++) {
        const int t0 = IT(e, 0), is_del = t0 & 1, qpos = t0 >> 1, nt = IT(e, 2), off = IT(e, 3);
        double ap = 0;
        for (int j = 0; j < nt; j++) {
            const double t = TM(off + j);
            if (t >= 0.) {
                ap += t;
            }
        }
        ap = 1 - ap;
        const int qv = (ap < 0.0 + 2.220446049250313e-16) ? 126 + 1 : ((int)(-10 * log10(ap)) + 33);   /* :55-56 */
        const uint8_t ch = (uint8_t)(qv < 33 ? '!' : (qv > 126 ? '~' : qv));
        (is_del ? ad : ai)[qpos - 1] = ch;
    }
#undef IT
#undef TM
#undef FQ
#undef BQ
#undef SQ
#undef RQ
#undef LQ
#undef RC
}

/* ------------------------------------------------------------------------------------------------------------------
 * Narrow-band reads, rows in REGISTERS (lfq_baq_reg_kernel).
 *
 * With the default band of 7 a row of the HMM is 15 cells x 3 states.  In the band-relative layout -- slot j of row i
 * holds reference position k = i - bw + j -- every dependency of the recurrences has a STATIC slot distance:
 *     forward   (i, k) <- (i-1, k-1) = slot j of the old row (match / all three states), (i-1, k) = slot j + 1
 *               (insertion), (i, k-1) = slot j - 1 of the new row (deletion);
 *     backward  (i, k) <- (i+1, k+1) = slot j, (i+1, k) = slot j - 1 of the old row, (i, k+1) = slot j + 1 of the new row.
 * So the row lives in 45 doubles of registers, updated in place (ascending j forward, descending j backward) by a
 * fully unrolled loop over the 15 slots: no LDS round trip inside the dependent chain, two wavefronts per SIMD.
 * Cells outside [max(1, i - bw), min(l_ref, i + bw)] are kept at exactly 0, which is what the reference reads from
 * its calloc'ed matrices there.  Same operations in the same order on the same doubles as kpa_ext_glocal
 * (kprobaln_ext.c:134-268; -ffp-contract=off): bit-identical state / quality per base.
 *
 * Schedule.  The row index i is the same for all 64 reads of a wavefront (a read shorter than the longest one idles
 * at the ends), and the rows are cut into three ranges: the interior rows, where every read of the wavefront has all
 * 15 cells (8 <= i, i + 7 <= l_ref, i <= l_query), run a branch-free body without the per-cell bounds; the few rows
 * before and after run the general body.  Separate loops, so that the masks of the general body cost the interior
 * loop no registers: it fits 2 wavefronts per SIMD without spilling.
 *
 * Inside the row loops only the forward matrix moves through HBM (write-once in the forward pass, read-once by the
 * MAP step: (match, insertion) of a slot as one 16-byte pair, 1 KiB contiguous per wavefront and slot):
 *   - base code and quality of every row: LDS ((code | quality << 8) per row and read, staged once), the quality's
 *     probability one row ahead from the 1 KiB table (cache resident);
 *   - the reference bases of the band: a 64-bit window of sixteen 4-bit codes that shifts by one base per row; the
 *     codes that enter it come 16 rows at a time, four 4-byte loads requested 16 rows before they are needed;
 *   - 1 / s[i] and expect[i-1] of the backward pass: four rows per batch, requested four to eight rows ahead;
 *   - the BAQ byte of a row goes into the LDS slot of a row that is done, the extended-BAQ passes run there and
 *     the read's bytes leave in one burst.
 */
#define LFQ_BAQ_NB 15              /* slots of the default band: 2 * 7 + 1 */
#define LFQ_BAQ_NB_WIDE 17         /* band 8: what a read with a deletion of odd length gets (bam_md_ext.c:353-356) */

/* the reference codes of a row's slots as 4-bit fields: 64 bits hold the 15 slots of band 7 and the code that enters
 * next; band 8 takes a 128-bit word */
template <int NB> struct LfqBaqWinT { typedef unsigned long long type; };
template <> struct LfqBaqWinT<LFQ_BAQ_NB_WIDE> { typedef unsigned __int128 type; };
template <int NB, typename W>
__device__ __forceinline__ W lfq_baq_nibbles(unsigned v)        /* v in each of the NB lowest nibbles */
{
    W w = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
        w |= (W)v << (4 * j);
    }
    return w;
}
struct alignas(16) LfqBaqPair {
    double x, y;
};
/* A stored forward row in HBM (and in its LDS staging buffer): NB slots of 64 lanes x 16 bytes.  Slot t < NB / 2 holds the
 * match cells 2 t and 2 t + 1, slot NB / 2 the match and the insertion cell NB - 1, the slots after it the insertion cells
 * 2 t and 2 t + 1: the row arrays live in consecutive registers, so a pair of NEIGHBOURING cells of one array is a 16-byte
 * store straight from where it was computed (the (match, insertion) pair of one cell cost four register moves per store). */
template <int NB>
__device__ __forceinline__ void lfq_baq_store_row(LfqBaqPair *fp, const double (&O0)[NB + 1], const double (&O1)[NB + 1])
{
    constexpr int H = NB / 2;
#pragma unroll
    for (int t = 0; t < H; t++) {
        fp[(size_t)t * 64] = LfqBaqPair{O0[2 * t], O0[2 * t + 1]};
    }
    fp[(size_t)H * 64] = LfqBaqPair{O0[NB - 1], O1[NB - 1]};
#pragma unroll
    for (int t = 0; t < H; t++) {
        fp[(size_t)(H + 1 + t) * 64] = LfqBaqPair{O1[2 * t], O1[2 * t + 1]};
    }
}
/* One wavefront per SIMD: the interior row needs ~330 registers (45 + 30 doubles of rows, the transition matrix,
 * the prefetch batches), which the unified 512-entry file of gfx950 holds (256 VGPRs + AGPRs as the overflow).  At two
 * wavefronts per SIMD (256 in all) the row loops spill into scratch, whose reloads queue behind the forward matrix's
 * stores: measured 12.8 ms against 7.0 ms per 400 K reads.  (Round 3, after the row bodies shrank to 310 registers:
 * still 203 spilled at 256, 364 B of scratch per lane -- not taken.) */
#define LFQ_BAQ_WAVES 1

#define LFQ_NZ(x_) ((((x_) & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | (x_))    /* bit 7 of every byte that is not zero */
/* lfq_baq_code of the four ASCII bytes of a dword as four 4-bit fields, without a branch: a byte is A, C, G or T in either
 * case iff it equals the letter after clearing bit 5 */
__device__ __forceinline__ uint32_t lfq_baq_codes4(uint32_t v)
{
    const uint32_t u = v & 0xDFDFDFDFu;
    const uint32_t na = LFQ_NZ(u ^ 0x41414141u), nc = LFQ_NZ(u ^ 0x43434343u);
    const uint32_t ng = LFQ_NZ(u ^ 0x47474747u), nt = LFQ_NZ(u ^ 0x54545454u);
    const uint32_t ec = ~nc & 0x80808080u, eg = ~ng & 0x80808080u, et = ~nt & 0x80808080u;
    const uint32_t none = na & nc & ng & nt & 0x80808080u;
    const uint32_t cb = (ec >> 7) | (eg >> 6) | (et >> 7) | (et >> 6) | (none >> 5);   /* 1, 2, 3, 4 (A: 0) per byte */
    const uint32_t y = (cb | (cb >> 4)) & 0x00FF00FFu;
    return (y | (y >> 8)) & 0xFFFFu;
}
/* sixteen 4-bit base codes out of four ASCII dwords (byte t of dword d = code 4 d + t) */
__device__ __forceinline__ unsigned long long lfq_baq_pack16(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3)
{
    const uint32_t lo = lfq_baq_codes4(d0) | (lfq_baq_codes4(d1) << 16);
    const uint32_t hi = lfq_baq_codes4(d2) | (lfq_baq_codes4(d3) << 16);
    return (unsigned long long)lo | ((unsigned long long)hi << 32);
}

/* the ASCII bytes of reference positions p .. p + 3 of a read's window (1-based; 'N' outside 1 .. l_ref) */
__device__ __forceinline__ uint32_t lfq_baq_ref4(const uint8_t *refw, int p, int l_ref)
{
    if (p >= 1 && p + 3 <= l_ref) {
        uint32_t v;
        __builtin_memcpy(&v, refw + p, 4);           /* unaligned dword */
        return v;
    }
    uint32_t v = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int k = p + t;
        v |= (uint32_t)((k >= 1 && k <= l_ref) ? refw[k] : (uint8_t)'N') << (8 * t);
    }
    return v;
}

/* one interior row of the forward pass (all 15 cells exist for every read of the wavefront); HASN: some read has an
 * N in its window or as its base */
template <int NB, bool HASN, bool IDAQ>
__device__ __forceinline__ void lfq_baq_fwd_row(double (&O0)[NB + 1], double (&O1)[NB + 1], double (&O2)[NB + 1],
                                                typename LfqBaqWinT<NB>::type win, int qyi, double e_eq,
                                                double e_ne, double rs, const double (&m)[9], double &sum_out)
{
    typedef typename LfqBaqWinT<NB>::type WinT;
    const WinT xq = win ^ (lfq_baq_nibbles<NB, WinT>(1u) * (WinT)(unsigned)(qyi & 3));
    double sum = 0., m_prev = 0., d_prev = 0.;
#pragma unroll
    for (int j = 0; j < NB; j++) {
        double e = ((unsigned)(xq >> (4 * j)) & 15u) == 0 ? e_eq : e_ne;
        if (HASN) {
            e = (((unsigned)(win >> (4 * j)) & 4u) != 0 || qyi > 3) ? 1. : e;
        }
        const double a0 = O0[j] * rs, a1 = O1[j] * rs, a2 = O2[j] * rs;
        const double c0 = O0[j + 1] * rs, c1 = O1[j + 1] * rs;
        const double f0 = e * (m[0] * a0 + m[3] * a1 + m[6] * a2);
        const double f1 = LFQ_BAQ_EI * (m[1] * c0 + m[4] * c1);
        const double f2 = m[2] * m_prev + m[8] * d_prev;
        m_prev = f0;
        d_prev = f2;
        sum += f0 + f1 + f2;
        O0[j] = f0;
        O1[j] = f1;
        O2[j] = f2;
    }
    sum_out = sum;
}

/* The forward cells (match, insertion) of an ODD row i >= 3 from the stored cells of row i - 1, inside the backward sweep:
 * the forward pass keeps only the even rows (and row 1) in HBM -- half the bytes of the kernel's dominant stream -- and
 * this redoes the forward step of the row in between with the operations of lfq_baq_fwd_row / the masked forward body
 * in the same order on the same values: the deletion cells of row i - 1 first (f2(k) = m2 f0(k-1) + m8 f2(k-1), what
 * the forward pass had in O2), then f0 / f1 of row i.  Bit-identical by construction (-ffp-contract=off, no
 * reassociation).  Cells a row does not have (beyond the end of the reference) come out as don't-care values: every
 * reader masks them, and they only ever feed cells that do not exist either. */
template <int NB, bool HASN>
__device__ __forceinline__ void lfq_baq_refwd_row(const double (&A0)[NB], const double (&A1)[NB],
                                                  typename LfqBaqWinT<NB>::type win, int qyi, double e_eq, double e_ne,
                                                  double rs, const double (&m)[9], double (&F0)[NB], double (&F1)[NB])
{
    typedef typename LfqBaqWinT<NB>::type WinT;
    const WinT xq = win ^ (lfq_baq_nibbles<NB, WinT>(1u) * (WinT)(unsigned)(qyi & 3));
    double mp = 0., dp = 0.;
#pragma unroll
    for (int j = 0; j < NB; j++) {
        double e = ((unsigned)(xq >> (4 * j)) & 15u) == 0 ? e_eq : e_ne;
        if (HASN) {
            e = (((unsigned)(win >> (4 * j)) & 4u) != 0 || qyi > 3) ? 1. : e;
        }
        const double a2u = m[2] * mp + m[8] * dp;   /* deletion cell j of row i - 1 (row i - 1 >= 2: it has them) */
        mp = A0[j];
        dp = a2u;
        const double a0 = A0[j] * rs, a1 = A1[j] * rs, a2 = a2u * rs;
        const double c0 = (j + 1 < NB ? A0[j + 1 < NB ? j + 1 : j] : 0.) * rs, c1 = (j + 1 < NB ? A1[j + 1 < NB ? j + 1 : j] : 0.) * rs;
        F0[j] = e * (m[0] * a0 + m[3] * a1 + m[6] * a2);
        F1[j] = LFQ_BAQ_EI * (m[1] * c0 + m[4] * c1);
    }
}

/* MAP step of row i (kprobaln_ext.c:254-281): z = f b over the cells in ascending k, match before insertion, the first
 * maximum wins (z > max).  ODD: the forward cells of the row are not in HBM -- each is recomputed right here from the
 * stored row below (G = row i - 1) exactly as lfq_baq_refwd_row does, and consumed at once, so that no second row of
 * forward cells occupies registers.  MASKED: the edge rows, cells outside jmin .. jmax count as absent. */
template <int NB, bool ODD, bool HASN, bool MASKED>
__device__ __forceinline__ void lfq_baq_map_row(const double (&O0)[NB + 1], const double (&O1)[NB + 1], const double (&G0)[NB],
                                                const double (&G1)[NB], typename LfqBaqWinT<NB>::type fwin, int f_qy,
                                                double f_eq, double f_ne, double f_rs, const double (&m)[9], double rsi,
                                                int jmin, int jmax, double &sum, double &max, int &max_u)
{
    typedef typename LfqBaqWinT<NB>::type WinT;
    const WinT xq = fwin ^ (lfq_baq_nibbles<NB, WinT>(1u) * (WinT)(unsigned)(f_qy & 3));
    double mp = 0., dp = 0.;
#pragma unroll
    for (int j = 0; j < NB; j++) {
        double f0, f1;
        if (ODD) {
            double e = ((unsigned)(xq >> (4 * j)) & 15u) == 0 ? f_eq : f_ne;
            if (HASN) {
                e = (((unsigned)(fwin >> (4 * j)) & 4u) != 0 || f_qy > 3) ? 1. : e;
            }
            const double a2u = m[2] * mp + m[8] * dp;
            mp = G0[j];
            dp = a2u;
            const double a0 = G0[j] * f_rs, a1 = G1[j] * f_rs, a2 = a2u * f_rs;
            const double c0 = (j + 1 < NB ? G0[j + 1 < NB ? j + 1 : j] : 0.) * f_rs;
            const double c1 = (j + 1 < NB ? G1[j + 1 < NB ? j + 1 : j] : 0.) * f_rs;
            f0 = e * (m[0] * a0 + m[3] * a1 + m[6] * a2);
            f1 = LFQ_BAQ_EI * (m[1] * c0 + m[4] * c1);
        } else {
            f0 = G0[j];
            f1 = G1[j];
        }
        const bool valid = !MASKED || (j >= jmin && j <= jmax);
        const double z0 = valid ? (f0 * rsi) * O0[j] : -1.;              /* -1.: never the maximum, adds 0. */
        max_u = z0 > max ? 4 * j : max_u;
        max = __builtin_fmax(z0, max);               /* = z0 > max ? z0 : max (no NaNs here), one instruction */
        sum += valid ? z0 : 0.;
        const double z1 = valid ? (f1 * rsi) * O1[j] : -1.;
        max_u = z1 > max ? 4 * j + 1 : max_u;
        max = __builtin_fmax(z1, max);
        sum += valid ? z1 : 0.;
    }
}

/* one interior row of the backward pass: O <- row i from row i + 1 (both scaled), all 15 cells */
template <int NB, bool HASN, bool IDAQ>
__device__ __forceinline__ void lfq_baq_bwd_row(double (&O0)[NB + 1], double (&O1)[NB + 1], double (&O2)[NB + 1],
                                                typename LfqBaqWinT<NB>::type win, int qy1, double e_eq,
                                                double e_ne, double ys, const double (&m)[9])
{
    typedef typename LfqBaqWinT<NB>::type WinT;
    const WinT xq = win ^ (lfq_baq_nibbles<NB, WinT>(1u) * (WinT)(unsigned)(qy1 & 3));
    double d01 = 0.;
#pragma unroll
    for (int j = NB - 1; j >= 0; --j) {
        double em = ((unsigned)(xq >> (4 * j)) & 15u) == 0 ? e_eq : e_ne;
        if (HASN) {
            em = (((unsigned)(win >> (4 * j)) & 4u) != 0 || qy1 > 3) ? 1. : em;
        }
        const double o101 = j > 0 ? O1[j > 0 ? j - 1 : 0] : 0.;
        const double e = em * O0[j];
        const double b0 = e * m[0] + LFQ_BAQ_EI * m[1] * o101 + m[2] * d01;
        const double b1 = e * m[3] + LFQ_BAQ_EI * m[4] * o101;
        const double b2 = e * m[6] + m[8] * d01;                          /* times y = 1. (i > 1) */
        O0[j] = b0 * ys;
        O1[j] = b1 * ys;
        if (IDAQ) {
            O2[j] = b2 * ys;
        }
        d01 = b2;
    }
}

/* IDAQ = false: lb only -- no indel table, no deletion row in the backward pass (it exists there only as the running
 * d01 of the recurrence) */
/* s[l_query + 1] (kprobaln_ext.c:184-189): the cells of the last row within the band limits, k ascending */
template <int NB>
__device__ __forceinline__ double lfq_baq_sfin(const double (&O0)[NB + 1], const double (&O1)[NB + 1], double rs,
                                               double sM, double sI, int l_query, int l_ref, int bw)
{
    double sum = 0.;
    const int xl = l_query - bw > 0 ? l_query - bw : 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const int k = l_query - bw + j;
        if (j < 2 * bw + 1 && k >= 1 && k <= l_ref && k >= xl && k <= xl + 2 * bw) {
            sum += (O0[j] * rs) * sM + (O1[j] * rs) * sI;
        }
    }
    return sum;
}

/* Which wavefronts of a launch of lfq_baq_reg_kernel meet an N at all -- among a read's bases or in its reference window?
 * Nearly none.  One byte per wavefront; the register kernel is instantiated twice, with and without the N case of the
 * emission in its row bodies (the same arithmetic on the same values where there is no N, about 7 % fewer instructions
 * per row), both are launched over the same grid, and a wavefront leaves at once in the instantiation that is not its own.
 * Four bytes at a time: a base code is N iff it is above 3; a reference byte is none of A, C, G, T in either case iff it
 * differs from all four after clearing bit 5 (lfq_baq_code). */
__global__ __launch_bounds__(64) void lfq_baq_nflag_kernel(LfqBaqArgs A, int64_t n_launch)
{
    const int lane = (int)threadIdx.x;
    const int64_t ridx = (int64_t)blockIdx.x * 64 + lane;
    bool any_n = false;
    if (ridx < n_launch) {
        const int64_t rid = A.order ? (int64_t)A.order[A.first_read + ridx] : A.first_read + ridx;
        const LfqBaqRead R = lfq_baq_read_of(A, rid);
        if (R.l_qseq > 0 && R.l_ref > 0) {
            const uint8_t *q = A.seq + A.seq_off[rid], *rp = A.ref + R.xb;
            /* sixteen bytes per load (the base array carries 16 bytes of padding; the window's tail goes byte by byte) */
            uint32_t bad = 0;
            int p = 0;
            for (; p < R.l_qseq; p += 16) {
                uint4 v;
                __builtin_memcpy(&v, q + p, 16);
                const int left = R.l_qseq - p;       /* bytes of this read in the load */
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int nb = left - 4 * t;
                    const uint32_t keep = nb >= 4 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u));
                    bad |= w[t] & keep & 0xFCFCFCFCu;
                }
            }
            any_n = bad != 0;
            bad = 0;
            for (p = 0; p + 16 <= R.l_ref; p += 16) {       /* (the window lies inside the contig) */
                uint4 v;
                __builtin_memcpy(&v, rp + p, 16);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const uint32_t u = w[t] & 0xDFDFDFDFu;
                    bad |= LFQ_NZ(u ^ 0x41414141u) & LFQ_NZ(u ^ 0x43434343u) & LFQ_NZ(u ^ 0x47474747u) & LFQ_NZ(u ^ 0x54545454u);
                }
            }
#undef LFQ_NZ
            for (; p < R.l_ref; p++) {
                any_n = any_n || lfq_baq_code(rp[p]) > 3;
            }
            any_n = any_n || (bad & 0x80808080u) != 0;
        }
    }
    const bool wave_n = __any(any_n) != 0;
    if (lane == 0) {
        A.nflag[blockIdx.x] = wave_n ? 1 : 0;
    }
}

template <int NB, bool IDAQ, bool HN = true>
__global__ __launch_bounds__(64, LFQ_BAQ_WAVES) void lfq_baq_reg_kernel(LfqBaqArgs A, int64_t n_launch)
{
    if (A.nflag && (A.nflag[blockIdx.x] != 0) != HN) {
        return;                                      /* the other instantiation's wavefront (lfq_baq_nflag_kernel) */
    }
    typedef typename LfqBaqWinT<NB>::type WinT;
    constexpr int BWF = (NB - 1) / 2;               /* the band this instantiation holds in full */
    /* dynamic LDS: [NB][64] (match, insertion) pairs = the stored forward row the backward sweep needs next, brought in by
     * global_load_lds (HBM -> LDS without passing through registers); then [lds_rows + 2][64] base code | quality << 8 of
     * row i, later the BAQ bytes */
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    LfqBaqPair *const s_g = reinterpret_cast<LfqBaqPair *>(s_dyn);
    __attribute__((address_space(3))) unsigned char *const s_g_lds = (__attribute__((address_space(3))) unsigned char *)s_dyn;
    uint16_t *const s_rowq = reinterpret_cast<uint16_t *>(s_dyn + (size_t)NB * 64 * sizeof(LfqBaqPair));
    const int lane = (int)threadIdx.x;
    const int64_t ridx = (int64_t)blockIdx.x * 64 + lane;
    const bool live = ridx < n_launch;
    const int64_t rid = A.order ? (int64_t)A.order[A.first_read + (live ? ridx : 0)] : A.first_read + (live ? ridx : 0);
    const LfqBaqRead R = lfq_baq_read_of(A, rid);
    const int W = A.W, rows = A.rows;
    double *F = A.scratch + (size_t)blockIdx.x * ((size_t)rows * W + 2 * (size_t)W + 2 * ((size_t)rows + 2)) * 64;
    double *S = F + (size_t)rows * W * 64 + 2 * (size_t)W * 64;
    int32_t *expect = A.expect + (size_t)blockIdx.x * rows * 64;
    /* forward row i in the scratch: the match and insertion cells as 16-byte pairs in the slot order of lfq_baq_store_row;
     * the deletion cells are not stored: the few an indel's quality needs are recomputed from the row's match cells */
    /* (only row 1 and the even rows are stored: row i sits at index i >> 1, NB pairs apart -- a wavefront's forward matrix is
     * one dense run of (rows / 2 + 1) x NB KiB at the start of its scratch slot, which is sized for the all-HBM kernel's rows;
     * the pitch of the rows -- 15, 16, 25.5 KiB, every row or every second -- makes no difference in time) */
#define FP(i_) ((LfqBaqPair *)(F + (size_t)((i_) >> 1) * (2 * NB) * 64) + lane)
#define SQ(i_) S[(size_t)(i_) * 64 + lane]
#define RQ(i_) S[(size_t)(rows + 2 + (i_)) * 64 + lane]
#define ROWQ(i_) ((int)s_rowq[(size_t)(i_) * 64 + lane])
    /* a lane without a read runs along on zeros (its scratch column is its own) and writes no result */
    const bool act = live && R.l_qseq > 0 && R.l_ref > 0;
    const int l_query = act ? R.l_qseq : 0, l_ref = act ? R.l_ref : 1;
    const int64_t s0 = A.seq_off[rid];
    const uint8_t *query = A.seq + s0 - 1, *iqual = A.qual + s0 - 1;     /* 1-based like the reference */
    const uint8_t *refw = A.ref + R.xb - 1;
    uint8_t *out = A.lb_out + s0;
    int Lmax = l_query;                              /* rows of the longest read of the wavefront */
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(Lmax, off);
        Lmax = o > Lmax ? o : Lmax;
    }
    /* (base code | quality << 8) of every row into LDS: 64 bases per step as eight unaligned 16-byte loads of the read's own
     * bytes, all in flight before the first LDS write (one byte pair per iteration waited out a full memory round trip
     * 150 times per wavefront: a quarter of the forward pass) */
    s_rowq[lane] = (uint16_t)4;                      /* row 0 */
    for (int i0 = 1; i0 <= Lmax + 1; i0 += 64) {
        uint4 qv[4], bv[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            qv[c] = bv[c] = make_uint4(0u, 0u, 0u, 0u);
            if (i0 + 16 * c <= l_query) {            /* the arrays carry 16 bytes of padding behind the last read */
                __builtin_memcpy(&qv[c], query + i0 + 16 * c, 16);
                __builtin_memcpy(&bv[c], iqual + i0 + 16 * c, 16);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t qw[4] = {qv[c].x, qv[c].y, qv[c].z, qv[c].w}, bw_[4] = {bv[c].x, bv[c].y, bv[c].z, bv[c].w};
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const int i = i0 + 16 * c + t;
                if (i <= Lmax + 1) {
                    const uint32_t qb = (qw[t >> 2] >> (8 * (t & 3))) & 0xffu, bb = (bw_[t >> 2] >> (8 * (t & 3))) & 0xffu;
                    s_rowq[(size_t)i * 64 + lane] = i <= l_query ? (uint16_t)(qb | (bb << 8)) : (uint16_t)4;
                }
            }
        }
    }
    /* the quality table in LDS: a row's lookup then waits on lgkmcnt only.  As a global load it counted in vmcnt, which
     * retires in order on this ISA -- every row waited for its table entry behind the forward matrix's stores (forward
     * pass) or behind the next stored row's loads (backward sweep), i.e. the full HBM latency the prefetching was there
     * to hide, once per row. */
    __shared__ double s_q2p[256];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        s_q2p[t * 64 + lane] = (double)A.qual2prob[t * 64 + lane];
    }
    __syncthreads();
    int bw = l_ref > l_query ? l_ref : l_query;                          /* kprobaln_ext.c:99-101 */
    if (bw > R.bw) bw = R.bw;
    if (bw < abs(l_ref - l_query)) bw = abs(l_ref - l_query);            /* <= 7: the host sends only such reads here */
    if (!act) bw = BWF;
    const int bw2 = bw * 2 + 1;
    const float par_d = A.par_d, par_e = A.par_e;                        /* kpa_ext_par_t: lfq_set_baq_hmm_params */
    double m[9];
    const double sM = 1. / (2 * l_query + 2), sI = sM;                   /* :127-132 */
    m[0] = (1 - par_d - par_d) * (1 - sM); m[1] = m[2] = par_d * (1 - sM);
    m[3] = (1 - par_e) * (1 - sI); m[4] = par_e * (1 - sI); m[5] = 0.;
    m[6] = 1 - par_e; m[7] = 0.; m[8] = par_e;
    const double bM = (1 - par_d) / l_ref, bI = par_d / l_ref;
    /* interior rows: [8, f_hi] forward, [8, b_hi] backward -- every read of the wavefront has all 15 cells there (and,
     * backward, is neither at its last row nor at the last reference position) */
    int f_hi = 0, b_hi = 0;
    {
        int fh = act ? (bw == BWF ? (l_query < l_ref - BWF ? l_query : l_ref - BWF) : 0) : 1 << 30;
        int bh = act ? (bw == BWF ? (l_query - 1 < l_ref - BWF - 1 ? l_query - 1 : l_ref - BWF - 1) : 0) : 1 << 30;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int o1 = __shfl_xor(fh, off), o2 = __shfl_xor(bh, off);
            fh = o1 < fh ? o1 : fh;
            bh = o2 < bh ? o2 : bh;
        }
        f_hi = fh < BWF + 1 ? 0 : fh;                /* no interior range: the masked body takes every row */
        b_hi = bh < BWF + 1 ? 0 : bh;
    }

    /* the row: slot j = k - i + bw; O[NB] is the always-zero slot beyond the band */
    double O0[NB + 1], O1[NB + 1], O2[NB + 1];
#pragma unroll
    for (int j = 0; j <= NB; j++) {
        O0[j] = O1[j] = O2[j] = 0.;
    }
    /* ---- forward (:134-190): rows >= 2 stay unscaled in the registers and in HBM, 1 / s[i] is applied by the reader ---- */
    RQ(0) = 1.;
    RQ(1) = 1.;
    double s_row1 = 1., s_last = 1., s_fin = 0.;     /* s[1], s[l_query], s[l_query + 1] */
    {
        /* row 1 (:141-157): k = 1 .. min(l_ref, bw + 1) -> slots bw .. 2 bw */
        double sum = 0.;
        const int end = l_ref < bw + 1 ? l_ref : bw + 1;
        const double ql = s_q2p[ROWQ(1) >> 8];
        const int qy1 = ROWQ(1) & 0xff;
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int k = 1 - bw + j;
            if (j < bw2 && k >= 1 && k <= end) {
                const double e = lfq_baq_emit(lfq_baq_code(refw[k]), qy1, ql);
                const double f0 = e * bM, f1 = LFQ_BAQ_EI * bI;
                O0[j] = f0;
                O1[j] = f1;
                sum += f0 + f1;
            }
        }
        s_row1 = sum;
        s_last = sum;
        if (IDAQ) {
            SQ(1) = sum;
        }
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int k = 1 - bw + j;
            if (j < bw2 && k >= 1 && k <= end) {
                O0[j] = O0[j] / sum;
                O1[j] = O1[j] / sum;
            }
        }
        lfq_baq_store_row<NB>(FP(1), O0, O1);        /* cells the row does not have: 0., never looked at */
    }
    /* codes of reference positions i - bw .. i - bw + NB of the row about to be computed, 4 bits each (slot j = nibble j,
     * nibble NB = the code that becomes slot NB - 1 of the next row); the code that enters after row i (position
     * i - bw + NB + 1) is nibble i & 15 of `nxt`, the codes of the 16 rows after that are on their way in `pd` */
    WinT win = 0;
    unsigned long long nxt;
    uint32_t pd0, pd1, pd2, pd3;
#pragma unroll
    for (int q = 0; q < (NB + 1 + 3) / 4; q++) {     /* positions 2 - bw .. 2 - bw + NB: the slots of row 2 and the lookahead */
        const uint32_t d = lfq_baq_ref4(refw, 2 - bw + 4 * q, l_ref);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (4 * q + t <= NB) {
                win |= (WinT)(unsigned)lfq_baq_code((int)((d >> (8 * t)) & 0xffu)) << (4 * (4 * q + t));
            }
        }
    }
    {
        const int p = NB + 1 - bw;                   /* the code entering after row i: position i - bw + NB + 1 */
        nxt = lfq_baq_pack16(lfq_baq_ref4(refw, p, l_ref), lfq_baq_ref4(refw, p + 4, l_ref), lfq_baq_ref4(refw, p + 8, l_ref),
                             lfq_baq_ref4(refw, p + 12, l_ref));
        pd0 = lfq_baq_ref4(refw, p + 16, l_ref); pd1 = lfq_baq_ref4(refw, p + 20, l_ref);
        pd2 = lfq_baq_ref4(refw, p + 24, l_ref); pd3 = lfq_baq_ref4(refw, p + 28, l_ref);
    }
    double rs_next = 1.;                             /* RQ(1) */
    /* base code and quality of a row come out of LDS one row ahead */
    int rq_next = ROWQ(2);
    double ql_next = s_q2p[rq_next >> 8];
    if (l_query == 1) {
        s_fin = lfq_baq_sfin<NB>(O0, O1, 1., sM, sI, l_query, l_ref, bw);
    }
    /* One row of the forward pass; INTERIOR: all 15 cells for every read of the wavefront.  The interior rows run in a loop
     * of their own below: with both bodies in one loop the row arrays met in different registers at its end and every row
     * paid ~30 register copies for it. */
    auto fwd_step = [&](const int i, auto interior_tag) __attribute__((always_inline)) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        double sum = 0.;
        const double qli = ql_next;
        const double rs = rs_next;                   /* pending scale of row i-1 */
        const int qyi = rq_next & 0xff;
        rq_next = ROWQ(i + 1);
        ql_next = s_q2p[rq_next >> 8];
        const int code_in = (int)((nxt >> (4 * (i & 15))) & 15ull);      /* enters the window for row i + 1 */
        const double e_eq = 1. - qli, e_ne = qli * LFQ_BAQ_EM;           /* lfq_baq_emit's two non-trivial values */
        LfqBaqPair *fp = FP(i);
        const bool store = (i & 1) == 0;             /* odd rows >= 3 are recomputed by the backward sweep (lfq_baq_refwd_row) */
        if constexpr (INTERIOR) {                    /* interior row: all 15 cells, for every read of the wavefront */
            const bool has_n = qyi > 3 || (win & lfq_baq_nibbles<NB, WinT>(4u)) != 0;
            /* ONE instantiation of the row (the N case always handled: a handful of instructions per slot): with two or
             * four variants of the body the compiler kept the row in different registers in each and paid ~100 register
             * moves per row at the joins.  The row is stored after it is complete, from the registers it lives in. */
            (void)has_n;
            lfq_baq_fwd_row<NB, HN, IDAQ>(O0, O1, O2, win, qyi, e_eq, e_ne, rs, m, sum);
            if (store) {                             /* wave-uniform: only the even rows go to HBM */
                lfq_baq_store_row<NB>(fp, O0, O1);
            }
        } else {
            /* the same arithmetic with the cells beyond the end of the reference masked out.  Cells before its start
             * (k < 1) need no mask: what they read of row i - 1 is 0, so they come out as exactly 0, like the
             * reference's untouched cells.  A read past its last row (i > l_query) has no valid cell. */
            const int jmax = i <= l_query ? (l_ref - i + bw < 2 * bw ? l_ref - i + bw : 2 * bw) : -1;
            double m_prev = 0., d_prev = 0.;         /* cell k-1 of this row (unscaled, like the reference at that point) */
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const bool valid = j <= jmax;
                const int r = (int)((unsigned)(win >> (4 * j)) & 15u);
                const double e = (r > 3 || qyi > 3) ? 1. : (r == qyi ? e_eq : e_ne);
                const double a0 = O0[j] * rs, a1 = O1[j] * rs, a2 = O2[j] * rs;
                const double c0 = O0[j + 1] * rs, c1 = O1[j + 1] * rs;
                const double f0 = e * (m[0] * a0 + m[3] * a1 + m[6] * a2);
                const double f1 = LFQ_BAQ_EI * (m[1] * c0 + m[4] * c1);
                const double f2 = m[2] * m_prev + m[8] * d_prev;
                m_prev = f0;
                d_prev = f2;
                sum += valid ? f0 + f1 + f2 : 0.;    /* + 0. leaves the sum as it is */
                O0[j] = valid ? f0 : 0.;
                O1[j] = valid ? f1 : 0.;
                O2[j] = valid ? f2 : 0.;
            }
            if (store) {                             /* the whole row, 0. where it has no cell (every reader masks those) */
                lfq_baq_store_row<NB>(fp, O0, O1);
            }
        }
        win = (win >> 4) | ((WinT)(unsigned)code_in << (4 * NB));
        if ((i & 15) == 15) {                        /* the next 16 codes have had 16 rows to arrive; request the ones after */
            nxt = lfq_baq_pack16(pd0, pd1, pd2, pd3);
            const int p = i + 18 + NB - bw;          /* row i + 17's code: position (i + 17) - bw + NB + 1 */
            pd0 = lfq_baq_ref4(refw, p, l_ref); pd1 = lfq_baq_ref4(refw, p + 4, l_ref);
            pd2 = lfq_baq_ref4(refw, p + 8, l_ref); pd3 = lfq_baq_ref4(refw, p + 12, l_ref);
        }
        if (i <= l_query) {
            if (IDAQ) {
                SQ(i) = sum;
            }
            s_last = sum;
            rs_next = 1. / sum;
            RQ(i) = rs_next;
        }
        if (__any(i == l_query)) {                   /* some read's last row: its s[l_query + 1] while the row is there */
            const double v = lfq_baq_sfin<NB>(O0, O1, rs_next, sM, sI, l_query, l_ref, bw);
            s_fin = i == l_query ? v : s_fin;
        }
    };
    for (int i = 2; i <= Lmax;) {
        if (i >= BWF + 1 && i <= f_hi) {
            do {
                fwd_step(i, std::true_type{});
                ++i;
            } while (i <= f_hi);
        } else {
            fwd_step(i, std::false_type{});
            ++i;
        }
    }

    /* ---- expected reference offset of every matched query base (bam_md_ext.c:409-447) ---- */
    for (int i = 0; i < l_query; i++) {
        expect[(size_t)i * 64 + lane] = INT32_MIN;       /* not in a match block (the offset itself can be negative) */
    }
    if (act) {
        const uint32_t *cg = A.cigar + R.cigar_off;
        int x = R.pos, y = 0;
        for (int k = 0; k < R.n_cigar; ++k) {
            const int op = cg[k] & 0xf, l = cg[k] >> 4;
            if (op == 0 || op == 7 || op == 8) {
                for (int i = y; i < y + l; ++i) {
                    expect[(size_t)i * 64 + lane] = x - R.xb + (i - y);
                }
                x += l; y += l;
            } else if (op == 4 || op == 1) {
                y += l;
            } else if (op == 2) {
                x += l;
            }
        }
    }

    /* ---- indel table for idaq (bam_md_ext.c:95-234): which posterior cells each indel needs ---- */
    int n_tab = 0, n_ins = 0, n_del = 0;
    int it_lo = 1 << 30, it_hi = -1;               /* rows at which some kept indel of this read needs a posterior cell */
    int32_t *itab = IDAQ && A.itab ? A.itab + (size_t)blockIdx.x * LFQ_BAQ_MAX_INDELS * 4 * 64 : nullptr;
    double *terms = A.terms ? A.terms + (size_t)blockIdx.x * LFQ_BAQ_MAX_TERMS * 64 : nullptr;
#define IT(e_, f_) itab[((size_t)(e_) * 4 + (f_)) * 64 + lane]
#define TM(t_) terms[(size_t)(t_) * 64 + lane]
    uint8_t *ai = A.ai_out ? A.ai_out + s0 : nullptr, *ad = A.ad_out ? A.ad_out + s0 : nullptr;
    if (IDAQ && act && itab) {
        const uint32_t *cg = A.cigar + R.cigar_off;
        const int xe = R.xb + l_ref;
        int x = R.pos, y = 0, n_terms = 0;
        for (int i = 0; i < l_query; i++) {
            ai[i] = '~';
            ad[i] = '~';
        }
        for (int k = 0; k < R.n_cigar; ++k) {
            const int op = cg[k] & 0xf, oplen = cg[k] >> 4;
            if (op == 0 || op == 7 || op == 8) {
                x += oplen; y += oplen;
            } else if (op == 2) {                   /* deletion; the reference's skips do not advance x (:112-114) */
                const int rpos = x, qpos = y;
                if (qpos == 0) continue;
                if (oplen > 16) continue;
                n_del += 1;
                x += oplen;
                int ref_i = x, rep = 0, rep_i = 0;
                while (ref_i < xe) {
                    if (A.ref[ref_i] != A.ref[rpos + rep_i]) break;
                    rep += 1; ref_i += 1; rep_i += 1;
                    if (rep_i >= oplen) rep_i = 0;
                }
                int nt = rep + 1;
                if (qpos + nt - 1 > l_query) nt = l_query - qpos + 1;          /* `if (qpos+j > l_qseq) break` */
                if (n_tab < LFQ_BAQ_MAX_INDELS && n_terms + nt <= LFQ_BAQ_MAX_TERMS) {
                    it_lo = min(it_lo, qpos);
                    it_hi = max(it_hi, qpos + nt - 1);
                    IT(n_tab, 0) = (qpos << 1) | 1;                            /* bit 0: deletion */
                    IT(n_tab, 1) = rpos - R.xb + 1;
                    IT(n_tab, 2) = nt;
                    IT(n_tab, 3) = n_terms;
                    for (int j = 0; j < nt; j++) {
                        TM(n_terms + j) = -1.;                                 /* "not added" */
                    }
                    n_terms += nt;
                    n_tab += 1;
                }
            } else if (op == 1) {                   /* insertion; the skips do not advance y (:181-183) */
                const int rpos = x, qpos = y;
                if (oplen > 16) continue;
                n_ins += 1;
                if (qpos == 0) continue;
                y += oplen;
                int ref_i = x, rep = 0, rep_i = 0;
                while (ref_i < xe) {
                    const unsigned b = query[1 + qpos + rep_i];                /* code -> seq_nt16_str letter */
                    if (A.ref[ref_i] != (uint8_t)lfq_seq_letter(b)) break;
                    rep += 1; ref_i += 1; rep_i += 1;
                    if (rep_i >= oplen) rep_i = 0;
                }
                int nt = rep + 1;
                if (qpos + nt > l_query) nt = l_query - qpos;                  /* `if (qpos+j+1 > l_qseq) break` */
                if (nt < 0) nt = 0;
                if (n_tab < LFQ_BAQ_MAX_INDELS && n_terms + nt <= LFQ_BAQ_MAX_TERMS) {
                    it_lo = min(it_lo, qpos + 1);
                    it_hi = max(it_hi, qpos + nt);
                    IT(n_tab, 0) = qpos << 1;
                    IT(n_tab, 1) = rpos - R.xb;
                    IT(n_tab, 2) = nt;
                    IT(n_tab, 3) = n_terms;
                    for (int j = 0; j < nt; j++) {
                        TM(n_terms + j) = -1.;
                    }
                    n_terms += nt;
                    n_tab += 1;
                }
            } else if (op == 4) {
                y += oplen;
            }
        }
        A.tag_flags[rid] = (uint8_t)((n_ins ? 1 : 0) | (n_del ? 2 : 0));
    }


    /* ---- backward (:206-238), with the MAP step of a row (:254-281) as soon as the row exists ---- */
    const double b_init0 = sM / s_last / s_fin, b_init1 = sI / s_last / s_fin;
    /* codes of positions i - bw + 1 .. i - bw + NB for the row about to be computed (the emission of cell k looks at
     * k + 1); after row i the code of position i - bw enters: nibble i & 15 of `nxt` */
    win = 0;
#pragma unroll
    for (int q = 0; q < (NB + 3) / 4; q++) {
        const uint32_t d = lfq_baq_ref4(refw, Lmax - bw + 1 + 4 * q, l_ref);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (4 * q + t < NB) {
                win |= (WinT)(unsigned)lfq_baq_code((int)((d >> (8 * t)) & 0xffu)) << (4 * (4 * q + t));
            }
        }
    }
    {
        const int p = ((Lmax >> 4) << 4) - bw;
        nxt = lfq_baq_pack16(lfq_baq_ref4(refw, p, l_ref), lfq_baq_ref4(refw, p + 4, l_ref), lfq_baq_ref4(refw, p + 8, l_ref),
                             lfq_baq_ref4(refw, p + 12, l_ref));
        pd0 = lfq_baq_ref4(refw, p - 16, l_ref); pd1 = lfq_baq_ref4(refw, p - 12, l_ref);
        pd2 = lfq_baq_ref4(refw, p - 8, l_ref); pd3 = lfq_baq_ref4(refw, p - 4, l_ref);
    }
    /* 1 / s[i] and expect[i-1] in batches of four rows (rows 4 q .. 4 q + 3), the batch after the current one requested
     * four to eight rows before its first use */
    double rA0, rA1, rA2, rA3, rB0, rB1, rB2, rB3;
    int eA0, eA1, eA2, eA3, eB0, eB1, eB2, eB3;
#define LFQ_BAQ_CLAMP(x_) ((x_) > l_query ? (l_query < 1 ? 1 : l_query) : ((x_) < 1 ? 1 : (x_)))
#define LFQ_BAQ_BATCH(q_, r0, r1, r2, r3, e0, e1, e2, e3) \
    do { \
        const int b_ = 4 * (q_); \
        const int i0_ = LFQ_BAQ_CLAMP(b_), i1_ = LFQ_BAQ_CLAMP(b_ + 1), i2_ = LFQ_BAQ_CLAMP(b_ + 2), i3_ = LFQ_BAQ_CLAMP(b_ + 3); \
        r0 = RQ(i0_); r1 = RQ(i1_); r2 = RQ(i2_); r3 = RQ(i3_); \
        e0 = expect[(size_t)(i0_ - 1) * 64 + lane]; e1 = expect[(size_t)(i1_ - 1) * 64 + lane]; \
        e2 = expect[(size_t)(i2_ - 1) * 64 + lane]; e3 = expect[(size_t)(i3_ - 1) * 64 + lane]; \
    } while (0)
    LFQ_BAQ_BATCH(Lmax >> 2, rA0, rA1, rA2, rA3, eA0, eA1, eA2, eA3);
    LFQ_BAQ_BATCH((Lmax >> 2) - 1, rB0, rB1, rB2, rB3, eB0, eB1, eB2, eB3);
    /* G: the forward cells of the stored row (even, or row 1) the sweep is at -- row i itself at a stored row, row i - 1 at
     * an odd row, whose own cells are recomputed from it.  A stored row travels HBM -> LDS by global_load_lds (no
     * registers on the way, so it can be requested two rows of arithmetic before it is needed: the registers of the row
     * in use stay untouched) and LDS -> G when the sweep reaches the odd row above it; at that moment the LDS buffer is
     * free again and the stored row after it is requested. */
    double G0[NB], G1[NB];
    int g_have = -1;
    /* (macros, not lambdas: a closure called from inside the row lambda below kept the row arrays in scratch memory) */
#define LFQ_BAQ_DMA_ROW(row_) do { \
        const LfqBaqPair *gp_ = FP(row_); \
        _Pragma("unroll") \
        for (int j_ = 0; j_ < NB; j_++) { \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp_ + (size_t)j_ * 64), \
                                             (__attribute__((address_space(3))) void *)(s_g_lds + j_ * 64 * sizeof(LfqBaqPair)), 16, 0, 0); \
        } \
    } while (0)
#define LFQ_BAQ_ENSURE_G(need_) do {                 /* need_: wave-uniform */ \
        const int need__ = (need_); \
        if (need__ != g_have) { \
            __builtin_amdgcn_s_waitcnt(0x0F70);      /* vmcnt(0): the row has landed in LDS */ \
            _Pragma("unroll") \
            for (int t_ = 0; t_ < NB / 2; t_++) {    /* the slots of lfq_baq_store_row */ \
                const LfqBaqPair v_ = s_g[t_ * 64 + lane], w_ = s_g[(NB / 2 + 1 + t_) * 64 + lane]; \
                G0[2 * t_] = v_.x; \
                G0[2 * t_ + 1] = v_.y; \
                G1[2 * t_] = w_.x; \
                G1[2 * t_ + 1] = w_.y; \
            } \
            { \
                const LfqBaqPair v_ = s_g[(NB / 2) * 64 + lane]; \
                G0[NB - 1] = v_.x; \
                G1[NB - 1] = v_.y; \
            } \
            g_have = need__; \
            const int next_ = need__ == 2 ? 1 : need__ - 2; \
            if (next_ >= 1) { \
                __builtin_amdgcn_s_waitcnt(0xC07F);  /* lgkmcnt(0): the reads above are done with the buffer */ \
                LFQ_BAQ_DMA_ROW(next_); \
            } \
        } \
    } while (0)
    const int Lsw = __builtin_amdgcn_readfirstlane(Lmax);   /* the same in every lane: the sweep's counter is scalar */
    LFQ_BAQ_DMA_ROW(((Lsw & 1) == 0 || Lsw == 1) ? Lsw : Lsw - 1);
    /* one row of the backward sweep: lfq_baq_sweep_row.inc; the interior rows in a loop of their own, like the forward
     * pass's */
    for (int i = Lsw; i >= 1;) {
        if (i >= BWF + 1 && i <= b_hi) {
            do {
#define LFQ_BAQ_ROW_INTERIOR 1
#include "lfq_baq_sweep_row.inc"
#undef LFQ_BAQ_ROW_INTERIOR
                --i;
            } while (i >= BWF + 1);
        } else {
#define LFQ_BAQ_ROW_INTERIOR 0
#include "lfq_baq_sweep_row.inc"
#undef LFQ_BAQ_ROW_INTERIOR
            --i;
        }
    }
#define OUTE(i0_) s_rowq[(size_t)((i0_) + 2) * 64 + lane]

    /* ---- extended BAQ: min of the running maxima from both ends of each match block (:437-446), in the LDS slots:
     * the low byte keeps out[i], the high byte takes the maximum from the left ---- */
    if (act && A.baq_extended) {
        const uint32_t *cg = A.cigar + R.cigar_off;
        int y = 0;
        for (int k = 0; k < R.n_cigar; ++k) {
            const int op = cg[k] & 0xf, l = cg[k] >> 4;
            if (op == 0 || op == 7 || op == 8) {
                if (l > 0) {
                    int run = OUTE(y) & 0xff;
                    OUTE(y) = (uint16_t)(run | (run << 8));
                    for (int i = y + 1; i < y + l; ++i) {
                        const int o = OUTE(i) & 0xff;
                        run = o > run ? o : run;
                        OUTE(i) = (uint16_t)(o | (run << 8));
                    }
                    run = 0;
                    for (int i = y + l - 1; i >= y; --i) {
                        const int v = OUTE(i), o = v & 0xff, lf = v >> 8;
                        run = o > run ? o : run;
                        OUTE(i) = (uint16_t)(lf < run ? lf : run);
                    }
                }
                y += l;
            } else if (op == 4 || op == 1) {
                y += l;
            }
        }
    }
    for (int i = 0; i + 4 <= l_query; i += 4) {                          /* :456-462, four bytes per (unaligned) store */
        uint32_t w = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int o = OUTE(i + t) & 0xff;
            w |= (uint32_t)((o > 93 ? 93 : o) + 33) << (8 * t);
        }
        __builtin_memcpy(out + i, &w, 4);
    }
    for (int i = l_query & ~3; i < l_query; ++i) {
        const int o = OUTE(i) & 0xff;
        out[i] = (uint8_t)((o > 93 ? 93 : o) + 33);
    }
    /* ---- idaq: sum each indel's terms in the reference's order (j ascending), 1 - sum -> phred char ---- */
    for (int e = 0; IDAQ && e < n_tab; e++) {
        const int t0 = IT(e, 0), is_del = t0 & 1, qpos = t0 >> 1, nt = IT(e, 2), off = IT(e, 3);
        double ap = 0;
        for (int j = 0; j < nt; j++) {
            const double t = TM(off + j);
            if (t >= 0.) {
                ap += t;
            }
        }
        ap = 1 - ap;
        const int qv = (ap < 0.0 + 2.220446049250313e-16) ? 126 + 1 : ((int)(-10 * log10(ap)) + 33);   /* :55-56 */
        const uint8_t ch = (uint8_t)(qv < 33 ? '!' : (qv > 126 ? '~' : qv));
        (is_del ? ad : ai)[qpos - 1] = ch;
    }
#undef IT
#undef TM
#undef SQ
#undef RQ
#undef FP
#undef ROWQ
#undef OUTE
#undef LFQ_BAQ_BATCH
#undef LFQ_BAQ_CLAMP
#undef LFQ_BAQ_DMA_ROW
#undef LFQ_BAQ_ENSURE_G
}

/* lds: 0 = the all-HBM kernel (any band), 1 = band <= 7 (rows in registers / LDS), 2 = band 8 (rows in registers) */
int lfq_launch_baq(const LfqBaqArgs &a, int64_t n_launch, int lds, void *stream, int nmode)
{
    if (n_launch <= 0) {
        return LFQ_OK;
    }
    const unsigned blocks = (unsigned)((n_launch + 63) / 64);
    if (lds) {
        /* (base | quality) of every row, later the BAQ bytes (see the kernel) */
        const size_t lds_bytes = ((size_t)a.lds_rows + 2) * 64 * 2 + (size_t)(lds == 2 ? LFQ_BAQ_NB_WIDE : LFQ_BAQ_NB) * 64 * 16;
        const hipStream_t st = (hipStream_t)stream;
        if (lds == 2 && a.itab) {
            hipLaunchKernelGGL((lfq_baq_reg_kernel<LFQ_BAQ_NB_WIDE, true>), dim3(blocks), dim3(64), lds_bytes, st, a, n_launch);
        } else if (lds == 2) {
            hipLaunchKernelGGL((lfq_baq_reg_kernel<LFQ_BAQ_NB_WIDE, false>), dim3(blocks), dim3(64), lds_bytes, st, a, n_launch);
        } else if (a.itab) {
            hipLaunchKernelGGL((lfq_baq_reg_kernel<LFQ_BAQ_NB, true>), dim3(blocks), dim3(64), lds_bytes, st, a, n_launch);
        } else if (a.nflag) {
            if (nmode != 2) {
                hipLaunchKernelGGL(lfq_baq_nflag_kernel, dim3(blocks), dim3(64), 0, st, a, n_launch);
                hipLaunchKernelGGL((lfq_baq_reg_kernel<LFQ_BAQ_NB, false, false>), dim3(blocks), dim3(64), lds_bytes, st, a, n_launch);
            }
            if (nmode != 1) {
                hipLaunchKernelGGL((lfq_baq_reg_kernel<LFQ_BAQ_NB, false, true>), dim3(blocks), dim3(64), lds_bytes, st, a, n_launch);
            }
        } else {
            hipLaunchKernelGGL((lfq_baq_reg_kernel<LFQ_BAQ_NB, false>), dim3(blocks), dim3(64), lds_bytes, st, a, n_launch);
        }
    } else {
        hipLaunchKernelGGL(lfq_baq_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, a, n_launch);
    }
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}
