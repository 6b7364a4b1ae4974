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
 * lock-step per wavefront.  Narrow-band reads (the default band of 7) keep the row the recurrences read in LDS,
 * updated in place, and send the forward matrix to HBM write-once for the MAP step (lfq_baq_kernel<true>);
 * everything else keeps every row in HBM, interleaved per wavefront ([row][cell][lane]: every access of a wave is
 * one coalesced 512-byte line), with two rotating backward rows (lfq_baq_kernel<false>).  The MAP step of a row
 * is done as soon as the backward row exists, so no backward matrix is stored.  See DESIGN.md 6b.
 *
 * -ffp-contract=off (Makefile): no FMA contraction, every operation rounds like the reference's SSE2 build.
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "lfq_internal.h"

#define LFQ_BAQ_EI .25
#define LFQ_BAQ_EM .33333333333

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

/* LDS = true: the row the recurrences read (row i-1 of the forward pass, row i+1 of the backward pass) lives in
 * LDS and is updated IN PLACE, for reads whose rows fit LFQ_BAQ_LDS_CELLS cells (the default band of 7: every read
 * whose alignment does not shift by more than 7 bases).  The forward matrix still goes to HBM (stores only) for the backward pass's MAP
 * step; nothing of the backward matrix touches HBM.  LDS = false: the general kernel, every row in HBM. */
#define LFQ_BAQ_LDS_W LFQ_BAQ_LDS_CELLS
template <bool LDS>
__global__ __launch_bounds__(64) void lfq_baq_kernel(LfqBaqArgs A, int64_t n_launch)
{
    __shared__ double s_row[LDS ? LFQ_BAQ_LDS_W * 64 : 1];
    extern __shared__ uint8_t s_ref[];          /* LDS variant: [l_ref / 2 + 1][64] base codes 0..4 of the reference window, two per byte */
    const int lane = (int)threadIdx.x;
    const int64_t ridx = (int64_t)blockIdx.x * 64 + lane;
    const bool live = ridx < n_launch;
    const int64_t rid = A.order ? (int64_t)A.order[A.first_read + (live ? ridx : 0)] : A.first_read + (live ? ridx : 0);
    const LfqBaqRead R = A.reads[rid];
#define LQ(u_) s_row[(size_t)(u_) * 64 + lane]
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
    if (LDS) {
        /* a global load inside the cell loops is an exposed L2 round trip per cell at this occupancy: the window's
         * base codes go to LDS once (index 1..l_ref like refw) */
        for (int k = 0; k <= R.l_ref; k += 2) {      /* two codes per byte: k even in the low nibble */
            const int lo = k >= 1 ? lfq_baq_code(refw[k]) : 0, hi = k + 1 <= R.l_ref ? lfq_baq_code(refw[k + 1]) : 0;
            s_ref[(size_t)(k >> 1) * 64 + lane] = (uint8_t)(lo | (hi << 4));
        }
    }
#define RC(k_) (LDS ? (int)((s_ref[(size_t)((k_) >> 1) * 64 + lane] >> (((k_) & 1) * 4)) & 15) : lfq_baq_code(refw[k_]))
    int bw = l_ref > l_query ? l_ref : l_query;                          /* kprobaln_ext.c:99-101 */
    if (bw > R.bw) bw = R.bw;
    if (bw < abs(l_ref - l_query)) bw = abs(l_ref - l_query);
    const int bw2 = bw * 2 + 1;
    const int Wr = bw2 * 3 + 6;                                          /* this read's row width (<= W) */
    const float par_d = 0.00001f, par_e = 0.4f;                          /* kpa_ext_par_lofreq_illumina */
    double m[9];
    const double sM = 1. / (2 * l_query + 2), sI = sM;                   /* :127-132 */
    m[0] = (1 - par_d - par_d) * (1 - sM); m[1] = m[2] = par_d * (1 - sM);
    m[3] = (1 - par_e) * (1 - sI); m[4] = par_e * (1 - sI); m[5] = 0.;
    m[6] = 1 - par_e; m[7] = 0.; m[8] = par_e;
    const double bM = (1 - par_d) / l_ref, bI = par_d / l_ref;

    /* LDS variant: the HBM copy of the forward matrix only serves the MAP step (match / insertion cells) and the
     * idaq terms (deletion cells of reads that have a deletion): deletion cells of other reads are not written */
    bool keep_f2 = true;
    if (LDS) {
        keep_f2 = false;
        const uint32_t *cg0 = A.cigar + R.cigar_off;
        for (int k = 0; k < R.n_cigar; ++k) {
            keep_f2 = keep_f2 || ((cg0[k] & 0xf) == 2);
        }
        keep_f2 = keep_f2 && A.itab != nullptr;
    }
    /* ---- forward (:134-190) ----
     * Rows >= 2 are stored UNSCALED; the reference's `fi[k] *= 1/sum` (:181) is applied by whoever reads the
     * cell (the same multiplication of the same two doubles: identical value), which saves one read + write
     * pass over the matrix.  SQ(i) keeps s[i], RQ(i) the reciprocal the reference multiplies with.
     * Only the cells next to the band are zeroed (the reference reads them from calloc'ed memory). */
#define RQ(i_) S[(size_t)(rows + 2 + (i_)) * 64 + lane]
    for (int u = 0; u < Wr; u++) {
        FQ(0, u) = 0.;
        FQ(1, u) = 0.;
        if (LDS) {
            LQ(u) = 0.;
        }
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
            if (LDS) {
                LQ(k) = v;
            }
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
        if (!LDS) {                                   /* (the LDS variant never reads HBM cells outside the band) */
            for (int u = (b_ >= 3 ? b_ - 3 : 0); u < b_; u++) {
                FQ(i, u) = 0.;
            }
            for (int u = e_ + 1; u < Wr && u <= e_ + 3; u++) {
                FQ(i, u) = 0.;
            }
        }
        double m_prev = 0., d_prev = 0.;             /* cell k-1 of this row (unscaled, like the reference at that point) */
        if (LDS) {
            /* in place: u(i-1, k) = u(i, k) + 3 sh, so cell k overwrites what cell k+1 would read as its (k-1)
             * neighbour when sh == 0 -- those three values travel in registers (a0..a2) */
            const int sh = (i - bw > 0) ? 1 : 0;
            const int ub = lfq_baq_u(bw, i, beg), v11b = ub + 3 * sh - 3;
            double a0 = LQ(v11b + 0) * rs, a1 = LQ(v11b + 1) * rs, a2 = LQ(v11b + 2) * rs;
            /* the three old cells at v10 of the NEXT k are requested before this k's values are written: they lie
             * above everything this k writes (u + 3 sh + 3 > u + 2), so the order does not matter */
            double n0 = LQ(ub + 3 * sh + 0), n1 = LQ(ub + 3 * sh + 1), n2 = LQ(ub + 3 * sh + 2);
            for (int k = beg; k <= end; ++k) {
                const int u = lfq_baq_u(bw, i, k), v10 = u + 3 * sh;
                const double e = lfq_baq_emit(RC(k), qyi, qli);
                const double c0 = n0 * rs, c1 = n1 * rs, c2 = n2 * rs;
                if (k < end) {
                    n0 = LQ(v10 + 3);
                    n1 = LQ(v10 + 4);
                    n2 = LQ(v10 + 5);
                }
                const double f0 = e * (m[0] * a0 + m[3] * a1 + m[6] * a2);
                const double f1 = LFQ_BAQ_EI * (m[1] * c0 + m[4] * c1);
                const double f2 = m[2] * m_prev + m[8] * d_prev;
                FQ(i, u + 0) = f0;
                FQ(i, u + 1) = f1;
                if (keep_f2) {
                    FQ(i, u + 2) = f2;
                }
                LQ(u + 0) = f0;
                LQ(u + 1) = f1;
                LQ(u + 2) = f2;
                a0 = c0; a1 = c1; a2 = c2;
                m_prev = f0;
                d_prev = f2;
                sum += f0 + f1 + f2;
            }
        } else
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
                    const int b = query[1 + qpos + rep_i];                     /* 0..4 -> seq_nt16_str letter */
                    if (A.ref[ref_i] != (uint8_t)"ACGTN"[b > 4 ? 4 : b]) break;
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
        if (LDS) {
            LQ(u) = 0.;
        } else {
            BQ(0, u) = 0.;
        }
    }
    {
        const double sl = SQ(l_query), sl1 = SQ(l_query + 1);
        for (int k = 1; k <= l_ref; ++k) {
            const int u = lfq_baq_u(bw, l_query, k);
            if (u < 3 || u >= bw2 * 3 + 3) continue;
            if (LDS) {
                LQ(u + 0) = sM / sl / sl1;
                LQ(u + 1) = sI / sl / sl1;
            } else {
                BQ(0, u + 0) = sM / sl / sl1;
                BQ(0, u + 1) = sI / sl / sl1;
            }
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
        /* LDS variant: the forward cells the MAP step of this row needs are requested now and land while the
         * backward row is computed (band <= 17 cells: registers, static indices) */
        double fz0[LDS ? LFQ_BAQ_LDS_BAND : 1], fz1[LDS ? LFQ_BAQ_LDS_BAND : 1];
        if (LDS) {
            int fb = 1, fe = l_ref, x;
            x = i - bw; fb = fb > x ? fb : x;
            x = i + bw; fe = fe < x ? fe : x;
            const int u0 = lfq_baq_u(bw, i, fb);
#pragma unroll
            for (int j = 0; j < LFQ_BAQ_LDS_BAND; j++) {
                const bool in = fb + j <= fe;
                fz0[j] = in ? FQ(i, u0 + 3 * j + 0) : 0.;
                fz1[j] = in ? FQ(i, u0 + 3 * j + 1) : 0.;
            }
        }
        if (i < l_query && LDS) {
            /* row i from row i+1, in place: u(i+1, k) = u(i, k) - 3 sh.  With sh == 0 the cell's own slot holds
             * what the next (lower) cell needs as its (k+1) neighbour: that value travels in `keep` */
            int beg = 1, end = l_ref, x;
            const double y = (i > 1), qli1 = c_ql;
            const int qyi1 = c_qy;
            x = i - bw; beg = beg > x ? beg : x;
            x = i + bw; end = end < x ? end : x;
            const int sh = (i + 1 - bw > 0) ? 1 : 0;
            double d01 = 0., keep = 0.;
            for (int k = end; k >= beg; --k) {
                const int u = lfq_baq_u(bw, i, k), v10 = u - 3 * sh, v11 = v10 + 3;
                const double o11 = (sh == 0 && k < end) ? keep : LQ(v11);
                const double o10 = LQ(v10), o101 = LQ(v10 + 1);
                const double e = (k >= l_ref ? 0 : lfq_baq_emit(RC(k + 1), qyi1, qli1)) * o11;
                const double b0 = e * m[0] + LFQ_BAQ_EI * m[1] * o101 + m[2] * d01;
                const double b1 = e * m[3] + LFQ_BAQ_EI * m[4] * o101;
                const double b2 = (e * m[6] + m[8] * d01) * y;
                LQ(u + 0) = b0;
                LQ(u + 1) = b1;
                LQ(u + 2) = b2;
                keep = o10;
                d01 = b2;
            }
            const int b_ = lfq_baq_u(bw, i, beg), e_ = lfq_baq_u(bw, i, end) + 2;
            const double ys = 1. / c_s;
            for (int k = b_; k <= e_; ++k) {
                LQ(k) = LQ(k) * ys;
            }
        } else
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
        if (LDS) {
            const int u0 = lfq_baq_u(bw, i, beg);
#pragma unroll
            for (int j = 0; j < LFQ_BAQ_LDS_BAND; j++) {
                const int k = beg + j, u = u0 + 3 * j;
                if (k <= end) {
                    double z;
                    z = (fz0[j] * rsi) * LQ(u + 0); if (z > max) max = z, max_k = (k - 1) << 2 | 0; sum += z;
                    z = (fz1[j] * rsi) * LQ(u + 1); if (z > max) max = z, max_k = (k - 1) << 2 | 1; sum += z;
                }
            }
        } else
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
            if (LDS) {
                /* outside the band of row i the reference's matrices hold 0 (calloc); the in-place row does not */
                const int kk = IT(e, 1) + j;
                TM(IT(e, 3) + j) = (kk >= beg && kk <= end) ? (FQ(i, u + st) * rsi) * LQ(u + st) * c_s : 0.;
            } else {
                TM(IT(e, 3) + j) = (FQ(i, u + st) * rsi) * BQ(cur, u + st) * c_s;
            }
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

int lfq_launch_baq(const LfqBaqArgs &a, int64_t n_launch, int lds, void *stream)
{
    if (n_launch <= 0) {
        return LFQ_OK;
    }
    const unsigned blocks = (unsigned)((n_launch + 63) / 64);
    if (lds) {
        const size_t ref_bytes = ((size_t)a.max_lref / 2 + 2) * 64;
        hipLaunchKernelGGL(lfq_baq_kernel<true>, dim3(blocks), dim3(64), ref_bytes, (hipStream_t)stream, a, n_launch);
    } else {
        hipLaunchKernelGGL(lfq_baq_kernel<false>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, a, n_launch);
    }
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}
