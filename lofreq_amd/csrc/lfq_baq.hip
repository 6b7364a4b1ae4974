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
 * lock-step per wavefront.  The forward matrix lives in HBM, interleaved per wavefront
 * ([row][cell][lane]: every access of a wave is one coalesced 512-byte line); the backward pass keeps two
 * rotating rows and does the MAP step of a row as soon as the row exists, so no backward matrix is stored.
 * Bound: HBM traffic of the forward matrix (~2 x 8 B per cell) -- see DESIGN.md.
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

__global__ __launch_bounds__(64) void lfq_baq_kernel(LfqBaqArgs A, int64_t n_launch)
{
    const int lane = (int)threadIdx.x;
    const int64_t ridx = (int64_t)blockIdx.x * 64 + lane;
    const bool live = ridx < n_launch;
    const LfqBaqRead R = A.reads[A.first_read + (live ? ridx : 0)];
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
    const int64_t s0 = A.seq_off[A.first_read + ridx];
    const uint8_t *query = A.seq + s0 - 1, *iqual = A.qual + s0 - 1;     /* 1-based like the reference */
    const uint8_t *refw = A.ref + R.xb - 1;
    uint8_t *out = A.lb_out + s0;
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
            const double e = lfq_baq_emit(lfq_baq_code(refw[k]), query[1], ql);
            const double f0 = e * bM, f1 = LFQ_BAQ_EI * bI;
            FQ(1, u + 0) = f0;
            FQ(1, u + 1) = f1;
            sum += f0 + f1;
        }
        SQ(1) = sum;
        const int b_ = lfq_baq_u(bw, 1, 1), e_ = lfq_baq_u(bw, 1, end) + 2;
        for (int k = b_; k <= e_; ++k) {
            FQ(1, k) = FQ(1, k) / sum;
        }
    }
    for (int i = 2; i <= l_query; ++i) {
        double sum = 0.;
        const double qli = A.qual2prob[iqual[i]];
        const double rs = RQ(i - 1);                 /* pending scale of row i-1 */
        const int qyi = query[i];
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
            const double e = lfq_baq_emit(lfq_baq_code(refw[k]), qyi, qli);
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
        RQ(i) = 1. / sum;
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
        A.tag_flags[A.first_read + ridx] = (uint8_t)((n_ins ? 1 : 0) | (n_del ? 2 : 0));
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
    for (int i = l_query; i >= 1; --i) {
        if (i < l_query) {
            const int nxt = cur ^ 1;                  /* row i goes to `nxt`, row i+1 is in `cur` */
            for (int u = 0; u < Wr; u++) {
                BQ(nxt, u) = 0.;
            }
            int beg = 1, end = l_ref, x;
            const double y = (i > 1), qli1 = A.qual2prob[iqual[i + 1]];
            const int qyi1 = query[i + 1];
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
            const double ys = 1. / SQ(i);
            for (int k = b_; k <= e_; ++k) {
                BQ(nxt, k) = BQ(nxt, k) * ys;
            }
            cur = nxt;
        }
        /* MAP of row i */
        double sum = 0., max = 0.;
        const double rsi = RQ(i);
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
            TM(IT(e, 3) + j) = (FQ(i, u + st) * rsi) * BQ(cur, u + st) * SQ(i);
        }
        max /= sum;
        int qk = (int)(-4.343 * log(1. - max) + .499);
        qk = qk > 100 ? 99 : qk;
        /* bam_md_ext.c:413-416 / :435-436: a base the HMM does not put where the CIGAR puts it gets 0 (extended
         * BAQ; the plain variant overwrites the 0 with q again -- reproduced); unaligned bases keep their BQ */
        const int ex = expect[(size_t)(i - 1) * 64 + lane];
        int bq = iqual[i];
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
}

int lfq_launch_baq(const LfqBaqArgs &a, int64_t n_launch, void *stream)
{
    if (n_launch <= 0) {
        return LFQ_OK;
    }
    const unsigned blocks = (unsigned)((n_launch + 63) / 64);
    hipLaunchKernelGGL(lfq_baq_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, a, n_launch);
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}
