/*
 * lofreq_oracle.c -- CPU restatement of LoFreq's per-column SNV calling path.
 *
 * TEST INFRASTRUCTURE ONLY (see lofreq_oracle.h).  Plain C99, scalar, one
 * thread.  Citations are reference paths relative to src/lofreq/.
 *
 * Build: see oracle/Makefile (gcc -O3 -std=gnu99, no -march, no -ffast-math:
 * the reference is built the same way, src/lofreq/Makefile.am:1, so that the
 * quality merge of snpcaller.c:334 is evaluated without FMA contraction).
 */
#define _GNU_SOURCE
#include "lofreq_oracle.h"

#include <errno.h>
#include <fenv.h>
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ORC_LOGZERO (-1e100)   /* snpcaller.c:66 */
#define ORC_MQ0_ERRPROB 0.5    /* snpcaller.c:64 */
#define ORC_FE_BAD (FE_INVALID | FE_DIVBYZERO | FE_OVERFLOW | FE_UNDERFLOW)

static const char ORC_NT4[5] = {'A', 'C', 'G', 'T', 'N'};   /* plp.c:49 bam_nt4_rev_table */

static double orc_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* snpcaller.c:627-651 (plus lofreq_call.c:84 for the test counter) */
void orc_conf_init(orc_conf *c)
{
    memset(c, 0, sizeof(*c));
    c->min_bq = 6;       /* defaults.h:47 */
    c->min_alt_bq = 6;   /* defaults.h:49 */
    c->def_alt_bq = 0;   /* defaults.h:50 */
    c->min_jq = 0;
    c->min_alt_jq = 0;
    c->def_alt_jq = 0;
    c->min_cov = 1;      /* defaults.h:64 */
    c->bonf_dynamic = 1;
    c->bonf_subst = 1;
    c->sig = 0.01;       /* defaults.h:73, stored as float (snpcaller.h:53) */
    c->flag = ORC_USE_MQ | ORC_USE_BAQ;
    c->raw_counts_after_minbq = 0;
    c->num_snv_tests = 0;
    c->bonf_indel = 1;      /* snpcaller.c:642 */
    c->num_indel_tests = 0;
    c->flag |= ORC_USE_IDAQ; /* snpcaller.c:647 */
    c->approx_threshold_n = -1; /* snpcaller.c:650 */
}

/* utils.h:42  PHREDQUAL_TO_PROB */
double orc_phred_to_prob(int q)
{
    if (q == INT_MAX) {
        return DBL_MIN;
    }
    return pow(10.0, -1.0 * q / 10.0);
}

/* utils.h:45  PROB_TO_PHREDQUAL: 80-bit log10l, truncation toward zero */
int orc_prob_to_phred(long double p)
{
    return (int)(-10.0 * log10l(p));
}

int orc_prob_to_phred_p(const long double *p)
{
    return orc_prob_to_phred(*p);
}

/* utils.h:46  PROB_TO_PHREDQUAL_SAFE */
int orc_prob_to_phred_safe(double p)
{
    if (p <= 0.0) {
        return INT_MAX;
    }
    return (int)(-10.0 * log10l(p));
}

/* snpcaller.c:303-341.  -1 means "track missing" -> probability 0; MQ 0 -> 0.5. */
double orc_merge_quals(int sq, int mq, int baq, int bq)
{
    double p_src = (sq == -1) ? 0.0 : orc_phred_to_prob(sq);
    double p_map;
    double p_aln = (baq == -1) ? 0.0 : orc_phred_to_prob(baq);
    double p_base = (bq == -1) ? 0.0 : orc_phred_to_prob(bq);

    if (mq == -1) {
        p_map = 0.0;
    } else if (mq == 0) {
        p_map = ORC_MQ0_ERRPROB;
    } else {
        p_map = orc_phred_to_prob(mq);
    }
    /* snpcaller.c:334, same association as the C expression there */
    return p_map + (1.0 - p_map) * p_src + (1 - p_map) * (1 - p_src) * p_aln
           + (1 - p_map) * (1 - p_src) * (1 - p_aln) * p_base;
}

static int orc_int_cmp(const void *a, const void *b)
{
    int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}

/* utils.c:436-457: sort a copy; even size -> truncated mean of the two middle values */
int orc_int_median(const int *data, int n)
{
    int *tmp;
    int med;
    if (n == 0) {
        return 0;
    }
    tmp = malloc(sizeof(int) * (size_t)n);
    memcpy(tmp, data, sizeof(int) * (size_t)n);
    qsort(tmp, (size_t)n, sizeof(int), orc_int_cmp);
    if ((n & 1) == 0) {
        med = (int)((tmp[n / 2] + tmp[n / 2 - 1]) / 2.0);
    } else {
        med = tmp[n / 2];
    }
    free(tmp);
    return med;
}

/* utils.c:66-76: absolute-epsilon tolerant comparator */
int orc_dbl_cmp(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    if (fabs(x - y) < DBL_EPSILON) {
        return 0;
    }
    return (x > y) - (x < y);
}

/* snpcaller.c:693-700 */
double orc_log_sum(double a, double b)
{
    if (a > b) {
        return a + log1p(exp(b - a));
    }
    return b + log1p(exp(a - b));
}

/* snpcaller.c:730-741: sequential left fold from `start` to len-1 */
double orc_probvec_tailsum(const double *pv, int start, int len)
{
    double acc = pv[start];
    int i;
    for (i = start + 1; i < len; i++) {
        acc = orc_log_sum(acc, pv[i]);
    }
    return acc;
}

/* the errno / floating-point-exception clamp applied after every expl()
 * (snpcaller.c:929-936, 1052-1059, 1174-1188) */
static long double orc_clamp_after_expl(long double pv, int errsv)
{
    if (errsv || fetestexcept(ORC_FE_BAD)) {
        return (pv < DBL_EPSILON) ? LDBL_MIN : LDBL_MAX;
    }
    return pv;
}

/* snpcaller.c:831-972.  Returns malloc'ed K+1 log-probabilities:
 * [k<K] = log P(X=k), [K] = log P(X>=K); early exit returns the row at which
 * P(X>=K)*bonf first exceeded sig. */
double *orc_pruned_calc_prob_dist(const double *ep, int n_ep, int kmax, long long bonf, double sig,
                                  int *rows_done)
{
    double *cur = malloc(sizeof(double) * (size_t)(kmax + 1));
    double *prev = malloc(sizeof(double) * (size_t)(kmax + 1));
    double *swp;
    int n;

    if (!cur || !prev) {
        free(cur);
        free(prev);
        return NULL;
    }
    prev[0] = 0.0;                                          /* :863 */
    for (n = 1; n <= n_ep; n++) {
        double pn = ep[n - 1];
        double l_hit, l_miss;
        int k, kstart;

        l_hit = (fabs(pn) < DBL_EPSILON) ? log(DBL_EPSILON) : log(pn);                  /* :872-876 */
        l_miss = (fabs(pn - 1.0) < DBL_EPSILON) ? log1p(-pn + DBL_EPSILON) : log1p(-pn); /* :877-881 */

        if (n < kmax) {
            prev[n] = ORC_LOGZERO;                          /* :888-890 */
        }
        kstart = (n < kmax - 1) ? n : kmax - 1;             /* MIN(n, K-1), :892 */
        for (k = kstart; k >= 1; k--) {
            cur[k] = orc_log_sum(prev[k] + l_miss, prev[k - 1] + l_hit);   /* :894 */
        }
        cur[0] = prev[0] + l_miss;                          /* :899 */

        if (n == kmax) {
            cur[kmax] = prev[kmax - 1] + l_hit;             /* :913 */
        } else if (n > kmax) {
            long double pv;
            int errsv;
            cur[kmax] = orc_log_sum(prev[kmax], prev[kmax - 1] + l_hit);   /* :922 */
            errno = 0;
            feclearexcept(FE_ALL_EXCEPT);
            pv = expl(cur[kmax]);
            errsv = errno;
            pv = orc_clamp_after_expl(pv, errsv);           /* :929-936 */
            if (pv * (double)bonf > sig) {                  /* :950 */
                free(prev);
                if (rows_done) {
                    *rows_done = n;
                }
                return cur;
            }
        }
        swp = cur;
        cur = prev;
        prev = swp;
    }
    free(cur);
    if (rows_done) {
        *rows_done = n_ep;
    }
    return prev;                                            /* :970 */
}

/* snpcaller.c:1020-1062 */
double *orc_poissbin(long double *pvalue, const double *ep, int n_ep, int kmax, long long bonf,
                     double sig, int *rows_done)
{
    double *pv;
    int errsv;
    *pvalue = LDBL_MAX;
    pv = orc_pruned_calc_prob_dist(ep, n_ep, kmax, bonf, sig, rows_done);
    if (!pv) {
        return NULL;
    }
    errno = 0;
    feclearexcept(FE_ALL_EXCEPT);
    *pvalue = expl(pv[kmax]);                               /* :1050 */
    errsv = errno;
    *pvalue = orc_clamp_after_expl(*pvalue, errsv);         /* :1052-1059 */
    return pv;
}

/* ---- the Poisson approximation gate of snpcaller (-t / --approx-threshold, snpcaller.c:1128-1142) ----------
 * PARITY UNPINNED: the branch needs libgsl (gsl_cdf_poisson_P), which is neither in the reference tree nor in this image,
 * and the 2.1.4 binary predates the option.  What is restated is the published definition GSL implements:
 *     gsl_cdf_poisson_P(k, mu) = P(X <= k) = Q(k + 1, mu)      (cdf/poisson.c: a = k + 1; gsl_cdf_gamma_Q(mu, a, 1))
 *     gsl_cdf_gamma_Q(x, a, 1): x < a ? 1 - gsl_sf_gamma_inc_P(a, x) : gsl_sf_gamma_inc_Q(a, x)   (cdf/gamma.c)
 * with the regularized incomplete gamma functions evaluated here in long double (series below a + 1, Lentz's continued
 * fraction above) and rounded to double where GSL returns a double, so that the two subtractions from 1 round the way
 * they do there.  Pinned against scipy.special.pdtr (tests/test_oracle_kat.py), not against GSL. */
static long double orc_gamma_inc_p_series(long double a, long double x)    /* P(a, x), x < a + 1 */
{
    long double sum = 1.0L, term = 1.0L, n = a;
    int i;
    for (i = 0; i < 1000000; i++) {
        n += 1.0L;
        term *= x / n;
        sum += term;
        if (term < sum * 1e-21L) {
            break;
        }
    }
    return sum * expl(a * logl(x) - x - lgammal(a + 1.0L));
}

static long double orc_gamma_inc_q_cf(long double a, long double x)        /* Q(a, x), x >= a + 1 (modified Lentz) */
{
    const long double tiny = 1e-4000L;
    long double b = x + 1.0L - a, c = 1.0L / tiny, d = 1.0L / b, h = d;
    int i;
    for (i = 1; i < 1000000; i++) {
        const long double an = -(long double)i * ((long double)i - a);
        long double del;
        b += 2.0L;
        d = an * d + b;
        if (fabsl(d) < tiny) d = tiny;
        c = b + an / c;
        if (fabsl(c) < tiny) c = tiny;
        d = 1.0L / d;
        del = d * c;
        h *= del;
        if (fabsl(del - 1.0L) < 1e-20L) {
            break;
        }
    }
    return h * expl(a * logl(x) - x - lgammal(a));
}

/* gsl_cdf_poisson_P(k, mu) as a double */
double orc_poisson_cdf(unsigned int k, double mu)
{
    const long double a = (long double)k + 1.0L, x = (long double)mu;
    if (!(mu > 0.0)) {
        return NAN;                                         /* GSL: domain error (its handler aborts by default) */
    }
    if (x < a) {
        const double P = (double)orc_gamma_inc_p_series(a, x);
        return 1.0 - P;                                     /* gsl_cdf_gamma_Q: Q = 1 - P */
    }
    if (x < a + 1.0L) {
        return (double)(1.0L - orc_gamma_inc_p_series(a, x));
    }
    return (double)orc_gamma_inc_q_cf(a, x);
}

/* snpcaller.c:1131-1140: 1 = the column is given up without running the DP */
int orc_approx_gate(const double *ep, int n_ep, int kmax, long long bonf, double sig, int approx_threshold_n,
                    long double *approx_out)
{
    long double mu = 0.0L, approx;
    int i;
    if (approx_out) {
        *approx_out = NAN;
    }
    if (!(approx_threshold_n > 0 && n_ep > approx_threshold_n)) {      /* :1131 */
        return 0;
    }
    for (i = 0; i < n_ep; ++i) {
        mu += ep[i];                                        /* :1133-1135 */
    }
    approx = 1 - orc_poisson_cdf((unsigned int)(kmax - 1), (double)mu);   /* :1136 (double arithmetic, then widened) */
    if (approx_out) {
        *approx_out = approx;
    }
    return approx * (double)bonf > sig;                     /* :1137 */
}

/* snpcaller.c:1075-1205 */
int orc_snpcaller_approx(long double pv_out[3], double logp_out[3], const double *ep, int n_ep,
                         const int counts[3], long long bonf, double sig, int approx_threshold_n, int *rows_done);
int orc_snpcaller(long double pv_out[3], double logp_out[3], const double *ep, int n_ep,
                  const int counts[3], long long bonf, double sig, int *rows_done)
{
    return orc_snpcaller_approx(pv_out, logp_out, ep, n_ep, counts, bonf, sig, -1, rows_done);
}

int orc_snpcaller_approx(long double pv_out[3], double logp_out[3], const double *ep, int n_ep,
                         const int counts[3], long long bonf, double sig, int approx_threshold_n, int *rows_done)
{
    double *probvec;
    long double pv;
    int i, kmax = 0;

    for (i = 0; i < 3; i++) {
        pv_out[i] = LDBL_MAX;                               /* :1101-1103 */
        if (logp_out) {
            logp_out[i] = NAN;
        }
        if (counts[i] > kmax) {
            kmax = counts[i];
        }
    }
    if (rows_done) {
        *rows_done = 0;
    }
    if (kmax == 0) {
        return 0;                                           /* :1113 */
    }
    if (orc_approx_gate(ep, n_ep, kmax, bonf, sig, approx_threshold_n, NULL)) {
        return 0;                                           /* :1137-1139 */
    }
    probvec = orc_poissbin(&pv, ep, n_ep, kmax, bonf, sig, rows_done);   /* :1144 */
    if (!probvec) {
        return -1;
    }
    if (pv * (double)bonf > sig) {                          /* :1155 */
        free(probvec);
        return 0;
    }
    for (i = 0; i < 3; i++) {
        double lp;
        int errsv;
        if (counts[i] == 0) {
            continue;
        }
        errno = 0;
        feclearexcept(FE_ALL_EXCEPT);
        lp = orc_probvec_tailsum(probvec, counts[i], kmax + 1);   /* :1172 */
        pv = expl(lp);
        errsv = errno;
        pv = orc_clamp_after_expl(pv, errsv);               /* :1174-1188 */
        pv_out[i] = pv;
        if (logp_out) {
            logp_out[i] = lp;
        }
    }
    free(probvec);
    return 0;
}

/* ---- ground truth for the tolerance story (NOT a restatement of reference code) --------------------
 * The tail probabilities P(X >= counts[i]) of the Poisson-binomial distribution the reference's
 * pruned_calc_prob_dist (snpcaller.c:831-972) iterates in log space, here as the plain linear-space
 * recurrence  v[k] <- v[k]*(1-p) + v[k-1]*p  with the absorbing tail at K = max(counts)
 * in x87 80-bit long double: 64-bit mantissa, one rounding of 2^-64 per operation, range down to 1e-4932.
 * No pruning, no Bonferroni factor, any row order (the distribution does not depend on it).  Used by the
 * tests to show how far the reference's own log-space chain (and the device's scaled-double recurrence)
 * are from the exact value.  Returns 0, or -1 when out of memory / K == 0.  tails_log[i] = logl(tail), NAN
 * where counts[i] == 0; a tail below LDBL_MIN (denormal / zero) is reported as -INFINITY. */
int orc_tail_truth(long double tails[3], double tails_log[3], const double *ep, int n_ep, const int counts[3])
{
    long double *v;
    int i, k, n, kmax = 0;

    for (i = 0; i < 3; i++) {
        tails[i] = 0.0L;
        tails_log[i] = NAN;
        if (counts[i] > kmax) {
            kmax = counts[i];
        }
    }
    if (kmax == 0) {
        return -1;
    }
    v = calloc((size_t)kmax + 1, sizeof(long double));
    if (!v) {
        return -1;
    }
    v[0] = 1.0L;
    for (n = 0; n < n_ep; n++) {
        long double p = (long double)ep[n], q = 1.0L - p;
        int top = n + 1 < kmax ? n + 1 : kmax;           /* cells above the row index are still 0 */
        if (top == kmax) {
            v[kmax] += v[kmax - 1] * p;                  /* absorbing tail: P(X >= K) */
            top = kmax - 1;
        }
        for (k = top; k >= 1; k--) {
            v[k] = v[k] * q + v[k - 1] * p;
        }
        v[0] *= q;
    }
    for (i = 0; i < 3; i++) {
        long double t = 0.0L;
        if (counts[i] == 0) {
            continue;
        }
        for (k = kmax; k >= counts[i]; k--) {            /* smallest terms first */
            t += v[k];
        }
        tails[i] = t;
        tails_log[i] = (t >= LDBL_MIN) ? (double)logl(t) : -INFINITY;
    }
    free(v);
    return 0;
}

/* orc_tail_truth for one packed column: the error probabilities come from orc_col_errprobs (the pinned
 * restatement of plp_to_errprobs), everything after it is the 80-bit recurrence above. */
int orc_col_tail_truth(long double tails[3], double tails_log[3], int counts_out[3],
                       const uint8_t *nt, const uint8_t *bq, const uint8_t *baq, const uint8_t *mq,
                       const uint8_t *sq, int64_t n_obs, char ref_base, const orc_conf *conf)
{
    double *ep = malloc(sizeof(double) * (size_t)(n_obs > 0 ? n_obs : 1));
    int n_ep = 0, alt_base[3], alt_raw[3], rc;
    if (!ep) {
        return -1;
    }
    rc = orc_col_errprobs(ep, &n_ep, alt_base, counts_out, alt_raw, nt, bq, baq, mq, sq, n_obs, ref_base, conf);
    if (rc == 0) {
        rc = orc_tail_truth(tails, tails_log, ep, n_ep, counts_out);
    }
    free(ep);
    return rc;
}

static int orc_unpack_q(uint8_t v)
{
    return (v == ORC_Q_MISSING) ? -1 : (int)v;
}

/* snpcaller.c:346-498 on one packed column.  The reference walks plp_col_t's
 * per-nucleotide arrays A,C,G,T (N skipped, :386); observations of one
 * nucleotide keep their pileup (= storage) order. */
int orc_col_errprobs(double *ep, int *n_ep, int alt_base[3], int alt_counts[3], int alt_raw[3],
                     const uint8_t *nt, const uint8_t *bq, const uint8_t *baq, const uint8_t *mq,
                     const uint8_t *sq, int64_t n_obs, char ref_base, const orc_conf *conf)
{
    int median_ref_bq = -1;
    int alt_idx = -1;
    int code;
    int64_t j;

    *n_ep = 0;
    if (conf->def_alt_bq == -1) {                           /* :363-379 */
        int ref_code = -1;
        int64_t cnt = 0;
        for (code = 0; code < 4; code++) {
            if (ORC_NT4[code] == ref_base) {
                ref_code = code;
            }
        }
        if (ref_code >= 0) {
            int *tmp = malloc(sizeof(int) * (size_t)(n_obs > 0 ? n_obs : 1));
            for (j = 0; j < n_obs; j++) {
                if ((nt[j] & 7) == ref_code) {
                    tmp[cnt++] = bq[j];
                }
            }
            if (cnt) {
                median_ref_bq = orc_int_median(tmp, (int)cnt);
            }
            free(tmp);
        }
    }

    for (code = 0; code < 4; code++) {                      /* :383-388, N skipped */
        int is_alt = (ORC_NT4[code] != ref_base);
        if (is_alt) {                                       /* :391-397 */
            alt_idx++;
            if (alt_idx > 2) {
                return -1;   /* ref_base not in ACGT: caller must gate (lofreq_call.c:754) */
            }
            alt_base[alt_idx] = ORC_NT4[code];
            alt_counts[alt_idx] = 0;
            alt_raw[alt_idx] = 0;
        }
        for (j = 0; j < n_obs; j++) {
            int q_base, q_aln = -1, q_map = -1, q_src = -1, q_joint;
            double p_joint;
            if ((nt[j] & 7) != code) {
                continue;
            }
            q_base = bq[j];
            if (is_alt && !conf->raw_counts_after_minbq) {
                alt_raw[alt_idx]++;                         /* :418-420 (HEAD: before the BQ filter) */
            }
            if (q_base < conf->min_bq) {                    /* :426 */
                continue;
            }
            if (is_alt && conf->raw_counts_after_minbq) {
                alt_raw[alt_idx]++;                         /* lofreq 2.1.4 placement */
            }
            if (is_alt) {                                   /* :431-441 */
                if (q_base < conf->min_alt_bq) {
                    continue;
                } else if (conf->def_alt_bq == -1) {
                    q_base = median_ref_bq;
                } else if (conf->def_alt_bq != 0) {
                    q_base = conf->def_alt_bq;
                }
            }
            if ((conf->flag & ORC_USE_BAQ) && baq) {        /* :444-446 */
                q_aln = orc_unpack_q(baq[j]);
            }
            if (conf->flag & ORC_USE_MQ) {                  /* :448-453 */
                q_map = mq[j];
                if (q_map == 255) {
                    q_map = -1;
                }
            }
            if ((conf->flag & ORC_USE_SQ) && sq) {          /* :461-463 */
                q_src = orc_unpack_q(sq[j]);
                if (q_src == 254) {                         /* packed track: 254 = source_qual's 49314 (plp.c:521) */
                    q_src = 49314;
                }
            }
            p_joint = orc_merge_quals(q_src, q_map, q_aln, q_base);   /* :465 */
            q_joint = orc_prob_to_phred_safe(p_joint);                /* :466 */
            if (q_joint < conf->min_jq) {                   /* :469 */
                continue;
            }
            if (is_alt) {                                   /* :473-490 */
                if (q_joint < conf->min_alt_jq) {
                    continue;
                } else if (conf->def_alt_jq == -1) {
                    return -2;   /* reference aborts: "median off ref joined q not implemented" */
                } else if (conf->def_alt_jq != 0) {
                    p_joint = orc_phred_to_prob(conf->def_alt_jq);
                }
                alt_counts[alt_idx]++;
            }
            ep[(*n_ep)++] = p_joint;                        /* :491 */
        }
    }
    return 0;
}

/* lofreq_call.c:735-879 over a batch; call_vars' ref gate (:892) included.
 * The consensus-indel gate of call_vars (:928-931) needs cons_base, which is a
 * pileup-side quantity: the caller expresses it through num_bases/coverage_plp. */
int orc_call_batch(const uint8_t *nt, const uint8_t *bq, const uint8_t *baq, const uint8_t *mq,
                   const uint8_t *sq, const uint64_t *col_off, const uint8_t *ref_base,
                   const int32_t *coverage_plp, const int32_t *num_bases, int64_t ncols,
                   orc_conf *conf, orc_col_result *out, orc_timing *timing)
{
    int64_t c;
    uint64_t max_obs = 1;
    double *ep;

    for (c = 0; c < ncols; c++) {
        uint64_t d = col_off[c + 1] - col_off[c];
        if (d > max_obs) {
            max_obs = d;
        }
    }
    ep = malloc(sizeof(double) * max_obs);
    if (!ep) {
        return -1;
    }
    if (timing) {
        timing->t_merge = timing->t_sort = timing->t_dp = 0.0;
    }

    for (c = 0; c < ncols; c++) {
        orc_col_result *r = &out[c];
        uint64_t o0 = col_off[c];
        int64_t n_obs = (int64_t)(col_off[c + 1] - o0);
        int cov = coverage_plp ? coverage_plp[c] : (int)n_obs;
        int nb = num_bases ? num_bases[c] : (int)n_obs;
        char ref = (char)ref_base[c];
        int i, rc, any_alt = 0;
        int64_t j;
        double t0, t1, t2, t3;

        memset(r, 0, sizeof(*r));
        for (i = 0; i < 3; i++) {
            r->pvalue[i] = LDBL_MAX;
            r->logp[i] = NAN;
            r->qual[i] = -1;
        }
        for (j = 0; j < n_obs; j++) {                       /* plp.c:1007-1011 */
            int code = nt[o0 + j] & 7;
            if (code > 4) {
                code = 4;
            }
            if (nt[o0 + j] & 8) {
                r->rv[code]++;
            } else {
                r->fw[code]++;
            }
        }
        if (ref == 'N' || !(ref == 'A' || ref == 'C' || ref == 'G' || ref == 'T')) {
            continue;                                       /* lofreq_call.c:754, 892; plp.c:819-823 */
        }
        if (nb * 2 < cov) {
            continue;                                       /* lofreq_call.c:930 */
        }
        if (nb < conf->min_cov) {
            continue;                                       /* lofreq_call.c:747 */
        }

        t0 = orc_now();
        rc = orc_col_errprobs(ep, &r->n_err_probs, r->alt_base, r->alt_counts, r->alt_raw_counts,
                              nt + o0, bq + o0, baq ? baq + o0 : NULL, mq + o0, sq ? sq + o0 : NULL,
                              n_obs, ref, conf);
        t1 = orc_now();
        if (rc) {
            free(ep);
            return rc;
        }
        for (i = 0; i < 3; i++) {
            if (r->alt_counts[i]) {
                any_alt = 1;
            }
        }
        if (timing) {
            timing->t_merge += t1 - t0;
        }
        if (!any_alt) {
            continue;                                       /* lofreq_call.c:768-780 */
        }
        qsort(ep, (size_t)r->n_err_probs, sizeof(double), orc_dbl_cmp);   /* lofreq_call.c:784 */
        t2 = orc_now();

        if (conf->bonf_dynamic) {                           /* lofreq_call.c:794-800 */
            if (conf->bonf_subst == 1) {
                conf->bonf_subst = 3;
            } else {
                conf->bonf_subst += 3;
            }
        }
        conf->num_snv_tests += 3;                           /* lofreq_call.c:801 */
        r->tested = 1;
        r->bonf_used = conf->bonf_subst;

        rc = orc_snpcaller_approx(r->pvalue, r->logp, ep, r->n_err_probs, r->alt_counts, conf->bonf_subst,
                                  (double)conf->sig, conf->approx_threshold_n, &r->dp_rows);   /* lofreq_call.c:807 */
        t3 = orc_now();
        if (timing) {
            timing->t_sort += t2 - t1;
            timing->t_dp += t3 - t2;
        }
        if (rc) {
            free(ep);
            return rc;
        }
        for (i = 0; i < 3; i++) {
            if (r->pvalue[i] * (double)conf->bonf_subst < conf->sig) {   /* lofreq_call.c:832 */
                r->emitted[i] = 1;
                r->qual[i] = orc_prob_to_phred(r->pvalue[i]);            /* lofreq_call.c:863 */
            }
        }
    }
    free(ep);
    return 0;
}

/* ---- indel path --------------------------------------------------------------------------- */

static int orc_nt4(char c)      /* plp.c:71-88 bam_nt4_table for the letters that matter here */
{
    switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return 4;
    }
}

/* plp_to_ins_errprobs / plp_to_del_errprobs (snpcaller.c:502-561 / 565-623): all non-event reads
 * (indel quality + mapping quality, no alignment quality), then the reads of EVERY event of this side;
 * the alignment quality is only used for the event being tested. */
static int orc_indel_errprobs(double *ep, const orc_indel_batch *b, int side, int64_t col, int64_t tested_ev,
                              const orc_conf *conf)
{
    int n = 0;
    int64_t i, e;
    for (i = b->ne_off[side][col]; i < b->ne_off[side][col + 1]; i++) {
        int q = b->ne_q[side][i], mq = -1;
        if (conf->flag & ORC_USE_MQ) {
            mq = b->ne_mq[side][i];                 /* no 255 -> -1 mapping here (snpcaller.c:523-526) */
        }
        ep[n++] = orc_merge_quals(-1, mq, -1, q);
    }
    for (e = b->ev_off[side][col]; e < b->ev_off[side][col + 1]; e++) {
        for (i = b->rd_off[side][e]; i < b->rd_off[side][e + 1]; i++) {
            int q = b->rd_q[side][i], aq = -1, mq = -1, sq = -1;
            if ((conf->flag & ORC_USE_IDAQ) && e == tested_ev) {     /* strcmp(it->key, key) == 0 */
                aq = b->rd_aq[side][i];
            }
            if (conf->flag & ORC_USE_MQ) {
                mq = b->rd_mq[side][i];
                if (mq == 255) {
                    mq = -1;
                }
            }
            if (conf->flag & ORC_USE_SQ) {
                sq = b->rd_sq[side][i];
            }
            ep[n++] = orc_merge_quals(sq, mq, aq, q);
        }
    }
    return n;
}

/* call_indels (lofreq_call.c:619-726) + call_alt_ins/del (:306-426) */
int orc_call_indels_batch(const orc_indel_batch *b, orc_conf *conf, orc_indel_test *out, int64_t cap,
                          int64_t *n_out)
{
    int64_t c, n_tests = 0, max_reads = 1;
    double *ep;
    for (c = 0; c < b->ncols; c++) {
        int side;
        for (side = 0; side < 2; side++) {
            int64_t e0 = b->ev_off[side][c], e1 = b->ev_off[side][c + 1];
            int64_t m = (b->ne_off[side][c + 1] - b->ne_off[side][c]) + (b->rd_off[side][e1] - b->rd_off[side][e0]);
            if (m > max_reads) {
                max_reads = m;
            }
        }
    }
    ep = malloc(sizeof(double) * (size_t)max_reads);
    for (c = 0; c < b->ncols; c++) {
        int ign[5] = {0, 0, 0, 0, 0};
        int side;
        const float denom = (float)b->coverage_plp[c] - b->num_tails[c];   /* lofreq_call.c:334, 391 */
        if (b->ref_base[c] == 'N') {
            continue;                                                       /* lofreq_call.c:892 */
        }
        if (b->num_non_indels[c] + b->num_ins[c] + b->num_dels[c] < conf->min_cov) {
            continue;                                                       /* :626 */
        }
        /* multi-allelic low-AF 1-bp A/T indels next to poly-AT (:649-681) */
        if (b->num_ins[c] && (b->ne_off[0][c + 1] - b->ne_off[0][c]) && b->num_dels[c]
            && (b->ne_off[1][c + 1] - b->ne_off[1][c])) {
            int dict[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
            int i;
            for (side = 0; side < 2; side++) {
                int64_t e;
                for (e = b->ev_off[side][c]; e < b->ev_off[side][c + 1]; e++) {
                    const char *key = b->key_chars[side] + b->key_off[side][e];
                    int64_t len = b->key_off[side][e + 1] - b->key_off[side][e];
                    if (len == 1 && (key[0] == 'A' || key[0] == 'T')) {
                        dict[side][orc_nt4(key[0])] = (int)(b->rd_off[side][e + 1] - b->rd_off[side][e]);
                    }
                }
            }
            for (i = 0; i < 5; i++) {
                if (dict[0][i] && dict[1][i]) {
                    float ins_af = dict[0][i] / ((float)(b->coverage_plp[c] - b->num_tails[c]));
                    float del_af = dict[1][i] / ((float)(b->coverage_plp[c] - b->num_tails[c]));
                    if (ins_af < 0.05f && del_af < 0.05f) {
                        ign[i] = 1;
                    }
                }
            }
        }
        for (side = 0; side < 2; side++) {
            int64_t e;
            if (!(side == 0 ? b->num_ins[c] : b->num_dels[c])) {
                continue;                                                   /* :684, :706 */
            }
            for (e = b->ev_off[side][c]; e < b->ev_off[side][c + 1]; e++) {
                const char *key = b->key_chars[side] + b->key_off[side][e];
                int64_t len = b->key_off[side][e + 1] - b->key_off[side][e];
                int counts[3] = {0, 0, 0};
                long double pv[3];
                double lp[3];
                int n, rows;
                orc_indel_test *t;
                if (len == 1 && ign[orc_nt4(key[0])]) {
                    continue;                                               /* :687-689 */
                }
                n = orc_indel_errprobs(ep, b, side, c, e, conf);
                qsort(ep, (size_t)n, sizeof(double), orc_dbl_cmp);         /* :692 */
                if (conf->bonf_dynamic) {
                    conf->bonf_indel += 1;                                  /* :693-695 */
                }
                conf->num_indel_tests += 1;
                counts[0] = (int)(b->rd_off[side][e + 1] - b->rd_off[side][e]);   /* it->count */
                if (orc_snpcaller_approx(pv, lp, ep, n, counts, conf->bonf_indel, (double)conf->sig, conf->approx_threshold_n, &rows)) {
                    free(ep);
                    return -1;
                }
                if (n_tests >= cap) {
                    free(ep);
                    return -4;
                }
                t = &out[n_tests++];
                memset(t, 0, sizeof(*t));
                t->col = c;
                t->side = side;
                t->event = (int32_t)e;
                t->n_err_probs = n;
                t->count = counts[0];
                t->bonf_used = conf->bonf_indel;
                t->logp = lp[0];
                t->pvalue = pv[0];
                t->qual = -1;
                if (pv[0] * conf->bonf_indel < conf->sig) {                 /* :326 / :384 */
                    t->emitted = 1;
                    t->qual = orc_prob_to_phred(pv[0]);
                    t->af = counts[0] / denom;
                    t->ref_fw = b->non_fw[side][c];
                    t->ref_rv = b->non_rv[side][c];
                    t->alt_fw = b->ev_fw[side][e];
                    t->alt_rv = b->ev_rv[side][e];
                    t->sb = orc_sb_phred(t->ref_fw, t->ref_rv, t->alt_fw, t->alt_rv);
                    t->dp = b->coverage_plp[c] - b->num_tails[c];           /* lofreq_call.c:132 */
                    t->hrun = b->hrun[c];
                }
            }
        }
    }
    free(ep);
    *n_out = n_tests;
    return 0;
}

int orc_format_indel(char *buf, int buflen, const char *chrom, long pos0, const char *ref, const char *alt,
                     int qual, int dp, float af, int sb, int ref_fw, int ref_rv, int alt_fw, int alt_rv, int hrun,
                     const char *filter)
{
    return snprintf(buf, (size_t)buflen, "%s\t%ld\t.\t%s\t%s\t%d\t%s\tDP=%d;AF=%f;SB=%d;DP4=%d,%d,%d,%d;INDEL;HRUN=%d\n",
                    chrom, pos0 + 1, ref, alt, qual, filter ? filter : ".", dp, af, sb, ref_fw, ref_rv, alt_fw,
                    alt_rv, hrun);
}

/* ---- fet.c: Fisher's exact test (samtools 0.1.18 kfunc) ------------------ */

static double orc_lchoose(int n, int k)                     /* fet.c:13-17 */
{
    if (k == 0 || n == k) {
        return 0;
    }
    return lgamma(n + 1) - lgamma(k + 1) - lgamma(n - k + 1);
}

static double orc_hyper(int n11, int n1_, int n_1, int n)   /* fet.c:26-29 */
{
    return exp(orc_lchoose(n1_, n11) + orc_lchoose(n - n1_, n_1 - n11) - orc_lchoose(n, n_1));
}

typedef struct {
    int n11, n1_, n_1, n;
    double p;
} orc_hyper_state;

/* fet.c:37-61: incremental hypergeometric; re-anchored every 11th n11 */
static double orc_hyper_step(int n11, int n1_, int n_1, int n, orc_hyper_state *s)
{
    if (n1_ || n_1 || n) {
        s->n11 = n11;
        s->n1_ = n1_;
        s->n_1 = n_1;
        s->n = n;
    } else {
        if (n11 % 11 && n11 + s->n - s->n1_ - s->n_1) {
            if (n11 == s->n11 + 1) {
                s->p *= (double)(s->n1_ - s->n11) / n11 * (s->n_1 - s->n11)
                        / (n11 + s->n - s->n1_ - s->n_1);
                s->n11 = n11;
                return s->p;
            }
            if (n11 == s->n11 - 1) {
                s->p *= (double)s->n11 / (s->n1_ - n11) * (s->n11 + s->n - s->n1_ - s->n_1)
                        / (s->n_1 - n11);
                s->n11 = n11;
                return s->p;
            }
        }
        s->n11 = n11;
    }
    s->p = orc_hyper(s->n11, s->n1_, s->n_1, s->n);
    return s->p;
}

/* fet.c:62-99 */
double orc_fisher_exact(int n11, int n12, int n21, int n22, double *left_out, double *right_out,
                        double *two_out)
{
    orc_hyper_state st;
    int row1 = n11 + n12, col1 = n11 + n21, tot = n11 + n12 + n21 + n22;
    int hi = (col1 < row1) ? col1 : row1;
    int lo = row1 + col1 - tot;
    int i, j;
    double p, q, left, right;

    if (lo < 0) {
        lo = 0;
    }
    *two_out = *left_out = *right_out = 1.;
    if (lo == hi) {
        return 1.;
    }
    q = orc_hyper_step(n11, row1, col1, tot, &st);
    p = orc_hyper_step(lo, 0, 0, 0, &st);
    for (left = 0., i = lo + 1; p < 0.99999999 * q; ++i) {
        left += p;
        p = orc_hyper_step(i, 0, 0, 0, &st);
    }
    --i;
    if (p < 1.00000001 * q) {
        left += p;
    } else {
        --i;
    }
    p = orc_hyper_step(hi, 0, 0, 0, &st);
    for (right = 0., j = hi - 1; p < 0.99999999 * q; --j) {
        right += p;
        p = orc_hyper_step(j, 0, 0, 0, &st);
    }
    ++j;
    if (p < 1.00000001 * q) {
        right += p;
    } else {
        ++j;
    }
    *two_out = left + right;
    if (*two_out > 1.) {
        *two_out = 1.;
    }
    if (abs(i - n11) < abs(j - n11)) {
        right = 1. - left + q;
    } else {
        left = 1.0 - right + q;
    }
    *left_out = left;
    *right_out = right;
    return q;
}

/* lofreq_call.c:117-129 */
int orc_sb_phred(int ref_fw, int ref_rv, int alt_fw, int alt_rv)
{
    double l, r, two;
    if ((ref_fw + ref_rv) == 0 && (alt_fw == 0 || alt_rv == 0)) {
        return INT_MAX;
    }
    (void)orc_fisher_exact(ref_fw, ref_rv, alt_fw, alt_rv, &l, &r, &two);
    return orc_prob_to_phred_safe(two);
}

/* vcf.c:469-497 + 608-629; AF as float division (lofreq_call.c:835) printed with %f */
int orc_format_snv(char *buf, int buflen, const char *chrom, long pos0, char ref, char alt,
                   int qual, int dp, int alt_raw_count, int sb, int ref_fw, int ref_rv,
                   int alt_fw, int alt_rv, int hqa, int with_hqa, const char *filter)
{
    float af = alt_raw_count / (float)dp;
    int n = snprintf(buf, (size_t)buflen, "%s\t%ld\t.\t%c\t%c\t%d\t%s\tDP=%d;AF=%f;SB=%d;DP4=%d,%d,%d,%d",
                     chrom, pos0 + 1, ref, alt, qual, filter ? filter : ".", dp, af, sb, ref_fw,
                     ref_rv, alt_fw, alt_rv);
    if (with_hqa && n < buflen) {
        n += snprintf(buf + n, (size_t)(buflen - n), ";HQA=%d", hqa);
    }
    if (n < buflen) {
        n += snprintf(buf + n, (size_t)(buflen - n), "\n");
    }
    return n;
}

/* ---- multtest.c ----------------------------------------------------------- */

typedef struct {
    double p;
    long i;
} orc_ixp;

static int orc_ixp_cmp(const void *a, const void *b)        /* multtest.c:51-57 */
{
    return orc_dbl_cmp(&((const orc_ixp *)a)->p, &((const orc_ixp *)b)->p);
}

void orc_bonf_corr(double *data, long n, long num_tests)    /* multtest.c:66-81 */
{
    long fac = (num_tests < 1) ? n : num_tests;
    long i;
    for (i = 0; i < n; i++) {
        data[i] *= fac;
    }
}

void orc_holm_bonf_corr(double *data, long n, double alpha, long num_tests)   /* multtest.c:91-136 */
{
    orc_ixp *ix = malloc(sizeof(orc_ixp) * (size_t)(n > 0 ? n : 1));
    long i, lp = (num_tests < 1) ? n : num_tests;
    double seen;
    for (i = 0; i < n; i++) {
        ix[i].i = i;
        ix[i].p = data[i];
    }
    qsort(ix, (size_t)n, sizeof(orc_ixp), orc_ixp_cmp);
    seen = n ? ix[0].p : 0.0;
    for (i = 0; i < n; i++) {
        double tp;
        if (orc_dbl_cmp(&ix[i].p, &seen) != 0) {
            lp = (num_tests < 1) ? n - i : num_tests - i;
            seen = ix[i].p;
        }
        tp = ix[i].p * 1. / lp;
        if (orc_dbl_cmp(&tp, &alpha) < 0) {
            data[ix[i].i] = ix[i].p * lp;
        }
    }
    free(ix);
}

/* multtest.c:148-189: Benjamini-Hochberg; note the float division in the threshold */
long orc_fdr(const double *data, long n, double alpha, long num_tests, long *rejected_idx)
{
    orc_ixp *ix = malloc(sizeof(orc_ixp) * (size_t)(n > 0 ? n : 1));
    long i, nrej = 0, m = (num_tests < 1) ? n : num_tests;
    for (i = 0; i < n; i++) {
        ix[i].i = i;
        ix[i].p = data[i];
    }
    qsort(ix, (size_t)n, sizeof(orc_ixp), orc_ixp_cmp);
    for (i = n; i > 0; i--) {
        if (ix[i - 1].p < (alpha * i / (float)m)) {
            nrej = i;
            break;
        }
    }
    if (rejected_idx) {
        for (i = 0; i < nrej; i++) {
            rejected_idx[i] = ix[i].i;
        }
    }
    free(ix);
    return nrej;
}

/* lofreq_call.c:1523-1527: float / long long division, then 80-bit log10l */
int orc_snvqual_thresh(float sig, long long bonf_subst)
{
    int t = INT_MAX;
    if (bonf_subst) {
        t = orc_prob_to_phred(sig / bonf_subst);
        if (t < 0) {
            t = 0;
        }
    }
    return t;
}

/* `lofreq filter` as run by `lofreq call` (lofreq_call.c:1506-1538) on SNV records:
 *   QUAL threshold (lofreq_filter.c:313-323), and unless --no-defaults:
 *   DP >= 10 (lofreq_filter.c:1095-1097, 270-305) and strand-bias FDR alpha 0.001 with the
 *   "alt mostly on one strand" compound rule (lofreq_filter.c:57, 210-236, 582-677, 1089-1094).
 * The SB multiple-testing pass runs over ALL input records (first pass of the VCF). */
int orc_default_filter(const int *qual, const int *dp, const int *sb, const int *alt_fw,
                       const int *alt_rv, long n, int snvqual_thresh, int apply_defaults, int *keep)
{
    long i;
    for (i = 0; i < n; i++) {
        keep[i] = 1;
    }
    if (apply_defaults && n > 0) {
        double *sbp = malloc(sizeof(double) * (size_t)n);
        long *rej = malloc(sizeof(long) * (size_t)n);
        long nrej;
        const double alpha = 0.001;
        for (i = 0; i < n; i++) {
            sbp[i] = orc_phred_to_prob(sb[i]);              /* lofreq_filter.c:611 */
        }
        nrej = orc_fdr(sbp, n, alpha, n, rej);              /* ntests = #variants, :620-621 */
        for (i = 0; i < nrej; i++) {
            long v = rej[i];
            float ratio = ((alt_fw[v] > alt_rv[v]) ? alt_fw[v] : alt_rv[v])
                          / (float)(alt_fw[v] + alt_rv[v]);  /* lofreq_filter.c:227 */
            if (ratio > 0.85) {                             /* ALT_STRAND_RATIO, :57, :231 */
                keep[v] = 0;
            }
        }
        free(sbp);
        free(rej);
        for (i = 0; i < n; i++) {
            if (dp[i] < 10) {                               /* :301-303 */
                keep[i] = 0;
            }
        }
    }
    if (snvqual_thresh) {
        for (i = 0; i < n; i++) {
            if (qual[i] > -1 && qual[i] < snvqual_thresh) { /* :319 */
                keep[i] = 0;
            }
        }
    }
    return 0;
}

/* ---- base alignment quality (BAQ) ------------------------------------------------------------
 * SURVEY 8(f) rank 1.  orc_kpa_glocal restates kpa_ext_glocal (kprobaln_ext.c:80-270, samtools 0.1.19
 * kprobaln with LoFreq's posterior-matrix extension left out: pd == NULL), the banded profile-HMM
 * forward/backward in scaled doubles; orc_baq_read restates the BAQ half of bam_prob_realn_core_ext
 * (bam_md_ext.c:260-491).  Pinned bitwise against the reference's own kprobaln_ext.c compiled unmodified into
 * oracle/_ref/libref_parts.so (tests/test_baq.py) and against `lb` tags written by the 2.1.4 binary's
 * `lofreq alnqual` (tests/golden/baq_*.json).  htslib's seq_nt16_table / seq_nt16_int (absent here; hts.c)
 * are used by the reference only to map bases to 0..3 / 4: A,C,G,T (either case) -> 0..3, anything else -> 4. */

#define ORC_EI .25
#define ORC_EM .33333333333

static inline int orc_band_u(int bw, int i, int k)      /* set_u, kprobaln_ext.c:46 */
{
    int x = i - bw;
    x = x > 0 ? x : 0;
    return (k - x + 1) * 3;
}

static inline double orc_emit(int r, int qy, double ql) /* the emission term of kprobaln_ext.c:143, 163, 222 */
{
    return (r > 3 || qy > 3) ? 1. : (r == qy ? 1. - ql : ql * ORC_EM);
}

/* pd (optional): posterior matrix, (l_query + 1) rows of (2 bw + 1) * 3 + 6 doubles with bw = *ret_bw
 * (kprobaln_ext.c:266-270); the caller sizes it for the largest possible band */
int orc_kpa_glocal_pd(const uint8_t *ref0, int l_ref, const uint8_t *query0, int l_query, const uint8_t *iqual,
                      float par_d, float par_e, int par_bw, int *state, uint8_t *q, double *pd, int *ret_bw)
{
    static float qual2prob[256];
    const uint8_t *ref = ref0 - 1, *query = query0 - 1;         /* 1-based, :98 */
    double *F, *B, *s, m[9], sI, sM, bI, bM;
    float *qual;
    int bw, bw2, W, i, k, Pr;
    if (l_ref <= 0 || l_query <= 0) {
        return 0;                                               /* :89 */
    }
    bw = l_ref > l_query ? l_ref : l_query;                     /* :99-101 */
    if (bw > par_bw) bw = par_bw;
    if (bw < abs(l_ref - l_query)) bw = abs(l_ref - l_query);
    if (ret_bw) *ret_bw = bw;
    bw2 = bw * 2 + 1;
    W = bw2 * 3 + 6;
    F = calloc((size_t)(l_query + 1) * W, sizeof(double));
    B = calloc((size_t)(l_query + 1) * W, sizeof(double));
    s = calloc((size_t)l_query + 2, sizeof(double));
    qual = calloc((size_t)l_query + 1, sizeof(float));
    if (qual2prob[0] == 0) {
        for (i = 0; i < 256; ++i) qual2prob[i] = pow(10, -i / 10.);   /* float table, :121-123 */
    }
    for (i = 1; i <= l_query; ++i) qual[i] = qual2prob[iqual ? iqual[i - 1] : 30];
    sM = sI = 1. / (2 * l_query + 2);                           /* :127-132; par_d / par_e are floats */
    m[0] = (1 - par_d - par_d) * (1 - sM); m[1] = m[2] = par_d * (1 - sM);
    m[3] = (1 - par_e) * (1 - sI); m[4] = par_e * (1 - sI); m[5] = 0.;
    m[6] = 1 - par_e; m[7] = 0.; m[8] = par_e;
    bM = (1 - par_d) / l_ref; bI = par_d / l_ref;
#define FI(i_) (F + (size_t)(i_) * W)
#define BI(i_) (B + (size_t)(i_) * W)
    /* forward, :134-190 */
    FI(0)[orc_band_u(bw, 0, 0)] = s[0] = 1.;
    {
        double *fi = FI(1), sum = 0.;
        int end = l_ref < bw + 1 ? l_ref : bw + 1, b_, e_;
        for (k = 1; k <= end; ++k) {
            const int u = orc_band_u(bw, 1, k);
            const double e = orc_emit(ref[k], query[1], qual[1]);
            fi[u + 0] = e * bM; fi[u + 1] = ORC_EI * bI;
            sum += fi[u] + fi[u + 1];
        }
        s[1] = sum;
        b_ = orc_band_u(bw, 1, 1); e_ = orc_band_u(bw, 1, end) + 2;
        for (k = b_; k <= e_; ++k) fi[k] /= sum;
    }
    for (i = 2; i <= l_query; ++i) {
        double *fi = FI(i), *fi1 = FI(i - 1), sum = 0., qli = qual[i];
        int beg = 1, end = l_ref, x, b_, e_;
        const int qyi = query[i];
        x = i - bw; beg = beg > x ? beg : x;
        x = i + bw; end = end < x ? end : x;
        for (k = beg; k <= end; ++k) {
            const int u = orc_band_u(bw, i, k), v11 = orc_band_u(bw, i - 1, k - 1), v10 = orc_band_u(bw, i - 1, k),
                      v01 = orc_band_u(bw, i, k - 1);
            const double e = orc_emit(ref[k], qyi, qli);
            fi[u + 0] = e * (m[0] * fi1[v11 + 0] + m[3] * fi1[v11 + 1] + m[6] * fi1[v11 + 2]);
            fi[u + 1] = ORC_EI * (m[1] * fi1[v10 + 0] + m[4] * fi1[v10 + 1]);
            fi[u + 2] = m[2] * fi[v01 + 0] + m[8] * fi[v01 + 2];
            sum += fi[u] + fi[u + 1] + fi[u + 2];
        }
        s[i] = sum;
        b_ = orc_band_u(bw, i, beg); e_ = orc_band_u(bw, i, end) + 2;
        for (k = b_, sum = 1. / sum; k <= e_; ++k) fi[k] *= sum;
    }
    {
        double sum = 0.;
        for (k = 1; k <= l_ref; ++k) {
            const int u = orc_band_u(bw, l_query, k);
            if (u < 3 || u >= bw2 * 3 + 3) continue;
            sum += FI(l_query)[u + 0] * sM + FI(l_query)[u + 1] * sI;
        }
        s[l_query + 1] = sum;
    }
    {   /* likelihood, :191-205 */
        double p = 1., Pr1 = 0.;
        for (i = 0; i <= l_query + 1; ++i) {
            p *= s[i];
            if (p < 1e-100) Pr1 += -4.343 * log(p), p = 1.;
        }
        Pr1 += -4.343 * log(p * l_ref * l_query);
        Pr = (int)(Pr1 + .499);
    }
    /* backward, :206-238 */
    for (k = 1; k <= l_ref; ++k) {
        const int u = orc_band_u(bw, l_query, k);
        double *bi = BI(l_query);
        if (u < 3 || u >= bw2 * 3 + 3) continue;
        bi[u + 0] = sM / s[l_query] / s[l_query + 1]; bi[u + 1] = sI / s[l_query] / s[l_query + 1];
    }
    for (i = l_query - 1; i >= 1; --i) {
        int beg = 1, end = l_ref, x, b_, e_;
        double *bi = BI(i), *bi1 = BI(i + 1), y = (i > 1), qli1 = qual[i + 1];
        const int qyi1 = query[i + 1];
        x = i - bw; beg = beg > x ? beg : x;
        x = i + bw; end = end < x ? end : x;
        for (k = end; k >= beg; --k) {
            const int u = orc_band_u(bw, i, k), v11 = orc_band_u(bw, i + 1, k + 1), v10 = orc_band_u(bw, i + 1, k),
                      v01 = orc_band_u(bw, i, k + 1);
            const double e = (k >= l_ref ? 0 : orc_emit(ref[k + 1], qyi1, qli1)) * bi1[v11];
            bi[u + 0] = e * m[0] + ORC_EI * m[1] * bi1[v10 + 1] + m[2] * bi[v01 + 2];
            bi[u + 1] = e * m[3] + ORC_EI * m[4] * bi1[v10 + 1];
            bi[u + 2] = (e * m[6] + m[8] * bi[v01 + 2]) * y;
        }
        b_ = orc_band_u(bw, i, beg); e_ = orc_band_u(bw, i, end) + 2;
        for (k = b_, y = 1. / s[i]; k <= e_; ++k) bi[k] *= y;
    }
    /* MAP, :254-281 */
    for (i = 1; i <= l_query; ++i) {
        double sum = 0., *fi = FI(i), *bi = BI(i), max = 0.;
        int beg = 1, end = l_ref, x, max_k = -1;
        x = i - bw; beg = beg > x ? beg : x;
        x = i + bw; end = end < x ? end : x;
        for (k = beg; k <= end; ++k) {
            const int u = orc_band_u(bw, i, k);
            double z;
            z = fi[u + 0] * bi[u + 0]; if (z > max) max = z, max_k = (k - 1) << 2 | 0; sum += z;
            z = fi[u + 1] * bi[u + 1]; if (z > max) max = z, max_k = (k - 1) << 2 | 1; sum += z;
            if (pd) {
                double *pdi = pd + (size_t)i * W;
                pdi[u + 0] = fi[u + 0] * bi[u + 0] * s[i];
                pdi[u + 1] = fi[u + 1] * bi[u + 1] * s[i];
                pdi[u + 2] = fi[u + 2] * bi[u + 2] * s[i];
            }
        }
        max /= sum;
        if (state) state[i - 1] = max_k;
        if (q) { k = (int)(-4.343 * log(1. - max) + .499); q[i - 1] = k > 100 ? 99 : k; }
    }
#undef FI
#undef BI
    free(F); free(B); free(s); free(qual);
    return Pr;
}

int orc_kpa_glocal(const uint8_t *ref0, int l_ref, const uint8_t *query0, int l_query, const uint8_t *iqual,
                   float par_d, float par_e, int par_bw, int *state, uint8_t *q)
{
    return orc_kpa_glocal_pd(ref0, l_ref, query0, l_query, iqual, par_d, par_e, par_bw, state, q, NULL, NULL);
}

static inline int orc_base_code(int ch)         /* seq_nt16_int[seq_nt16_table[ch]] */
{
    switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

/* BAQ of one read: bam_prob_realn_core_ext with baq_flag = 1, idaq_flag = 0, no pre-existing tags
 * (bam_md_ext.c:330-470).  cigar: BAM encoding (len << 4 | op; M0 I1 D2 N3 S4 H5 P6 =7 X8).  seq: 0..4.
 * out[l_qseq]: the bytes of the `lb` tag (BAQ + 33).  Returns 1 if a tag was computed, 0 if the read is
 * skipped. */
/* prob_to_sangerq + encode_q, bam_md_ext.c:55-56 */
static inline uint8_t orc_ap_to_char(double p)
{
    const int q = (p < 0.0 + DBL_EPSILON) ? 126 + 1 : ((int)(-10 * log10(p)) + 33);
    return (uint8_t)(q < 33 ? '!' : (q > 126 ? '~' : q));
}

/* idaq (bam_md_ext.c:73-248): indel alignment qualities from the posterior matrix.  iaq / daq: l_qseq bytes
 * ('~' = nothing); returns bit 0 = an `ai` tag is written (n_ins > 0), bit 1 = an `ad` tag (n_del > 0).
 * The read's bases are ORC_SEQ_LETTERS[code] (0..4 = ACGTN, 5..15 = the other letters of seq_nt16_str, "=MRSVWYHKDB": an
 * ambiguity code is its own letter in the repeat scan, :197, as in the reference; everywhere else it behaves like N). */
static int orc_idaq(int pos, const uint32_t *cigar, int n_cigar, const uint8_t *seq, int l_qseq, const char *ref,
                    const double *pd, int W, int xe, int xb, int bw, uint8_t *iaq, uint8_t *daq)
{
    int k, x, y, n_ins = 0, n_del = 0;
    const int bw2 = bw * 2 + 1;
    memset(iaq, '~', (size_t)l_qseq);
    memset(daq, '~', (size_t)l_qseq);
    for (k = 0, x = pos, y = 0; k < n_cigar; ++k) {
        int j;
        const int op = cigar[k] & 0xf, oplen = cigar[k] >> 4;
        if (op == 0 || op == 7 || op == 8) {
            x += oplen; y += oplen;
        } else if (op == 2) {                               /* :107-171; note: the skips do not advance x */
            const int rpos = x, qpos = y;
            int ref_i, del_rep = 0, rep_i = 0;
            double ap = 0;
            const char *del_seq;
            if (qpos == 0) continue;
            if (oplen > 16) continue;
            n_del += 1;
            del_seq = ref + x;
            x += oplen;
            ref_i = x;
            while (ref_i < xe) {
                if (ref[ref_i] != del_seq[rep_i]) break;
                del_rep += 1; ref_i += 1; rep_i += 1;
                if (rep_i >= oplen) rep_i = 0;
            }
            for (j = 0; j < del_rep + 1; j++) {
                int u;
                if (qpos + j > l_qseq) break;
                u = orc_band_u(bw, qpos + j, rpos - xb + 1 + j);
                if (u < 3 || u >= bw2 * 3 + 3) continue;
                ap += pd[(size_t)(qpos + j) * W + u + 2];
            }
            ap = 1 - ap;
            daq[qpos - 1] = orc_ap_to_char(ap);
        } else if (op == 1) {                               /* :172-233; note: the skips do not advance y */
            const int rpos = x, qpos = y;
            int ref_i, ins_rep = 0, rep_i = 0;
            double ap = 0;
            char ins_seq[17];
            if (oplen > 16) continue;
            n_ins += 1;
            if (qpos == 0) continue;
            for (j = 0; j < oplen; j++) {
                ins_seq[j] = ORC_SEQ_LETTER(seq[y]);
                y++;
            }
            ref_i = x;
            while (ref_i < xe) {
                if (ref[ref_i] != ins_seq[rep_i]) break;
                ins_rep += 1; ref_i += 1; rep_i += 1;
                if (rep_i >= oplen) rep_i = 0;
            }
            for (j = 0; j < ins_rep + 1; j++) {
                int u;
                if (qpos + j + 1 > l_qseq) break;
                u = orc_band_u(bw, qpos + j + 1, rpos - xb + j);
                if (u < 3 || u >= bw2 * 3 + 3) continue;
                ap += pd[(size_t)(qpos + j + 1) * W + u + 1];
            }
            ap = 1 - ap;
            iaq[qpos - 1] = orc_ap_to_char(ap);
        } else if (op == 4) {
            y += oplen;
        }
    }
    return (n_ins ? 1 : 0) | (n_del ? 2 : 0);
}

/* kpa_ext_par_t.d / .e handed to the HMM by orc_baq_idaq_read: kpa_ext_par_lofreq_illumina (kprobaln_ext.c:50,
 * bam_md_ext.c:275) unless orc_set_baq_hmm_params says otherwise (a -DPACBIO_REALN build: kpa_ext_par_lofreq_pacbio =
 * { 0.1, 0.4 }, kprobaln_ext.c:51, bam_md_ext.c:268-273).  Process-wide: set it before the worker threads start. */
static float g_baq_par_d = 0.00001f, g_baq_par_e = 0.4f;
void orc_set_baq_hmm_params(float d, float e)
{
    g_baq_par_d = d;
    g_baq_par_e = e;
}

/* BAQ + IDAQ of one read; iaq / daq may be NULL (then only the lb tag is computed).  Returns bit 0 = lb computed,
 * bit 1 = ai tag present, bit 2 = ad tag present. */
int orc_baq_idaq_read(int pos, const uint32_t *cigar, int n_cigar, const uint8_t *seq, const uint8_t *qual, int l_qseq,
                      const char *ref, int64_t ref_len, int baq_extended, uint8_t *out, uint8_t *iaq, uint8_t *daq);

int orc_baq_read(int pos, const uint32_t *cigar, int n_cigar, const uint8_t *seq, const uint8_t *qual, int l_qseq,
                 const char *ref, int64_t ref_len, int baq_extended, uint8_t *out)
{
    return orc_baq_idaq_read(pos, cigar, n_cigar, seq, qual, l_qseq, ref, ref_len, baq_extended, out, NULL, NULL) & 1;
}

int orc_baq_idaq_read(int pos, const uint32_t *cigar, int n_cigar, const uint8_t *seq, const uint8_t *qual, int l_qseq,
                      const char *ref, int64_t ref_len, int baq_extended, uint8_t *out, uint8_t *iaq, uint8_t *daq)
{
    int k, i, bw, x, y, yb, ye, xb, xe, has_indel = 0, hmm_bw = 0, rc = 1;
    uint8_t *r, *q, *bq;
    int *state;
    double *pd = NULL;
    if (l_qseq == 0) {
        return 0;
    }
    x = pos; y = 0; yb = ye = xb = xe = -1;                     /* :312-340 */
    for (k = 0; k < n_cigar; ++k) {
        const int op = cigar[k] & 0xf, l = cigar[k] >> 4;
        if (op == 0 || op == 7 || op == 8) {
            if (yb < 0) yb = y;
            if (xb < 0) xb = x;
            ye = y + l; xe = x + l;
            x += l; y += l;
        } else if (op == 4 || op == 1) {
            y += l;
            if (op == 1) has_indel = 1;
        } else if (op == 2 || op == 3) {
            x += l;
            if (op == 2) has_indel = 1;
        }
    }
    bw = 7;                                                     /* :372-380 */
    if (abs((xe - xb) - (ye - yb)) > bw) bw = abs((xe - xb) - (ye - yb)) + 3;
    xb -= yb + bw / 2; if (xb < 0) xb = 0;
    xe += l_qseq - ye + bw / 2;
    if (xe - xb - l_qseq > bw) {
        xb += (xe - xb - l_qseq - bw) / 2, xe -= (xe - xb - l_qseq - bw) / 2;
    }
    bq = calloc((size_t)l_qseq + 1, 1);
    memcpy(bq, qual, (size_t)l_qseq);                           /* :391-392: bases outside match blocks keep their BQ */
    r = calloc((size_t)(xe - xb > 0 ? xe - xb : 1), 1);
    for (i = xb; i < xe; ++i) {                                 /* :396-399; ref[] ends with NUL */
        if (i >= ref_len || ref[i] == 0) { xe = i; break; }
        r[i - xb] = (uint8_t)orc_base_code((unsigned char)ref[i]);
    }
    state = calloc((size_t)l_qseq, sizeof(int));
    q = calloc((size_t)l_qseq, 1);
    if (iaq && daq && has_indel) {
        const int lr = xe - xb, bmax = (lr > l_qseq ? lr : l_qseq) + abs(lr - l_qseq) + bw;
        pd = calloc((size_t)(l_qseq + 1) * ((size_t)(2 * bmax + 1) * 3 + 6), sizeof(double));
    }
    orc_kpa_glocal_pd(r, xe - xb, seq, l_qseq, qual, g_baq_par_d, g_baq_par_e, bw, state, q, pd, &hmm_bw);
    if (!baq_extended) {                                        /* :409-426 */
        for (k = 0, x = pos, y = 0; k < n_cigar; ++k) {
            const int op = cigar[k] & 0xf, l = cigar[k] >> 4;
            if (op == 0 || op == 7 || op == 8) {
                for (i = y; i < y + l; ++i) {
                    if ((state[i] & 3) != 0 || state[i] >> 2 != x - xb + (i - y)) bq[i] = 0;
                    bq[i] = q[i];
                }
                x += l; y += l;
            } else if (op == 4 || op == 1) {
                y += l;
            } else if (op == 2) {
                x += l;
            }
        }
    } else {                                                    /* :431-451 */
        uint8_t *left = calloc((size_t)l_qseq, 1), *rght = calloc((size_t)l_qseq, 1);
        for (k = 0, x = pos, y = 0; k < n_cigar; ++k) {
            const int op = cigar[k] & 0xf, l = cigar[k] >> 4;
            if (op == 0 || op == 7 || op == 8) {
                for (i = y; i < y + l; ++i) {
                    bq[i] = ((state[i] & 3) != 0 || state[i] >> 2 != x - xb + (i - y)) ? 0 : q[i];
                }
                for (left[y] = bq[y], i = y + 1; i < y + l; ++i) left[i] = bq[i] > left[i - 1] ? bq[i] : left[i - 1];
                for (rght[y + l - 1] = bq[y + l - 1], i = y + l - 2; i >= y; --i) {
                    rght[i] = bq[i] > rght[i + 1] ? bq[i] : rght[i + 1];
                }
                for (i = y; i < y + l; ++i) bq[i] = left[i] < rght[i] ? left[i] : rght[i];
                x += l; y += l;
            } else if (op == 4 || op == 1) {
                y += l;
            } else if (op == 2) {
                x += l;
            }
        }
        free(left); free(rght);
    }
    for (i = 0; i < l_qseq; ++i) {                              /* :456-462 */
        if (bq[i] > 93) bq[i] = 93;
        out[i] = (uint8_t)(bq[i] + 33);
    }
    if (pd) {                                                   /* :476-478 */
        rc |= orc_idaq(pos, cigar, n_cigar, seq, l_qseq, ref, pd, (hmm_bw * 2 + 1) * 3 + 6, xe, xb, hmm_bw, iaq, daq) << 1;
        free(pd);
    }
    free(bq); free(r); free(q); free(state);
    return rc;
}

/* ---- source quality ---------------------------------------------------------------------
 * orc_count_cigar_ops restates count_cigar_ops (samutils.c:437-614), orc_source_qual restates source_qual
 * (plp.c:427-593) as mplp_func calls it (plp.c:727-730; the caller then stores max(sq, 0) in the `sq` tag).
 * Pinned against the SQ track printed by the 2.1.4 binary's `lofreq plpsummary -s` (tests/golden/srcq_*.json).
 * seq holds codes 0..4 (A,C,G,T,N) and 5..15 (the other letters of seq_nt16_str): seq_nt16_str of the BAM base is compared
 * with the reference letter as is (:486-489), so an N base matches an N reference and an R an R.
 * ign (optional): one byte per reference position, != 0 where the -S/--ign-vcf list holds a variant
 * (var_in_ign_list, plp.c:305-323, keyed by chrom and pos only). */
#define ORC_INDEL_QUAL_DEFAULT 45                                   /* samutils.c:51 */

int orc_count_cigar_ops(int counts[4], int *quals[4], int pos, const uint32_t *cigar, int n_cigar,
                        const uint8_t *seq, const uint8_t *qual, const char *ref, int64_t ref_len, int min_bq,
                        const uint8_t *ign)
{
    int64_t tpos = pos;
    int qpos = 0, k, i, num_ops = 0;
    memset(counts, 0, 4 * sizeof(int));
    for (k = 0; k < n_cigar; ++k) {                                 /* :472 */
        const int op = cigar[k] & 0xf;
        const int64_t l = cigar[k] >> 4;
        if (op == 0 || op == 8) {                                   /* BAM_CMATCH, BAM_CDIFF :481 */
            int64_t t;
            for (t = tpos; t < tpos + l; t++) {
                const char ref_nt = (t >= 0 && t < ref_len) ? ref[t] : '\0';
                const char read_nt = ORC_SEQ_LETTER(seq[qpos]);
                const int bq = qual[qpos];
                const int actual = (ref_nt != read_nt || op == 8) ? 1 : 0;      /* :489-493 */
                if (bq < min_bq) {                                  /* :496-502 */
                    qpos += 1;
                    continue;
                }
                if (ign && actual == 1 && t >= 0 && t < ref_len && ign[t]) {    /* :505-519 */
                    qpos += 1;
                    continue;
                }
                counts[actual] += 1;
                if (quals) {
                    quals[actual][counts[actual] - 1] = bq;
                }
                qpos += 1;
            }
            tpos += l;
        } else if (op == 1 || op == 2) {                            /* BAM_CINS, BAM_CDEL :533 */
            const int64_t vpos = op == 1 ? tpos - 1 : tpos;         /* :542-545 */
            if (ign && vpos >= 0 && vpos < ref_len && ign[vpos]) {  /* :547-555 */
                if (op == 1) {
                    qpos += (int)l;
                }
                continue;                                           /* NB a skipped deletion does not advance tpos */
            }
            if (op == 1) {
                counts[2] += 1;                                     /* one operation per indel :563 */
                if (quals) {
                    quals[2][counts[2] - 1] = ORC_INDEL_QUAL_DEFAULT;
                }
                qpos += (int)l;
            } else {
                counts[3] += 1;
                if (quals) {
                    quals[3][counts[3] - 1] = ORC_INDEL_QUAL_DEFAULT;
                }
                tpos += l;
            }
        } else if (op == 3) {                                       /* BAM_CREF_SKIP :581 */
            tpos += l;
        } else if (op == 4) {                                       /* BAM_CSOFT_CLIP :584 */
            qpos += (int)l;
        }                                                           /* H, P, =: nothing moves (:590-593) */
    }
    for (i = 0; i < 4; i++) {
        num_ops += counts[i];
    }
    return num_ops;
}

int orc_source_qual(int pos, const uint32_t *cigar, int n_cigar, const uint8_t *seq, const uint8_t *qual, int l_qseq,
                    const char *ref, int64_t ref_len, int nonmatch_qual, int min_bq, const uint8_t *ign)
{
    int counts[4], *quals[4], i, j, n_ep, idx = 0, nonmatch = 0, src_qual = -1;
    double *ep = NULL, *probvec, src_prob;
    long double unused;
    for (i = 0; i < 4; i++) {
        quals[i] = (int *)malloc((size_t)(l_qseq + n_cigar + 1) * sizeof(int));
    }
    n_ep = orc_count_cigar_ops(counts, quals, pos, cigar, n_cigar, seq, qual, ref, ref_len, min_bq, ign);
    if (n_ep < 1) {                                                 /* :468-474 */
        goto done;
    }
    ep = (double *)malloc((size_t)n_ep * sizeof(double));
    for (i = 0; i < 4; i++) {                                       /* :486-509 */
        if (i != 0) {
            nonmatch += counts[i];
        }
        for (j = 0; j < counts[i]; j++) {
            const int q = nonmatch_qual >= 0 ? nonmatch_qual : quals[i][j];     /* every category, :500-504 */
            ep[idx++] = orc_phred_to_prob(q);
        }
    }
    if (nonmatch > 0) {                                             /* :514-516 */
        nonmatch -= 1;
    }
    if (nonmatch == 0) {                                            /* :517-524 */
        src_qual = orc_prob_to_phred(LDBL_MIN);
        goto done;
    }
    qsort(ep, (size_t)n_ep, sizeof(double), orc_dbl_cmp);           /* :551 */
    probvec = orc_poissbin(&unused, ep, n_ep, nonmatch, 1, 0.05, NULL);         /* :552-553, bonf 1.0 -> 1 */
    errno = 0;                                                      /* :555-564 */
    feclearexcept(FE_ALL_EXCEPT);
    src_prob = exp(probvec[nonmatch - 1]);
    if (errno || fetestexcept(FE_INVALID | FE_DIVBYZERO | FE_OVERFLOW | FE_UNDERFLOW)) {
        src_prob = src_prob < DBL_EPSILON ? DBL_MIN : DBL_MAX;
    }
    free(probvec);
    src_qual = orc_prob_to_phred(1.0 - src_prob);                   /* :567 */
done:
    for (i = 0; i < 4; i++) {
        free(quals[i]);
    }
    free(ep);
    return src_qual;
}


/* lofreq_uniq.c:262-268: an AF parsed from the VCF that is out of bounds is logged ("LOG_FATAL", which does not exit)
 * and RESET -- af < 0 -> 0.01, af > 1 -> 1.0 -- and the variant is processed with the new value */
static float orc_uniq_af(float af)
{
    if (af < 0.0 || af > 1.0) {
        float new_af = af < 0.0 ? 0.01 : 1.0;
        af = new_af;
    }
    return af;
}

/* ---- `lofreq uniq --use-det-lim`: uniq_snv, lofreq_uniq.c:222-333 ------------------------
 * per column: default varcall_conf, plp_to_errprobs, alt_counts = {af * num_err_probs (float product, truncated),
 * 0, 0}, snpcaller(bonf 1, alpha (double)0.01f, -1); flag[col] = pvalues[0] * (float)bonf < alpha (:314), the
 * condition for the UNIQ tag.  Pinned against `lofreq uniq --use-det-lim --output-all` of the 2.1.4 binary
 * (tests/golden/uniq_*.json). */
int orc_uniq_detlim_batch(const uint8_t *nt, const uint8_t *bq, const uint8_t *baq, const uint8_t *mq,
                          const uint8_t *sq, const uint64_t *col_off, const uint8_t *ref_base, int64_t ncols,
                          const float *af, uint8_t *flag, long double *pvalue)
{
    int64_t c;
    for (c = 0; c < ncols; c++) {
        const uint64_t o0 = col_off[c];
        const int64_t n_obs = (int64_t)(col_off[c + 1] - o0);
        orc_conf conf;
        double *ep;
        int n_ep = 0, alt_base[3], alt_counts[3], alt_raw[3];
        long double pv[3];
        const int bonf = 1;
        const float alpha = 0.01;                                   /* :286-287 */
        flag[c] = 0;
        if (pvalue) {
            pvalue[c] = LDBL_MAX;
        }
        if (n_obs < 1 || ref_base[c] == 'N') {                      /* :254-256; an 'N' reference has no variant */
            continue;
        }
        orc_conf_init(&conf);                                       /* init_varcall_conf, :289 */
        ep = (double *)malloc((size_t)n_obs * sizeof(double));
        if (orc_col_errprobs(ep, &n_ep, alt_base, alt_counts, alt_raw, nt + o0, bq + o0, baq ? baq + o0 : NULL,
                             mq + o0, sq ? sq + o0 : NULL, n_obs, (char)ref_base[c], &conf)) {
            free(ep);
            return -1;
        }
        /* NB no qsort here: uniq_snv hands the probabilities to snpcaller in plp_to_errprobs order (:293-305) */
        alt_counts[0] = orc_uniq_af(af[c]) * n_ep;                  /* :262-268, :300 */
        alt_counts[1] = alt_counts[2] = 0;
        orc_snpcaller(pv, NULL, ep, n_ep, alt_counts, bonf, alpha, NULL);       /* :303 */
        if (pvalue) {
            pvalue[c] = pv[0];
        }
        flag[c] = (pv[0] * (float)bonf < alpha) ? 1 : 0;            /* :314 */
        free(ep);
    }
    return 0;
}

/* ---- `lofreq uniq`, default (binomial) mode ------------------------------------------------
 *
 * binom() (binom.c:52-69) calls cdfbin(which = 1) of cdflib90 -- THIRD-PARTY code vendored in the reference tree
 * (src/cdflib90/dcdflib.c:1727; argument checks :1880-1935) -- which hands P(X <= s), X ~ Binomial(xn, pr), to cumbin
 * (:4966-5031: Abramowitz & Stegun 26.5.24) = the regularised incomplete beta function I_{1-pr}(xn - s, s + 1),
 * evaluated by bratio (TOMS 708).  This restatement evaluates the same quantity from its DEFINITION, the sum of the
 * binomial probabilities of 0..s, in 80-bit arithmetic; it agrees with the reference's own compiled cdflib
 * (oracle/_ref/libref_parts.so: binom.c + dcdflib.c + ipmpar.c unmodified) to better than 1e-11 relative wherever
 * that returns a normal double (tests/test_uniq.py), and the phred value uniq_snv derives from it is an integer.
 * Returns cdfbin's status: 0, -5 (xn <= 0), -4 (s outside [0, xn]), -6 (pr outside [0, 1]). */
int orc_binom_cdf(double *p, int num_trials, int num_success, double prob_success)
{
    const long double n = (long double)num_trials, pr = (long double)prob_success;
    long double sum = 0.0L, lq, lp, lgn;
    int i;
    *p = 0.0;
    if (num_trials <= 0) {
        return -5;
    }
    if (num_success < 0 || num_success > num_trials) {
        return -4;
    }
    if (prob_success < 0.0 || prob_success > 1.0) {
        return -6;
    }
    if (num_success >= num_trials || prob_success <= 0.0) {          /* cumbin :5021-5028; cumbet's x <= 0 exit */
        *p = 1.0;
        return 0;
    }
    if (prob_success >= 1.0) {                                       /* cumbet's y <= 0 exit: all mass at xn > s */
        *p = 0.0;
        return 0;
    }
    lp = logl(pr);
    lq = log1pl(-pr);
    lgn = lgammal(n + 1.0L);
    for (i = 0; i <= num_success; i++) {
        const long double li = (long double)i;
        sum += expl(lgn - lgammal(li + 1.0L) - lgammal(n - li + 1.0L) + li * lp + (n - li) * lq);
    }
    *p = (double)(sum > 1.0L ? 1.0L : sum);
    return 0;
}

/* uniq_snv's default branch (lofreq_uniq.c:254-256, 335-393) for a batch of columns: coverage = coverage_plp
 * (here: the observations of the column; SNVs only), alt_count = base_count(p, alt) = every base of that
 * nucleotide in the column whatever its quality (plp.c:128-132), pvalue = binom(coverage, alt_count, af) -- one-sided:
 * "the other sample shows at most this many alt bases although the variant has frequency af" --
 * UQ = PROB_TO_PHREDQUAL_SAFE(pvalue) (:386; utils.h:46).  uq[col] = -1 where the reference adds no UQ tag
 * (coverage < 1, :254; binom() failed, :381-384). */
int orc_uniq_binom_batch(const uint8_t *nt, const uint64_t *col_off, const int32_t *coverage_plp_or_null, int64_t ncols,
                         const float *af, const char *alt_base, int32_t *uq, double *pvalue_or_null)
{
    int64_t c;
    for (c = 0; c < ncols; c++) {
        const uint64_t o0 = col_off[c], o1 = col_off[c + 1];
        const int coverage = coverage_plp_or_null ? coverage_plp_or_null[c] : (int)(o1 - o0);
        const char ab = alt_base[c];
        const int code = (ab == 'A' || ab == 'a') ? 0 : (ab == 'C' || ab == 'c') ? 1 : (ab == 'G' || ab == 'g') ? 2
                         : (ab == 'T' || ab == 't') ? 3 : 4;          /* bam_nt4_table */
        int alt_count = 0;
        uint64_t o;
        double pv = 0.0;
        uq[c] = -1;
        if (pvalue_or_null) {
            pvalue_or_null[c] = -1.0;
        }
        if (coverage < 1) {
            continue;
        }
        for (o = o0; o < o1; o++) {
            alt_count += ((nt[o] & 7) == code);
        }
        if (orc_binom_cdf(&pv, coverage, alt_count, (double)orc_uniq_af(af[c])) != 0) {       /* :262-268, :381 */
            continue;
        }
        uq[c] = (pv <= 0.0) ? INT_MAX : (int)(-10.0 * log10l(pv));
        if (pvalue_or_null) {
            pvalue_or_null[c] = pv;
        }
    }
    return 0;
}

/* apply_uniq_filter_mtc (lofreq_uniq.c:140-206): uniq_probs = PHREDQUAL_TO_PROB(UQ) -- of the INTEGER phred value, 0
 * where a variant carries no UQ tag (uniq_phred_from_var, :110-121) -- corrected by bonf / holm / fdr (multtest.c) over
 * ntests (0 = the number of variants, :155-157); a variant is filtered when its corrected value exceeds alpha (fdr: the
 * rejected ones are set to -1 first, :186-191).  mtc_type: 1 bonf, 2 holm, 3 fdr (multtest.h).  pass[i] = 1: not filtered. */
int orc_uniq_mtc(const int32_t *uq, long n, int mtc_type, double alpha, long ntests, uint8_t *pass)
{
    double *pr = (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    long i;
    if (!pr) {
        return -1;
    }
    if (!ntests) {
        ntests = n;
    }
    for (i = 0; i < n; i++) {
        const int q = uq[i] < 0 ? 0 : uq[i];
        pr[i] = (q == INT_MAX) ? DBL_MIN : pow(10.0, -1.0 * q / 10.0);          /* utils.h:42 */
    }
    if (mtc_type == 1) {
        orc_bonf_corr(pr, n, ntests);
    } else if (mtc_type == 2) {
        orc_holm_bonf_corr(pr, n, alpha, ntests);
    } else if (mtc_type == 3) {
        long *idx = (long *)malloc((size_t)(n > 0 ? n : 1) * sizeof(long));
        const long nrej = orc_fdr(pr, n, alpha, ntests, idx);
        for (i = 0; i < nrej; i++) {
            pr[idx[i]] = -1;
        }
        free(idx);
    } else {
        free(pr);
        return -1;
    }
    for (i = 0; i < n; i++) {
        pass[i] = pr[i] > alpha ? 0 : 1;
    }
    free(pr);
    return 0;
}

/* orc_baq_idaq_read over reads [r0, r1) of packed read arrays (the layout of orc_reads): lb / ai / ad bytes per base
 * ('~' = 126 where a read gets no ai / ad, like the tags the reference leaves out), flags[r] bit 0 / 1 = read has an
 * ai / ad tag.  What mplp_func does per read before the pileup (plp.c:667-683). */
int orc_baq_idaq_reads(const orc_reads *rd, int64_t r0, int64_t r1, int baq_extended, int want_idaq,
                       uint8_t *lb, uint8_t *ai, uint8_t *ad, uint8_t *flags)
{
    int64_t r;
    for (r = r0; r < r1; r++) {
        const int64_t so = rd->seq_off[r], l = rd->seq_off[r + 1] - so;
        const int64_t c0 = rd->cigar_off[r];
        const int nc = (int)(rd->cigar_off[r + 1] - c0);
        int rc;
        if (want_idaq) {
            memset(ai + so, 126, (size_t)l);
            memset(ad + so, 126, (size_t)l);
        }
        rc = orc_baq_idaq_read(rd->pos[r], rd->cigar + c0, nc, rd->seq + so, rd->qual + so, (int)l, rd->ref, rd->ref_len,
                               baq_extended, lb + so, want_idaq ? ai + so : NULL, want_idaq ? ad + so : NULL);
        if (!(rc & 1)) {
            memset(lb + so, 33, (size_t)l);         /* (zero-length reads aside, the tag is always written) */
        }
        if (flags) {
            flags[r] = (uint8_t)((rc >> 1) & 3);
        }
    }
    return 0;
}

