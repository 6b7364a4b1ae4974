/*
 * lfq_dp.hip -- the Poisson-binomial tail test on CDNA4: snpcaller() / poissbin() /
 * pruned_calc_prob_dist() of the reference (snpcaller.c:831-1205).
 *
 * Recurrence, per kept observation with error probability p:
 *     cell[k] <- cell[k]*(1-p) + cell[k-1]*p      k = 0..K-1   (P(X = k))
 *     tail    <- tail + cell[K-1]*p                              (P(X >= K), absorbing)
 * and the pruning test tail*bonf > sig (snpcaller.c:950, 1155).
 *
 * Mapping: cells across lanes, C consecutive cells per lane (blocked), 64*C cells per strip.  The
 * left neighbour arrives by DPP wave_shr:1, the row's p by v_readlane (SGPR broadcast).  Every lane
 * carries a binary exponent e for its C cells (value = v*2^e), renormalised every 8 rows, so the
 * 1e-4932-range tails the reference reaches in log space are representable.
 *
 *   lfq_dp_wave_kernel<1>  light columns, K < 64: one wavefront per column, one cell per lane.
 *   lfq_dp_wave_kernel<8>  mid columns, 64 <= K < 505: one wavefront per column, 8 cells per lane.
 *                        Both: persistent grid, static column striding, no strip exchange.
 *   lfq_dp_big_kernel    K >= 505: one 16-wave workgroup per column, C = 8, strip w on wave w.  Strips
 *                        run as a software pipeline over 64-row chunks (wave w works on chunk t-w at
 *                        step t); the boundary cell of strip w reaches strip w+1 through a
 *                        double-buffered LDS slab, one s_barrier per step.  More than 16 strips
 *                        (K > 8191) run in passes with the pass boundary in global scratch.
 *
 * The three kernels run concurrently on three HIP streams (lfq_api.hip).
 */
#include "lfq_device.h"

#define LFQ_LN2_HI 6.93147180369123816490e-01
#define LFQ_LN2_LO 1.90821492927058770002e-10
/* glibc's exp(x) raises FE_UNDERFLOW (result below DBL_MIN) for x < ln(2^-1022); pinned by
 * tests/test_oracle_kat.py::test_exp_underflow_threshold */
#define LFQ_EXP_UNDERFLOW_X (-708.3964185322641)
#define LFQ_DBL_EPS 2.220446049250313e-16

#define LFQ_HEAVY_WAVES 16
#define LFQ_HEAVY_C 8

struct LfqColCtx {
    int col;
    uint64_t off0;
    int64_t n_obs;
    int ref_code;
    int median_ref_bq;
    int K;
    int64_t bonf;
    double bonf_d;
    double sig_s;
};

__device__ __forceinline__ void lfq_col_setup(LfqColCtx &cx, int col, const LfqTracksDev &T, const LfqParams &P,
                                              const lfq_col_counts &cnt, const LfqWork &W)
{
    cx.col = col;
    cx.off0 = T.col_off[col];
    cx.n_obs = (int64_t)(T.col_off[col + 1] - cx.off0);
    const uint32_t rb = T.ref_base[col];
    cx.ref_code = (rb == 'A') ? 0 : (rb == 'C') ? 1 : (rb == 'G') ? 2 : 3;
    cx.median_ref_bq = cnt.median_ref_bq;
    cx.K = cnt.kmax;
    /* running Bonferroni factor at this column (lofreq_call.c:794-800) */
    int64_t bonf = P.bonf_base;
    if (P.bonf_dynamic) {
        bonf = ((P.bonf_base == 1) ? 0 : P.bonf_base) + 3 * (int64_t)W.tested_prefix[col];
    }
    cx.bonf = bonf;
    cx.bonf_d = (double)bonf;
    cx.sig_s = P.sig * (1.0 + P.prune_slack);
}

/* evaluate the 64 observations of chunk `ch`: keep mask + effective p and 1-p per lane, with the
 * reference's guards against log(0) (snpcaller.c:872-881) expressed on the probabilities */
__device__ __forceinline__ uint64_t lfq_eval_chunk(const LfqColCtx &cx, int64_t ch, const LfqTracksDev &T,
                                                   const LfqParams &P, const LfqLuts *L, double *ps, double *qf)
{
    const int64_t idx = ch * 64 + lfq_lane();
    LfqObs o;
    o.keep = false;
    o.p = 0.0;
    if (idx < cx.n_obs) {
        const uint64_t g = cx.off0 + (uint64_t)idx;
        o = lfq_eval_obs(T.nt[g], T.bq[g], T.baq ? T.baq[g] : 255u, T.mq[g], T.sq ? T.sq[g] : 255u,
                         cx.ref_code, cx.median_ref_bq, P, L);
    }
    *ps = (fabs(o.p) < LFQ_DBL_EPS) ? LFQ_DBL_EPS : o.p;
    *qf = (fabs(o.p - 1.0) < LFQ_DBL_EPS) ? 1.0 + (-o.p + LFQ_DBL_EPS) : 1.0 - o.p;
    return __ballot(o.keep);
}

template <int C>
struct LfqStrip {
    double v[C];
    int e, de, e_in, rows;
    bool all_zero;     /* wave-uniform: nothing has entered this strip yet */
};

template <int C>
__device__ __forceinline__ void lfq_strip_init(LfqStrip<C> &S, bool first_strip, int shift)
{
    const int lane = lfq_lane();
#pragma unroll
    for (int j = 0; j < C; j++) {
        S.v[j] = (first_strip && lane == 0 && j == shift) ? 1.0 : 0.0;
    }
    S.e = S.de = S.e_in = S.rows = 0;
    S.all_zero = !first_strip;
}

/* Advance one strip over the kept rows of one chunk.  (bv,be): per-lane incoming boundary of row
 * `lane` (value, exponent) when has_in; (ov,oe): per-lane outgoing boundary when has_out.
 * Returns true when the pruning test fires (only evaluated on the strip that owns the tail). */
template <int C>
__device__ __forceinline__ bool lfq_strip_chunk(LfqStrip<C> &S, uint64_t km, double ps, double qf, bool has_in,
                                                double bv, int be, bool has_out, double &ov, int &oe,
                                                bool is_tail, bool owns_tail, int lt, double bonf_d, double sig_s)
{
    const int lane = lfq_lane();
    while (km) {
        const int i = __builtin_ctzll(km);
        km &= km - 1;
        const double p = lfq_rl_f64(ps, i);
        const double q = lfq_rl_f64(qf, i);
        double x = lfq_shr1_f64(S.v[C - 1]);
        int dei = S.de;
        if (has_in) {
            const double xb = lfq_rl_f64(bv, i);
            const int eb = lfq_rl_i32(be, i);
            S.e_in = eb;
            if (S.all_zero) {
                S.e = eb;                   /* adopt the producer's scale while empty */
                S.de = 0;
                dei = 0;
                if (xb == 0.0) {
                    if (has_out && lane == i) {
                        ov = 0.0;
                        oe = eb;
                    }
                    S.rows++;
                    continue;
                }
                S.all_zero = false;
            }
            if (lane == 0) {
                x = xb;
                dei = eb - S.e;
            }
        }
        if (has_out) {
            const double v63 = lfq_rl_f64(S.v[C - 1], 63);
            const int e63 = lfq_rl_i32(S.e, 63);
            if (lane == i) {
                ov = v63;
                oe = e63;
            }
        }
        const double xs = ldexp(x, dei);
        const double ph = is_tail ? 0.0 : p;     /* nothing flows past the absorbing tail cell */
        const double q0 = is_tail ? 1.0 : q;
#pragma unroll
        for (int j = C - 1; j >= 1; j--) {
            S.v[j] = fma(S.v[j - 1], ph, S.v[j] * q);
        }
        S.v[0] = fma(xs, p, S.v[0] * q0);
        S.rows++;

        if ((S.rows & 7) == 0) {
            /* renormalise: lane maximum to [0.5,1), exponent into e */
            double m = S.v[0];
#pragma unroll
            for (int j = 1; j < C; j++) {
                m = fmax(m, S.v[j]);
            }
            const bool nzl = m > 0.0;
            const uint64_t nz = __ballot(nzl);
            if (nzl) {
                const int ex = __builtin_amdgcn_frexp_exp(m);
#pragma unroll
                for (int j = 0; j < C; j++) {
                    S.v[j] = ldexp(S.v[j], -ex);
                }
                S.e += ex;
            }
            /* empty lanes adopt the scale of the nearest non-empty lane to their left */
            const uint64_t below = nz & ((lane == 0) ? 0ull : (~0ull >> (64 - lane)));
            const int src = below ? (63 - __builtin_clzll(below)) : lane;
            const int e_src = __shfl(S.e, src, 64);
            if (!nzl) {
                S.e = below ? e_src : (has_in ? S.e_in : S.e);
            }
            S.de = lfq_shr1_i32(S.e) - S.e;
            if (owns_tail) {
                const double tv = lfq_rl_f64(S.v[0], lt);
                const int te = lfq_rl_i32(S.e, lt);
                if (ldexp(tv, te) * bonf_d > sig_s) {
                    return true;
                }
            }
        }
    }
    return false;
}

template <int C>
__device__ __forceinline__ bool lfq_strip_final_prune(const LfqStrip<C> &S, int lt, double bonf_d, double sig_s)
{
    const double tv = lfq_rl_f64(S.v[0], lt);
    const int te = lfq_rl_i32(S.e, lt);
    return ldexp(tv, te) * bonf_d > sig_s;
}

/* natural logs of a strip's cells -> probvec[k] (layout of poissbin()'s return array) */
template <int C>
__device__ __forceinline__ void lfq_strip_store_logs(const LfqStrip<C> &S, int gl, int shift, int K,
                                                     double *probvec)
{
    const double ed = (double)S.e;
#pragma unroll
    for (int j = 0; j < C; j++) {
        const int k = gl * C + j - shift;
        if (k >= 0 && k <= K && (k < K || j == 0)) {
            probvec[k] = (S.v[j] > 0.0) ? (ed * LFQ_LN2_HI + (ed * LFQ_LN2_LO + log(S.v[j]))) : -INFINITY;
        }
    }
}

__device__ __forceinline__ double lfq_logaddexp(double a, double b)
{
    const double hi = fmax(a, b), lo = fmin(a, b);
    if (lo == -INFINITY) {
        return hi;
    }
    return hi + log1p(exp(lo - hi));
}

/* probvec_tailsum (snpcaller.c:730-741) as a wave-parallel prefix scan, plus detection of the exp()
 * underflow inside the reference's sequential log_sum chain.  One wavefront. */
__device__ double lfq_tailsum(const double *probvec, int start, int K, bool *fe_flag)
{
    const int lane = lfq_lane();
    double carry = -INFINITY;
    bool flag = false;
    for (int base = start; base <= K; base += 64) {
        const int idx = base + lane;
        const double x = (idx <= K) ? probvec[idx] : -INFINITY;
        double incl = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double y = __shfl_up(incl, d, 64);
            if (lane >= d) {
                incl = lfq_logaddexp(incl, y);
            }
        }
        double excl = __shfl_up(incl, 1, 64);
        excl = (lane == 0) ? carry : lfq_logaddexp(carry, excl);
        if (idx <= K && idx > start) {
            /* the reference evaluates exp(min - max) of (running sum, probvec[idx]) */
            if (-fabs(x - excl) < LFQ_EXP_UNDERFLOW_X) {
                flag = true;
            }
        }
        carry = lfq_logaddexp(carry, lfq_rl_f64(incl, 63));
    }
    *fe_flag = __any(flag);
    return carry;
}

/* per-allele p-values (snpcaller.c:1166-1196) from probvec and the sparse-output append.  One wavefront.
 * `kp` is the K the recurrence was run with (normally cnt.kmax).  `uf_mask` marks alleles whose p-value
 * is proven to be below the 80-bit underflow threshold (the reference returns LDBL_MIN for them);
 * `force_fe` marks every computed tail as "the reference's log_sum chain underflows" (see the shortcut
 * in lfq_dp_big_kernel); `have_probvec` is false when the kp-recurrence was pruned or not needed. */
__device__ void lfq_emit_pvals(const LfqColCtx &cx, const lfq_col_counts &cnt, const double *probvec, int kp,
                               bool have_probvec, unsigned uf_mask, const double *uf_bound, bool force_fe,
                               int rows, const LfqWork &W, lfq_col_pvals *__restrict__ pvals,
                               int64_t pvals_capacity)
{
    double logp[3];
    int status[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const int c = cnt.alt_counts[a];
        logp[a] = 0.0;
        status[a] = LFQ_PV_NONE;
        if (c == 0) {
            continue;
        }
        if (uf_mask & (1u << a)) {
            logp[a] = uf_bound[a];
            status[a] = LFQ_PV_UNDERFLOW;
        } else if (have_probvec) {
            if (c == kp) {
                logp[a] = probvec[kp];
                status[a] = force_fe ? LFQ_PV_LOG_FECLAMP : LFQ_PV_LOG;
            } else if (c < kp) {
                bool fe = false;
                logp[a] = lfq_tailsum(probvec, c, kp, &fe);
                status[a] = (fe || force_fe) ? LFQ_PV_LOG_FECLAMP : LFQ_PV_LOG;
            }
        }
    }
    if (lfq_lane() == 0) {
        const int slot = atomicAdd(&W.counters[LFQ_CNT_PVALS], 1);
        if ((int64_t)slot < pvals_capacity) {
            lfq_col_pvals r;
            r.col = cx.col;
            r.bonf = cx.bonf;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                r.logp[a] = logp[a];
                r.status[a] = (uint8_t)status[a];
            }
#pragma unroll
            for (int i = 0; i < 5; i++) {
                r.pad_[i] = 0;
            }
            r.counts = cnt;
            r.dp_rows = rows;
            r.pad2_ = 0;
            r.reserved_ = 0;
            pvals[slot] = r;
        } else {
            W.counters[LFQ_CNT_OVERFLOW] = 1;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* wave-per-column kernel: light (K < 64, C = 1) and mid (64 <= K < 505, C = 8) columns          */
/* ------------------------------------------------------------------------------------------ */

template <int C>
__global__ __launch_bounds__(256) void lfq_dp_wave_kernel(LfqTracksDev T, LfqParams P,
                                                          const LfqLuts *__restrict__ g_luts,
                                                          const lfq_col_counts *__restrict__ counts, LfqWork W,
                                                          const int32_t *__restrict__ queue, int count_idx,
                                                          lfq_col_pvals *__restrict__ pvals,
                                                          int64_t pvals_capacity, int n_waves)
{
    __shared__ LfqLuts s_luts;
    __shared__ double s_probvec[4][64 * C];
    {
        const double *src = reinterpret_cast<const double *>(g_luts);
        double *dst = reinterpret_cast<double *>(&s_luts);
        for (int i = threadIdx.x; i < (int)(sizeof(LfqLuts) / sizeof(double)); i += blockDim.x) {
            dst[i] = src[i];
        }
    }
    __syncthreads();
    const int lane = lfq_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wave_id = (int)blockIdx.x * 4 + wave;
    const int n_work = W.counters[count_idx];
    double *probvec = s_probvec[wave];

    for (int w = wave_id; w < n_work; w += n_waves) {
        const int col = __builtin_amdgcn_readfirstlane(queue[w]);
        const lfq_col_counts cnt = counts[col];
        LfqColCtx cx;
        lfq_col_setup(cx, col, T, P, cnt, W);
        const int K = cx.K;
        const int shift = (C - K % C) % C;
        const int lt = (K + shift) / C;             /* lane that owns the tail cell (<= 63 by class) */
        const bool is_tail = (lane == lt);
        LfqStrip<C> S;
        lfq_strip_init<C>(S, true, shift);
        const int64_t n_chunks = (cx.n_obs + 63) / 64;
        bool pruned = false;
        double ov = 0.0;
        int oe = 0;
        for (int64_t ch = 0; ch < n_chunks; ch++) {
            double ps, qf;
            const uint64_t km = lfq_eval_chunk(cx, ch, T, P, &s_luts, &ps, &qf);
            if (lfq_strip_chunk<C>(S, km, ps, qf, false, 0.0, 0, false, ov, oe, is_tail, true, lt, cx.bonf_d,
                                   cx.sig_s)) {
                pruned = true;
                break;
            }
        }
        if (pruned || lfq_strip_final_prune<C>(S, lt, cx.bonf_d, cx.sig_s)) {
            continue;
        }
        lfq_strip_store_logs<C>(S, lane, shift, K, probvec);
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        lfq_emit_pvals(cx, cnt, probvec, K, true, 0u, nullptr, false, S.rows, W, pvals, pvals_capacity);
        __builtin_amdgcn_wave_barrier();
    }
}

/* ------------------------------------------------------------------------------------------ */
/* big columns: K >= 505, one 16-wave workgroup per column                                     */
/* ------------------------------------------------------------------------------------------ */

__global__ __launch_bounds__(LFQ_HEAVY_WAVES * 64) void lfq_dp_big_kernel(
    LfqTracksDev T, LfqParams P, const LfqLuts *__restrict__ g_luts, const lfq_col_counts *__restrict__ counts,
    LfqWork W, lfq_col_pvals *__restrict__ pvals, int64_t pvals_capacity, double *__restrict__ scratch,
    int64_t scratch_per_block)
{
    constexpr int C = LFQ_HEAVY_C;
    constexpr int NW = LFQ_HEAVY_WAVES;
    __shared__ LfqLuts s_luts;
    __shared__ double s_bv[2][NW][64];
    __shared__ int s_be[2][NW][64];
    __shared__ int s_col, s_pruned;
    __shared__ double s_mu[NW];
    {
        const double *src = reinterpret_cast<const double *>(g_luts);
        double *dst = reinterpret_cast<double *>(&s_luts);
        for (int i = threadIdx.x; i < (int)(sizeof(LfqLuts) / sizeof(double)); i += blockDim.x) {
            dst[i] = src[i];
        }
    }
    const int lane = lfq_lane();
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n_big = W.counters[LFQ_CNT_BIG];
    double *bnd = scratch + (int64_t)blockIdx.x * scratch_per_block;   /* pass boundary: 2 doubles / obs */

    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            s_col = atomicAdd(&W.counters[LFQ_CNT_HEAD], 1);
            s_pruned = 0;
        }
        __syncthreads();
        const int h = s_col;
        if (h >= n_big) {
            break;
        }
        const int col = W.q_big[h];
        const lfq_col_counts cnt = counts[col];
        LfqColCtx cx;
        lfq_col_setup(cx, col, T, P, cnt, W);
        const int64_t n_chunks = (cx.n_obs + 63) / 64;

        /* Shortcut for p-values below the 80-bit range.  P(X >= c) <= e_c(p) <= mu^c / c!  (union
         * bound + Maclaurin), mu = sum of the error probabilities.  If that bound is below e^-12200
         * the reference's expl() underflows and it reports LDBL_MIN (snpcaller.c:1047-1059, SURVEY
         * App. A.6) whatever the exact value is.  For the remaining alleles of such a column the
         * reference's log_sum chain provably underflows too (the chain spans > 708 in log space), so
         * their p-values are clamped by value: they only need the recurrence up to the largest
         * non-underflowing count. */
        double part = 0.0;
        for (int64_t ch = w; ch < n_chunks; ch += NW) {
            double ps, qf;
            const uint64_t km = lfq_eval_chunk(cx, ch, T, P, &s_luts, &ps, &qf);
            part += ((km >> lane) & 1ull) ? ps : 0.0;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            part += __shfl_xor(part, d, 64);
        }
        if (lane == 0) {
            s_mu[w] = part;
        }
        __syncthreads();
        double mu = 0.0;
        for (int i = 0; i < NW; i++) {
            mu += s_mu[i];
        }
        const double lmu = log(mu);
        unsigned uf_mask = 0;
        double uf_bound[3];
        int kp = 0;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const int c = cnt.alt_counts[a];
            uf_bound[a] = (c > 0) ? (double)c * lmu - lgamma((double)c + 1.0) : 0.0;
            if (c > 0 && uf_bound[a] < -12200.0) {
                uf_mask |= 1u << a;
            } else if (c > kp) {
                kp = c;
            }
        }
        const bool force_fe = uf_mask != 0;
        if (force_fe) {
            /* pruned alleles become LDBL_MAX; that equals the reference's clamp only while the pruning
             * threshold sig/bonf stays above DBL_EPSILON */
            cx.sig_s = fmax(cx.sig_s, 4.5e-16 * cx.bonf_d);
        }
        if (kp == 0) {
            if (w == 0) {
                lfq_emit_pvals(cx, cnt, nullptr, 0, false, uf_mask, uf_bound, true, 0, W, pvals, pvals_capacity);
            }
            continue;
        }
        const int K = kp;
        const int shift = (C - K % C) % C;
        const int Lt = (K + shift) / C;            /* global lane owning the tail cell at j = 0 */
        const int n_strips = Lt / 64 + 1;
        const int lt = Lt % 64;
        double *probvec = bnd + 2 * cx.n_obs + 2;
        bool pruned = false;
        int rows_tail = 0;

        for (int s0 = 0; s0 < n_strips && !pruned; s0 += NW) {
            const int nwp = min(NW, n_strips - s0);       /* strips in this pass */
            const int s = s0 + w;
            const bool active = w < nwp;
            const int gl = s * 64 + lane;
            const bool is_tail = (gl == Lt);
            const bool owns_tail = active && (s == n_strips - 1);
            const bool has_in = (s > 0);
            const bool in_global = has_in && (w == 0);    /* first strip of a later pass */
            const bool has_out = active && (s < n_strips - 1);
            const bool out_global = has_out && (w == nwp - 1);
            LfqStrip<C> S;
            lfq_strip_init<C>(S, s == 0, shift);

            const int64_t n_steps = n_chunks + nwp - 1;
            for (int64_t t = 0; t < n_steps; t++) {
                const int64_t ch = t - w;
                if (active && ch >= 0 && ch < n_chunks) {
                    double ps, qf;
                    const uint64_t km = lfq_eval_chunk(cx, ch, T, P, &s_luts, &ps, &qf);
                    double bv = 0.0, ov = 0.0;
                    int be = 0, oe = 0;
                    const int64_t idx = ch * 64 + lane;
                    if (has_in) {
                        if (in_global) {
                            if (idx < cx.n_obs) {
                                bv = bnd[2 * idx];
                                be = (int)bnd[2 * idx + 1];
                            }
                        } else {
                            bv = s_bv[(t - 1) & 1][w - 1][lane];
                            be = s_be[(t - 1) & 1][w - 1][lane];
                        }
                    }
                    if (lfq_strip_chunk<C>(S, km, ps, qf, has_in, bv, be, has_out, ov, oe, is_tail, owns_tail, lt,
                                           cx.bonf_d, cx.sig_s)) {
                        s_pruned = 1;
                    }
                    if (has_out) {
                        if (out_global) {
                            if (idx < cx.n_obs) {
                                bnd[2 * idx] = ov;
                                bnd[2 * idx + 1] = (double)oe;
                            }
                        } else {
                            s_bv[t & 1][w][lane] = ov;
                            s_be[t & 1][w][lane] = oe;
                        }
                    }
                }
                __syncthreads();
                if (s_pruned) {
                    pruned = true;
                    break;
                }
            }
            if (!pruned && owns_tail) {
                if (lfq_strip_final_prune<C>(S, lt, cx.bonf_d, cx.sig_s)) {
                    s_pruned = 1;
                }
                rows_tail = S.rows;
            }
            __threadfence_block();
            __syncthreads();
            if (s_pruned) {
                pruned = true;
            }
            if (!pruned && active) {
                lfq_strip_store_logs<C>(S, gl, shift, K, probvec);
            }
            __threadfence_block();
            __syncthreads();
        }
        if ((!pruned || uf_mask) && w == ((n_strips - 1) % NW)) {
            /* the wave that owned the tail strip finishes the column */
            lfq_emit_pvals(cx, cnt, probvec, K, !pruned, uf_mask, uf_bound, force_fe, rows_tail, W, pvals,
                           pvals_capacity);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* launchers                                                                                   */
/* ------------------------------------------------------------------------------------------ */

int lfq_launch_dp_light(const LfqTracksDev &t, const LfqParams &p, const LfqLuts *d_luts,
                        const lfq_col_counts *d_counts, const LfqWork &w, lfq_col_pvals *d_pvals,
                        int64_t pvals_capacity, int n_waves, void *stream)
{
    if (t.ncols <= 0 || n_waves <= 0) {
        return LFQ_OK;
    }
    const unsigned blocks = (unsigned)((n_waves + 3) / 4);
    hipLaunchKernelGGL(lfq_dp_wave_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, p, d_luts,
                       d_counts, w, (const int32_t *)w.q_light, LFQ_CNT_LIGHT, d_pvals, pvals_capacity,
                       (int)(blocks * 4));
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}

int lfq_launch_dp_mid(const LfqTracksDev &t, const LfqParams &p, const LfqLuts *d_luts,
                      const lfq_col_counts *d_counts, const LfqWork &w, lfq_col_pvals *d_pvals,
                      int64_t pvals_capacity, int n_waves, void *stream)
{
    if (t.ncols <= 0 || n_waves <= 0) {
        return LFQ_OK;
    }
    const unsigned blocks = (unsigned)((n_waves + 3) / 4);
    hipLaunchKernelGGL(lfq_dp_wave_kernel<LFQ_HEAVY_C>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, p,
                       d_luts, d_counts, w, (const int32_t *)w.q_mid, LFQ_CNT_MID, d_pvals, pvals_capacity,
                       (int)(blocks * 4));
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}

int lfq_launch_dp_big(const LfqTracksDev &t, const LfqParams &p, const LfqLuts *d_luts,
                      const lfq_col_counts *d_counts, const LfqWork &w, lfq_col_pvals *d_pvals,
                      int64_t pvals_capacity, double *d_scratch, int64_t scratch_doubles_per_block,
                      int n_blocks, void *stream)
{
    if (t.ncols <= 0 || n_blocks <= 0) {
        return LFQ_OK;
    }
    hipLaunchKernelGGL(lfq_dp_big_kernel, dim3((unsigned)n_blocks), dim3(LFQ_HEAVY_WAVES * 64), 0,
                       (hipStream_t)stream, t, p, d_luts, d_counts, w, d_pvals, pvals_capacity, d_scratch,
                       scratch_doubles_per_block);
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}
