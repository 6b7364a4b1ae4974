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
 * left neighbour arrives by DPP (wave_shr:1 / row_shr:1), the row's (p, 1-p) from an LDS broadcast.  Every
 * lane carries a binary exponent e for its C cells (value = v*2^e), renormalised every 8 rows, so the
 * 1e-4932-range tails the reference reaches in log space are representable.
 *
 * Kernels, three chains on three streams after the scan:
 *   light class   lfq_dp_quad_kernel<8>   eight columns per wavefront (8-lane groups, K <= 7); prunes only
 *                 lfq_dp_retry_kernel     the <1 % the quad kernel could not finish: one wavefront per column
 *   mid class     lfq_dp_wave_kernel<4>   one wavefront per column, 1 or 4 cells per lane, first 2048 rows
 *                 lfq_dp_seg_kernel<0>    row segments of the survivors (one wavefront each, from the identity)
 *                 lfq_dp_combine_kernel<0> convolution fold of the segments + emission
 *   big class     lfq_dp_big_prep_kernel  bounds / underflow shortcut / split decision
 *                 lfq_dp_seg_kernel<1>    row segments at 8, 16 or 32 cells per lane
 *                 lfq_dp_combine_kernel<1>
 *                 lfq_dp_big_kernel       what cannot be split: 8-wave strip pipeline over 64-row chunks,
 *                                         boundary cell through a double-buffered LDS slab, passes through
 *                                         global scratch beyond 8 strips
 *   lfq_dp_wave_kernel<1> (one light column per wavefront) is kept as the A/B reference of the quad kernel.
 */
#include <atomic>

#include "lfq_device.h"

#define LFQ_LN2_HI 6.93147180369123816490e-01
#define LFQ_LN2_LO 1.90821492927058770002e-10
/* glibc's exp(x) raises FE_UNDERFLOW (result below DBL_MIN) for x < ln(2^-1022); pinned by
 * tests/test_oracle_kat.py::test_exp_underflow_threshold */
#define LFQ_EXP_UNDERFLOW_X (-708.3964185322641)
#define LFQ_DBL_EPS 2.220446049250313e-16

#ifndef LFQ_HEAVY_WAVES
#define LFQ_HEAVY_WAVES 8
#endif
/* wavefronts per SIMD the 512-thread kernels of the long-column chains are compiled for (their register bound: 3 -> 168,
 * 4 -> 128): what a workgroup of theirs needs per SIMD, 2 x that, has to fit beside the next batch's count kernel */
#ifndef LFQ_DP512_WAVES
#define LFQ_DP512_WAVES 2
#endif

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

__device__ __forceinline__ void lfq_col_setup(LfqColCtx &cx, const LfqEntry &en, const LfqParams &P)
{
    cx.col = en.col;
    cx.off0 = en.off0;
    cx.n_obs = en.n_obs;
    cx.ref_code = en.ref_code;
    cx.median_ref_bq = en.median_ref_bq;
    cx.K = en.kmax;
    /* running Bonferroni factor at this column (lofreq_call.c:794-800) */
    int64_t bonf = P.bonf_base;
    if (P.bonf_dynamic) {
        bonf = ((P.bonf_reset_first && P.bonf_base == 1) ? 0 : P.bonf_base) + (int64_t)P.bonf_step * en.prefix;
    }
    cx.bonf = bonf;
    cx.bonf_d = (double)bonf;
    cx.sig_s = P.sig * (1.0 + P.prune_slack);
}

/* wave-uniform load of a work-list record (32 bytes) */
__device__ __forceinline__ LfqEntry lfq_load_entry(const LfqEntry *list, int i)
{
    const uint4 *p = reinterpret_cast<const uint4 *>(list + i);
    const uint4 a = p[0], b = p[1];
    LfqEntry e;
    e.off0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)a.y) << 32)
             | (uint32_t)__builtin_amdgcn_readfirstlane((int)a.x);
    e.n_obs = __builtin_amdgcn_readfirstlane((int)a.z);
    e.col = __builtin_amdgcn_readfirstlane((int)a.w);
    e.prefix = __builtin_amdgcn_readfirstlane((int)b.x);
    e.kmax = __builtin_amdgcn_readfirstlane((int)b.y);
    const int m = __builtin_amdgcn_readfirstlane((int)b.z);
    e.median_ref_bq = (int16_t)(m & 0xffff);
    e.ref_code = (uint8_t)((m >> 16) & 0xff);
    e.pad_ = 0;
    e.pad2_ = 0;
    return e;
}

/* the quality tables (8 KB) into a workgroup's LDS: every thread's loads issued before the first store (a loop of
 * load - wait - store paid two to four memory round trips at the start of every DP kernel); 128 threads or more */
__device__ __forceinline__ void lfq_luts_to_lds(LfqLuts *dst, const LfqLuts *__restrict__ src)
{
    static_assert(sizeof(LfqLuts) % 16 == 0, "copied as 16-byte words");
    constexpr int N = (int)(sizeof(LfqLuts) / 16);
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    const int t = (int)threadIdx.x, nt = (int)blockDim.x;
    uint4 tmp[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = t + k * nt;
        tmp[k] = s[i < N ? i : N - 1];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = t + k * nt;
        if (i < N) {
            d[i] = tmp[k];
        }
    }
}

/* ---- DP work accounting (SURVEY 8d secondary figure) -----------------------------------------------
 * cells(column) = sum over the kept rows n = 1..N* of min(n, K), N* = the row this implementation stopped at.
 * Batch-wide uint64 counters next to the sparse-output counter; read back by lfq_batch_finish. */
__device__ __forceinline__ unsigned long long lfq_cells_of(long long n, long long K)
{
    return (unsigned long long)((n <= K) ? n * (n + 1) / 2 : K * (K + 1) / 2 + (n - K) * K);
}

/* one lane adds a column's work */
__device__ __forceinline__ void lfq_account(const LfqWork &W, long long rows, long long K)
{
    atomicAdd(reinterpret_cast<unsigned long long *>(W.gcounters + LFQ_GC_CELLS), lfq_cells_of(rows, K));
    atomicAdd(reinterpret_cast<unsigned long long *>(W.gcounters + LFQ_GC_ROWS), (unsigned long long)rows);
}

/* the packed bytes of one observation per lane (chunk `ch` of the column); 0xffffffff = past the end */
struct LfqRaw {
    uint32_t w;      /* nt | bq << 8 | baq << 16 | mq << 24 */
    uint32_t sq;
};

__device__ __forceinline__ LfqRaw lfq_load_chunk_at(uint64_t off0, int64_t n_obs, int64_t ch, const LfqTracksDev &T)
{
    /* Every lane loads -- a lane past the end of the column from the column's last observation -- and the bytes of the
     * lanes past the end are replaced afterwards.  With the loads inside `if (idx < n_obs)` the bytes were packed, hence
     * waited for, inside the branch: a caller that asks for several chunks before it uses the first (the big-class prep
     * kernel: four per wavefront) paid one memory round trip per chunk instead of one for all of them. */
    const int64_t idx = ch * 64 + lfq_lane();
    LfqRaw r;
    r.w = 0x00000004u;          /* N base: ignored */
    r.sq = 255u;
    const bool in = idx < n_obs;
    /* (no column in a work list is empty: n_obs >= 1 and the clamped index is one of its observations) */
    const uint64_t g = off0 + (uint64_t)(in ? idx : (n_obs > 0 ? n_obs - 1 : 0));
    /* (no branch around a load either: the layout of the nt track and the absent tracks select addresses and values,
     * an absent track is read from the bq track's bytes and replaced) */
    const bool pk = T.nt_packed != 0;
    const uint8_t *p_baq = T.baq ? T.baq : T.bq, *p_sq = T.sq ? T.sq : T.bq;
    const uint32_t nb = T.nt[pk ? ((g >> 3) * 4 + (g & 3u)) : g];                 /* lfq_nt_at */
    const uint32_t bq = T.bq[g], baq_b = p_baq[g], mq = T.mq[g], sq_b = p_sq[g];
    const uint32_t nt = pk ? ((g & 4u) ? (nb >> 4) : (nb & 15u)) : nb;
    const uint32_t baq = T.baq ? baq_b : 255u, sq = T.sq ? sq_b : 255u;
    r.w = in ? (nt | (bq << 8) | (baq << 16) | (mq << 24)) : 0x00000004u;
    r.sq = in ? sq : 255u;
    return r;
}

__device__ __forceinline__ LfqRaw lfq_load_chunk(const LfqColCtx &cx, int64_t ch, const LfqTracksDev &T)
{
    return lfq_load_chunk_at(cx.off0, cx.n_obs, ch, T);
}

/* evaluate the 64 observations of a chunk: keep mask + effective p and 1-p per lane, with the
 * reference's guards against log(0) (snpcaller.c:872-881) expressed on the probabilities */
__device__ __forceinline__ uint64_t lfq_eval_raw(const LfqColCtx &cx, const LfqRaw &r, const LfqParams &P,
                                                 const LfqLuts *L, double *ps, double *qf)
{
    const LfqObs o = lfq_eval_obs(r.w & 0xffu, (r.w >> 8) & 0xffu, (r.w >> 16) & 0xffu, r.w >> 24, r.sq,
                                  cx.ref_code, cx.median_ref_bq, P, L);
    *ps = (fabs(o.p) < LFQ_DBL_EPS) ? LFQ_DBL_EPS : o.p;
    *qf = (fabs(o.p - 1.0) < LFQ_DBL_EPS) ? 1.0 + (-o.p + LFQ_DBL_EPS) : 1.0 - o.p;
    return __ballot(o.keep);
}

__device__ __forceinline__ uint64_t lfq_eval_chunk(const LfqColCtx &cx, int64_t ch, const LfqTracksDev &T,
                                                   const LfqParams &P, const LfqLuts *L, double *ps, double *qf)
{
    const LfqRaw r = lfq_load_chunk(cx, ch, T);
    return lfq_eval_raw(cx, r, P, L, ps, qf);
}

template <int C>
struct LfqStrip {
    double v[C];
    int e, de, e_in, rows;
    bool all_zero;     /* wave-uniform: nothing has entered this strip yet */
    double sc1, sc2;   /* lfq_strip_chunk2: 2^(e of the lane one / two to the left - e), fixed between renormalisations */
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
    S.sc1 = S.sc2 = 1.0;
}

/* one row of the chunk as the DP consumes it: effective p and 1-p.  Observations that do not
 * contribute an error probability are stored as (0, 1): an exact identity row. */
struct LfqRow {
    double p, q;
};

/* Advance one strip over the 64 rows of one chunk, 8 rows at a time with no per-row branches.
 *   rows     64 (p,q) pairs in LDS, read with a wave-uniform address (broadcast)
 *   in_v/e   incoming boundary (value, exponent) per row in LDS when has_in (strip > 0)
 *   out_v/e  outgoing boundary per row in LDS when has_out (strip is not the last one)
 *   tflag    1.0 on the lane that owns the absorbing tail cell (at j = 0), else 0.0: its miss factor
 *            becomes q + p instead of q (cells beyond the tail hold don't-care values that are never read)
 * Returns true when the pruning test fires (evaluated every 8 rows on the strip that owns the tail). */
template <int C, bool IO>
__device__ __forceinline__ bool lfq_strip_chunk(LfqStrip<C> &S, const LfqRow *rows, uint64_t km, bool has_in,
                                                const double *in_v, const int *in_e, bool has_out,
                                                double *out_v, int *out_e, double tflag, bool owns_tail,
                                                int lt, double bonf_d, double sig_s)
{
    const int lane = lfq_lane();
    constexpr int R = (C == 1) ? 8 : 4;     /* rows per unrolled group (register pressure vs. branch cost) */
    /* boundary I/O without branches or EXEC changes inside the row loop (IO = strip pipeline only):
     *   in : every lane reads the row's incoming (value, exponent) with a wave-uniform address; only lane 0
     *        of a strip > 0 uses it (in_sel)
     *   out: every lane stores its last cell; lane 63 of a strip that has a successor stores to the slab
     *        slot of the row, all other lanes to a private dump slot (out_stride = 0) */
    const bool in_sel = IO && has_in && lane == 0;
    const int out_slot0 = (IO && has_out && lane == 63) ? 0 : 64 + lane;
    const int out_stride = (IO && has_out && lane == 63) ? 1 : 0;
#pragma unroll 1
    for (int g = 0; g < 64 / R; g++) {
        const int r0 = g * R;
        if (IO && has_in && S.all_zero && (r0 & 7) == 0) {
            /* nothing has reached this strip yet: skip whole groups whose incoming values are all zero */
            const double xb = in_v[r0 + (lane & 7)];
            const int eb_last = in_e[r0 + 7];
            if (!__any(xb != 0.0)) {
                if (has_out && lane < 8) {
                    out_v[r0 + lane] = 0.0;
                    out_e[r0 + lane] = in_e[r0 + lane];
                }
                S.e = eb_last;
                S.e_in = eb_last;
                g += 8 / R - 1;         /* the whole 8-row block */
                continue;
            }
            S.all_zero = false;
            S.e = in_e[r0];             /* adopt the producer's scale */
            S.de = 0;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const LfqRow pq = rows[r0 + r];
            double x = lfq_shr1_f64(S.v[C - 1]);
            int dei = S.de;
            if (IO) {
                const double xb = in_v[r0 + r];
                const int eb = in_e[r0 + r];
                S.e_in = eb;
                x = in_sel ? xb : x;
                dei = in_sel ? min(eb - S.e, 1000) : dei;
                out_v[out_slot0 + (r0 + r) * out_stride] = S.v[C - 1];
                out_e[out_slot0 + (r0 + r) * out_stride] = S.e;
            }
            const double pe = ldexp(pq.p, dei);             /* off the row-to-row dependency chain */
            const double q0 = fma(tflag, pq.p, pq.q);
#pragma unroll
            for (int j = C - 1; j >= 1; j--) {
                S.v[j] = fma(S.v[j - 1], pq.p, S.v[j] * pq.q);
            }
            S.v[0] = fma(x, pe, S.v[0] * q0);
        }
        if (((r0 + R) & 7) != 0) {
            continue;
        }

        /* every 8 rows: renormalise (lane maximum to [0.5,1), exponent into e) ... */
        double m = S.v[0];
#pragma unroll
        for (int j = 1; j < C; j++) {
            m = fmax(m, S.v[j]);
        }
        const bool nzl = m > 0.0;
        const uint64_t nz = __ballot(nzl);
        const int ex = nzl ? __builtin_amdgcn_frexp_exp(m) : 0;
#pragma unroll
        for (int j = 0; j < C; j++) {
            S.v[j] = ldexp(S.v[j], -ex);
        }
        S.e += ex;
        /* ... empty lanes (always a suffix: every cell left of the frontier is positive) adopt the scale
         * of the frontier lane, so the first value that reaches them is representable ... */
        if (nz != ~0ull) {
            if (nz) {
                const int e_front = lfq_rl_i32(S.e, 63 - __builtin_clzll(nz));
                S.e = nzl ? S.e : e_front;
            } else if (IO && has_in) {
                S.e = S.e_in;
            }
        }
        /* lane 0 has no left neighbour (the shift brings in value 0 and exponent 0): once its own exponent is below -1024 --
         * P(X < C) of a column with more than ~700 expected errors, or after twenty observations of error probability 1 --
         * 2^(0 - e) is no longer a double and 0 * inf would poison the strip.  Any finite scale does for a zero. */
        S.de = min(lfq_shr1_i32(S.e) - S.e, 1000);
        /* ... and test the pruning condition on the tail cell */
        if (owns_tail) {
            const uint64_t over = __ballot(ldexp(S.v[0], S.e) * bonf_d > sig_s);
            if ((over >> lt) & 1ull) {
                S.rows = r0 + R;
                return true;
            }
        }
    }
    S.rows = 64;
    return false;
}

/* per-lane (p,q) of a chunk into the wave's LDS row buffer; returns the keep mask */
__device__ __forceinline__ uint64_t lfq_stage_rows(const LfqColCtx &cx, const LfqRaw &raw, const LfqParams &P,
                                                   const LfqLuts *L, LfqRow *rows)
{
    double ps, qf;
    const uint64_t km = lfq_eval_raw(cx, raw, P, L, &ps, &qf);
    const bool keep = (km >> lfq_lane()) & 1ull;
    LfqRow r;
    r.p = keep ? ps : 0.0;
    r.q = keep ? qf : 1.0;
    rows[lfq_lane()] = r;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return km;
}

/* ---- two rows per step ------------------------------------------------------------------------------------
 * Two consecutive rows (p1, q1), (p2, q2) applied at once:
 *     cell[k] <- cell[k] q1 q2 + cell[k-1] (p1 q2 + q1 p2) + cell[k-2] p1 p2
 * -- three multiply-adds per cell and TWO rows instead of four, and everything a row costs besides its cells (the
 * neighbour exchange, the LDS read of the row, the tail cell's special case) is paid once per pair.  With the
 * renormalisation every 16 rows instead of 8 a C = 8 segment goes from ~63 to ~30 instructions per row.  The
 * absorbing tail cell keeps its mass (coefficient 1 instead of q1 q2) and collects the inflow of both rows:
 *     tail <- tail + cell[K-1] (p1 + q1 p2) + cell[K-2] p1 p2,      p1 + q1 p2 = (p1 q2 + q1 p2) + p1 p2.
 * Rows that contribute no probability are (0, 1): the pair degenerates to the other row.  Same recurrence, same
 * products of the same probabilities; the rounding differs from row-at-a-time at the 1e-16 level. */
struct LfqRow2 {
    double A, B, C2, omA;       /* q1 q2, p1 q2 + q1 p2, p1 p2, 1 - q1 q2 */
};

__device__ __forceinline__ uint64_t lfq_stage_rows2(const LfqColCtx &cx, const LfqRaw &raw, const LfqParams &P,
                                                    const LfqLuts *L, LfqRow2 *rows2)
{
    double ps, qf;
    const uint64_t km = lfq_eval_raw(cx, raw, P, L, &ps, &qf);
    const int lane = lfq_lane();
    const bool keep = (km >> lane) & 1ull;
    const double p = keep ? ps : 0.0, q = keep ? qf : 1.0;
    const double p2 = __shfl_xor(p, 1, 64), q2 = __shfl_xor(q, 1, 64);       /* the other row of the pair */
    if (!(lane & 1)) {
        LfqRow2 r;
        r.A = q * q2;
        r.B = fma(p, q2, q * p2);
        r.C2 = p * p2;
        r.omA = 1.0 - r.A;
        rows2[lane >> 1] = r;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return km;
}

/* one strip over the 64 rows of a chunk, two rows per step, renormalisation + pruning test every 16 rows; single
 * strip (no boundary exchange with other wavefronts).  Returns true when the pruning test fires. */
template <int C>
__device__ __forceinline__ bool lfq_strip_chunk2(LfqStrip<C> &S, const LfqRow2 *rows2, double tflag, int lt,
                                                 double bonf_d, double sig_s)
{
#pragma unroll 1
    for (int g = 0; g < 4; g++) {
        const double sc1 = S.sc1, sc2 = S.sc2;
#pragma unroll
        for (int st = 0; st < 8; st++) {
            const LfqRow2 r = rows2[g * 8 + st];
            const double x1 = lfq_shr1_f64(S.v[C - 1]);                              /* cell -1 */
            double x2;                                                               /* cell -2 */
            if constexpr (C >= 2) {
                x2 = lfq_shr1_f64(S.v[C - 2]);
            } else {
                x2 = lfq_shr1_f64(x1);
            }
            const double A0 = fma(tflag, r.omA, r.A);           /* the tail cell keeps its mass ... */
            const double B0 = fma(tflag, r.C2, r.B);            /* ... and takes the inflow of both rows */
            const double Bs = B0 * sc1;
            const double Cs = r.C2 * ((C >= 2) ? sc1 : sc2);
#pragma unroll
            for (int j = C - 1; j >= 2; j--) {
                S.v[j] = fma(S.v[j - 2], r.C2, fma(S.v[j - 1], r.B, S.v[j] * r.A));
            }
            if constexpr (C >= 2) {
                S.v[1] = fma(x1, r.C2 * sc1, fma(S.v[0], r.B, S.v[1] * r.A));
            }
            S.v[0] = fma(x2, Cs, fma(x1, Bs, S.v[0] * A0));
        }
        /* every 16 rows: renormalise (lane maximum to [0.5,1), exponent into e) ... */
        double m = S.v[0];
#pragma unroll
        for (int j = 1; j < C; j++) {
            m = fmax(m, S.v[j]);
        }
        const bool nzl = m > 0.0;
        const uint64_t nz = __ballot(nzl);
        const int ex = nzl ? __builtin_amdgcn_frexp_exp(m) : 0;
#pragma unroll
        for (int j = 0; j < C; j++) {
            S.v[j] = ldexp(S.v[j], -ex);
        }
        S.e += ex;
        /* ... empty lanes (a suffix) adopt the scale of the frontier lane ... */
        if (nz != ~0ull && nz) {
            const int e_front = lfq_rl_i32(S.e, 63 - __builtin_clzll(nz));
            S.e = nzl ? S.e : e_front;
        }
        const int e_l1 = lfq_shr1_i32(S.e), e_l2 = lfq_shr1_i32(e_l1);
        S.de = min(e_l1 - S.e, 1000);                 /* (lane 0: see lfq_strip_chunk) */
        /* lanes 0 (and 1) read zeros from beyond the strip: any finite scale will do */
        S.sc1 = ldexp(1.0, max(-1000, min(1000, e_l1 - S.e)));
        S.sc2 = ldexp(1.0, max(-1000, min(1000, e_l2 - S.e)));
        /* ... and test the pruning condition on the tail cell */
        const uint64_t over = __ballot(ldexp(S.v[0], S.e) * bonf_d > sig_s);
        if ((over >> lt) & 1ull) {
            S.rows = g * 16 + 16;
            return true;
        }
    }
    S.rows = 64;
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

/* exponent stored with a zero mantissa: sums of two of them stay far below any real exponent, so the fold's
 * running-maximum alignment needs no special case for empty cells */
#define LFQ_EXT_ZERO_E (-(1 << 24))

/* a strip's cells in (mantissa, exponent) form -> the segment pool (cells 0..K-1 and the tail cell K) */
template <int C>
__device__ __forceinline__ void lfq_strip_store_cells(const LfqStrip<C> &S, int gl, int shift, int K, LfqSegCell *out)
{
#pragma unroll
    for (int j = 0; j < C; j++) {
        const int k = gl * C + j - shift;
        if (k >= 0 && k <= K && (k < K || j == 0)) {
            const double v = S.v[j];
            LfqSegCell c;
            c.v = (v > 0.0) ? __builtin_amdgcn_frexp_mant(v) : 0.0;
            c.e = (v > 0.0) ? S.e + __builtin_amdgcn_frexp_exp(v) : LFQ_EXT_ZERO_E;   /* 0 = 0 * 2^(very small) */
            c.pad_ = 0;
            out[k] = c;
        }
    }
}

/* How many NEW row segments to cut `rem_chunks` remaining chunks of a K-cell column into (0 = do not split).
 * Bounds: LFQ_SEG_MAX segments in total, no segment shorter than LFQ_SEG_MIN_CHUNKS chunks (LFQ_SEG_MIN_CHUNKS_SHORT for a
 * column of less than LFQ_SEG_SHORT_BELOW chunks: a 1000x column with hundreds of alt bases is a chain of a thousand dependent
 * rows of a third of a microsecond on a wavefront of its own, the longest thing in a 1000x batch), and the
 * (segments - 1) convolutions of K^2/2 terms must stay below a quarter of the rows * K recurrence work -- except that up to
 * three segments are always allowed: two convolutions are microseconds, and a segment that starts from the identity has
 * min(row, K) cells like the column itself, so short segments of a wide column are less work than the column in one piece. */
__device__ __forceinline__ int lfq_split_plan(int K, int64_t rem_chunks, int phase1, int seg_max)
{
    if (K > LFQ_SPLIT_MAX_K) {
        return 0;
    }
    int64_t n_new = seg_max - phase1;
    n_new = min(n_new, rem_chunks / (rem_chunks < LFQ_SEG_SHORT_BELOW ? LFQ_SEG_MIN_CHUNKS_SHORT : LFQ_SEG_MIN_CHUNKS));
    n_new = min(n_new, max(rem_chunks * 64 / (2 * (int64_t)max(K, 1)) + 1 - phase1, (int64_t)3 - phase1));
    return n_new >= 2 ? (int)n_new : 0;
}

__device__ __forceinline__ int lfq_seg_class(int K)
{
    return K <= 63 ? 0 : (K <= 252 ? 1 : (K <= 504 ? 2 : (K <= 1008 ? 3 : 4)));
}

/* reserve pool cells and a record slot for a column that is about to be split; nullptr = run it unsplit.
 * Called by one lane. */
__device__ __forceinline__ LfqLong *lfq_long_reserve(const LfqWork &W, int K, int n_seg, int64_t *cell0)
{
    const int cells = 2 * n_seg * (K + 1);          /* the segments + the intermediate results of the fold tree */
    const int c0 = atomicAdd(&W.counters[LFQ_CNT_POOL], cells);
    if (c0 + cells > W.pool_cells) {
        return nullptr;
    }
    const int cls = lfq_seg_class(K);
    const int per_class = W.long_cap / LFQ_SEG_CLASSES;
    const int slot = atomicAdd(&W.counters[LFQ_CNT_LONG0 + cls], 1);
    if (slot >= per_class) {
        return nullptr;
    }
    *cell0 = c0;
    return W.longs + cls * per_class + slot;
}

__device__ __forceinline__ void lfq_ctx_from_long(LfqColCtx &cx, const LfqLong &r, const LfqParams &P)
{
    cx.col = r.col;
    cx.off0 = r.off0;
    cx.n_obs = r.n_obs;
    cx.ref_code = r.ref_code;
    cx.median_ref_bq = r.median_ref_bq;
    cx.K = r.K;
    cx.bonf = r.bonf;
    cx.bonf_d = (double)r.bonf;
    cx.sig_s = P.sig * (1.0 + P.prune_slack);
    if (r.force_fe) {
        cx.sig_s = fmax(cx.sig_s, 4.5e-16 * cx.bonf_d);     /* see lfq_dp_big_kernel */
    }
}

/* chunk range [c0, c1) of new segment `r` (r >= phase1) of a row-split column */
__device__ __forceinline__ void lfq_seg_range(const LfqLong &rec, int r, int64_t *c0, int64_t *c1)
{
    const int64_t n_chunks = ((int64_t)rec.n_obs + 63) / 64;
    const int64_t rem = n_chunks - rec.ch_begin;
    const int n_new = rec.n_seg - rec.phase1;
    *c0 = rec.ch_begin + rem * (r - rec.phase1) / n_new;
    *c1 = rec.ch_begin + rem * (r - rec.phase1 + 1) / n_new;
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
__device__ __forceinline__ void lfq_emit_pvals(const LfqColCtx &cx, const lfq_col_counts &cnt, const double *probvec, int kp,
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
        const int slot = atomicAdd(&W.gcounters[LFQ_GC_PVALS], 1);
        if ((int64_t)slot < pvals_capacity) {
            lfq_col_pvals r;
            r.col = cx.col;
            r.bonf = cx.bonf;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                r.logp[a] = logp[a];
                r.status[a] = (uint8_t)status[a];
            }
            r.ref_base = (uint8_t)"ACGT"[cx.ref_code & 3];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                r.pad_[i] = 0;
            }
            r.counts = cnt;
            r.dp_rows = rows;
            r.pad2_ = 0;
            r.reserved_ = 0;
            pvals[slot] = r;
        } else {
            W.gcounters[LFQ_GC_OVERFLOW] = 1;
        }
    }
}

/* ---- tail sums without transcendentals (wave-per-column kernels) ---------------------------------
 * Values are (mantissa, binary exponent) pairs; sums align exponents with ldexp.  probvec_tailsum
 * (snpcaller.c:730-741) becomes a prefix sum over the cells c..K held in registers; the exp()
 * underflow inside the reference's log_sum chain (exp(min-max) < DBL_MIN) becomes
 * "ratio of the smaller to the larger of (running sum, next cell) below 2^-1022". */
struct LfqExt {
    double v;
    int e;
};

__device__ __forceinline__ LfqExt lfq_ext_add(LfqExt a, LfqExt b)
{
    if (a.v == 0.0) {
        return b;
    }
    if (b.v == 0.0) {
        return a;
    }
    LfqExt r;
    r.e = max(a.e, b.e);
    r.v = ldexp(a.v, a.e - r.e) + ldexp(b.v, b.e - r.e);
    return r;
}

/* true if min(a,b)/max(a,b) < 2^-1022 */
__device__ __forceinline__ bool lfq_ext_ratio_underflows(LfqExt a, LfqExt b)
{
    if (a.v == 0.0 || b.v == 0.0) {
        return a.v != b.v;
    }
    /* compare a and b */
    const int fa = __builtin_amdgcn_frexp_exp(a.v) + a.e, fb = __builtin_amdgcn_frexp_exp(b.v) + b.e;
    const double ma = __builtin_amdgcn_frexp_mant(a.v), mb = __builtin_amdgcn_frexp_mant(b.v);
    const bool a_small = (fa < fb) || (fa == fb && ma < mb);
    const double ms = a_small ? ma : mb, ml = a_small ? mb : ma;
    const int es = a_small ? fa : fb, el = a_small ? fb : fa;
    /* (ms/ml) * 2^(es-el) < 2^-1022  <=>  ms/ml < 2^(-1022-es+el) */
    const int d = es - el;                 /* <= 0 */
    if (d > -1021) {
        return false;
    }
    if (d < -1024) {
        return true;
    }
    return (ms / ml) < ldexp(1.0, -1022 - d);
}

template <int C>
__device__ void lfq_emit_linear(const LfqColCtx &cx, const lfq_col_counts &cnt, const LfqStrip<C> &S, int shift,
                                int lt, const LfqWork &W, lfq_col_pvals *__restrict__ pvals,
                                int64_t pvals_capacity)
{
    const int lane = lfq_lane();
    const int K = cx.K;
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
        LfqExt tot;
        bool fe = false;
        if (c == K) {
            tot.v = lfq_rl_f64(S.v[0], lt);
            tot.e = lfq_rl_i32(S.e, lt);
        } else {
            /* lane totals over the cells k in [c, K] (the tail cell k = K sits at j = 0 of lane lt) */
            LfqExt mine;
            mine.v = 0.0;
            mine.e = S.e;
#pragma unroll
            for (int j = 0; j < C; j++) {
                const int k = lane * C + j - shift;
                if (k >= c && (k < K || (k == K && j == 0))) {
                    mine.v += S.v[j];
                }
            }
            /* exclusive prefix across lanes */
            LfqExt incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                LfqExt y;
                y.v = __shfl_up(incl.v, d, 64);
                y.e = __shfl_up(incl.e, d, 64);
                if (lane >= d) {
                    incl = lfq_ext_add(incl, y);
                }
            }
            LfqExt run;
            run.v = __shfl_up(incl.v, 1, 64);
            run.e = __shfl_up(incl.e, 1, 64);
            if (lane == 0) {
                run.v = 0.0;
                run.e = 0;
            }
            /* walk this lane's cells in k order: the reference's fold state before cell k is `run` */
#pragma unroll
            for (int j = 0; j < C; j++) {
                const int k = lane * C + j - shift;
                if (k >= c && (k < K || (k == K && j == 0))) {
                    LfqExt cell;
                    cell.v = S.v[j];
                    cell.e = S.e;
                    if (k > c && lfq_ext_ratio_underflows(run, cell)) {
                        fe = true;
                    }
                    run = lfq_ext_add(run, cell);
                }
            }
            fe = __any(fe);
            tot.v = lfq_rl_f64(incl.v, 63);
            tot.e = lfq_rl_i32(incl.e, 63);
        }
        const double ed = (double)tot.e;
        logp[a] = (tot.v > 0.0) ? (ed * LFQ_LN2_HI + (ed * LFQ_LN2_LO + log(tot.v))) : -INFINITY;
        status[a] = fe ? LFQ_PV_LOG_FECLAMP : LFQ_PV_LOG;
    }
    if (lane == 0) {
        const int slot = atomicAdd(&W.gcounters[LFQ_GC_PVALS], 1);
        if ((int64_t)slot < pvals_capacity) {
            lfq_col_pvals r;
            r.col = cx.col;
            r.bonf = cx.bonf;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                r.logp[a] = logp[a];
                r.status[a] = (uint8_t)status[a];
            }
            r.ref_base = (uint8_t)"ACGT"[cx.ref_code & 3];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                r.pad_[i] = 0;
            }
            r.counts = cnt;
            r.dp_rows = S.rows;
            r.pad2_ = 0;
            r.reserved_ = 0;
            pvals[slot] = r;
        } else {
            W.gcounters[LFQ_GC_OVERFLOW] = 1;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* wave-per-column kernel: light (C = 1) and mid (K < 250, C = 1 or 4) columns                  */
/* ------------------------------------------------------------------------------------------ */

/* Claim `n` consecutive work-list slots for this wavefront: one returning device-scope atomic from
 * lane 0, result broadcast.  Written as inline asm on purpose: with ROCm 7.2's hipcc the plain
 * `if (lane == 0) atomicAdd(head, 1)` form in this loop (next to the sparse-output atomicAdd) produced a
 * kernel that never terminated; adding 2, or issuing the atomic from all lanes, did not. */
__device__ __forceinline__ int lfq_claim(int32_t *head, int n)
{
    int old = 0;
    if (lfq_lane() == 0) {
        asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)"
                     : "=v"(old)
                     : "v"(head), "v"(n)
                     : "memory");
    }
    return __builtin_amdgcn_readfirstlane(old);
}

/* what the column pipeline wants loaded while this column computes */
struct LfqPrefetch {
    bool want_raw, want_entry;
    uint64_t off0;
    int64_t n_obs;
    const LfqEntry *entry_ptr;
    LfqRaw *raw;
    LfqEntry *entry;
};

/* one column on one wavefront: C cells per lane, single strip ((K + C) / C <= 64) */
template <int C, bool PREFETCH, bool SPLIT>
__device__ __forceinline__ void lfq_wave_column(const LfqColCtx &cx, LfqRaw raw, const LfqTracksDev &T,
                                                const LfqParams &P, const LfqLuts *L, LfqRow *rows,
                                                const lfq_col_counts *__restrict__ counts, const LfqWork &W,
                                                lfq_col_pvals *__restrict__ pvals, int64_t pvals_capacity,
                                                const LfqPrefetch &pf)
{
    const int lane = lfq_lane();
    const int K = cx.K;
    const int shift = (C - K % C) % C;
    const int lt = (K + shift) / C;             /* lane that owns the tail cell (<= 63 by class) */
    const double tflag = (lane == lt) ? 1.0 : 0.0;
    LfqStrip<C> S;
    lfq_strip_init<C>(S, true, shift);
    const int64_t n_chunks = (cx.n_obs + 63) / 64;
    bool pruned = false;
    int n_rows = 0;
#ifdef LFQ_PROFILE
    long long t_stage = 0, t_rows = 0, t_load = 0;
#endif
    for (int64_t ch = 0; ch < n_chunks; ch++) {
#ifdef LFQ_PROFILE
        const long long c0 = clock64();
#endif
        const uint64_t km = lfq_stage_rows2(cx, raw, P, L, reinterpret_cast<LfqRow2 *>(rows));
#ifdef LFQ_PROFILE
        const long long c1 = clock64();
        t_stage += c1 - c0;
#endif
        if (ch == 0) {
            /* next column's first chunk and the record of the column after it */
            if (pf.want_raw) {
                *pf.raw = lfq_load_chunk_at(pf.off0, pf.n_obs, 0, T);
            }
            if (pf.want_entry) {
                *pf.entry = lfq_load_entry(pf.entry_ptr, 0);
            }
        }
        /* the next chunk's loads are issued AFTER this chunk has been consumed into LDS and BEFORE its
         * rows run: exactly one batch of loads is in flight, and it lands while the recurrence computes */
        if (PREFETCH && ch + 1 < n_chunks) {
            raw = lfq_load_chunk(cx, ch + 1, T);
        }
#ifdef LFQ_PROFILE
        const long long c2 = clock64();
        t_load += c2 - c1;
#endif
        const bool hit = lfq_strip_chunk2<C>(S, reinterpret_cast<const LfqRow2 *>(rows), tflag, lt, cx.bonf_d, cx.sig_s);
#ifdef LFQ_PROFILE
        t_rows += clock64() - c2;
#endif
        n_rows += __popcll(S.rows >= 64 ? km : (km & ((1ull << S.rows) - 1ull)));
        if (hit) {
            pruned = true;
            break;
        }
        if (SPLIT && ch + 1 == P.phase1_chunks) {
            /* still alive after the first stretch of rows: cut the rest into concurrent row segments;
             * the state reached here becomes segment 0 (lfq_dp_segw_kernel, lfq_dp_combine_kernel) */
            /* fewer, longer segments when many columns are long anyway: the fold costs (segments - 1) convolutions */
            const int n_new = lfq_split_plan(K, n_chunks - (ch + 1), 1,
                                             min(P.seg_max_mid, max(2, P.seg_budget_mid / max(W.counters[LFQ_CNT_MID], 1))));
            if (n_new > 0) {
                const int n_seg = n_new + 1;
                const int cells = 2 * n_seg * (K + 1);      /* segments + intermediates of the fold tree */
                const int cls = lfq_seg_class(K);
                const int per_class = W.long_cap / LFQ_SEG_CLASSES;
                const int c0 = lfq_claim(&W.counters[LFQ_CNT_POOL], cells);
                if (c0 + cells <= W.pool_cells) {
                    const int slot = lfq_claim(&W.counters[LFQ_CNT_LONG0 + cls], 1);
                    if (slot < per_class) {
                        lfq_strip_store_cells<C>(S, lane, shift, K, W.pool + c0);
                        if (lane == 0) {
                            LfqLong r;
                            r.off0 = cx.off0;
                            r.bonf = cx.bonf;
                            r.cell0 = c0;
                            r.n_obs = (int32_t)cx.n_obs;
                            r.col = cx.col;
                            r.K = K;
                            r.n_seg = n_seg;
                            r.ch_begin = (int32_t)(ch + 1);
                            r.phase1 = 1;
                            r.uf_mask = 0;
                            r.force_fe = 0;
                            r.pruned = 0;
                            r.rows = n_rows;
                            r.median_ref_bq = (int16_t)cx.median_ref_bq;
                            r.ref_code = (uint8_t)cx.ref_code;
                            r.pad0_ = 0;
                            r.pad1_ = 0;
                            r.uf_bound[0] = r.uf_bound[1] = r.uf_bound[2] = 0.0;
                            r.pad_[0] = r.pad_[1] = r.pad_[2] = r.pad_[3] = 0;
                            W.longs[cls * per_class + slot] = r;
                        }
                        return;
                    }
                }
            }
        }
        if (!PREFETCH && ch + 1 < n_chunks) {
            raw = lfq_load_chunk(cx, ch + 1, T);
        }
    }
#ifdef LFQ_PROFILE
    if (lane == 0) {
        atomicAdd(&W.counters[8], (int)(t_stage >> 8));
        atomicAdd(&W.counters[9], (int)(t_load >> 8));
        atomicAdd(&W.counters[10], (int)(t_rows >> 8));
        atomicAdd(&W.counters[11], n_rows);
    }
#endif
#ifdef LFQ_TRACE
    if (lane == 0) printf("col %d K %d C %d: rows done pruned=%d n_rows=%d\n", cx.col, K, C, (int)pruned, n_rows);
#endif
    if (lane == 0) {
        lfq_account(W, n_rows, K);
    }
    if (!pruned && !lfq_strip_final_prune<C>(S, lt, cx.bonf_d, cx.sig_s)) {
#ifdef LFQ_TRACE
        if (lane == 0) printf("col %d: emitting\n", cx.col);
#endif
        const lfq_col_counts cnt = counts[cx.col];
        S.rows = n_rows;
        lfq_emit_linear<C>(cx, cnt, S, shift, lt, W, pvals, pvals_capacity);
    }
}

template <int MAXC>
__global__ __launch_bounds__(256) void lfq_dp_wave_kernel(LfqTracksDev T, LfqParams P,
                                                          const LfqLuts *__restrict__ g_luts,
                                                          const lfq_col_counts *__restrict__ counts, LfqWork W,
                                                          int base_idx, int count_idx,
                                                          lfq_col_pvals *__restrict__ pvals,
                                                          int64_t pvals_capacity, int batch, int only_if_gl64)
{
    __shared__ LfqLuts s_luts;
    __shared__ LfqRow s_rows[4][64];
    (void)only_if_gl64;
    if (MAXC > 1) {
        /* few, long, latency-bound columns sharing SIMDs with the throughput-bound light kernel:
         * win the issue arbitration (MI355X_MICROARCH "two waves per SIMD", item 2) */
        __builtin_amdgcn_s_setprio(3);
    }
    lfq_luts_to_lds(&s_luts, g_luts);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n_work = W.counters[count_idx];
    /* list layout [light | mid | big] */
    const LfqEntry *list = W.entries + ((base_idx >= 0) ? W.counters[base_idx] : 0);
    LfqRow *rows = s_rows[wave];

    /* Dynamic distribution: a wavefront claims BATCH consecutive work-list records at a time (so the
     * kernel's duration does not depend on how many of its wavefronts are resident), and inside a batch
     * runs a software pipeline over columns: while column i computes, the first chunk of column i+1 and
     * the record of column i+2 are in flight. */
    const int BATCH = batch;            /* run-time on purpose, see lfq_claim */
    int32_t *head = &W.counters[(MAXC == 1) ? LFQ_CNT_HEAD_LIGHT : LFQ_CNT_HEAD_MID];
    for (;;) {
        int b0 = lfq_claim(head, BATCH);
        if (b0 >= n_work) {
            break;
        }
        const int b1 = min(b0 + BATCH, n_work);
        int w = b0;
        LfqEntry en = lfq_load_entry(list, w);
        LfqRaw raw = lfq_load_chunk_at(en.off0, en.n_obs, 0, T);
        LfqEntry en_next = en;
        bool have_next = (w + 1) < b1;
        if (have_next) {
            en_next = lfq_load_entry(list, w + 1);
        }
        for (;;) {
            LfqColCtx cx;
            lfq_col_setup(cx, en, P);
            LfqRaw raw_next = raw;
            LfqEntry en_next2 = en_next;
            const bool have_next2 = have_next && (w + 2) < b1;
            LfqPrefetch pf;
            pf.want_raw = have_next;
            pf.off0 = en_next.off0;
            pf.n_obs = en_next.n_obs;
            pf.want_entry = have_next2;
            pf.entry_ptr = list + (have_next2 ? (w + 2) : w);
            pf.raw = &raw_next;
            pf.entry = &en_next2;
            if (MAXC == 1 || cx.K < 64) {
                lfq_wave_column<1, (MAXC > 1), (MAXC > 1)>(cx, raw, T, P, &s_luts, rows, counts, W, pvals, pvals_capacity, pf);
            } else {
                /* (more cells-per-lane variants were measured -- C = 2 and 4 are ~30 % faster per row for
                 * K < 250 -- but four inlined variants push this kernel into SGPR spilling) */
                lfq_wave_column<4, true, true>(cx, raw, T, P, &s_luts, rows, counts, W, pvals, pvals_capacity, pf);
            }
            if (!have_next) {
                break;
            }
            w += 1;
            en = en_next;
            raw = raw_next;
            en_next = en_next2;
            have_next = have_next2;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* light columns, four at a time                                                               */
/* ------------------------------------------------------------------------------------------ */


/* the columns the screen kernel flagged: static partition of the light list over the wavefronts */
__global__ __launch_bounds__(256) void lfq_dp_retry_kernel(LfqTracksDev T, LfqParams P,
                                                           const LfqLuts *__restrict__ g_luts,
                                                           const lfq_col_counts *__restrict__ counts, LfqWork W,
                                                           const uint8_t *__restrict__ retry,
                                                           lfq_col_pvals *__restrict__ pvals, int64_t pvals_capacity)
{
    __shared__ LfqLuts s_luts;
    __shared__ LfqRow s_rows[4][64];
    lfq_luts_to_lds(&s_luts, g_luts);
    __syncthreads();
    const int lane = lfq_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n_work = W.counters[LFQ_CNT_LIGHT];
    const int n_waves_total = (int)gridDim.x * 4;
    LfqRow *rows = s_rows[wave];
    /* 16 list entries per wavefront and pass: flagged columns cluster (deep or noisy stretches), so a fine
     * interleave balances better than 64-entry blocks */
    for (int base = ((int)blockIdx.x * 4 + wave) * 16; base < n_work; base += n_waves_total * 16) {
        const int i = base + lane;
        uint64_t m = __ballot(lane < 16 && i < n_work && retry[i] != 0);
        while (m != 0ull) {
            const int j = __builtin_ctzll(m);
            m &= m - 1ull;
            const LfqEntry en = lfq_load_entry(W.entries, base + j);
            LfqColCtx cx;
            lfq_col_setup(cx, en, P);
            const LfqRaw raw = lfq_load_chunk_at(en.off0, en.n_obs, 0, T);
            LfqPrefetch pf;
            pf.want_raw = false;
            pf.want_entry = false;
            pf.off0 = 0;
            pf.n_obs = 0;
            pf.entry_ptr = nullptr;
            pf.raw = nullptr;
            pf.entry = nullptr;
            lfq_wave_column<1, true, false>(cx, raw, T, P, &s_luts, rows, counts, W, pvals, pvals_capacity, pf);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* light columns, one per LANE: the screen kernel                                              */
/* ------------------------------------------------------------------------------------------ */

/* A light column is decided by its first few dozen rows: a handful of mismatches, and a tail probability that
 * crosses the pruning threshold sig / bonf as soon as a couple of low-quality reads have gone by (C3: 21 rows on
 * average, 99.9 % within 250).  That is far too little work per column for cells-across-lanes -- the lane-group
 * kernels above spend their time waiting for one dependent byte gather per 8 rows -- so here ONE LANE owns one
 * column: its K + 1 <= KREG cells live in registers (cell k at register KREG - 1 - K + k, i.e. the absorbing tail
 * always in the last register and the registers below cell 0 hold zeros that stay zero, so every lane runs the
 * same KREG - 1 FMAs whatever its K), and it reads its column 16 observations at a time with one 16-byte load per
 * track (8 bytes of the packed nt track) from the 16-aligned window around the column start.  No neighbour
 * exchange, no LDS rows, no renormalisation (plain doubles: a few hundred rows of factors >= 1e-10 at K <= 31 stay
 * far inside the double range, and a cell that did underflow would only make the tail larger, which errs towards
 * pruning LATER, never earlier... see the bound below), 64 columns per wavefront in lock step.
 *
 * The kernel only PRUNES (like the lane-group kernels): a lane whose tail * bonf exceeds sig * (1 + slack) drops its
 * column -- the reference's own early exit (snpcaller.c:950), monotone in the rows, so order does not matter --
 * and takes the next one of its wavefront's slice of the light list.  Columns that are still alive after
 * `max_rounds` windows or at their end, and columns with K >= KREG, are flagged in `retry` for
 * lfq_dp_retry_kernel (whole wavefront, emission).  Underflow cannot cause a false prune: cells only lose mass by
 * rounding to zero, the tail is a sum of products of cells and probabilities, so the computed tail is never above
 * the exact one by more than rounding (1e-16 relative per operation), which the slack of 1e-6 covers. */
template <int KREG>
__device__ __forceinline__ void lfq_screen_row(double (&v)[KREG], double p, double q)
{
    v[KREG - 1] = fma(v[KREG - 2], p, v[KREG - 1]);         /* absorbing tail: P(X >= K) */
#pragma unroll
    for (int j = KREG - 2; j >= 1; j--) {
        v[j] = fma(v[j - 1], p, v[j] * q);
    }
    v[0] = v[0] * q;
}

#define LFQ_SCREEN_BYTE(w4, j) \
    (((((j) >> 2) == 0 ? (w4).x : ((j) >> 2) == 1 ? (w4).y : ((j) >> 2) == 2 ? (w4).z : (w4).w) >> (8 * ((j) & 3))) & 0xffu)

#define LFQ_SCREEN_CLAIM 128       /* work-list entries per dequeue */

/* LB: with the default filters (no merged-quality filter, an alt base keeps its own quality) the screen works on a
 * LOWER BOUND of every error probability, jp >= pm + (1 - pm) pb -- the alignment- and source-quality terms of the merge
 * (snpcaller.c:334) only ever add -- read from two of the five tracks and evaluated with one FMA.  P(X >= K) is
 * monotone in every p, so a tail computed from lower bounds that exceeds the pruning threshold proves that the exact
 * one does: the column is pruned a row or two later than it could be (the dropped terms are ~1 % of the error mass of
 * a typical column), never wrongly.  Everything the screen does not prune is redone exactly by the retry kernel. */
template <int KREG, bool LB>
__global__ __launch_bounds__(256) void lfq_dp_screen_kernel(LfqTracksDev T, LfqParams P,
                                                            const LfqLuts *__restrict__ g_luts, LfqWork W,
                                                            uint8_t *__restrict__ retry, int max_rounds)
{
    constexpr int MAXK = KREG - 1;
    __shared__ LfqLuts s_luts;
    lfq_luts_to_lds(&s_luts, g_luts);
    __syncthreads();
    const int lane = lfq_lane();
    const int n_work = W.counters[LFQ_CNT_LIGHT];
    const LfqEntry *list = W.entries;               /* the light class leads the work list */
    const double sig_s = P.sig * (1.0 + P.prune_slack);
    const LfqEvalMasks EM = lfq_eval_masks(P);
    /* Work distribution: the light list is cut into eight slices, one per XCD, each with its own dequeue head; a
     * wavefront claims LFQ_SCREEN_CLAIM consecutive entries at a time from its XCD's slice (blockIdx % 8: placement is a speed
     * matter only) and moves on to the other slices when that one is empty.  One head for everybody would
     * saturate (~88 dequeues per microsecond on one word); a static slice per wavefront leaves most lanes idle
     * while the last columns of each slice run out (measured: 35 % lane utilisation). */
    const int xcd0 = (int)(blockIdx.x & 7u);
    int xcd_try = 0;                                /* slices found empty so far */
    int q_next = 0, q_end = 0;                      /* this wavefront's claimed, not yet assigned entries */

    bool active = false;
    uint64_t off0 = 0;
    int n_obs = 0, rel = 0, K = 0, ref_code = 0, med = 0, rounds = 0, list_idx = 0, n_kept = 0;
    double bonf_d = 1.0;
    double v[KREG];
#pragma unroll
    for (int j = 0; j < KREG; j++) {
        v[j] = 0.0;
    }
    unsigned long long acc_cells = 0, acc_rows = 0;
    int n_retry = 0;

    for (;;) {
        /* ---- lanes without a column take the next claimed entries ---- */
        uint64_t need = __ballot(!active);
        while (need != 0ull && (q_next < q_end || xcd_try < 8)) {
            if (q_next >= q_end) {
                const int x = (xcd0 + xcd_try) & 7;
                const int lo = (int)((int64_t)n_work * x / 8), hi = (int)((int64_t)n_work * (x + 1) / 8);
                int32_t *head = &W.counters[LFQ_CNT_XHEAD + 32 * x];
                /* an empty slice is recognised with a plain L2 load: only claims that can succeed pay for an atomic */
                int b = __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (b < hi - lo) {
                    b = lfq_claim(head, LFQ_SCREEN_CLAIM);
                }
                if (b >= hi - lo) {
                    xcd_try++;
                    continue;
                }
                q_next = lo + b;
                q_end = min(lo + b + LFQ_SCREEN_CLAIM, hi);
            }
            const int idx = q_next + __popcll(need & ((1ull << lane) - 1ull));
            q_next = min(q_next + __popcll(need), q_end);
            if (!active && idx < q_end) {
                const uint4 *ep = reinterpret_cast<const uint4 *>(list + idx);
                const uint4 a = ep[0], b = ep[1];
                K = (int)b.y;
                if (K > MAXK) {
                    retry[idx] = 1;                 /* needs more cells than this variant keeps in registers */
                    n_retry++;
                } else {
                    active = true;
                    off0 = ((uint64_t)a.y << 32) | a.x;
                    n_obs = (int)a.z;
                    int64_t bonf = P.bonf_base;
                    if (P.bonf_dynamic) {           /* lfq_col_setup */
                        bonf = ((P.bonf_reset_first && P.bonf_base == 1) ? 0 : P.bonf_base)
                               + (int64_t)P.bonf_step * (int)b.x;
                    }
                    bonf_d = (double)bonf;
                    med = (int)(int16_t)(b.z & 0xffffu);
                    ref_code = (int)((b.z >> 16) & 0xffu);
                    rel = -(int)(off0 & 15u);       /* window start relative to the column start */
                    rounds = 0;
                    n_kept = 0;
                    list_idx = idx;
#pragma unroll
                    for (int j = 0; j < KREG; j++) {
                        v[j] = (j == MAXK - K) ? 1.0 : 0.0;
                    }
                }
            }
            need = __ballot(!active);
        }
        if (__ballot(active) == 0ull) {
            break;                                  /* nothing left to claim either */
        }

        /* ---- this lane's window: 16 observations, one load per track ---- */
        uint4 bqw = make_uint4(0, 0, 0, 0), baqw = make_uint4(~0u, ~0u, ~0u, ~0u), mqw = make_uint4(0, 0, 0, 0);
        uint4 sqw = make_uint4(~0u, ~0u, ~0u, ~0u), ntw = make_uint4(0, 0, 0, 0);
        if (active) {
            const uint64_t wbase = (uint64_t)((int64_t)off0 + rel);        /* multiple of 16 */
            bqw = *reinterpret_cast<const uint4 *>(T.bq + wbase);
            mqw = *reinterpret_cast<const uint4 *>(T.mq + wbase);
            if (!LB && T.baq) {
                baqw = *reinterpret_cast<const uint4 *>(T.baq + wbase);
            }
            if (!LB && T.sq) {
                sqw = *reinterpret_cast<const uint4 *>(T.sq + wbase);
            }
            if (T.nt_packed) {
                const uint2 n2 = *reinterpret_cast<const uint2 *>(T.nt + (wbase >> 1));
                ntw.x = n2.x;
                ntw.y = n2.y;
            } else {
                ntw = *reinterpret_cast<const uint4 *>(T.nt + wbase);
            }
        }
        /* ---- 16 rows ---- */
        const bool packed = T.nt_packed != 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            /* packed layout: observation j of the window sits in dword j / 8, byte j % 4, nibble (j % 8) / 4 (lfq_nt_at) */
            const uint32_t nib = (((j >> 3) ? ntw.y : ntw.x) >> (8 * (j & 3) + 4 * ((j & 7) >> 2))) & 15u;
            const uint32_t ntb = packed ? nib : LFQ_SCREEN_BYTE(ntw, j);
            if (LB) {
                const uint32_t code = ntb & 7u, bqb = LFQ_SCREEN_BYTE(bqw, j);
                const bool valid = code <= 3u;
                const bool is_alt = valid & (code != (uint32_t)ref_code);
                const bool bq_ok = (int)bqb >= (is_alt ? P.min_alt_bq4 : P.min_bq4);
                const double pm = s_luts.mq[LFQ_SCREEN_BYTE(mqw, j) | EM.off_mq];
                const double pl = fma(1.0 - pm, s_luts.bq[bqb], pm);
                const bool keep = active & valid & bq_ok & ((unsigned)(rel + j) < (unsigned)n_obs);
                lfq_screen_row<KREG>(v, keep ? pl : 0.0, keep ? 1.0 - pl : 1.0);
                n_kept += keep ? 1 : 0;
                continue;
            }
            const LfqObs o = lfq_eval_obs_flat(ntb, LFQ_SCREEN_BYTE(bqw, j), LFQ_SCREEN_BYTE(baqw, j), LFQ_SCREEN_BYTE(mqw, j),
                                               LFQ_SCREEN_BYTE(sqw, j), ref_code, med, P, EM, &s_luts);
            const bool keep = active & o.keep & ((unsigned)(rel + j) < (unsigned)n_obs);
            /* the reference's guards against log(0) (lfq_eval_raw) as two maxima: 0 <= p <= 1 here, and within
             * DBL_EPSILON of the ends the exact form differs by less than the pruning slack cares about */
            const double ps = fmax(o.p, LFQ_DBL_EPS);
            const double qf = fmax(1.0 - o.p, LFQ_DBL_EPS);
            lfq_screen_row<KREG>(v, keep ? ps : 0.0, keep ? qf : 1.0);
            n_kept += keep ? 1 : 0;
        }
        rel += 16;
        rounds += 1;
        if (active) {
            const bool pruned = v[KREG - 1] * bonf_d > sig_s;
            const bool give_up = !pruned && (rel >= n_obs || rounds >= max_rounds);
            if (pruned || give_up) {
                active = false;
                acc_cells += lfq_cells_of(n_kept, K);
                acc_rows += (unsigned long long)n_kept;
                if (give_up) {
                    retry[list_idx] = 1;            /* survivor: the whole-wavefront kernel finishes (and emits) it */
                    n_retry++;
                }
            }
        }
    }
    /* work accounting: one set of atomics per wavefront */
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        acc_cells += __shfl_xor(acc_cells, d, 64);
        acc_rows += __shfl_xor(acc_rows, d, 64);
        n_retry += __shfl_xor(n_retry, d, 64);
    }
    if (lane == 0 && acc_rows) {
        atomicAdd(reinterpret_cast<unsigned long long *>(W.gcounters + LFQ_GC_CELLS), acc_cells);
        atomicAdd(reinterpret_cast<unsigned long long *>(W.gcounters + LFQ_GC_ROWS), acc_rows);
    }
    if (lane == 0 && n_retry) {
        atomicAdd(&W.gcounters[LFQ_GC_SCREEN_RETRY], n_retry);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* big columns: K >= 250, one 8-wave workgroup per column                                     */
/* ------------------------------------------------------------------------------------------ */

struct LfqBigShared {
    LfqLuts luts;
    LfqRow rows[LFQ_HEAVY_WAVES][64];
    double bv[2][LFQ_HEAVY_WAVES][128];     /* strip boundary slabs, double-buffered across steps; entries */
    int be[2][LFQ_HEAVY_WAVES][128];        /* 64..127 of each slab are per-lane dump slots (see lfq_strip_chunk) */
    double zero_v[64];                      /* "no predecessor" input of the first strip */
    int zero_e[64];
    double gv[LFQ_HEAVY_WAVES][64];         /* pass boundary staged from global scratch */
    int ge[LFQ_HEAVY_WAVES][64];
    double mu[LFQ_HEAVY_WAVES];
    int col, pruned, split;
};

/* mu = sum of the column's error probabilities (all wavefronts of the workgroup), then the per-allele
 * upper bounds log(mu^c / c!) and the recurrence size kp that is still needed.  Kept out of line: its
 * registers (log, lgamma) must not inflate the strip pipeline's allocation. */
__device__ __forceinline__ void lfq_big_bounds_impl(const LfqColCtx &cx, const lfq_col_counts &cnt,
                                                    const LfqTracksDev &T, const LfqParams &P, const LfqLuts *luts,
                                                    double *mu_sh, int NW, unsigned *uf_mask_out, double *uf_bound,
                                                    int *kp_out)
{
    const int lane = lfq_lane();
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t n_chunks = (cx.n_obs + 63) / 64;
    double part = 0.0;
    /* four chunks of loads in flight per wavefront: this loop is pure memory latency otherwise */
    for (int64_t ch = w; ch < n_chunks; ch += 4 * NW) {
        LfqRaw r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            r[u] = lfq_load_chunk(cx, ch + u * NW, T);      /* past the end = ignored observations */
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            double ps, qf;
            const uint64_t km = lfq_eval_raw(cx, r[u], P, luts, &ps, &qf);
            part += ((km >> lane) & 1ull) ? ps : 0.0;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        part += __shfl_xor(part, d, 64);
    }
    if (lane == 0) {
        mu_sh[w] = part;
    }
    __syncthreads();
    double mu = 0.0;
    for (int i = 0; i < NW; i++) {
        mu += mu_sh[i];
    }
    const double lmu = log(mu);
    unsigned uf_mask = 0;
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
    *uf_mask_out = uf_mask;
    *kp_out = kp;
}

__device__ __forceinline__ void lfq_big_bounds(const LfqColCtx &cx, const lfq_col_counts &cnt, const LfqTracksDev &T,
                                            const LfqParams &P, const LfqLuts *luts, double *mu_sh, int NW,
                                            unsigned *uf_mask_out, double *uf_bound, int *kp_out)
{
    lfq_big_bounds_impl(cx, cnt, T, P, luts, mu_sh, NW, uf_mask_out, uf_bound, kp_out);
}

/* the strip pipeline of one big column with C cells per lane (C chosen so that the strips fit the
 * workgroup's wavefronts in one pass whenever possible: fewer cells per lane = shorter rows) */
/* ch0..ch1: the chunk range to run (the whole column) */
template <int C>
__device__ __forceinline__ void lfq_big_column(LfqColCtx &cx, const lfq_col_counts *cntp, int kp, unsigned uf_mask,
                                               const double *uf_bound, bool force_fe, double *bnd,
                                               const LfqTracksDev &T, const LfqParams &P, LfqBigShared &sh,
                                               const LfqWork &W, lfq_col_pvals *__restrict__ pvals,
                                               int64_t pvals_capacity, int64_t ch0, int64_t ch1)
{
    constexpr int NW = LFQ_HEAVY_WAVES;
    const int lane = lfq_lane();
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t n_chunks = ch1 - ch0;
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
        const double tflag = (gl == Lt) ? 1.0 : 0.0;
        const bool owns_tail = active && (s == n_strips - 1);
        const bool has_in = (s > 0);
        const bool in_global = has_in && (w == 0);    /* first strip of a later pass */
        const bool has_out = active && (s < n_strips - 1);
        const bool out_global = has_out && (w == nwp - 1);
        LfqStrip<C> S;
        lfq_strip_init<C>(S, s == 0, shift);

        const int64_t n_steps = n_chunks + nwp - 1;
#ifdef LFQ_PROFILE
        long long pt_stage = 0, pt_rows = 0, pt_bar = 0;
        const long long pstart = clock64();
#endif
        LfqRaw raw = lfq_load_chunk(cx, ch0, T);
        for (int64_t t = 0; t < n_steps; t++) {
            const int64_t ch = ch0 + t - w;
            if (active && ch >= ch0 && ch < ch1) {
                const int64_t idx = ch * 64 + lane;
                const double *in_v = sh.zero_v;
                const int *in_e = sh.zero_e;
                if (has_in) {
                    if (in_global) {
                        sh.gv[w][lane] = (idx < cx.n_obs) ? bnd[2 * idx] : 0.0;
                        sh.ge[w][lane] = (idx < cx.n_obs) ? (int)bnd[2 * idx + 1] : 0;
                        in_v = sh.gv[w];
                        in_e = sh.ge[w];
                    } else {
                        in_v = sh.bv[(t - 1) & 1][w - 1];
                        in_e = sh.be[(t - 1) & 1][w - 1];
                    }
                }
#ifdef LFQ_PROFILE
                const long long pc0 = clock64();
#endif
                const uint64_t km = lfq_stage_rows(cx, raw, P, &sh.luts, sh.rows[w]);
                if (ch + 1 < ch1) {
                    raw = lfq_load_chunk(cx, ch + 1, T);   /* lands while this step's rows run */
                }
#ifdef LFQ_PROFILE
                const long long pc1 = clock64();
                pt_stage += pc1 - pc0;
#endif
                if (lfq_strip_chunk<C, true>(S, sh.rows[w], km, has_in, in_v, in_e, has_out, sh.bv[t & 1][w],
                                       sh.be[t & 1][w], tflag, owns_tail, lt, cx.bonf_d, cx.sig_s)) {
                    sh.pruned = 1;
                }
                rows_tail += __popcll(km);
#ifdef LFQ_PROFILE
                pt_rows += clock64() - pc1;
#endif
                if (out_global) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (idx < cx.n_obs) {
                        bnd[2 * idx] = sh.bv[t & 1][w][lane];
                        bnd[2 * idx + 1] = (double)sh.be[t & 1][w][lane];
                    }
                }
            }
#ifdef LFQ_PROFILE
            const long long pb0 = clock64();
#endif
            __syncthreads();
#ifdef LFQ_PROFILE
            pt_bar += clock64() - pb0;
#endif
            if (sh.pruned) {
                pruned = true;
                break;
            }
        }
#ifdef LFQ_PROFILE
        if (lane == 0 && w == 0) {
            atomicAdd(&W.counters[8], (int)(pt_stage >> 8));
            atomicAdd(&W.counters[9], (int)(pt_bar >> 8));
            atomicAdd(&W.counters[10], (int)(pt_rows >> 8));
            atomicAdd(&W.counters[11], (int)((clock64() - pstart) >> 8));
            atomicAdd(&W.counters[14], (int)n_steps);
        }
#endif
        if (!pruned && owns_tail) {
            if (lfq_strip_final_prune<C>(S, lt, cx.bonf_d, cx.sig_s)) {
                sh.pruned = 1;
            }
        }
        __threadfence_block();
        __syncthreads();
        if (sh.pruned) {
            pruned = true;
        }
        if (!pruned && active) {
            lfq_strip_store_logs<C>(S, gl, shift, K, probvec);
        }
        __threadfence_block();
        __syncthreads();
    }
    if (w == ((n_strips - 1) % NW) && lane == 0) {
        lfq_account(W, rows_tail, K);
    }
    if ((!pruned || uf_mask) && w == ((n_strips - 1) % NW)) {
        /* the wave that owned the tail strip finishes the column */
        lfq_emit_pvals(cx, *cntp, probvec, K, !pruned, uf_mask, uf_bound, force_fe, rows_tail, W, pvals,
                       pvals_capacity);
    }
}

__global__ __launch_bounds__(LFQ_HEAVY_WAVES * 64, LFQ_DP512_WAVES) void lfq_dp_big_kernel(
    LfqTracksDev T, LfqParams P, const LfqLuts *__restrict__ g_luts, const lfq_col_counts *__restrict__ counts,
    LfqWork W, lfq_col_pvals *__restrict__ pvals, int64_t pvals_capacity, double *__restrict__ scratch,
    int64_t scratch_per_block)
{
    constexpr int NW = LFQ_HEAVY_WAVES;
    __shared__ LfqBigShared sh;
    __builtin_amdgcn_s_setprio(3);
    lfq_luts_to_lds(&sh.luts, g_luts);
    if (threadIdx.x < 64) {
        sh.zero_v[threadIdx.x] = 0.0;
        sh.zero_e[threadIdx.x] = 0;
    }
    const int n_unsplit = W.counters[LFQ_CNT_UNSPLIT];
    double *bnd = scratch + (int64_t)blockIdx.x * scratch_per_block;   /* pass boundary: 2 doubles / obs */

    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            sh.col = atomicAdd(&W.counters[LFQ_CNT_HEAD], 1);
            sh.pruned = 0;
        }
        __syncthreads();
        if (sh.col >= n_unsplit) {
            break;
        }
        /* the columns lfq_dp_big_prep_kernel could not (or need not) cut into row segments */
        const int h = W.unsplit[sh.col];
        const LfqEntry en = lfq_load_entry(W.entries + W.counters[LFQ_CNT_LIGHT] + W.counters[LFQ_CNT_MID], h);
        const lfq_col_counts cnt = counts[en.col];
        LfqColCtx cx;
        lfq_col_setup(cx, en, P);
        const int64_t n_chunks = (cx.n_obs + 63) / 64;
        unsigned uf_mask = 0;
        double uf_bound[3];
        int kp = 0;
        lfq_big_bounds(cx, cnt, T, P, &sh.luts, sh.mu, NW, &uf_mask, uf_bound, &kp);     /* see the prep kernel */
        const bool force_fe = uf_mask != 0;
        if (force_fe) {
            cx.sig_s = fmax(cx.sig_s, 4.5e-16 * cx.bonf_d);
        }
        if (kp == 0) {
            continue;                   /* already emitted by the prep kernel */
        }
        if (kp < 128 * NW - 1) {
            lfq_big_column<2>(cx, &counts[en.col], kp, uf_mask, uf_bound, force_fe, bnd, T, P, sh, W, pvals, pvals_capacity, 0,
                              n_chunks);
        } else {
            /* 4 cells per lane: up to K = 2044 in one pass; deeper columns run in passes.  (8 cells per
             * lane would halve the passes but doubles the kernel's register footprint, which decides
             * whether these workgroups can be resident beside the light kernel.) */
            lfq_big_column<4>(cx, &counts[en.col], kp, uf_mask, uf_bound, force_fe, bnd, T, P, sh, W, pvals, pvals_capacity, 0,
                              n_chunks);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* big columns, step 1: bounds, shortcut, and the decision to split                            */
/* ------------------------------------------------------------------------------------------ */

#define LFQ_PREP_WAVES 8

/* One 4-wave workgroup per big column (K >= LFQ_BIG_K).
 *
 * Shortcut for p-values below the 80-bit range.  P(X >= c) <= e_c(p) <= mu^c / c!  (union bound +
 * Maclaurin), mu = sum of the error probabilities.  If that bound is below e^-12200 the reference's
 * expl() underflows and it reports LDBL_MIN (snpcaller.c:1047-1059) whatever the exact
 * value is.  For the remaining alleles of such a column the reference's log_sum chain provably underflows
 * too (the chain spans > 708 in log space), so their p-values are clamped by value: they only need the
 * recurrence up to the largest non-underflowing count kp.
 *
 * kp == 0: the record is emitted here.  Otherwise the column is cut into row segments (a record in the
 * class list of its kp) or, if it is too short / too wide / the pool is full, queued for the unsplit
 * strip-pipeline kernel. */
__global__ __launch_bounds__(LFQ_PREP_WAVES * 64, LFQ_DP512_WAVES) void lfq_dp_big_prep_kernel(
    LfqTracksDev T, LfqParams P, const LfqLuts *__restrict__ g_luts, const lfq_col_counts *__restrict__ counts,
    LfqWork W, lfq_col_pvals *__restrict__ pvals, int64_t pvals_capacity)
{
    __shared__ LfqLuts s_luts;
    __shared__ double s_mu[LFQ_PREP_WAVES];
    __shared__ int s_col;
    lfq_luts_to_lds(&s_luts, g_luts);
    __builtin_amdgcn_s_setprio(3);
    const int n_big = W.counters[LFQ_CNT_BIG];
    bool first = true;
    __syncthreads();                                /* the tables are in LDS */
    for (;;) {
        /* a workgroup's first column is its own index (there are about as many workgroups as big columns: no atomic
         * round trip before the first load), the following ones are claimed behind those */
        int h = (int)blockIdx.x;
        if (!first && n_big <= (int)gridDim.x) {
            break;                                  /* every column was some workgroup's first: nothing to claim */
        }
        if (!first) {
            __syncthreads();
            if (threadIdx.x == 0) {
                s_col = (int)gridDim.x + atomicAdd(&W.counters[LFQ_CNT_HEAD_PREP], 1);
            }
            __syncthreads();
            h = s_col;
        }
        first = false;
        if (h >= n_big) {
            break;
        }
        const LfqEntry en = lfq_load_entry(W.entries + W.counters[LFQ_CNT_LIGHT] + W.counters[LFQ_CNT_MID], h);
        const lfq_col_counts cnt = counts[en.col];
        LfqColCtx cx;
        lfq_col_setup(cx, en, P);
        const int64_t n_chunks = (cx.n_obs + 63) / 64;
        unsigned uf_mask = 0;
        double uf_bound[3];
        int kp = 0;
        lfq_big_bounds_impl(cx, cnt, T, P, &s_luts, s_mu, LFQ_PREP_WAVES, &uf_mask, uf_bound, &kp);
        const bool force_fe = uf_mask != 0;
        if (force_fe) {
            /* pruned alleles become LDBL_MAX; that equals the reference's clamp only while the pruning
             * threshold sig/bonf stays above DBL_EPSILON */
            cx.sig_s = fmax(cx.sig_s, 4.5e-16 * cx.bonf_d);
        }
        if (kp == 0) {
            /* every allele is an 80-bit underflow: the record is complete (same layout as lfq_emit_pvals) */
            if (threadIdx.x == 0) {
                const int slot = atomicAdd(&W.gcounters[LFQ_GC_PVALS], 1);
                if ((int64_t)slot < pvals_capacity) {
                    lfq_col_pvals r;
                    r.col = cx.col;
                    r.bonf = cx.bonf;
#pragma unroll
                    for (int a = 0; a < 3; a++) {
                        const bool uf = (uf_mask >> a) & 1u;
                        r.logp[a] = uf ? uf_bound[a] : 0.0;
                        r.status[a] = (uint8_t)(uf ? LFQ_PV_UNDERFLOW : LFQ_PV_NONE);
                    }
                    r.ref_base = (uint8_t)"ACGT"[cx.ref_code & 3];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        r.pad_[i] = 0;
                    }
                    r.counts = cnt;
                    r.dp_rows = 0;
                    r.pad2_ = 0;
                    r.reserved_ = 0;
                    pvals[slot] = r;
                } else {
                    W.gcounters[LFQ_GC_OVERFLOW] = 1;
                }
            }
            continue;
        }
        if (threadIdx.x == 0) {
            const int n_new = lfq_split_plan(kp, n_chunks, 0, min(P.seg_max, max(2, P.seg_budget_big / max(n_big, 1))));
            int64_t cell0 = 0;
            LfqLong *slot = (n_new > 0) ? lfq_long_reserve(W, kp, n_new, &cell0) : nullptr;
            if (slot) {
                LfqLong r;
                r.off0 = cx.off0;
                r.bonf = cx.bonf;
                r.cell0 = cell0;
                r.n_obs = (int32_t)cx.n_obs;
                r.col = cx.col;
                r.K = kp;
                r.n_seg = n_new;
                r.ch_begin = 0;
                r.phase1 = 0;
                r.uf_mask = uf_mask;
                r.force_fe = force_fe ? 1 : 0;
                r.pruned = 0;
                r.rows = 0;
                r.median_ref_bq = (int16_t)cx.median_ref_bq;
                r.ref_code = (uint8_t)cx.ref_code;
                r.pad0_ = 0;
                r.pad1_ = 0;
                r.uf_bound[0] = uf_bound[0];
                r.uf_bound[1] = uf_bound[1];
                r.uf_bound[2] = uf_bound[2];
                r.pad_[0] = r.pad_[1] = r.pad_[2] = r.pad_[3] = 0;
                *slot = r;
            } else {
                W.unsplit[atomicAdd(&W.counters[LFQ_CNT_UNSPLIT], 1)] = h;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* row segments of split columns: one wavefront each                                           */
/* ------------------------------------------------------------------------------------------ */

/* chunks [ch0, ch1) of a column from the identity distribution, C cells per lane ((K + C) / C <= 64) */
template <int C>
__device__ __forceinline__ void lfq_wave_segment(const LfqColCtx &cx, int64_t ch0, int64_t ch1, const LfqTracksDev &T,
                                                 const LfqParams &P, const LfqLuts *L, LfqRow *rows,
                                                 LfqSegCell *out)
{
    const int lane = lfq_lane();
    const int K = cx.K;
    const int shift = (C - K % C) % C;
    const int lt = (K + shift) / C;
    const double tflag = (lane == lt) ? 1.0 : 0.0;
    LfqStrip<C> S;
    lfq_strip_init<C>(S, true, shift);
    bool pruned = false;
    int n_rows = 0;
    LfqRaw raw = lfq_load_chunk(cx, ch0, T);
    for (int64_t ch = ch0; ch < ch1; ch++) {
        const uint64_t km = lfq_stage_rows2(cx, raw, P, L, reinterpret_cast<LfqRow2 *>(rows));
        if (ch + 1 < ch1) {
            raw = lfq_load_chunk(cx, ch + 1, T);
        }
        const bool hit = lfq_strip_chunk2<C>(S, reinterpret_cast<const LfqRow2 *>(rows), tflag, lt, cx.bonf_d, cx.sig_s);
        n_rows += __popcll(S.rows >= 64 ? km : (km & ((1ull << S.rows) - 1ull)));
        if (hit) {
            pruned = true;
            break;
        }
    }
    if (!pruned && lfq_strip_final_prune<C>(S, lt, cx.bonf_d, cx.sig_s)) {
        pruned = true;
    }
    /* a segment's own tail is a lower bound of the column's: pruned here = pruned for good.  The flag and
     * the row count travel in the pad field of the segment's cell 0 (read by the combine kernel). */
    lfq_strip_store_cells<C>(S, lane, shift, K, out);
    if (lane == 0) {
        __builtin_nontemporal_store(n_rows | (pruned ? (int)0x40000000 : 0), &out[0].pad_);
    }
}

/* MODE 0: the classes fed by the mid kernel (1 and 4 cells per lane); MODE 1: the classes fed by the big
 * prep kernel (8, 16, 32).  Items = (record, segment) pairs, class-major, static stride over the
 * wavefronts: the loop is free of atomics. */
template <int MODE>
__global__ __launch_bounds__(256) void lfq_dp_seg_kernel(LfqTracksDev T, LfqParams P,
                                                         const LfqLuts *__restrict__ g_luts, LfqWork W)
{
    __shared__ LfqLuts s_luts;
    __shared__ LfqRow s_rows[4][64];
    __builtin_amdgcn_s_setprio(3);
    lfq_luts_to_lds(&s_luts, g_luts);
    __syncthreads();
    constexpr int C_LO = MODE ? 2 : 0, C_N = MODE ? 3 : 2;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int per_class = W.long_cap / LFQ_SEG_CLASSES;
    int n_items[C_N], total = 0;
#pragma unroll
    for (int c = 0; c < C_N; c++) {
        n_items[c] = min(W.counters[LFQ_CNT_LONG0 + C_LO + c], per_class) * LFQ_SEG_MAX;
        total += n_items[c];
    }
    LfqRow *rows = s_rows[wave];
    const int n_waves_total = (int)gridDim.x * 4;
    for (int idx = (int)blockIdx.x * 4 + wave; idx < total; idx += n_waves_total) {
        int c = 0, i = idx;
#pragma unroll
        for (int k = 0; k < C_N - 1; k++) {
            if (c == k && i >= n_items[k]) {
                i -= n_items[k];
                c = k + 1;
            }
        }
        const int r = i % LFQ_SEG_MAX;
        const LfqLong R = W.longs[(C_LO + c) * per_class + i / LFQ_SEG_MAX];
        if (r < R.phase1 || r >= R.n_seg) {
            continue;
        }
        LfqColCtx cx;
        lfq_ctx_from_long(cx, R, P);
        int64_t c0, c1;
        lfq_seg_range(R, r, &c0, &c1);
        LfqSegCell *out = W.pool + R.cell0 + (int64_t)r * (R.K + 1);
        if (MODE == 0) {
            if (c == 0) {
                lfq_wave_segment<1>(cx, c0, c1, T, P, &s_luts, rows, out);
            } else {
                lfq_wave_segment<4>(cx, c0, c1, T, P, &s_luts, rows, out);
            }
        } else {
            if (c == 0) {
                lfq_wave_segment<8>(cx, c0, c1, T, P, &s_luts, rows, out);
            } else if (c == 1) {
                lfq_wave_segment<16>(cx, c0, c1, T, P, &s_luts, rows, out);
            } else {
                lfq_wave_segment<32>(cx, c0, c1, T, P, &s_luts, rows, out);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* combine: fold the segment distributions of a split column, then finish it like the big kernel */
/* ------------------------------------------------------------------------------------------ */

#ifndef LFQ_COMB_THREADS
#define LFQ_COMB_THREADS 512
#endif
#define LFQ_FOLD_MAX_CLASS 3        /* cells-per-lane classes 0..3 (K <= 1008) are folded by lfq_dp_fold_kernel */
#define LFQ_COMB_CELLS 2048
#define LFQ_COMB_PER_THREAD (LFQ_COMB_CELLS / LFQ_COMB_THREADS)

struct LfqCombShared {
    double av[LFQ_COMB_CELLS];
    double bv[LFQ_COMB_CELLS];
    int ae[LFQ_COMB_CELLS];
    int be[LFQ_COMB_CELLS];
    double rv[LFQ_COMB_THREADS / 64];
    int re[LFQ_COMB_THREADS / 64];
    double cv[LFQ_COMB_CELLS / 64 + 1];     /* chunk totals of the suffix scan */
    int ce[LFQ_COMB_CELLS / 64 + 1];
    double tot_v;
    int tot_e;
    int idx, pruned;
    /* fast fold (see lfq_comb_plan): both distributions as plain doubles under one exponential tilt */
    double as[LFQ_COMB_CELLS];
    double bs[LFQ_COMB_CELLS];
    int red_lo[LFQ_COMB_THREADS / 64], red_hi[LFQ_COMB_THREADS / 64];
    int plan_fast, plan_t, plan_sa, plan_sb;
};

__device__ __forceinline__ LfqExt lfq_ext_norm(double v, int e)
{
    LfqExt r;
    r.v = (v > 0.0) ? __builtin_amdgcn_frexp_mant(v) : 0.0;
    r.e = (v > 0.0) ? e + __builtin_amdgcn_frexp_exp(v) : LFQ_EXT_ZERO_E;
    return r;
}

/* ---- fast fold --------------------------------------------------------------------------------------------
 * The convolution of two count distributions costs K^2 / 2 products, and with a binary exponent per cell every
 * product drags an exponent addition, a running maximum and two ldexp behind it (~10 instructions).  But both
 * factors are distributions of sums of independent Bernoulli trials: log-concave, their exponents fall almost
 * linearly in the cell index (cell k ~ mu^k / k!).  Multiplying cell k by 2^(t k) with ONE integer tilt t for both
 * factors leaves the convolution structure intact -- (a_i 2^(t i)) (b_j 2^(t j)) = a_i b_j 2^(t (i + j)) -- and, with
 * t = minus the average slope, flattens both sequences into the range of a plain double (C3's K ~ 500 segments:
 * ~550 binary orders after the tilt against ~2900 before).  Scaling by powers of two is exact, so the fold becomes
 * what it looks like: one FMA per product on pre-scaled LDS arrays, and one frexp per OUTPUT cell to get back to
 * (mantissa, exponent).  lfq_comb_plan picks t and the two shifts and checks that every non-zero cell of both
 * factors (tail cells included) lands within 2^+-LFQ_COMB_SPAN of the centre; if not (very wide K against short
 * segments), the fold of that pair runs the exact-exponent loops below. */
#define LFQ_COMB_SPAN 800          /* binary orders a tilted factor may span */
#define LFQ_COMB_TOP 380           /* largest tilted cell = 2^TOP: products <= 2^760, sums of 2048 of them < 2^1023 */

/* block-wide min / max over the threads' (lo, hi); every thread gets the result */
__device__ __forceinline__ void lfq_comb_minmax(LfqCombShared &sh, int &lo, int &hi)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo = min(lo, __shfl_xor(lo, d, 64));
        hi = max(hi, __shfl_xor(hi, d, 64));
    }
    const int w = (int)(threadIdx.x >> 6);
    if (lfq_lane() == 0) {
        sh.red_lo[w] = lo;
        sh.red_hi[w] = hi;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < LFQ_COMB_THREADS / 64; i++) {
        lo = min(lo, sh.red_lo[i]);
        hi = max(hi, sh.red_hi[i]);
    }
    __syncthreads();
}

/* tilt + shifts for the pair (a, b) in sh.av/ae, sh.bv/be (cells 0..K, cell K = absorbing tail); true = fast fold */
__device__ __forceinline__ bool lfq_comb_plan(LfqCombShared &sh, int K)
{
    const int tid = threadIdx.x;
    /* slope of the exponents between cell 0 and the last non-zero regular cell of each factor */
    int last_a = -1, last_b = -1;
    for (int k = tid; k < K; k += LFQ_COMB_THREADS) {
        if (sh.av[k] > 0.0) last_a = k;
        if (sh.bv[k] > 0.0) last_b = k;
    }
    {
        int dummy = 0;
        lfq_comb_minmax(sh, dummy, last_a);
        dummy = 0;
        lfq_comb_minmax(sh, dummy, last_b);
    }
    if (last_a < 0 || last_b < 0 || !(sh.av[0] > 0.0) || !(sh.bv[0] > 0.0)) {
        return false;
    }
    const double sl_a = last_a > 0 ? (double)(sh.ae[last_a] - sh.ae[0]) / last_a : 0.0;
    const double sl_b = last_b > 0 ? (double)(sh.be[last_b] - sh.be[0]) / last_b : 0.0;
    const int t = -(int)lrint(0.5 * (sl_a + sl_b));
    int lo_a = INT_MAX, hi_a = INT_MIN, lo_b = INT_MAX, hi_b = INT_MIN;
    for (int k = tid; k <= K; k += LFQ_COMB_THREADS) {
        if (sh.av[k] > 0.0) {
            const int E = sh.ae[k] + t * k;
            lo_a = min(lo_a, E);
            hi_a = max(hi_a, E);
        }
        if (sh.bv[k] > 0.0) {
            const int E = sh.be[k] + t * k;
            lo_b = min(lo_b, E);
            hi_b = max(hi_b, E);
        }
    }
    lfq_comb_minmax(sh, lo_a, hi_a);
    lfq_comb_minmax(sh, lo_b, hi_b);
    if ((long long)hi_a - lo_a > LFQ_COMB_SPAN || (long long)hi_b - lo_b > LFQ_COMB_SPAN) {
        return false;
    }
    if (tid == 0) {
        sh.plan_t = t;
        sh.plan_sa = hi_a - LFQ_COMB_TOP;
        sh.plan_sb = hi_b - LFQ_COMB_TOP;
    }
    return true;
}

/* Footprint: this kernel mostly finds nothing to do (the tree fold finishes every column of the usual batches), but it is
 * a link of the chain, and when the count kernel of the next batch fills the machine its workgroups only start where
 * they fit into what ONE retiring count workgroup frees: 2 wavefronts per SIMD x <= 168 registers (at 204 they waited
 * for a CU with no count workgroup at all -- 1.2-1.6 ms for 20 us of work, the longest link of the chain).  The LDS
 * block is requested at launch (dynamic) so that the compiler's occupancy estimate, which its 82 KB would pin at two
 * wavefronts per SIMD, does not overrule the register bound. */
template <int MODE>
__global__ __launch_bounds__(LFQ_COMB_THREADS, LFQ_DP512_WAVES) void lfq_dp_combine_kernel(LfqParams P,
                                                                             const lfq_col_counts *__restrict__ counts,
                                                                             LfqWork W, lfq_col_pvals *__restrict__ pvals,
                                                                             int64_t pvals_capacity, int only_flagged)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lfq_comb_lds[];
    LfqCombShared &sh = *reinterpret_cast<LfqCombShared *>(lfq_comb_lds);
    const int tid = threadIdx.x;
    const int lane = lfq_lane();
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    /* MODE as in lfq_dp_seg_kernel: 0 = classes 0..1 (mid kernel's columns), 1 = classes 2..4 (big class) */
    constexpr int C_LO = MODE ? 2 : 0, C_HI = MODE ? LFQ_SEG_CLASSES : 2;
    const int per_class = W.long_cap / LFQ_SEG_CLASSES;
    int n_cls[LFQ_SEG_CLASSES], n_all = 0;
#pragma unroll
    for (int c = 0; c < LFQ_SEG_CLASSES; c++) {
        n_cls[c] = (c >= C_LO && c < C_HI) ? min(W.counters[LFQ_CNT_LONG0 + c], per_class) : 0;
        n_all += n_cls[c];
    }
    for (;;) {
        __syncthreads();
        if (tid == 0) {
            sh.idx = atomicAdd(&W.counters[LFQ_CNT_HEAD_COMB + MODE], 1);
        }
        __syncthreads();
        int h = sh.idx;
        if (h >= n_all) {
            break;
        }
        int cls = LFQ_SEG_CLASSES - 1;            /* widest class first: its folds take longest */
        while (h >= n_cls[cls]) {
            h -= n_cls[cls];
            cls--;
        }
        const LfqLong R = W.longs[cls * per_class + h];
        const int K = R.K;
        if (only_flagged && cls < LFQ_FOLD_MAX_CLASS + 1 && R.pad1_ == 0) {
            continue;                               /* lfq_dp_fold_kernel finished this column */
        }
#ifdef LFQ_PROFILE
        const long long pw_rec = wall_clock64();
#endif
        LfqColCtx cx;
        lfq_ctx_from_long(cx, R, P);
        /* row counts and pruned flags of the segments the segment kernels wrote (pad of their cell 0) */
        int rows_total = R.rows;
        bool pruned = R.pruned != 0;
        for (int sgi = R.phase1; sgi < R.n_seg; sgi++) {
            const int f = W.pool[R.cell0 + (int64_t)sgi * (K + 1)].pad_;
            rows_total += f & 0x3fffffff;
            pruned = pruned || (f & 0x40000000) != 0;
        }
        if (tid == 0) {
            lfq_account(W, rows_total, K);
        }
        if (!pruned) {
            const LfqSegCell *seg = W.pool + R.cell0;
            for (int k = tid; k <= K; k += LFQ_COMB_THREADS) {
                const LfqSegCell c = seg[k];
                sh.av[k] = c.v;
                sh.ae[k] = c.e;
            }
#ifdef LFQ_PROFILE
            long long pt[6] = {0, 0, 0, 0, 0, 0};
            const long long pc_start = clock64();
            const long long pw_start = wall_clock64();
#define LFQ_PT(i) do { const long long t_ = wall_clock64(); pt[i] += t_ - pt_last; pt_last = t_; } while (0)
            long long pt_last = pw_start;
#else
#define LFQ_PT(i)
#endif
            for (int s = 1; s < R.n_seg; s++) {
                seg += K + 1;
                for (int k = tid; k <= K; k += LFQ_COMB_THREADS) {
                    const LfqSegCell c = seg[k];
                    sh.bv[k] = c.v;
                    sh.be[k] = c.e;
                }
                __syncthreads();
                LFQ_PT(0);
                const bool fast = lfq_comb_plan(sh, K);                 /* block-uniform */
                LfqExt out[LFQ_COMB_PER_THREAD];
                int f_t = 0, f_sa = 0, f_sb = 0;
                if (fast) {
                    __syncthreads();
                    f_t = sh.plan_t;
                    f_sa = sh.plan_sa;
                    f_sb = sh.plan_sb;
                    for (int k = tid; k <= K; k += LFQ_COMB_THREADS) {
                        sh.as[k] = (sh.av[k] > 0.0) ? ldexp(sh.av[k], sh.ae[k] + f_t * k - f_sa) : 0.0;
                        sh.bs[k] = (sh.bv[k] > 0.0) ? ldexp(sh.bv[k], sh.be[k] + f_t * k - f_sb) : 0.0;
                    }
                    __syncthreads();
                    /* c'_k = sum_i a'_i b'_(k-i): a'_i is a broadcast read, b'_(k-i) consecutive across the lanes */
#pragma unroll
                    for (int m = 0; m < LFQ_COMB_PER_THREAD; m++) {
                        const int k = tid + m * LFQ_COMB_THREADS;
                        double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
                        if (k < K) {
                            int i = 0;
                            for (; i + 3 <= k; i += 4) {
                                c0 = fma(sh.as[i], sh.bs[k - i], c0);
                                c1 = fma(sh.as[i + 1], sh.bs[k - i - 1], c1);
                                c2 = fma(sh.as[i + 2], sh.bs[k - i - 2], c2);
                                c3 = fma(sh.as[i + 3], sh.bs[k - i - 3], c3);
                            }
                            for (; i <= k; i++) {
                                c0 = fma(sh.as[i], sh.bs[k - i], c0);
                            }
                        }
                        out[m] = lfq_ext_norm((c0 + c1) + (c2 + c3), f_sa + f_sb - f_t * k);
                    }
                }
                /* c_k = sum_i a_i b_(k-i), k < K: aligned to a running maximum exponent */
#pragma unroll
                for (int m = 0; m < LFQ_COMB_PER_THREAD; m++) {
                    if (fast) {
                        break;
                    }
                    const int k = tid + m * LFQ_COMB_THREADS;
                    /* four independent accumulators: the LDS reads and the ldexp/add chains of consecutive
                     * terms overlap instead of serialising */
                    double acc[4] = {0.0, 0.0, 0.0, 0.0};
                    int me[4] = {4 * LFQ_EXT_ZERO_E, 4 * LFQ_EXT_ZERO_E, 4 * LFQ_EXT_ZERO_E, 4 * LFQ_EXT_ZERO_E};
                    if (k < K) {
                        int i = 0;
                        for (; i + 3 <= k; i += 4) {
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                const double t = sh.av[i + u] * sh.bv[k - i - u];
                                const int x = sh.ae[i + u] + sh.be[k - i - u];
                                const int nm = max(me[u], x);
                                acc[u] = ldexp(acc[u], me[u] - nm) + ldexp(t, x - nm);
                                me[u] = nm;
                            }
                        }
                        for (; i <= k; i++) {
                            const double t = sh.av[i] * sh.bv[k - i];
                            const int x = sh.ae[i] + sh.be[k - i];
                            const int nm = max(me[0], x);
                            acc[0] = ldexp(acc[0], me[0] - nm) + ldexp(t, x - nm);
                            me[0] = nm;
                        }
                    }
                    LfqExt s01, s23;
                    s01.e = max(me[0], me[1]);
                    s01.v = ldexp(acc[0], me[0] - s01.e) + ldexp(acc[1], me[1] - s01.e);
                    s23.e = max(me[2], me[3]);
                    s23.v = ldexp(acc[2], me[2] - s23.e) + ldexp(acc[3], me[3] - s23.e);
                    const int se = max(s01.e, s23.e);
                    out[m] = lfq_ext_norm(ldexp(s01.v, s01.e - se) + ldexp(s23.v, s23.e - se), se);
                }
                __syncthreads();
                LFQ_PT(1);
                /* in place: b[j] <- S_B(j) = b[j] + ... + b[K-1] + tail_B.  Two levels: every wavefront
                 * suffix-scans 64-cell chunks (chunk c = cells K-64c-63 .. K-64c, lane 0 = highest index), the chunk
                 * totals are suffix-combined by wave 0, then added back. */
                const int n_chunks_b = (K + 64) / 64;
                for (int c = w; c < n_chunks_b; c += LFQ_COMB_THREADS / 64) {
                    const int j = K - 64 * c - lane;
                    LfqExt incl;
                    incl.v = (j >= 0) ? sh.bv[j] : 0.0;
                    incl.e = (j >= 0) ? sh.be[j] : LFQ_EXT_ZERO_E;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        LfqExt y;
                        y.v = __shfl_up(incl.v, d, 64);
                        y.e = __shfl_up(incl.e, d, 64);
                        if (lane >= d) {
                            incl = lfq_ext_add(incl, y);
                        }
                    }
                    if (j >= 0) {
                        sh.bv[j] = incl.v;
                        sh.be[j] = incl.e;
                    }
                    if (lane == 63) {
                        sh.cv[c] = incl.v;              /* chunk total */
                        sh.ce[c] = incl.e;
                    }
                }
                __syncthreads();
                if (w == 0) {
                    /* exclusive suffix over the chunk totals: carry[c] = sum of chunks 0..c-1 (the higher cells) */
                    LfqExt t;
                    t.v = (lane < n_chunks_b) ? sh.cv[lane] : 0.0;
                    t.e = (lane < n_chunks_b) ? sh.ce[lane] : LFQ_EXT_ZERO_E;
                    LfqExt incl = t;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        LfqExt y;
                        y.v = __shfl_up(incl.v, d, 64);
                        y.e = __shfl_up(incl.e, d, 64);
                        if (lane >= d) {
                            incl = lfq_ext_add(incl, y);
                        }
                    }
                    LfqExt ex;
                    ex.v = __shfl_up(incl.v, 1, 64);
                    ex.e = __shfl_up(incl.e, 1, 64);
                    if (lane == 0) {
                        ex.v = 0.0;
                        ex.e = LFQ_EXT_ZERO_E;
                    }
                    if (lane < n_chunks_b) {
                        sh.cv[lane] = ex.v;
                        sh.ce[lane] = ex.e;
                    }
                }
                __syncthreads();
                for (int j = tid; j <= K; j += LFQ_COMB_THREADS) {
                    const int c = (K - j) >> 6;
                    LfqExt a, b;
                    a.v = sh.bv[j];
                    a.e = sh.be[j];
                    b.v = sh.cv[c];
                    b.e = sh.ce[c];
                    a = lfq_ext_add(a, b);
                    sh.bv[j] = a.v;
                    sh.be[j] = a.e;
                }
                __syncthreads();
                LFQ_PT(2);
                /* tail_C = tail_A + sum_(i<K) a_i S_B(K-i) */
                LfqExt part;
                part.v = 0.0;
                part.e = 0;
                /* the same under the tilt: S'_B(j) = S_B(j) 2^(t j - s_b) >= b'_j and normally of its order (the factors
                 * fall off beyond their mode; below it S_B ~ 1 against b_j >= b_0 = e^-mu); a'_i S'_B(K - i) carries
                 * 2^(t K - s_a - s_b) whatever i is.  If some S'_B would leave the safe range, this pair's tail takes
                 * the exact-exponent loop. */
                bool fast_tail = fast;
                if (fast) {
                    int lo_s = INT_MAX, hi_s = INT_MIN;
                    for (int j = tid + 1; j <= K; j += LFQ_COMB_THREADS) {
                        if (sh.bv[j] > 0.0) {
                            hi_s = max(hi_s, sh.be[j] + f_t * j - f_sb);
                        }
                    }
                    lfq_comb_minmax(sh, lo_s, hi_s);
                    fast_tail = hi_s <= LFQ_COMB_TOP + 200;
                }
                if (fast_tail) {
                    for (int j = tid; j <= K; j += LFQ_COMB_THREADS) {
                        sh.bs[j] = (sh.bv[j] > 0.0) ? ldexp(sh.bv[j], sh.be[j] + f_t * j - f_sb) : 0.0;
                    }
                    __syncthreads();
                    double acc = 0.0;
                    for (int i = tid; i < K; i += LFQ_COMB_THREADS) {
                        acc = fma(sh.as[i], sh.bs[K - i], acc);
                    }
                    part = lfq_ext_norm(acc, f_sa + f_sb - f_t * K);
                    if (!(acc > 0.0)) {
                        part.v = 0.0;
                        part.e = 0;
                    }
                }
                for (int i = tid; i < K && !fast_tail; i += LFQ_COMB_THREADS) {
                    LfqExt t;
                    t.v = sh.av[i] * sh.bv[K - i];
                    t.e = sh.ae[i] + sh.be[K - i];
                    part = lfq_ext_add(part, t);
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    LfqExt y;
                    y.v = __shfl_xor(part.v, d, 64);
                    y.e = __shfl_xor(part.e, d, 64);
                    part = lfq_ext_add(part, y);
                }
                if (lane == 0) {
                    sh.rv[w] = part.v;
                    sh.re[w] = part.e;
                }
                __syncthreads();
                if (tid == 0) {
                    LfqExt tot;
                    tot.v = sh.av[K];
                    tot.e = sh.ae[K];
                    for (int i = 0; i < LFQ_COMB_THREADS / 64; i++) {
                        LfqExt y;
                        y.v = sh.rv[i];
                        y.e = sh.re[i];
                        tot = lfq_ext_add(tot, y);
                    }
                    tot = lfq_ext_norm(tot.v, tot.e);
                    sh.tot_v = tot.v;
                    sh.tot_e = tot.e;
                }
                __syncthreads();
#pragma unroll
                for (int m = 0; m < LFQ_COMB_PER_THREAD; m++) {
                    const int k = tid + m * LFQ_COMB_THREADS;
                    if (k < K) {
                        sh.av[k] = out[m].v;
                        sh.ae[k] = out[m].e;
                    }
                }
                if (tid == 0) {
                    sh.av[K] = sh.tot_v;
                    sh.ae[K] = sh.tot_e;
                }
                __syncthreads();
                LFQ_PT(3);
            }
#ifdef LFQ_PROFILE
            if (tid == 0) {
                atomicAdd(&W.counters[8], (int)pt[0]);
                atomicAdd(&W.counters[9], (int)pt[1]);
                atomicAdd(&W.counters[10], (int)pt[2]);
                atomicAdd(&W.counters[11], (int)pt[3]);
                atomicAdd(&W.counters[12], (int)((clock64() - pc_start) >> 8));
                atomicAdd(&W.counters[13], (int)(wall_clock64() - pw_start));
                atomicAdd(&W.counters[14], 1);
            }
#endif
            if (R.n_seg <= 1) {
                __syncthreads();
            }
            pruned = ldexp(sh.av[K], sh.ae[K]) * cx.bonf_d > cx.sig_s;
        }
        if (!pruned) {
            /* natural logs in poissbin()'s probvec layout, into the b arrays */
            for (int k = tid; k <= K; k += LFQ_COMB_THREADS) {
                const double ed = (double)sh.ae[k];
                const double v = sh.av[k];
                sh.bv[k] = (v > 0.0) ? (ed * LFQ_LN2_HI + (ed * LFQ_LN2_LO + log(v))) : -INFINITY;
            }
        }
        __syncthreads();
#ifdef LFQ_PROFILE
        const long long pw_emit = wall_clock64();
#endif
        if ((!pruned || R.uf_mask) && w == 0) {
            const lfq_col_counts cnt = counts[R.col];
            lfq_emit_pvals(cx, cnt, sh.bv, K, !pruned, R.uf_mask, R.uf_bound, R.force_fe != 0, rows_total, W, pvals,
                           pvals_capacity);
        }
#ifdef LFQ_PROFILE
        if (tid == 0) {
            atomicAdd(&W.counters[29], (int)(wall_clock64() - pw_emit));      /* emit */
            atomicAdd(&W.counters[30], (int)(wall_clock64() - pw_rec));       /* whole record */
        }
#endif
    }
}

/* ------------------------------------------------------------------------------------------ */
/* fold tree: one 4-wavefront workgroup per column, one wavefront per fold                      */
/* ------------------------------------------------------------------------------------------ */

/* lfq_dp_combine_kernel folds a column's segments one after the other with a 512-thread workgroup and ~20 block
 * barriers per fold: ~40 us per fold whatever the arithmetic costs.  Here the folds of a column form a balanced
 * TREE -- (0,1) (2,3) (4,5) (6,7), then (01,23) (45,67), then the root -- and every fold is done by ONE wavefront
 * with the tilted plain-FMA arithmetic of lfq_comb_plan: the four wavefronts of the workgroup work on the four
 * pairs of the first level at the same time, two on the second, one on the root, with one block barrier per level.
 * The critical path of 8 segments is 3 folds instead of 7, and cheap folds are what lets the segment kernels run
 * enough wavefronts per SIMD to leave the dependent-issue regime.
 *
 * The tree is also what makes the single tilt work: both factors of a fold cover about the same number of rows, so
 * their exponents fall at about the same rate (a running total after j segments falls log2(j) bits per cell more
 * slowly than the next segment: +-700 binary orders at K = 500, j = 7 -- no common tilt flattens both).  For equal
 * factors the tilted span is ~0.53 K binary orders whatever the error rate: K <= 1008 always fits.
 *
 * Per fold: exact power-of-two scaling of both factors into LDS (zero-padded on both sides, so the product loop has
 * no bounds), c'_k = sum_i a'_i b'_(k-i) as plain FMAs (a'_i a broadcast read, b'_(k-i) consecutive across the lanes),
 * one frexp per output cell; the absorbing tail through S'_B(j) = b'_j + 2^-t S'_B(j + 1), a scan with a constant
 * ratio.  Intermediate distributions live in the pool behind the column's segments (which stay untouched).  MODE 0
 * serves cells-per-lane classes 0-1 (K <= 252), MODE 1 classes 2-3 (K <= 1008); class 4 (K <= 2016: the tilted span
 * can exceed a double's range) and any pair that fails the span / tilt check are flagged (LfqLong::pad1_) and left to
 * the block kernel, which runs behind this one with `only_flagged` and refolds them from the segments. */
#define LFQ_FOLD_PAD 64
#define LFQ_FOLD_WAVES 4

template <int KMAX>
struct LfqFoldWave {
    double as[LFQ_FOLD_PAD + KMAX + 1 + LFQ_FOLD_PAD];
    double bs[LFQ_FOLD_PAD + KMAX + 1 + LFQ_FOLD_PAD];
    double xv[KMAX + 1];        /* the left factor in exact form while the tilt is planned, then the result.  (The right
                                 * factor's exact form sits in the bs / as slots of its own cells until they are scaled.) */
    int xe[KMAX + 1];
};

__device__ __forceinline__ int lfq_wave_min_i32(int x)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        x = min(x, __shfl_xor(x, d, 64));
    }
    return x;
}

__device__ __forceinline__ int lfq_wave_max_i32(int x)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        x = max(x, __shfl_xor(x, d, 64));
    }
    return x;
}

__device__ __forceinline__ void lfq_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int KMAX, int NM>
__device__ __forceinline__ void lfq_fold_conv(LfqFoldWave<KMAX> &L, int K, int shift, int t)
{
    constexpr int PAD = LFQ_FOLD_PAD;
    const int lane = lfq_lane();
    double c[NM];
#pragma unroll
    for (int m = 0; m < NM; m++) {
        c[m] = 0.0;
    }
    const double *ap = L.as + PAD, *bp = L.bs + PAD + lane;
#pragma unroll 2
    for (int r = 0; r < 64; r++) {
        double a[NM], b[NM];
#pragma unroll
        for (int u = 0; u < NM; u++) {
            a[u] = ap[64 * u + r];
            b[u] = bp[64 * u - r];
        }
#pragma unroll
        for (int m = 0; m < NM; m++) {
#pragma unroll
            for (int u = 0; u <= m; u++) {
                c[m] = fma(a[u], b[m - u], c[m]);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < NM; m++) {
        const int k = lane + 64 * m;
        if (k < K) {                                /* x is no longer needed: the result takes its place */
            const LfqExt o = lfq_ext_norm(c[m], shift - t * k);
            L.xv[k] = o.v;
            L.xe[k] = o.e;
        }
    }
}

/* one fold by one wavefront: src (x) (*) seg (y) -> dst (pool) and L.xv / L.xe; false = this pair needs the exact path */
template <int KMAX>
__device__ __forceinline__ bool lfq_fold_pair(LfqFoldWave<KMAX> &L, const LfqSegCell *src, const LfqSegCell *seg,
                                              LfqSegCell *dst, int K, int32_t *prof)
{
#ifdef LFQ_PROFILE
    long long pt_last = wall_clock64();
#define LFQ_FP(i) do { const long long t_ = wall_clock64(); if (lfq_lane() == 0) atomicAdd(&prof[52 + (i)], (int)(t_ - pt_last)); pt_last = t_; } while (0)
#else
#define LFQ_FP(i)
#endif
    constexpr int PAD = LFQ_FOLD_PAD;
    const int lane = lfq_lane();
    /* both factors into LDS in exact form (one pass each over the pool, all loads in flight together) */
    for (int k = lane; k <= K; k += 64) {
        const LfqSegCell a = src[k], b = seg[k];
        L.xv[k] = a.v;
        L.xe[k] = a.e;
        L.bs[PAD + k] = b.v;                                        /* y: mantissa in its b' slot ... */
        reinterpret_cast<int *>(&L.as[PAD + k])[0] = b.e;           /* ... exponent in its a' slot */
    }
    lfq_wave_sync();
    LFQ_FP(0);
#define LFQ_YV(k) (L.bs[PAD + (k)])
#define LFQ_YE(k) (reinterpret_cast<const int *>(&L.as[PAD + (k)])[0])
    /* tilt from the exponent slopes of both factors (regular cells 0 .. K-1) */
    int last_a = -1, last_b = -1, ea_l = 0, eb_l = 0;
    for (int k = lane; k < K; k += 64) {
        if (L.xv[k] > 0.0) {
            last_a = k;
            ea_l = L.xe[k];
        }
        if (LFQ_YV(k) > 0.0) {
            last_b = k;
            eb_l = LFQ_YE(k);
        }
    }
    last_a = lfq_wave_max_i32(last_a);
    last_b = lfq_wave_max_i32(last_b);
    if (last_a < 0 || last_b < 0 || !(L.xv[0] > 0.0) || !(LFQ_YV(0) > 0.0)) {
        return false;
    }
    const int ea_last = lfq_rl_i32(ea_l, last_a & 63), eb_last = lfq_rl_i32(eb_l, last_b & 63);
    const double sl_a = last_a > 0 ? (double)(ea_last - L.xe[0]) / last_a : 0.0;
    const double sl_b = last_b > 0 ? (double)(eb_last - LFQ_YE(0)) / last_b : 0.0;
    const int t = -(int)lrint(0.5 * (sl_a + sl_b));
    int lo_a = INT_MAX, hi_a = INT_MIN, lo_b = INT_MAX, hi_b = INT_MIN;
    for (int k = lane; k <= K; k += 64) {
        if (L.xv[k] > 0.0) {
            const int E = L.xe[k] + t * k;
            lo_a = min(lo_a, E);
            hi_a = max(hi_a, E);
        }
        if (LFQ_YV(k) > 0.0) {
            const int E = LFQ_YE(k) + t * k;
            lo_b = min(lo_b, E);
            hi_b = max(hi_b, E);
        }
    }
    lo_a = lfq_wave_min_i32(lo_a);
    hi_a = lfq_wave_max_i32(hi_a);
    lo_b = lfq_wave_min_i32(lo_b);
    hi_b = lfq_wave_max_i32(hi_b);
    /* (|t| bounded so that the 2^(-64 t) of the scan below stays a finite double; t < 0: cells still rising at K, e.g. the
     * few-cell recurrences left of an underflow-shortcut column) */
    if (t < -14 || t > 60 || (long long)hi_a - lo_a > LFQ_COMB_SPAN || (long long)hi_b - lo_b > LFQ_COMB_SPAN) {
        return false;
    }
    const int sa = hi_a - LFQ_COMB_TOP, sb = hi_b - LFQ_COMB_TOP;
    LFQ_FP(1);
    const LfqExt tail_a = {L.xv[K], L.xe[K]};
    const double tail_b = (LFQ_YV(K) > 0.0) ? ldexp(LFQ_YV(K), LFQ_YE(K) + t * K - sb) : 0.0;    /* inside the checked range */
    lfq_wave_sync();                                /* (everybody has read y[K] before its slot is zeroed below) */
    /* tilted factors, regular cells only; everything else of the arrays is zero */
    for (int k = lane; k < KMAX + 1 + PAD; k += 64) {
        double a_s = 0.0, b_s = 0.0;
        if (k < K) {
            const double yv = LFQ_YV(k);
            const int ye = LFQ_YE(k);
            a_s = (L.xv[k] > 0.0) ? ldexp(L.xv[k], L.xe[k] + t * k - sa) : 0.0;
            b_s = (yv > 0.0) ? ldexp(yv, ye + t * k - sb) : 0.0;
        }
        L.as[PAD + k] = a_s;
        L.bs[PAD + k] = b_s;
    }
#undef LFQ_YV
#undef LFQ_YE
    lfq_wave_sync();
    LFQ_FP(2);
    /* c'_k = sum_(i <= k) a'_i b'_(k-i).  Cells in chunks of 64: k = 64 m + l (l = lane), i = 64 u + r:
     *     c'[64 m + l] = sum_r sum_(u <= m) a'[64 u + r] b'[64 (m - u) + l - r]
     * so for one r a lane needs NM values of a' (broadcast reads) and NM values of b' (consecutive across the lanes; the
     * zero border below b'_0 absorbs l < r) for all NM (NM + 1) / 2 products of that r: 0.44 LDS reads per multiply-add
     * at NM = 8 instead of 2 -- a wavefront on its own gets a fraction of the LDS rate and has nothing to hide the
     * latency behind, which made the one-product-per-two-reads loop 60 us per fold.  Cells >= K of both tilted arrays
     * are zero, so no bounds anywhere; outputs >= K are not stored. */
    if constexpr (KMAX <= 256) {
        if (K <= 128) {
            lfq_fold_conv<KMAX, 2>(L, K, sa + sb, t);
        } else {
            lfq_fold_conv<KMAX, 4>(L, K, sa + sb, t);
        }
    } else {
        const int nm = (K + 63) / 64;               /* wave-uniform */
        if (nm <= 6) {
            lfq_fold_conv<KMAX, 6>(L, K, sa + sb, t);
        } else if (nm <= 8) {
            lfq_fold_conv<KMAX, 8>(L, K, sa + sb, t);
        } else if (nm <= 9) {
            lfq_fold_conv<KMAX, 9>(L, K, sa + sb, t);
        } else if (nm <= 10) {
            lfq_fold_conv<KMAX, 10>(L, K, sa + sb, t);
        } else if (nm <= 12) {
            lfq_fold_conv<KMAX, 12>(L, K, sa + sb, t);
        } else {
            lfq_fold_conv<KMAX, 16>(L, K, sa + sb, t);
        }
    }
    lfq_wave_sync();
    LFQ_FP(3);
    /* S'_B(j) = sum_(m >= j) b'_m 2^(-t (m - j)), cell K = the factor's tail: chunks of 64 from the top, lane l of chunk c
     * holds j = K - 64 c - l; y_l = x_l + rho y_(l-1), rho = 2^-t (2^(-t d) may underflow to 0 for large t d: what it
     * would have multiplied is then below every cell of the checked span anyway) */
    {
        double carry = 0.0;
        for (int c = 0; c * 64 <= K; c++) {
            const int j = K - 64 * c - lane;
            double y = (j == K) ? tail_b : (j >= 0 ? L.bs[PAD + j] : 0.0);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const double up = __shfl_up(y, d, 64);
                if (lane >= d) {
                    y = fma(up, ldexp(1.0, -t * d), y);
                }
            }
            y = fma(carry, ldexp(1.0, -t * (lane + 1)), y);
            carry = lfq_rl_f64(y, 63);
            if (j >= 0) {
                L.bs[PAD + j] = y;
            }
        }
    }
    lfq_wave_sync();
    LFQ_FP(4);
    /* tail_C = tail_A + 2^(sa + sb - t K) sum_(i < K) a'_i S'_B(K - i) */
    double acc = 0.0;
    for (int i = lane; i < K; i += 64) {
        acc = fma(L.as[PAD + i], L.bs[PAD + K - i], acc);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        acc += __shfl_xor(acc, d, 64);
    }
    if (!(acc < 1.7e308)) {                         /* S'_B left the double range: the block kernel's exact path */
        return false;
    }
    LfqExt tot = tail_a;
    if (acc > 0.0) {
        tot = lfq_ext_add(tot, lfq_ext_norm(acc, sa + sb - t * K));
    }
    tot = lfq_ext_norm(tot.v, tot.e);
    if (lane == (K & 63)) {
        L.xv[K] = tot.v;
        L.xe[K] = tot.e;
    }
    lfq_wave_sync();
    for (int k = lane; k <= K; k += 64) {
        LfqSegCell c;
        c.v = L.xv[k];
        c.e = L.xe[k];
        c.pad_ = 0;
        dst[k] = c;
    }
    LFQ_FP(5);
    if (lfq_lane() == 0) {
        LFQ_FP(6);
    }
#ifdef LFQ_PROFILE
    if (lfq_lane() == 0) atomicAdd(&prof[60], 1);
#endif
    return true;
}

template <int MODE>
__global__ __launch_bounds__(64 * LFQ_FOLD_WAVES) void lfq_dp_fold_kernel(LfqParams P,
                                                                        const lfq_col_counts *__restrict__ counts,
                                                                        LfqWork W, lfq_col_pvals *__restrict__ pvals,
                                                                        int64_t pvals_capacity)
{
    constexpr int KMAX = MODE ? 1008 : 252;
    constexpr int NW = LFQ_FOLD_WAVES;
    __shared__ LfqFoldWave<KMAX> s_wave[NW];
    __shared__ int s_slot[LFQ_SEG_MAX];
    __shared__ int s_idx, s_slow;
    const int lane = lfq_lane();
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    LfqFoldWave<KMAX> &L = s_wave[wv];
    constexpr int C_LO = MODE ? 2 : 0, C_HI = MODE ? LFQ_FOLD_MAX_CLASS + 1 : 2;
    const int per_class = W.long_cap / LFQ_SEG_CLASSES;
    int n_cls[LFQ_SEG_CLASSES], n_all = 0;
#pragma unroll
    for (int c = 0; c < LFQ_SEG_CLASSES; c++) {
        n_cls[c] = (c >= C_LO && c < C_HI) ? min(W.counters[LFQ_CNT_LONG0 + c], per_class) : 0;
        n_all += n_cls[c];
    }
    /* the zero borders of the tilted arrays never change */
    L.as[lane] = 0.0;
    L.bs[lane] = 0.0;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            s_idx = atomicAdd(&W.counters[LFQ_CNT_HEAD_FOLD + MODE], 1);
            s_slow = 0;
        }
        if (threadIdx.x < LFQ_SEG_MAX) {
            s_slot[threadIdx.x] = (int)threadIdx.x; /* where the distribution of tree node i of this level lives */
        }
        __syncthreads();
        int h = s_idx;
        if (h >= n_all) {
            break;
        }
        int cls = C_HI - 1;                         /* widest class first: its folds take longest */
        while (h >= n_cls[cls]) {
            h -= n_cls[cls];
            cls--;
        }
        LfqLong *Rp = W.longs + cls * per_class + h;
        const LfqLong R = *Rp;
        const int K = R.K;
        LfqColCtx cx;
        lfq_ctx_from_long(cx, R, P);
        int rows_total = R.rows;
        bool pruned = R.pruned != 0;
        for (int sgi = R.phase1; sgi < R.n_seg; sgi++) {
            const int f = W.pool[R.cell0 + (int64_t)sgi * (K + 1)].pad_;
            rows_total += f & 0x3fffffff;
            pruned = pruned || (f & 0x40000000) != 0;
        }
        bool slow = K > KMAX;
        LfqSegCell *base = W.pool + R.cell0;
        if (!pruned && !slow) {
            int next_slot = R.n_seg;                /* intermediates go behind the segments, which stay untouched */
            for (int stride = 1; stride < R.n_seg; stride *= 2) {
                /* the pairs of this level, NW at a time */
                const int n_pairs = (R.n_seg - stride + 2 * stride - 1) / (2 * stride);
                for (int p0 = 0; p0 < n_pairs; p0 += NW) {
                    const int pr = p0 + wv, s0 = pr * 2 * stride;
                    if (pr < n_pairs) {
                        const bool ok = lfq_fold_pair<KMAX>(L, base + (int64_t)s_slot[s0] * (K + 1),
                                                            base + (int64_t)s_slot[s0 + stride] * (K + 1),
                                                            base + (int64_t)(next_slot + pr) * (K + 1), K, W.counters);
                        if (!ok && lane == 0) {
                            s_slow = 1;
                        }
                    }
                    __threadfence_block();
                    __syncthreads();
                    if (pr < n_pairs && lane == 0) {
                        s_slot[s0] = next_slot + pr;
                    }
                    __syncthreads();
                }
                next_slot += n_pairs;
                if (s_slow) {
                    break;
                }
            }
            slow = s_slow != 0;
        }
        if (slow) {
            if (threadIdx.x == 0) {
                Rp->pad1_ = 1;                      /* lfq_dp_combine_kernel(only_flagged) takes it from here */
            }
            continue;
        }
        if (wv != 0) {
            continue;                               /* the root of the tree was folded by wavefront 0 */
        }
        if (!pruned) {
            if (R.n_seg <= 1) {                     /* a single segment: nothing was folded */
                for (int k = lane; k <= K; k += 64) {
                    const LfqSegCell c = base[k];
                    L.xv[k] = c.v;
                    L.xe[k] = c.e;
                }
                lfq_wave_sync();
            }
            pruned = ldexp(L.xv[K], L.xe[K]) * cx.bonf_d > cx.sig_s;
        }
        if (lane == 0) {
            lfq_account(W, rows_total, K);
        }
        if (!pruned) {
            /* natural logs in poissbin()'s probvec layout, into the a' array */
            for (int k = lane; k <= K; k += 64) {
                const double ed = (double)L.xe[k];
                const double v = L.xv[k];
                L.as[k] = (v > 0.0) ? (ed * LFQ_LN2_HI + (ed * LFQ_LN2_LO + log(v))) : -INFINITY;
            }
            lfq_wave_sync();
        }
        if (!pruned || R.uf_mask) {
            const lfq_col_counts cnt = counts[R.col];
            lfq_emit_pvals(cx, cnt, L.as, K, !pruned, R.uf_mask, R.uf_bound, R.force_fe != 0, rows_total, W, pvals,
                           pvals_capacity);
        }
        /* the probvec sat in the low border of a': zero it again */
        lfq_wave_sync();
        L.as[lane] = 0.0;
        lfq_wave_sync();
    }
}

/* ------------------------------------------------------------------------------------------ */
/* launchers                                                                                   */
/* ------------------------------------------------------------------------------------------ */

/* ------------------------------------------------------------------------------------------ */
/* -t / --approx-threshold: the Poisson gate in front of the DP (snpcaller.c:1128-1142)          */
/* ------------------------------------------------------------------------------------------ */
/* A column with more than approx_n error probabilities is given up without the DP when the tail of the Poisson
 * distribution with the same mean, 1 - gsl_cdf_poisson_P(K - 1, mu), times the Bonferroni factor exceeds sig.  GSL is
 * not part of the reference tree: what is evaluated is the definition it implements (cdf/poisson.c, cdf/gamma.c:
 * P(X <= k) = Q(k + 1, mu), computed as 1 - P(k + 1, mu) below the mean), including the two roundings of "1 - (1 - P)"
 * in double -- the decision differs from a GSL build's only where the approximation lies within rounding of sig / bonf
 * (parity unpinned).  The Bonferroni bump of the column is not affected (lofreq_call.c:794-801 runs first).
 * On this implementation the gate is a parity feature, not a shortcut: mu costs a pass over all tracks of the gated
 * columns, more than the DP of most of them. */

/* pass 1, one wavefront per listed column: mu = sum of its error probabilities (:1132-1135), < 0 = not gated */
__global__ __launch_bounds__(256) void lfq_approx_mu_kernel(LfqTracksDev T, LfqParams P, const LfqLuts *__restrict__ luts,
                                                            const lfq_col_counts *__restrict__ counts, LfqWork W,
                                                            double *__restrict__ mu_out)
{
    const int n_list = W.counters[LFQ_CNT_LIGHT] + W.counters[LFQ_CNT_MID] + W.counters[LFQ_CNT_BIG];
    const int n_waves = (int)gridDim.x * 4;
    const int lane = lfq_lane();
    for (int i = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); i < n_list; i += n_waves) {
        const LfqEntry en = lfq_load_entry(W.entries, i);
        if (counts[en.col].n_err_probs <= P.approx_n) {
            if (lane == 0) {
                mu_out[i] = -1.0;
            }
            continue;
        }
        LfqColCtx cx;
        lfq_col_setup(cx, en, P);
        double s = 0.;
        const int64_t n_chunks = (cx.n_obs + 63) / 64;
        for (int64_t ch = 0; ch < n_chunks; ch++) {
            const LfqRaw r = lfq_load_chunk(cx, ch, T);
            const LfqObs o = lfq_eval_obs(r.w & 0xffu, (r.w >> 8) & 0xffu, (r.w >> 16) & 0xffu, r.w >> 24, r.sq,
                                          cx.ref_code, cx.median_ref_bq, P, luts);
            s += o.keep ? o.p : 0.;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            s += __shfl_xor(s, d, 64);
        }
        if (lane == 0) {
            mu_out[i] = s;
        }
    }
}

/* lgamma(a + 1) - (a log a - a): small next to its two parts, which cancel against a log(mu) - mu below */
__device__ __forceinline__ double lfq_stirling_rest(double a)
{
    if (a < 16.) {
        return lgamma(a + 1.) - (a * log(a) - a);
    }
    const double r = 1. / a, r2 = r * r;
    return 0.5 * log(6.283185307179586477 * a) + r * (1. / 12. - r2 * (1. / 360. - r2 * (1. / 1260. - r2 * (1. / 1680.))));
}

/* 1 - gsl_cdf_poisson_P(k - 1, mu) for k >= 1, mu > 0 */
__device__ double lfq_poisson_tail(int k, double mu)
{
    const double a = (double)k;                     /* (k - 1) + 1 */
    /* x^a e^-x / Gamma(a + 1) */
    const double pre = exp(a * log(mu / a) + (a - mu) - lfq_stirling_rest(a));
    if (mu < a + 1.) {
        double sum = 1., term = 1., n = a;           /* P(a, x) = pre * sum_{j >= 0} x^j / ((a + 1) .. (a + j)) */
        for (int i = 0; i < 200000; i++) {
            n += 1.;
            term *= mu / n;
            sum += term;
            if (term < sum * 1e-17) {
                break;
            }
        }
        const double Pl = pre * sum;
        return 1. - (1. - Pl);                      /* gsl_cdf_gamma_Q: 1 - P, then snpcaller.c:1136: 1 - that */
    }
    /* Q(a, x) by the continued fraction (modified Lentz); x^a e^-x / Gamma(a) = pre * a */
    const double tiny = 1e-300;
    double b = mu + 1. - a, c = 1. / tiny, d = 1. / b, h = d;
    for (int i = 1; i < 200000; i++) {
        const double an = -(double)i * ((double)i - a);
        b += 2.;
        d = an * d + b;
        d = fabs(d) < tiny ? tiny : d;
        c = b + an / c;
        c = fabs(c) < tiny ? tiny : c;
        d = 1. / d;
        const double del = d * c;
        h *= del;
        if (fabs(del - 1.) < 2e-16) {
            break;
        }
    }
    return 1. - h * pre * a;
}

/* pass 2, one lane per listed column: the gate (:1136-1139); a column given up leaves the work list (flag byte 0) */
__global__ __launch_bounds__(256) void lfq_approx_gate_kernel(LfqParams P, LfqWork W, const double *__restrict__ mu_in,
                                                              uint8_t *__restrict__ flags)
{
    const int n_list = W.counters[LFQ_CNT_LIGHT] + W.counters[LFQ_CNT_MID] + W.counters[LFQ_CNT_BIG];
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= n_list) {
        return;
    }
    const double mu = mu_in[i];
    if (!(mu > 0.)) {
        return;                                     /* not gated (mu == 0 cannot happen: a phred value is a probability > 0) */
    }
    const LfqEntry en = W.entries[i];
    LfqColCtx cx;
    lfq_col_setup(cx, en, P);
    const double approx = lfq_poisson_tail(en.kmax, mu);
    if (approx * cx.bonf_d > P.sig) {
        flags[en.col] = 0;
        atomicAdd(&W.gcounters[LFQ_GC_APPROX_PRUNED], 1);
    }
}

int lfq_launch_approx_gate(const LfqTracksDev &t, const LfqParams &p, const LfqLuts *d_luts, const lfq_col_counts *d_counts,
                           const LfqWork &w, int64_t ncols_seg, double *d_mu, uint8_t *d_flags, void *stream)
{
    if (ncols_seg <= 0) {
        return LFQ_OK;
    }
    const unsigned waves = (unsigned)std::min<int64_t>(ncols_seg, 256 * 32);
    hipLaunchKernelGGL(lfq_approx_mu_kernel, dim3((waves + 3) / 4), dim3(256), 0, (hipStream_t)stream, t, p, d_luts, d_counts,
                       w, d_mu);
    hipLaunchKernelGGL(lfq_approx_gate_kernel, dim3((unsigned)((ncols_seg + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       p, w, (const double *)d_mu, d_flags);
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}

int lfq_launch_dp_light(const LfqTracksDev &t, const LfqParams &p, const LfqLuts *d_luts,
                        const lfq_col_counts *d_counts, const LfqWork &w, lfq_col_pvals *d_pvals,
                        int64_t pvals_capacity, int n_waves, void *stream)
{
    if (t.ncols <= 0 || n_waves <= 0) {
        return LFQ_OK;
    }
    const unsigned blocks = (unsigned)((n_waves + 3) / 4);
    hipLaunchKernelGGL(lfq_dp_wave_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, p, d_luts,
                       d_counts, w, -1, LFQ_CNT_LIGHT, d_pvals, pvals_capacity, 32, 0);
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
    hipLaunchKernelGGL(lfq_dp_wave_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, p,
                       d_luts, d_counts, w, LFQ_CNT_LIGHT, LFQ_CNT_MID, d_pvals, pvals_capacity, 1, 0);
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


int lfq_launch_dp_big_prep(const LfqTracksDev &t, const LfqParams &p, const LfqLuts *d_luts,
                           const lfq_col_counts *d_counts, const LfqWork &w, lfq_col_pvals *d_pvals,
                           int64_t pvals_capacity, int n_blocks, void *stream)
{
    if (t.ncols <= 0 || n_blocks <= 0) {
        return LFQ_OK;
    }
    hipLaunchKernelGGL(lfq_dp_big_prep_kernel, dim3((unsigned)n_blocks), dim3(LFQ_PREP_WAVES * 64), 0,
                       (hipStream_t)stream, t, p, d_luts, d_counts, w, d_pvals, pvals_capacity);
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}

int lfq_launch_dp_seg(int mode, const LfqTracksDev &t, const LfqParams &p, const LfqLuts *d_luts,
                      const LfqWork &w, int n_waves, void *stream)
{
    if (t.ncols <= 0 || n_waves <= 0) {
        return LFQ_OK;
    }
    const dim3 grid((unsigned)((n_waves + 3) / 4)), block(256);
    if (mode == 0) {
        hipLaunchKernelGGL(lfq_dp_seg_kernel<0>, grid, block, 0, (hipStream_t)stream, t, p, d_luts, w);
    } else {
        hipLaunchKernelGGL(lfq_dp_seg_kernel<1>, grid, block, 0, (hipStream_t)stream, t, p, d_luts, w);
    }
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}

int lfq_launch_dp_combine(int mode, const LfqParams &p, const lfq_col_counts *d_counts, const LfqWork &w,
                          lfq_col_pvals *d_pvals, int64_t pvals_capacity, int n_blocks, void *stream)
{
    if (n_blocks <= 0) {
        return LFQ_OK;
    }
    /* the tree fold first; the block kernel behind it takes what the fold flagged (tilt out of range, K beyond its
     * classes) and skips the columns it finished */
    {                                                   /* > 64 KB of dynamic LDS has to be allowed once per device */
        static std::atomic<unsigned long long> allowed{0ull};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) {
            return LFQ_ERR_HIP;
        }
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(allowed.load(std::memory_order_acquire) & bit)) {
            const int bytes = (int)sizeof(LfqCombShared);
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&lfq_dp_combine_kernel<0>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void *>(&lfq_dp_combine_kernel<1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
                return LFQ_ERR_HIP;
            }
            allowed.fetch_or(bit, std::memory_order_release);
        }
    }
    if (mode == 0) {
        hipLaunchKernelGGL(lfq_dp_fold_kernel<0>, dim3((unsigned)n_blocks * 4), dim3(64 * LFQ_FOLD_WAVES), 0,
                           (hipStream_t)stream, p, d_counts, w, d_pvals, pvals_capacity);
        hipLaunchKernelGGL(lfq_dp_combine_kernel<0>, dim3((unsigned)n_blocks), dim3(LFQ_COMB_THREADS), sizeof(LfqCombShared),
                           (hipStream_t)stream, p, d_counts, w, d_pvals, pvals_capacity, 1);
    } else {
        hipLaunchKernelGGL(lfq_dp_fold_kernel<1>, dim3((unsigned)n_blocks), dim3(64 * LFQ_FOLD_WAVES), 0,
                           (hipStream_t)stream, p, d_counts, w, d_pvals, pvals_capacity);
        hipLaunchKernelGGL(lfq_dp_combine_kernel<1>, dim3((unsigned)n_blocks), dim3(LFQ_COMB_THREADS), sizeof(LfqCombShared),
                           (hipStream_t)stream, p, d_counts, w, d_pvals, pvals_capacity, 1);
    }
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}

int lfq_launch_dp_quad(const LfqTracksDev &t, const LfqParams &p, const LfqLuts *d_luts,
                       const lfq_col_counts *d_counts, const LfqWork &w, uint8_t *d_retry, lfq_col_pvals *d_pvals,
                       int64_t pvals_capacity, int n_waves, int kreg_hint, void *stream, int phase)
{
    if (t.ncols <= 0 || n_waves <= 0) {
        return LFQ_OK;
    }
    const dim3 grid((unsigned)((n_waves + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const LfqKnobs &kn = lfq_knobs();
    int force = 0;
    if (phase != 2) {
        /* one light column per lane: ONE variant is launched, chosen on the host from the K histogram the scan of this
         * context's previous batch left behind (a batch cannot wait for its own scan without stalling the host).  Any
         * choice is correct: a column with more alt bases than the variant has cells goes to the retry kernel. */
        if (!force) {
            force = kreg_hint > 0 ? kreg_hint : 8;
        }
        /* lower-bound evaluation where the filters allow it (see the kernel) */
        const bool lb = !p.general && p.def_alt_bq == 0 && p.def_alt_jp < 0.0;
#define LFQ_LAUNCH_SCREEN(KR)                                                                                        \
    do {                                                                                                             \
        if (lb) {                                                                                                    \
            hipLaunchKernelGGL((lfq_dp_screen_kernel<KR, true>), grid, block, 0, st, t, p, d_luts, w, d_retry,       \
                               kn.screen_rounds);                                                                    \
        } else {                                                                                                     \
            hipLaunchKernelGGL((lfq_dp_screen_kernel<KR, false>), grid, block, 0, st, t, p, d_luts, w, d_retry,      \
                               kn.screen_rounds);                                                                    \
        }                                                                                                            \
    } while (0)
        if (force <= 6) LFQ_LAUNCH_SCREEN(6);
        else if (force <= 8) LFQ_LAUNCH_SCREEN(8);
        else if (force <= 10) LFQ_LAUNCH_SCREEN(10);
        else if (force <= 12) LFQ_LAUNCH_SCREEN(12);
        else if (force <= 16) LFQ_LAUNCH_SCREEN(16);
        else if (force <= 24) LFQ_LAUNCH_SCREEN(24);
        else if (force <= 32) LFQ_LAUNCH_SCREEN(32);
        else {
            hipLaunchKernelGGL(lfq_dp_wave_kernel<1>, grid, block, 0, st, t, p, d_luts, d_counts, w, -1, LFQ_CNT_LIGHT,
                               d_pvals, pvals_capacity, 32, 0);
        }
#undef LFQ_LAUNCH_SCREEN
    }
    if (phase != 1) {
        hipLaunchKernelGGL(lfq_dp_retry_kernel, grid, block, 0, st, t, p, d_luts, d_counts, w, d_retry, d_pvals,
                           pvals_capacity);
    }
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}
