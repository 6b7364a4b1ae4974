/*
 * lfq_kernels.hip -- CDNA4 / gfx950 kernels of the LoFreq per-column SNV calling path.
 *
 *   lfq_count_kernel   plp_to_errprobs()'s integer outputs (snpcaller.c:346-498): per column the
 *                      number of error probabilities, filtered / raw alt counts, strand counts and
 *                      K = max filtered alt count.  One wavefront per column, 16-byte coalesced loads
 *                      of the nt and bq tracks, byte-parallel (SWAR) compares + popcounts.  HBM-bound.
 *   lfq_scan_*         inclusive prefix count of "tested" columns = the reference's running dynamic
 *                      Bonferroni factor (lofreq_call.c:794-801), and compaction of the tested columns
 *                      into a heavy (K >= 64) and a light work list.
 *   lfq_dp_kernel      snpcaller()/poissbin()/pruned_calc_prob_dist() (snpcaller.c:831-1205): the
 *                      Poisson-binomial recurrence with the absorbing tail cell and the Bonferroni
 *                      pruning test.  One wavefront per column, cells across lanes (C cells per lane,
 *                      strip-mined for K+1 > 64*C), neighbour exchange by DPP wave_shr:1, rows
 *                      broadcast by v_readlane.  FP64 VALU bound.
 *
 * Arithmetic note (DESIGN.md "DP arithmetic"): the reference runs the recurrence in log space
 * (log_sum = max + log1p(exp(min-max)), snpcaller.c:693).  Here each cell is a double mantissa with
 * a per-lane extended binary exponent (value = v * 2^e), renormalised every 8 rows: the same
 * recurrence, two FMAs per cell instead of exp+log1p, relative error <= ~1e-12 for depth 1e4 (all
 * terms positive, no cancellation).  Results are converted to natural logs at the end; p-values stay
 * in log space until the host turns them into 80-bit long doubles exactly like the reference.
 *
 * Compile with -ffp-contract=off: the quality merge (snpcaller.c:334) must round like the
 * reference's x86-64 build (no FMA); the recurrence uses explicit fma().
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "lfq_internal.h"
#include "lofreq_synth.h"

#define LFQ_WAVE 64

/* ------------------------------------------------------------------------------------------ */
/* wave helpers                                                                                */
/* ------------------------------------------------------------------------------------------ */

__device__ __forceinline__ int lfq_lane() { return (int)(threadIdx.x & 63u); }

/* lane i receives lane i-1's value, lane 0 receives 0 (DPP wave_shr:1, VALU, no LDS) */
__device__ __forceinline__ int lfq_shr1_i32(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ double lfq_shr1_f64(double x)
{
    int lo = lfq_shr1_i32(__double2loint(x));
    int hi = lfq_shr1_i32(__double2hiint(x));
    return __hiloint2double(hi, lo);
}

/* broadcast lane `i` (wave-uniform index) */
__device__ __forceinline__ int lfq_rl_i32(int x, int i) { return __builtin_amdgcn_readlane(x, i); }

__device__ __forceinline__ double lfq_rl_f64(double x, int i)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(x), i);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), i);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ uint32_t lfq_wave_sum_u32(uint32_t x)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        x += (uint32_t)__shfl_xor((int)x, d, 64);
    }
    return x;
}

/* ------------------------------------------------------------------------------------------ */
/* per-observation evaluation == the body of plp_to_errprobs' inner loop (snpcaller.c:399-496) */
/* ------------------------------------------------------------------------------------------ */

struct LfqObs {
    bool keep;     /* contributes an error probability */
    bool is_alt;
    double p;      /* merged error probability */
};

__device__ __forceinline__ LfqObs lfq_eval_obs(uint32_t ntb, uint32_t bqb, uint32_t baqb, uint32_t mqb,
                                               uint32_t sqb, int ref_code, int median_ref_bq,
                                               const LfqParams &P, const LfqLuts *L)
{
    LfqObs o;
    const uint32_t code = ntb & 7u;
    o.keep = false;
    o.is_alt = (code != (uint32_t)ref_code);
    o.p = 0.0;
    if (code > 3u) {                       /* N is ignored entirely, snpcaller.c:386-388 */
        o.is_alt = false;
        return o;
    }
    int bq = (int)bqb;
    if (bq < P.min_bq4) {                  /* snpcaller.c:426 */
        return o;
    }
    double pb;
    if (o.is_alt) {                        /* snpcaller.c:431-441 */
        if (bq < P.min_alt_bq4) {
            return o;
        }
        if (P.def_alt_bq == -1) {
            pb = (median_ref_bq < 0) ? 0.0 : L->bq[median_ref_bq & 255];
        } else if (P.def_alt_bq != 0) {
            pb = L->bq[P.def_alt_bq & 255];
        } else {
            pb = L->bq[bq];
        }
    } else {
        pb = L->bq[bq];
    }
    const double pa = L->baq[P.use_baq ? baqb : 255u];   /* snpcaller.c:444-446 */
    const double pm = L->mq[P.use_mq ? mqb : 255u];      /* snpcaller.c:448-453, 313-319 */
    const double ps = L->sq[P.use_sq ? sqb : 255u];      /* snpcaller.c:461-463 */
    /* snpcaller.c:334, identical association; -ffp-contract=off keeps every rounding */
    const double om = 1.0 - pm, os = 1.0 - ps, oa = 1.0 - pa;
    double jp = pm + om * ps + om * os * pa + om * os * oa * pb;
    if (jp > P.jq_reject_above) {          /* merged_qual < min_jq, snpcaller.c:469 */
        return o;
    }
    if (o.is_alt) {                        /* snpcaller.c:473-490 */
        if (jp > P.alt_jq_reject_above) {
            return o;
        }
        if (P.def_alt_jp >= 0.0) {
            jp = P.def_alt_jp;
        }
    }
    o.keep = true;
    o.p = jp;
    return o;
}

/* ------------------------------------------------------------------------------------------ */
/* count kernel                                                                                */
/* ------------------------------------------------------------------------------------------ */

__device__ __forceinline__ uint32_t lfq_eq4(uint32_t code4, uint32_t x4)
{
    /* 0x80 in every byte where code == x (all bytes < 0x80) */
    const uint32_t t = code4 ^ x4;
    return ((t + 0x7F7F7F7Fu) & 0x80808080u) ^ 0x80808080u;
}

__device__ __forceinline__ uint32_t lfq_bytes_mask(int lo, int hi, int d)
{
    /* 0x80 for bytes [lo,hi) of the 16-byte chunk that fall into dword d */
    int l = lo - 4 * d, h = hi - 4 * d;
    l = l < 0 ? 0 : (l > 4 ? 4 : l);
    h = h < 0 ? 0 : (h > 4 ? 4 : h);
    const uint32_t below_h = (h >= 4) ? 0xFFFFFFFFu : ((1u << (8 * h)) - 1u);
    const uint32_t below_l = (l >= 4) ? 0xFFFFFFFFu : ((1u << (8 * l)) - 1u);
    return below_h & ~below_l & 0x80808080u;
}

struct LfqAcc {
    uint32_t raw[4], fw[4], filt[4];
};

__device__ __forceinline__ void lfq_count_dword(LfqAcc &a, uint32_t ntw, uint32_t bqw, uint32_t vm,
                                                uint32_t minbq4, uint32_t minalt4, int ref_code)
{
    const uint32_t code = ntw & 0x07070707u;
    const uint32_t fwd = ~(ntw << 4);                             /* bit 7 set where forward strand */
    const uint32_t hi = bqw | 0x80808080u;
    const uint32_t ge_min = (hi - minbq4) & vm;                   /* bit 7: bq >= min_bq */
    const uint32_t ge_alt = (hi - minalt4) & ge_min;              /* ... and >= min_alt_bq */
#pragma unroll
    for (int x = 0; x < 4; x++) {
        const uint32_t eq = lfq_eq4(code, 0x01010101u * (uint32_t)x) & vm;
        a.raw[x] += __popc(eq);
        a.fw[x] += __popc(eq & fwd);
        a.filt[x] += __popc(eq & ((x == ref_code) ? ge_min : ge_alt));
    }
}

__global__ __launch_bounds__(256) void lfq_count_kernel(LfqTracksDev T, LfqParams P,
                                                        const LfqLuts *__restrict__ luts,
                                                        lfq_col_counts *__restrict__ out,
                                                        uint8_t *__restrict__ flags,
                                                        int32_t *__restrict__ counters)
{
    __shared__ uint32_t s_hist[4][128];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = lfq_lane();
    const int64_t col = (int64_t)blockIdx.x * 4 + wave;
    if (col >= T.ncols) {
        return;
    }
    const uint64_t off0 = T.col_off[col], off1 = T.col_off[col + 1];
    const int64_t n_obs = (int64_t)(off1 - off0);
    const int cov = T.coverage_plp ? T.coverage_plp[col] : (int)n_obs;
    const int nb = T.num_bases ? T.num_bases[col] : (int)n_obs;
    const uint32_t rb = T.ref_base[col];
    const int ref_code = (rb == 'A') ? 0 : (rb == 'C') ? 1 : (rb == 'G') ? 2 : (rb == 'T') ? 3 : -1;

    lfq_col_counts r;
    r.n_err_probs = 0;
    for (int i = 0; i < 3; i++) {
        r.alt_counts[i] = r.alt_raw_counts[i] = r.alt_fw[i] = 0;
    }
    r.ref_fw = r.ref_rv = 0;
    r.kmax = 0;
    r.tested = 0;
    r.pad_[0] = r.pad_[1] = 0;
    r.median_ref_bq = -1;
    r.coverage = cov;
    /* gates: lofreq_call.c:892/754 (ref N; non-ACGT refs are N, plp.c:819-823), :930, :747 */
    r.gated = (ref_code < 0) || ((int64_t)nb * 2 < (int64_t)cov) || (nb < P.min_cov);

    LfqAcc a;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        a.raw[x] = a.fw[x] = a.filt[x] = 0;
    }

    if (!r.gated && !P.general) {
        /* fast path: only the nt and bq tracks decide the counts */
        const uint32_t minbq4 = 0x01010101u * (uint32_t)P.min_bq4;
        const uint32_t minalt4 = 0x01010101u * (uint32_t)P.min_alt_bq4;
        const uint4 *nt16 = reinterpret_cast<const uint4 *>(T.nt);
        const uint4 *bq16 = reinterpret_cast<const uint4 *>(T.bq);
        const int64_t c0 = (int64_t)(off0 >> 4), c1 = (int64_t)((off1 + 15) >> 4);
        for (int64_t ch = c0 + lane; ch < c1; ch += LFQ_WAVE) {
            const uint4 n4 = nt16[ch];
            const uint4 b4 = bq16[ch];
            const int64_t base = ch << 4;
            const int lo = (int64_t)off0 > base ? (int)((int64_t)off0 - base) : 0;
            const int hi = (int64_t)off1 < base + 16 ? (int)((int64_t)off1 - base) : 16;
            if (lo == 0 && hi == 16) {
                lfq_count_dword(a, n4.x, b4.x, 0x80808080u, minbq4, minalt4, ref_code);
                lfq_count_dword(a, n4.y, b4.y, 0x80808080u, minbq4, minalt4, ref_code);
                lfq_count_dword(a, n4.z, b4.z, 0x80808080u, minbq4, minalt4, ref_code);
                lfq_count_dword(a, n4.w, b4.w, 0x80808080u, minbq4, minalt4, ref_code);
            } else {
                lfq_count_dword(a, n4.x, b4.x, lfq_bytes_mask(lo, hi, 0), minbq4, minalt4, ref_code);
                lfq_count_dword(a, n4.y, b4.y, lfq_bytes_mask(lo, hi, 1), minbq4, minalt4, ref_code);
                lfq_count_dword(a, n4.z, b4.z, lfq_bytes_mask(lo, hi, 2), minbq4, minalt4, ref_code);
                lfq_count_dword(a, n4.w, b4.w, lfq_bytes_mask(lo, hi, 3), minbq4, minalt4, ref_code);
            }
        }
    } else if (!r.gated) {
        /* general path: merged-quality filters and/or the median-of-reference-BQ override
         * (snpcaller.c:363-379) need the full per-observation evaluation */
        int median = -1;
        if (P.def_alt_bq == -1) {
            for (int i = lane; i < 128; i += LFQ_WAVE) {
                s_hist[wave][i] = 0;
            }
            __builtin_amdgcn_wave_barrier();
            for (int64_t i = lane; i < n_obs; i += LFQ_WAVE) {
                const uint32_t ntb = T.nt[off0 + i];
                if ((int)(ntb & 7u) == ref_code) {
                    atomicAdd(&s_hist[wave][T.bq[off0 + i] & 127u], 1u);
                }
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            /* int_median (utils.c:436-457) from the histogram: lane-serial, <= 128 bins */
            uint32_t total = 0;
            for (int i = 0; i < 128; i++) {
                total += s_hist[wave][i];
            }
            if (total) {
                const uint32_t r_hi = total / 2, r_lo = (total & 1u) ? r_hi : r_hi - 1;
                int q_lo = -1, q_hi = -1;
                uint32_t run = 0;
                for (int i = 0; i < 128; i++) {
                    run += s_hist[wave][i];
                    if (q_lo < 0 && run > r_lo) q_lo = i;
                    if (q_hi < 0 && run > r_hi) q_hi = i;
                }
                median = (q_lo + q_hi) / 2;     /* (a+b)/2.0 truncated; odd size: q_lo == q_hi */
            }
        }
        r.median_ref_bq = median;
        for (int64_t i = lane; i < n_obs; i += LFQ_WAVE) {
            const uint32_t ntb = T.nt[off0 + i];
            const uint32_t code = ntb & 7u;
            if (code > 3u) {
                continue;
            }
            const LfqObs o = lfq_eval_obs(ntb, T.bq[off0 + i], T.baq ? T.baq[off0 + i] : 255u,
                                          T.mq[off0 + i], T.sq ? T.sq[off0 + i] : 255u, ref_code, median,
                                          P, luts);
            a.raw[code] += 1;
            a.fw[code] += (ntb & 8u) ? 0u : 1u;
            a.filt[code] += o.keep ? 1u : 0u;
        }
    }

    uint32_t raw[4], fw[4], filt[4];
#pragma unroll
    for (int x = 0; x < 4; x++) {
        raw[x] = lfq_wave_sum_u32(a.raw[x]);
        fw[x] = lfq_wave_sum_u32(a.fw[x]);
        filt[x] = lfq_wave_sum_u32(a.filt[x]);
    }

    if (lane == 0) {
        uint8_t flag = 0;
        if (!r.gated) {
            /* the three non-reference nucleotides in A,C,G,T order (snpcaller.c:391-397) */
            const int x0 = (ref_code == 0) ? 1 : 0;
            const int x1 = (ref_code <= 1) ? 2 : 1;
            const int x2 = (ref_code <= 2) ? 3 : 2;
#define LFQ_PICK(arr, x) ((x) == 0 ? arr[0] : (x) == 1 ? arr[1] : (x) == 2 ? arr[2] : arr[3])
            r.ref_fw = (int)LFQ_PICK(fw, ref_code);
            r.ref_rv = (int)(LFQ_PICK(raw, ref_code) - LFQ_PICK(fw, ref_code));
            r.alt_counts[0] = (int)LFQ_PICK(filt, x0);
            r.alt_counts[1] = (int)LFQ_PICK(filt, x1);
            r.alt_counts[2] = (int)LFQ_PICK(filt, x2);
            r.alt_raw_counts[0] = (int)LFQ_PICK(raw, x0);
            r.alt_raw_counts[1] = (int)LFQ_PICK(raw, x1);
            r.alt_raw_counts[2] = (int)LFQ_PICK(raw, x2);
            r.alt_fw[0] = (int)LFQ_PICK(fw, x0);
            r.alt_fw[1] = (int)LFQ_PICK(fw, x1);
            r.alt_fw[2] = (int)LFQ_PICK(fw, x2);
#undef LFQ_PICK
            r.n_err_probs = (int)(filt[0] + filt[1] + filt[2] + filt[3]);
            const int kmax = max(r.alt_counts[0], max(r.alt_counts[1], r.alt_counts[2]));
            r.kmax = kmax;
            r.tested = kmax > 0;                     /* lofreq_call.c:768-780 */
            flag = (uint8_t)((r.tested ? 1 : 0) | ((kmax >= LFQ_HEAVY_K) ? 2 : 0));
        }
        out[col] = r;
        flags[col] = flag;
        if (n_obs > (int64_t)counters[LFQ_CNT_MAXDEPTH]) {
            atomicMax(&counters[LFQ_CNT_MAXDEPTH], (int)min(n_obs, (int64_t)0x7fffffff));
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* scan kernels: running Bonferroni prefix + work-list compaction                              */
/* ------------------------------------------------------------------------------------------ */

#define LFQ_SCAN_THREADS 1024
#define LFQ_SCAN_ITEMS 4
#define LFQ_SCAN_TILE (LFQ_SCAN_THREADS * LFQ_SCAN_ITEMS)

/* block-wide exclusive scan of two counters packed as (tested | heavy << 32) */
__device__ uint64_t lfq_block_excl_scan(uint64_t x, uint64_t *total, uint64_t *s_wave /*[16]*/)
{
    const int lane = lfq_lane(), wave = (int)(threadIdx.x >> 6);
    uint64_t incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t y = ((uint64_t)(uint32_t)__shfl_up((int)(incl >> 32), d, 64) << 32)
                     | (uint32_t)__shfl_up((int)(incl & 0xffffffffu), d, 64);
        if (lane >= d) {
            incl += y;
        }
    }
    if (lane == 63) {
        s_wave[wave] = incl;
    }
    __syncthreads();
    uint64_t wave_off = 0, tot = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) {
        const uint64_t v = s_wave[w];
        if (w < wave) {
            wave_off += v;
        }
        tot += v;
    }
    __syncthreads();
    *total = tot;
    return wave_off + incl - x;
}

__global__ __launch_bounds__(LFQ_SCAN_THREADS) void lfq_scan_tiles_kernel(int64_t ncols,
                                                                         const uint8_t *__restrict__ flags,
                                                                         uint64_t *__restrict__ tile_sums)
{
    __shared__ uint64_t s_wave[16];
    const int64_t base = (int64_t)blockIdx.x * LFQ_SCAN_TILE + (int64_t)threadIdx.x * LFQ_SCAN_ITEMS;
    uint64_t x = 0;
    for (int i = 0; i < LFQ_SCAN_ITEMS; i++) {
        if (base + i < ncols) {
            const uint32_t f = flags[base + i];
            x += (uint64_t)(f & 1u) | ((uint64_t)((f >> 1) & 1u) << 32);
        }
    }
    uint64_t total;
    (void)lfq_block_excl_scan(x, &total, s_wave);
    if (threadIdx.x == 0) {
        tile_sums[blockIdx.x] = total;
    }
}

/* single block: exclusive scan over the tile sums, totals into the counters */
__global__ __launch_bounds__(LFQ_SCAN_THREADS) void lfq_scan_sums_kernel(int64_t ntiles,
                                                                        uint64_t *__restrict__ tile_sums,
                                                                        int32_t *__restrict__ counters)
{
    __shared__ uint64_t s_wave[16];
    uint64_t carry = 0;
    for (int64_t b = 0; b < ntiles; b += LFQ_SCAN_THREADS) {
        const int64_t i = b + threadIdx.x;
        const uint64_t x = (i < ntiles) ? tile_sums[i] : 0;
        uint64_t total;
        const uint64_t ex = lfq_block_excl_scan(x, &total, s_wave);
        if (i < ntiles) {
            tile_sums[i] = carry + ex;
        }
        carry += total;
    }
    if (threadIdx.x == 0) {
        const int32_t n_tested = (int32_t)(carry & 0xffffffffu), n_heavy = (int32_t)(carry >> 32);
        counters[LFQ_CNT_TESTED] = n_tested;
        counters[LFQ_CNT_HEAVY] = n_heavy;
        counters[LFQ_CNT_LIGHT] = n_tested - n_heavy;
    }
}

__global__ __launch_bounds__(LFQ_SCAN_THREADS) void lfq_scan_apply_kernel(int64_t ncols,
                                                                         const uint8_t *__restrict__ flags,
                                                                         const uint64_t *__restrict__ tile_sums,
                                                                         LfqWork W)
{
    __shared__ uint64_t s_wave[16];
    const int64_t base = (int64_t)blockIdx.x * LFQ_SCAN_TILE + (int64_t)threadIdx.x * LFQ_SCAN_ITEMS;
    uint32_t f[LFQ_SCAN_ITEMS];
    uint64_t x = 0;
    for (int i = 0; i < LFQ_SCAN_ITEMS; i++) {
        f[i] = (base + i < ncols) ? flags[base + i] : 0u;
        x += (uint64_t)(f[i] & 1u) | ((uint64_t)((f[i] >> 1) & 1u) << 32);
    }
    uint64_t total;
    uint64_t ex = lfq_block_excl_scan(x, &total, s_wave) + tile_sums[blockIdx.x];
    uint32_t n_t = (uint32_t)(ex & 0xffffffffu), n_h = (uint32_t)(ex >> 32);
    for (int i = 0; i < LFQ_SCAN_ITEMS; i++) {
        if (base + i >= ncols) {
            break;
        }
        if (f[i] & 1u) {
            if (f[i] & 2u) {
                W.q_heavy[n_h] = (int32_t)(base + i);
                n_h++;
            } else {
                W.q_light[n_t - n_h] = (int32_t)(base + i);
            }
            n_t++;
        }
        W.tested_prefix[base + i] = (int32_t)n_t;     /* inclusive */
    }
}

/* ------------------------------------------------------------------------------------------ */
/* DP kernel                                                                                   */
/* ------------------------------------------------------------------------------------------ */

#define LFQ_LN2_HI 6.93147180369123816490e-01
#define LFQ_LN2_LO 1.90821492927058770002e-10
/* exp(x) raises FE_UNDERFLOW in glibc (result below DBL_MIN) for x < ln(2^-1022); pinned in
 * tests/test_oracle_kat.py::test_exp_underflow_threshold */
#define LFQ_EXP_UNDERFLOW_X (-708.3964185322641)

__device__ __forceinline__ double lfq_logaddexp(double a, double b)
{
    const double hi = fmax(a, b), lo = fmin(a, b);
    if (lo == -INFINITY) {
        return hi;
    }
    return hi + log1p(exp(lo - hi));
}

struct LfqColCtx {
    int64_t col;
    uint64_t off0;
    int64_t n_obs;
    int ref_code;
    int median_ref_bq;
    int K;
    double bonf_d;
    double sig_s;
};

/* Runs the recurrence for one column.  Returns true if the column was pruned
 * (P(X>=K)*bonf > sig, snpcaller.c:950/1155).  Otherwise probvec[0..K] holds natural logs of
 * P(X=k) (k<K) and P(X>=K) (k=K), like the array poissbin() returns (snpcaller.c:1020-1062). */
template <int C>
__device__ bool lfq_dp_run(const LfqColCtx &cx, const LfqTracksDev &T, const LfqParams &P,
                           const LfqLuts *L, double *__restrict__ bnd, double *__restrict__ probvec,
                           int *rows_out)
{
    const int lane = lfq_lane();
    const int K = cx.K;
    const int shift = (C - K % C) % C;
    const int Lt = (K + shift) / C;           /* global lane that owns the tail cell at j = 0 */
    const int n_strips = Lt / LFQ_WAVE + 1;
    const int lt = Lt % LFQ_WAVE;
    const int64_t n_chunks = (cx.n_obs + LFQ_WAVE - 1) / LFQ_WAVE;
    bool pruned = false;
    int rows = 0;

    for (int s = 0; s < n_strips && !pruned; s++) {
        const bool last = (s == n_strips - 1);
        const int gl = s * LFQ_WAVE + lane;
        const bool is_tail = (gl == Lt);
        double v[C];
#pragma unroll
        for (int j = 0; j < C; j++) {
            v[j] = (s == 0 && lane == 0 && j == shift) ? 1.0 : 0.0;
        }
        int e = 0, de = 0;
        bool all_zero = (s > 0);               /* wave-uniform: nothing has entered this strip yet */
        int e_in = 0;
        rows = 0;

        for (int64_t ch = 0; ch < n_chunks && !pruned; ch++) {
            const int64_t idx = ch * LFQ_WAVE + lane;
            const bool valid = idx < cx.n_obs;
            LfqObs o;
            o.keep = false;
            o.p = 0.0;
            if (valid) {
                const uint64_t g = cx.off0 + (uint64_t)idx;
                o = lfq_eval_obs(T.nt[g], T.bq[g], T.baq ? T.baq[g] : 255u, T.mq[g], T.sq ? T.sq[g] : 255u,
                                 cx.ref_code, cx.median_ref_bq, P, L);
            }
            /* the reference's guards against log(0) (snpcaller.c:872-881) as effective p and 1-p */
            const double ps = (fabs(o.p) < 2.220446049250313e-16) ? 2.220446049250313e-16 : o.p;
            const double qf = (fabs(o.p - 1.0) < 2.220446049250313e-16)
                                  ? 1.0 + (-o.p + 2.220446049250313e-16)
                                  : 1.0 - o.p;
            double bv = 0.0;
            int be = 0;
            if (s > 0 && valid && o.keep) {
                bv = bnd[2 * idx];
                be = (int)bnd[2 * idx + 1];
            }
            uint64_t km = __ballot(o.keep);
            while (km) {
                const int i = __builtin_ctzll(km);
                km &= km - 1;
                const double p = lfq_rl_f64(ps, i);
                const double q = lfq_rl_f64(qf, i);
                double x = lfq_shr1_f64(v[C - 1]);
                int dei = de;
                if (s > 0) {
                    const double xb = lfq_rl_f64(bv, i);
                    const int eb = lfq_rl_i32(be, i);
                    e_in = eb;
                    if (all_zero) {
                        e = eb;                 /* adopt the producer's scale while empty */
                        de = 0;
                        dei = 0;
                        if (xb == 0.0) {
                            if (!last && lane == 63) {
                                bnd[2 * (ch * LFQ_WAVE + i)] = 0.0;
                                bnd[2 * (ch * LFQ_WAVE + i) + 1] = (double)e;
                            }
                            rows++;
                            continue;
                        }
                        all_zero = false;
                    }
                    if (lane == 0) {
                        x = xb;
                        dei = eb - e;
                    }
                }
                if (!last && lane == 63) {
                    bnd[2 * (ch * LFQ_WAVE + i)] = v[C - 1];
                    bnd[2 * (ch * LFQ_WAVE + i) + 1] = (double)e;
                }
                const double xs = ldexp(x, dei);
                const double ph = is_tail ? 0.0 : p;
                const double q0 = is_tail ? 1.0 : q;
#pragma unroll
                for (int j = C - 1; j >= 1; j--) {
                    v[j] = fma(v[j - 1], ph, v[j] * q);
                }
                v[0] = fma(xs, p, v[0] * q0);
                rows++;

                if ((rows & 7) == 0) {
                    /* renormalise: lane maximum to [0.5,1), exponent into e */
                    double m = v[0];
#pragma unroll
                    for (int j = 1; j < C; j++) {
                        m = fmax(m, v[j]);
                    }
                    const bool nzl = m > 0.0;
                    const uint64_t nz = __ballot(nzl);
                    if (nzl) {
                        const int ex = __builtin_amdgcn_frexp_exp(m);
#pragma unroll
                        for (int j = 0; j < C; j++) {
                            v[j] = ldexp(v[j], -ex);
                        }
                        e += ex;
                    }
                    /* empty lanes adopt the scale of the nearest non-empty lane to their left */
                    const uint64_t below = nz & ((lane == 0) ? 0ull : (~0ull >> (64 - lane)));
                    const int src = below ? (63 - __builtin_clzll(below)) : lane;
                    const int e_src = __shfl(e, src, 64);
                    if (!nzl) {
                        e = below ? e_src : ((s > 0) ? e_in : e);
                    }
                    de = lfq_shr1_i32(e) - e;
                    if (last) {
                        const double tv = lfq_rl_f64(v[0], lt);
                        const int te = lfq_rl_i32(e, lt);
                        if (ldexp(tv, te) * cx.bonf_d > cx.sig_s) {
                            pruned = true;
                            break;
                        }
                    }
                }
            }
        }
        if (last && !pruned) {
            const double tv = lfq_rl_f64(v[0], lt);
            const int te = lfq_rl_i32(e, lt);
            if (ldexp(tv, te) * cx.bonf_d > cx.sig_s) {
                pruned = true;
            }
        }
        if (!pruned) {
            /* natural logs of this strip's cells */
#pragma unroll
            for (int j = 0; j < C; j++) {
                const int k = gl * C + j - shift;
                if (k >= 0 && k <= K && (k < K || j == 0)) {
                    const double ed = (double)e;
                    probvec[k] = (v[j] > 0.0) ? (ed * LFQ_LN2_HI + (ed * LFQ_LN2_LO + log(v[j]))) : -INFINITY;
                }
            }
        }
        __threadfence_block();
    }
    *rows_out = rows;
    return pruned;
}

/* probvec_tailsum (snpcaller.c:730-741) as a wave-parallel prefix scan, plus detection of the
 * exp() underflow inside the reference's sequential log_sum chain. */
__device__ double lfq_tailsum(const double *__restrict__ probvec, int start, int K, bool *fe_flag)
{
    const int lane = lfq_lane();
    double carry = -INFINITY;
    bool flag = false;
    for (int base = start; base <= K; base += LFQ_WAVE) {
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

__global__ __launch_bounds__(256) void lfq_dp_kernel(LfqTracksDev T, LfqParams P,
                                                     const LfqLuts *__restrict__ luts,
                                                     const lfq_col_counts *__restrict__ counts,
                                                     LfqWork W, lfq_col_pvals *__restrict__ pvals,
                                                     int64_t pvals_capacity, double *__restrict__ scratch,
                                                     int64_t scratch_per_wave, int n_waves)
{
    const int lane = lfq_lane();
    const int wave_id = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (wave_id >= n_waves) {
        return;
    }
    const int n_heavy = W.counters[LFQ_CNT_HEAVY];
    const int n_light = W.counters[LFQ_CNT_LIGHT];
    double *bnd = scratch + (int64_t)wave_id * scratch_per_wave;

    int light_next = wave_id;
    bool heavy_phase = true;
    for (;;) {
        int col;
        if (heavy_phase) {
            int h = 0;
            if (lane == 0) {
                h = atomicAdd(&W.counters[LFQ_CNT_HEAD], 1);
            }
            h = __builtin_amdgcn_readfirstlane(h);
            if (h < n_heavy) {
                col = W.q_heavy[h];
            } else {
                heavy_phase = false;
                continue;
            }
        } else {
            if (light_next >= n_light) {
                break;
            }
            col = W.q_light[light_next];
            light_next += n_waves;
        }
        col = __builtin_amdgcn_readfirstlane(col);

        const lfq_col_counts cnt = counts[col];
        LfqColCtx cx;
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
            const int64_t t = W.tested_prefix[col];
            bonf = ((P.bonf_base == 1) ? 0 : P.bonf_base) + 3 * t;
        }
        cx.bonf_d = (double)bonf;
        cx.sig_s = P.sig * (1.0 + P.prune_slack);

        double *probvec = bnd + 2 * cx.n_obs + 2;
        int rows = 0;
        bool pruned;
        if (cx.K < LFQ_HEAVY_K) {
            pruned = lfq_dp_run<1>(cx, T, P, luts, bnd, probvec, &rows);
        } else {
            pruned = lfq_dp_run<8>(cx, T, P, luts, bnd, probvec, &rows);
        }
        if (pruned) {
            continue;
        }

        /* per-allele p-values (snpcaller.c:1166-1196) */
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
            if (c == cx.K) {
                logp[a] = probvec[cx.K];
                status[a] = LFQ_PV_LOG;
            } else {
                bool fe = false;
                logp[a] = lfq_tailsum(probvec, c, cx.K, &fe);
                status[a] = fe ? LFQ_PV_LOG_FECLAMP : LFQ_PV_LOG;
            }
        }
        if (lane == 0) {
            const int slot = atomicAdd(&W.counters[LFQ_CNT_PVALS], 1);
            if ((int64_t)slot < pvals_capacity) {
                lfq_col_pvals r;
                r.col = col;
                r.bonf = bonf;
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    r.logp[a] = logp[a];
                    r.status[a] = (uint8_t)status[a];
                }
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
}

/* ------------------------------------------------------------------------------------------ */
/* synthetic workload generator (include/lofreq_synth.h)                                       */
/* ------------------------------------------------------------------------------------------ */

__global__ __launch_bounds__(256) void lfq_synth_kernel(lfq_synth_spec S, int64_t col_begin, int64_t ncols,
                                                        uint8_t *__restrict__ nt, uint8_t *__restrict__ bq,
                                                        uint8_t *__restrict__ baq, uint8_t *__restrict__ mq,
                                                        uint64_t *__restrict__ col_off,
                                                        uint8_t *__restrict__ ref_base)
{
    const int64_t total = ncols * (int64_t)S.depth;
    const int64_t n16 = (total + 15) / 16;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n16; t += stride) {
        uint32_t wn[4] = {0, 0, 0, 0}, wb[4] = {0, 0, 0, 0}, wa[4] = {0, 0, 0, 0}, wm[4] = {0, 0, 0, 0};
        for (int b = 0; b < 16; b++) {
            const int64_t g = t * 16 + b;
            lfq_synth_obs o;
            if (g < total) {
                const int64_t c = g / S.depth;
                o = lfq_synth_observation(&S, (uint64_t)(col_begin + c), (uint64_t)(g - c * S.depth));
            } else {
                o.nt = 4;
                o.bq = 0;
                o.baq = 0;
                o.mq = 0;
            }
            wn[b >> 2] |= (uint32_t)o.nt << (8 * (b & 3));
            wb[b >> 2] |= (uint32_t)o.bq << (8 * (b & 3));
            wa[b >> 2] |= (uint32_t)o.baq << (8 * (b & 3));
            wm[b >> 2] |= (uint32_t)o.mq << (8 * (b & 3));
        }
        reinterpret_cast<uint4 *>(nt)[t] = make_uint4(wn[0], wn[1], wn[2], wn[3]);
        reinterpret_cast<uint4 *>(bq)[t] = make_uint4(wb[0], wb[1], wb[2], wb[3]);
        reinterpret_cast<uint4 *>(baq)[t] = make_uint4(wa[0], wa[1], wa[2], wa[3]);
        reinterpret_cast<uint4 *>(mq)[t] = make_uint4(wm[0], wm[1], wm[2], wm[3]);
    }
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c <= ncols; c += stride) {
        col_off[c] = (uint64_t)c * S.depth;
        if (c < ncols) {
            ref_base[c] = (uint8_t)("ACGT"[lfq_synth_ref_code((uint64_t)(col_begin + c))]);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* launchers                                                                                   */
/* ------------------------------------------------------------------------------------------ */

#define LFQ_HIP_TRY(expr)                \
    do {                                 \
        hipError_t e_ = (expr);          \
        if (e_ != hipSuccess) {          \
            return LFQ_ERR_HIP;          \
        }                                \
    } while (0)

int lfq_launch_count(const LfqTracksDev &t, const LfqParams &p, const LfqLuts *d_luts,
                     lfq_col_counts *d_counts, uint8_t *d_flags, int32_t *d_counters, void *stream)
{
    if (t.ncols <= 0) {
        return LFQ_OK;
    }
    const unsigned blocks = (unsigned)((t.ncols + 3) / 4);
    hipLaunchKernelGGL(lfq_count_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, p, d_luts,
                       d_counts, d_flags, d_counters);
    LFQ_HIP_TRY(hipGetLastError());
    return LFQ_OK;
}

int lfq_launch_scan(int64_t ncols, const uint8_t *d_flags, const LfqWork &w, void *stream)
{
    if (ncols <= 0) {
        return LFQ_OK;
    }
    const int64_t ntiles = (ncols + LFQ_SCAN_TILE - 1) / LFQ_SCAN_TILE;
    uint64_t *tile_sums = reinterpret_cast<uint64_t *>(w.block_sums);
    hipLaunchKernelGGL(lfq_scan_tiles_kernel, dim3((unsigned)ntiles), dim3(LFQ_SCAN_THREADS), 0,
                       (hipStream_t)stream, ncols, d_flags, tile_sums);
    hipLaunchKernelGGL(lfq_scan_sums_kernel, dim3(1), dim3(LFQ_SCAN_THREADS), 0, (hipStream_t)stream, ntiles,
                       tile_sums, w.counters);
    hipLaunchKernelGGL(lfq_scan_apply_kernel, dim3((unsigned)ntiles), dim3(LFQ_SCAN_THREADS), 0,
                       (hipStream_t)stream, ncols, d_flags, (const uint64_t *)tile_sums, w);
    LFQ_HIP_TRY(hipGetLastError());
    return LFQ_OK;
}

int lfq_launch_dp(const LfqTracksDev &t, const LfqParams &p, const LfqLuts *d_luts,
                  const lfq_col_counts *d_counts, const LfqWork &w, lfq_col_pvals *d_pvals,
                  int64_t pvals_capacity, double *d_scratch, int64_t scratch_doubles_per_wave, int n_waves,
                  void *stream)
{
    if (t.ncols <= 0 || n_waves <= 0) {
        return LFQ_OK;
    }
    const unsigned blocks = (unsigned)((n_waves + 3) / 4);
    hipLaunchKernelGGL(lfq_dp_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, p, d_luts, d_counts,
                       w, d_pvals, pvals_capacity, d_scratch, scratch_doubles_per_wave, n_waves);
    LFQ_HIP_TRY(hipGetLastError());
    return LFQ_OK;
}

int lfq_launch_synth(const lfq_synth_spec *spec, int64_t col_begin, int64_t ncols, uint8_t *d_nt,
                     uint8_t *d_bq, uint8_t *d_baq, uint8_t *d_mq, uint64_t *d_col_off, uint8_t *d_ref_base,
                     void *stream)
{
    if (ncols <= 0) {
        return LFQ_OK;
    }
    const int64_t n16 = (ncols * (int64_t)spec->depth + 15) / 16;
    int64_t blocks = (n16 + 255) / 256;
    if (blocks > 256 * 32) {
        blocks = 256 * 32;
    }
    if (blocks < 1) {
        blocks = 1;
    }
    hipLaunchKernelGGL(lfq_synth_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *spec,
                       col_begin, ncols, d_nt, d_bq, d_baq, d_mq, d_col_off, d_ref_base);
    LFQ_HIP_TRY(hipGetLastError());
    return LFQ_OK;
}
