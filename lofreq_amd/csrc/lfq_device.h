/*
 * lfq_device.h -- device-side helpers shared by the HIP translation units (wave primitives and the
 * per-observation evaluation that both the count and the DP kernels use).
 */
#ifndef LFQ_DEVICE_H
#define LFQ_DEVICE_H

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "lfq_internal.h"

#define LFQ_WAVE 64

/* ------------------------------------------------------------------------------------------ */
/* wave helpers                                                                                */
/* ------------------------------------------------------------------------------------------ */

__device__ __forceinline__ int lfq_lane() { return (int)(threadIdx.x & 63u); }

/* nt byte (code | strand << 3) of observation g.  Packed layout: groups of 8 observations in 4 bytes, byte k of a
 * group holding observation k in its low and observation 4 + k in its high nibble -- so that the even / odd
 * nibbles of a dword line up with the two bq dwords of the group (lfq_count_chunks). */
__device__ __forceinline__ uint32_t lfq_nt_at(const LfqTracksDev &T, uint64_t g)
{
    if (!T.nt_packed) {
        return T.nt[g];
    }
    const uint32_t b = T.nt[(g >> 3) * 4 + (g & 3u)];
    return (g & 4u) ? (b >> 4) : (b & 15u);
}

/* lane i receives lane i-1's value, lane 0 receives 0 (DPP wave_shr:1, VALU, no LDS) */
__device__ __forceinline__ int lfq_shr1_i32(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, true);   /* bound_ctrl: lane 0 reads 0 */
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

/* Sums over groups of lanes by DPP (no LDS crossbar, no lgkmcnt wait): quad_perm [1,0,3,2] and [2,3,0,1] give every
 * lane its quad's sum, row_half_mirror (lane i <-> 7 - i) adds the other quad of the half row, row_mirror (i <-> 15 - i)
 * the other half.  Uniform control flow only (a disabled lane contributes nothing and receives nothing). */
template <int CTRL>
__device__ __forceinline__ uint32_t lfq_dpp_u32(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);
}

template <int LANES>        /* 4, 8 or 16 adjacent lanes, aligned: every lane of the group gets the group's sum */
__device__ __forceinline__ uint32_t lfq_group_sum_u32(uint32_t x)
{
    x += lfq_dpp_u32<0xB1>(x);
    x += lfq_dpp_u32<0x4E>(x);
    if (LANES >= 8) {
        x += lfq_dpp_u32<0x141>(x);
    }
    if (LANES >= 16) {
        x += lfq_dpp_u32<0x140>(x);
    }
    return x;
}

__device__ __forceinline__ uint32_t lfq_wave_sum_u32(uint32_t x)
{
    x = lfq_group_sum_u32<16>(x);                    /* the four rows, then their sums through the scalar unit */
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 0) + (uint32_t)__builtin_amdgcn_readlane((int)x, 16)
           + (uint32_t)__builtin_amdgcn_readlane((int)x, 32) + (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
}

/* the wavefront's sum in lane 63 (other lanes: partial sums): the four row sums, then row_bcast15 into rows 1 and 3 and
 * row_bcast31 into rows 2 and 3 -- six DPP adds, nothing through the scalar unit */
__device__ __forceinline__ uint32_t lfq_wave_sum_lane63_u32(uint32_t x)
{
    x = lfq_group_sum_u32<16>(x);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
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

/* The same evaluation without control flow: every filter is a predicate, every table entry is fetched
 * unconditionally (with a harmless index where the reference would not look), the result is selected; the
 * wave-uniform switches of LfqParams are turned into masks once (LfqEvalMasks) so that they cost an AND / OR per
 * observation instead of a scalar branch.  For kernels that evaluate one observation per LANE of different columns
 * (lfq_dp_screen_kernel): there the early returns of lfq_eval_obs become EXEC-mask regions that every wavefront
 * walks through anyway -- ~70 scalar instructions and both sides of every branch per observation.  Identical
 * values (the same operations on the kept path). */
struct LfqEvalMasks {
    uint32_t own, fixed, med;       /* all-ones on exactly one: an alt base's BQ is its own / def_alt_bq / the median */
    uint32_t fixed_q;
    uint32_t off_baq, off_mq, off_sq;   /* 255 where the track is switched off (index 255 = "missing"), else 0 */
    bool has_def_jp;
};

__device__ __forceinline__ LfqEvalMasks lfq_eval_masks(const LfqParams &P)
{
    LfqEvalMasks m;
    m.med = (P.def_alt_bq == -1) ? ~0u : 0u;
    m.fixed = (P.def_alt_bq > 0) ? ~0u : 0u;
    m.own = ~(m.med | m.fixed);
    m.fixed_q = (uint32_t)P.def_alt_bq & 255u;
    m.off_baq = P.use_baq ? 0u : 255u;
    m.off_mq = P.use_mq ? 0u : 255u;
    m.off_sq = P.use_sq ? 0u : 255u;
    m.has_def_jp = P.def_alt_jp >= 0.0;
    return m;
}

__device__ __forceinline__ LfqObs lfq_eval_obs_flat(uint32_t ntb, uint32_t bqb, uint32_t baqb, uint32_t mqb,
                                                    uint32_t sqb, int ref_code, int median_ref_bq,
                                                    const LfqParams &P, const LfqEvalMasks &M, const LfqLuts *L)
{
    const uint32_t code = ntb & 7u;
    const bool valid = code <= 3u;
    const bool is_alt = valid & (code != (uint32_t)ref_code);
    const bool bq_ok = (int)bqb >= (is_alt ? P.min_alt_bq4 : P.min_bq4);    /* min_alt_bq4 = max(min_bq, min_alt_bq) */
    /* snpcaller.c:431-441: the quality that stands in for an alt base's own */
    const uint32_t alt_idx = (bqb & M.own) | (M.fixed_q & M.fixed) | ((uint32_t)median_ref_bq & 255u & M.med);
    const uint32_t bq_idx = is_alt ? alt_idx : bqb;
    double pb = L->bq[bq_idx];
    pb = (is_alt & (M.med != 0u) & (median_ref_bq < 0)) ? 0.0 : pb;
    const double pa = L->baq[baqb | M.off_baq];
    const double pm = L->mq[mqb | M.off_mq];
    const double ps = L->sq[sqb | M.off_sq];
    const double om = 1.0 - pm, os = 1.0 - ps, oa = 1.0 - pa;
    double jp = pm + om * ps + om * os * pa + om * os * oa * pb;              /* snpcaller.c:334 */
    const bool jq_bad = (jp > P.jq_reject_above) | (is_alt & (jp > P.alt_jq_reject_above));
    jp = (is_alt & M.has_def_jp) ? P.def_alt_jp : jp;
    LfqObs o;
    o.keep = valid & bq_ok & !jq_bad;
    o.is_alt = is_alt;
    o.p = jp;
    return o;
}

#endif
