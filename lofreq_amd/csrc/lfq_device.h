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

#endif
