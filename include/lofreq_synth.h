/*
 * lofreq_synth.h -- specification of the synthetic pileup workload.
 *
 * A pure function (seed, column, observation index) -> one packed observation, written with
 * integer arithmetic only so that the host C build (oracle/synth_ref.c, used for the CPU
 * baseline and parity samples) and the device build (lofreq_amd/csrc/lfq_synth.hip, used to
 * fill HBM for the benchmark) produce byte-identical tracks.
 *
 * Distribution (per observation): BQ ~ round(35 + 5 z) clipped to [2,41], z = Irwin-Hall(12)-6;
 * MQ = 60 w.p. 0.95 else U{0..59}; BAQ = 93 w.p. 0.9 else U{20..92}; strand alternates;
 * base = reference unless a sequencing error (probability 10^(-BQ/10), thresholds supplied by
 * the host) picks one of the three alternatives uniformly.  Every `plant_period`-th column
 * carries a planted SNV whose allele frequency cycles through 0.5 %, 1 %, 5 %, 50 %.
 * Reference base of column c is "ACGT"[c & 3].
 */
#ifndef LOFREQ_SYNTH_H
#define LOFREQ_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define LFQ_SYNTH_FN __host__ __device__ static inline
#else
#define LFQ_SYNTH_FN static inline
#endif

typedef struct lfq_synth_spec {
    uint64_t seed;
    uint32_t depth;          /* observations per column */
    uint32_t plant_period;   /* 0 = no planted variants; SURVEY 8d uses 997 */
    uint64_t err_thresh[64]; /* floor(10^(-q/10) * 2^64), q = 0..63, filled by lfq_synth_init_spec */
} lfq_synth_spec;

typedef struct lfq_synth_obs {
    uint8_t nt;  /* bits 0..2 nt4 code, bit 3 reverse strand */
    uint8_t bq, baq, mq;
} lfq_synth_obs;

/* splitmix64 finaliser */
LFQ_SYNTH_FN uint64_t lfq_synth_mix(uint64_t x)
{
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}

LFQ_SYNTH_FN uint8_t lfq_synth_ref_code(uint64_t col) { return (uint8_t)(col & 3); }

/* planted allele-frequency thresholds as fractions of 2^64: 0.5 %, 1 %, 5 %, 50 % */
LFQ_SYNTH_FN uint64_t lfq_synth_af_thresh(uint32_t which)
{
    switch (which & 3u) {
    case 0: return 0x0147AE147AE147AEULL;   /* 0.005 */
    case 1: return 0x028F5C28F5C28F5CULL;   /* 0.01  */
    case 2: return 0x0CCCCCCCCCCCCCCCULL;   /* 0.05  */
    default: return 0x8000000000000000ULL;  /* 0.5   */
    }
}

LFQ_SYNTH_FN lfq_synth_obs lfq_synth_observation(const lfq_synth_spec *s, uint64_t col, uint64_t i)
{
    const uint64_t G = 0x9E3779B97F4A7C15ULL;
    uint64_t base = lfq_synth_mix(s->seed + col * G) ^ (i * 0xD1B54A32D192ED03ULL);
    uint64_t r0 = lfq_synth_mix(base + 1 * G), r1 = lfq_synth_mix(base + 2 * G);
    uint64_t r2 = lfq_synth_mix(base + 3 * G), r3 = lfq_synth_mix(base + 4 * G);
    uint64_t r4 = lfq_synth_mix(base + 5 * G), r5 = lfq_synth_mix(base + 6 * G);
    uint64_t r6 = lfq_synth_mix(base + 7 * G);
    lfq_synth_obs o;
    uint32_t sum = 0, k;
    int32_t bq;
    uint32_t ref = lfq_synth_ref_code(col), code = ref;

    /* twelve 16-bit uniforms -> Irwin-Hall; bq = round(35 + 5 * (sum/65535 - 6)) */
    for (k = 0; k < 4; k++) {
        sum += (uint32_t)((r0 >> (16 * k)) & 0xFFFF) + (uint32_t)((r1 >> (16 * k)) & 0xFFFF)
               + (uint32_t)((r2 >> (16 * k)) & 0xFFFF);
    }
    /* (35 + 5*(sum - 393210)/65535) rounded half up, all in integers */
    bq = (int32_t)((2 * 35 * 65535 + 10 * ((int64_t)sum - 393210) + 65535) / (2 * 65535));
    if (bq < 2) bq = 2;
    if (bq > 41) bq = 41;
    o.bq = (uint8_t)bq;

    o.mq = ((uint32_t)r3 < 4080218931u) ? 60 : (uint8_t)((r3 >> 32) % 60);
    o.baq = ((uint32_t)r4 < 3865470566u) ? 93 : (uint8_t)(20 + (r4 >> 32) % 73);

    if (s->plant_period && (col % s->plant_period) == 0) {
        uint64_t pidx = col / s->plant_period;
        if (r6 < lfq_synth_af_thresh((uint32_t)pidx)) {
            uint32_t a = (uint32_t)(pidx % 3);          /* planted alt = a-th non-ref code */
            code = (a >= ref) ? a + 1 : a;
        }
    }
    if (code == ref && r5 < s->err_thresh[bq]) {        /* sequencing error */
        uint32_t a = (uint32_t)((r6 >> 11) % 3);
        code = (a >= ref) ? a + 1 : a;
    }
    o.nt = (uint8_t)(code | ((i & 1) ? 8u : 0u));
    return o;
}

#endif
