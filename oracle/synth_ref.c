/*
 * synth_ref.c -- host (CPU) generator of the synthetic pileup workload.
 * TEST / BENCH INFRASTRUCTURE (cpu_baseline sample and parity inputs); the spec itself
 * lives in include/lofreq_synth.h and is shared with the device generator.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "lofreq_synth.h"

void orc_synth_init_spec(lfq_synth_spec *s, uint64_t seed, uint32_t depth, uint32_t plant_period)
{
    int q;
    memset(s, 0, sizeof(*s));
    s->seed = seed;
    s->depth = depth;
    s->plant_period = plant_period;
    for (q = 0; q < 64; q++) {
        long double p = powl(10.0L, -(long double)q / 10.0L);
        long double t = floorl(p * 18446744073709551616.0L);
        s->err_thresh[q] = (t >= 18446744073709551615.0L) ? UINT64_MAX : (uint64_t)t;
    }
}

/* fills columns [col_begin, col_begin+ncols): tracks of ncols*depth bytes each */
void orc_synth_fill(const lfq_synth_spec *s, int64_t col_begin, int64_t ncols, uint8_t *nt,
                    uint8_t *bq, uint8_t *baq, uint8_t *mq, uint64_t *col_off, uint8_t *ref_base)
{
    static const char acgt[4] = {'A', 'C', 'G', 'T'};
    int64_t c;
    uint64_t i;
    for (c = 0; c < ncols; c++) {
        uint64_t col = (uint64_t)(col_begin + c);
        uint64_t o = (uint64_t)c * s->depth;
        col_off[c] = o;
        ref_base[c] = (uint8_t)acgt[lfq_synth_ref_code(col)];
        for (i = 0; i < s->depth; i++) {
            lfq_synth_obs ob = lfq_synth_observation(s, col, i);
            nt[o + i] = ob.nt;
            bq[o + i] = ob.bq;
            baq[o + i] = ob.baq;
            mq[o + i] = ob.mq;
        }
    }
    col_off[ncols] = (uint64_t)ncols * s->depth;
}
