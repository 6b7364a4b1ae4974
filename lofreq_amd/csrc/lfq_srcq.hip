/*
 * lfq_srcq.hip -- source quality of a batch of reads: source_qual (plp.c:427-593) over
 * count_cigar_ops (samutils.c:437-614), the per-read pre-step behind `lofreq call -s`.
 *
 * One read per wavefront:
 *   count   the 64 lanes stride over every M / X operation of the CIGAR, compare read and reference letters and
 *           histogram the base qualities of all counted operations in LDS (256 bins: the error probabilities are
 *           pow(10, -q/10) of an integer, so the reference's qsort by probability IS a counting sort by quality);
 *           an insertion / deletion is one operation of quality 45.
 *   DP      K = non-matches - 1.  The Poisson-binomial recurrence of pruned_calc_prob_dist (snpcaller.c:831-972)
 *           over the rows in ascending probability (bins 255 .. 0), cells 0..K double-buffered in LDS (K < 768)
 *           or in an HBM scratch slice of the wavefront (longer reads), lane k handles cells k, k + 64, ...
 *           Unlike in the SNV path the pruning rule is part of the RESULT here: source_qual reads cell K - 1 of
 *           whatever row poissbin stopped at (bonf 1, sig 0.05), so the row loop stops exactly where the
 *           reference's does (first row n > K whose tail exceeds 0.05).
 * The cells are plain doubles (linear space; the reference works in logs): anything that underflows is below
 * 1e-308 and the result only matters above 1e-17 (src_qual = (int)(-10 log10l(1 - P(X = K-1)))).  The device
 * hands back P(X = K-1); the host applies the long-double phred conversion (lfq_api.hip).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lfq_internal.h"
#include "lfq_device.h"

#define LFQ_SRCQ_WAVES 4
#define LFQ_SRCQ_INDEL_QUAL 45              /* INDEL_QUAL_DEFAULT, samutils.c:51 */

__device__ __forceinline__ void lfq_srcq_sync()
{
    /* orders this wavefront's LDS / scratch writes before the other lanes' reads */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

/* rows in ascending probability until poissbin's early exit; returns cell K - 1 of the row it stopped at */
template <typename P>
__device__ __forceinline__ double lfq_srcq_dp(P prev, P cur, const int *hist, int K, int lane, const LfqLuts *__restrict__ luts)
{
    for (int k = lane; k <= K; k += 64) {
        prev[k] = k == 0 ? 1.0 : 0.0;
        cur[k] = 0.0;
    }
    lfq_srcq_sync();
    int n = 0;
    for (int q = 255; q >= 0; --q) {
        const int cnt = hist[q];
        if (cnt == 0) {
            continue;
        }
        const double p = luts->bq[q];
        const double ps = fabs(p) < LFQ_DBL_EPSILON ? LFQ_DBL_EPSILON : p;                 /* snpcaller.c:872-876 */
        const double pf = fabs(p - 1.0) < LFQ_DBL_EPSILON ? LFQ_DBL_EPSILON : 1.0 - p;     /* :877-881 */
        for (int i = 0; i < cnt; ++i) {
            n++;
            const int top = n < K ? n : K;
            for (int k = lane; k <= top; k += 64) {
                const double a = prev[k], b = k > 0 ? prev[k - 1] : 0.0;
                cur[k] = k == K ? fma(b, ps, a) : fma(a, pf, b * ps);                     /* :892-899, 912-922 */
            }
            lfq_srcq_sync();
            P t = prev;
            prev = cur;
            cur = t;
            if (n > K && prev[K] > 0.05) {                              /* early exit, :950-957 (bonf 1, sig 0.05) */
                return prev[K - 1];
            }
        }
    }
    return prev[K - 1];
}

__global__ __launch_bounds__(LFQ_SRCQ_WAVES * 64) void lfq_srcq_kernel(LfqSrcqArgs A, const LfqLuts *__restrict__ luts)
{
    __shared__ int s_hist[LFQ_SRCQ_WAVES][256];
    __shared__ double s_cells[LFQ_SRCQ_WAVES][2][LFQ_SRCQ_LDS_CELLS];
    const int wave = (int)(threadIdx.x >> 6), lane = lfq_lane();
    const int64_t gw = (int64_t)blockIdx.x * LFQ_SRCQ_WAVES + wave, n_waves = (int64_t)gridDim.x * LFQ_SRCQ_WAVES;
    int *hist = s_hist[wave];
    for (int64_t r = gw; r < A.n_reads; r += n_waves) {
        for (int q = lane; q < 256; q += 64) {
            hist[q] = 0;
        }
        lfq_srcq_sync();
        const uint32_t *cg = A.cigar + A.cigar_off[r];
        const int n_cigar = (int)(A.cigar_off[r + 1] - A.cigar_off[r]);
        const uint8_t *seq = A.seq + A.seq_off[r], *qual = A.qual + A.seq_off[r];
        int64_t tpos = A.pos[r];
        int qpos = 0;
        uint32_t n_ops = 0, n_non = 0;          /* per lane, summed below */
        for (int k = 0; k < n_cigar; ++k) {
            const int op = (int)(cg[k] & 0xfu), l = (int)(cg[k] >> 4);
            if (op == 0 || op == 8) {                                   /* samutils.c:481-531 */
                for (int j = lane; j < l; j += 64) {
                    const int64_t t = tpos + j;
                    const int bq = qual[qpos + j];
                    const char ref_nt = (t >= 0 && t < A.ref_len) ? A.ref[t] : '\0';
                    const bool mism = (ref_nt != lfq_seq_letter(seq[qpos + j])) || op == 8;       /* letters, samutils.c:486-489 */
                    if (bq < A.min_bq) {
                        continue;
                    }
                    if (mism && A.ign && t >= 0 && t < A.ref_len && A.ign[t]) {
                        continue;
                    }
                    atomicAdd(&hist[A.nonmatch_qual >= 0 ? A.nonmatch_qual : bq], 1);
                    n_ops++;
                    n_non += mism ? 1u : 0u;
                }
                tpos += l;
                qpos += l;
            } else if (op == 1 || op == 2) {                            /* :533-579 */
                const int64_t v = op == 1 ? tpos - 1 : tpos;
                const bool ignored = A.ign && v >= 0 && v < A.ref_len && A.ign[v];
                if (!ignored && lane == 0) {
                    atomicAdd(&hist[A.nonmatch_qual >= 0 ? A.nonmatch_qual : LFQ_SRCQ_INDEL_QUAL], 1);
                    n_ops++;
                    n_non++;
                }
                if (op == 1) {
                    qpos += l;
                } else if (!ignored) {
                    tpos += l;                                          /* an ignored deletion leaves tpos alone (:547-555) */
                }
            } else if (op == 3) {
                tpos += l;
            } else if (op == 4) {
                qpos += l;
            }                                                           /* H, P, =: nothing moves (:590-593) */
        }
        const int n_ep = (int)lfq_wave_sum_u32(n_ops);
        int K = (int)lfq_wave_sum_u32(n_non);
        lfq_srcq_sync();
        if (n_ep < 1) {                                                 /* plp.c:468-474 */
            if (lane == 0) {
                A.status[r] = LFQ_SRCQ_NA;
                A.prob[r] = 0.0;
            }
            continue;
        }
        K = K > 0 ? K - 1 : 0;                                          /* :514-516 */
        if (K == 0) {                                                   /* :517-524 */
            if (lane == 0) {
                A.status[r] = LFQ_SRCQ_PERFECT;
                A.prob[r] = 0.0;
            }
            continue;
        }
        double res;
        if (K < LFQ_SRCQ_LDS_CELLS) {       /* two calls: the address space of the cells is static in each */
            res = lfq_srcq_dp(s_cells[wave][0], s_cells[wave][1], hist, K, lane, luts);
        } else {
            double *g = A.scratch + (size_t)gw * 2 * (size_t)A.scratch_cells;
            res = lfq_srcq_dp(g, g + A.scratch_cells, hist, K, lane, luts);
        }
        if (lane == 0) {
            A.status[r] = LFQ_SRCQ_VALUE;
            A.prob[r] = res;
        }
        lfq_srcq_sync();
    }
}

int lfq_launch_srcq(const LfqSrcqArgs &a, const LfqLuts *d_luts, int n_blocks, void *stream)
{
    if (a.n_reads <= 0) {
        return LFQ_OK;
    }
    hipLaunchKernelGGL(lfq_srcq_kernel, dim3((unsigned)n_blocks), dim3(LFQ_SRCQ_WAVES * 64), 0, (hipStream_t)stream, a,
                       d_luts);
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}
