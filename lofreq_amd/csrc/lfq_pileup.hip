/*
 * lfq_pileup.hip -- reads -> packed pileup columns on the device: the SNV tracks that
 * compile_plp_col (plp.c:797-1017) builds per column on the CPU, for a batch of reads of one region.
 *
 * Two passes of one thread per read over its CIGAR:
 *   count    coverage_plp (every alignment overlapping the column, deletions and reference skips included:
 *            n_plp of mpileup) and num_bases (bases that enter the arrays: not deleted / skipped, BQ >= min_plp_bq,
 *            plp.c:937-941, 1019-1022) per reference position, with atomics;
 *   scatter  after a prefix sum over the covered positions, every kept base goes to a slot of its column
 *            (atomic cursor): nt4 code | strand, BQ capped at 93 (plp.c:948-952), BAQ from the lb tag (255 =
 *            missing, plp.c:956-962), MAPQ.
 * The order of the observations inside a column is arbitrary; nothing downstream depends on it (the counts are
 * sums, the Poisson-binomial recurrence is order-independent).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lfq_internal.h"

__global__ __launch_bounds__(256) void lfq_pileup_count_kernel(LfqPileupArgs A)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.n_reads) {
        return;
    }
    const uint32_t *cg = A.cigar + A.cigar_off[r];
    const int n_cigar = (int)(A.cigar_off[r + 1] - A.cigar_off[r]);
    const uint8_t *qual = A.qual + A.seq_off[r];
    int64_t x = A.pos[r];
    int y = 0;
    for (int k = 0; k < n_cigar; ++k) {
        const int op = cg[k] & 0xf, l = cg[k] >> 4;
        if (op == 0 || op == 7 || op == 8) {
            for (int j = 0; j < l; j++) {
                const int64_t c = x + j - A.begin;
                if (c >= 0 && c < A.width) {
                    atomicAdd(&A.cov[c], 1);
                    if ((int)qual[y + j] >= A.min_plp_bq) {
                        atomicAdd(&A.nb[c], 1);
                    }
                }
            }
            x += l; y += l;
        } else if (op == 2 || op == 3) {            /* is_del / is_refskip: part of n_plp, no base */
            for (int j = 0; j < l; j++) {
                const int64_t c = x + j - A.begin;
                if (c >= 0 && c < A.width) {
                    atomicAdd(&A.cov[c], 1);
                }
            }
            x += l;
        } else if (op == 1 || op == 4) {
            y += l;
        }
    }
}

__global__ __launch_bounds__(256) void lfq_pileup_scatter_kernel(LfqPileupArgs A)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.n_reads) {
        return;
    }
    const uint32_t *cg = A.cigar + A.cigar_off[r];
    const int n_cigar = (int)(A.cigar_off[r + 1] - A.cigar_off[r]);
    const int64_t s0 = A.seq_off[r];
    const uint8_t *seq = A.seq + s0, *qual = A.qual + s0, *lb = A.baq ? A.baq + s0 : nullptr;
    const uint32_t strand = A.reverse[r] ? 8u : 0u, mq = A.mapq[r];
    int64_t x = A.pos[r];
    int y = 0;
    for (int k = 0; k < n_cigar; ++k) {
        const int op = cg[k] & 0xf, l = cg[k] >> 4;
        if (op == 0 || op == 7 || op == 8) {
            for (int j = 0; j < l; j++) {
                const int64_t c = x + j - A.begin;
                const int bq = qual[y + j];
                if (c >= 0 && c < A.width && bq >= A.min_plp_bq) {
                    const int ci = A.col_index[c];
                    const uint64_t slot = A.col_off[ci] + (uint64_t)atomicAdd(&A.cursor[c], 1);
                    A.t_nt[slot] = (uint8_t)((seq[y + j] > 4 ? 4 : seq[y + j]) | strand);
                    A.t_bq[slot] = (uint8_t)(bq > 93 ? 93 : bq);                       /* plp.c:948-952 */
                    A.t_baq[slot] = lb ? (uint8_t)(lb[y + j] >= 33 ? lb[y + j] - 33 : 255) : (uint8_t)255;
                    A.t_mq[slot] = (uint8_t)mq;
                    if (A.t_sq) {
                        A.t_sq[slot] = A.sq[r];                                         /* plp.c:975-977 */
                    }
                }
            }
            x += l; y += l;
        } else if (op == 2 || op == 3) {
            x += l;
        } else if (op == 1 || op == 4) {
            y += l;
        }
    }
}

int lfq_launch_pileup_count(const LfqPileupArgs &a, void *stream)
{
    if (a.n_reads <= 0) {
        return LFQ_OK;
    }
    hipLaunchKernelGGL(lfq_pileup_count_kernel, dim3((unsigned)((a.n_reads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}

int lfq_launch_pileup_scatter(const LfqPileupArgs &a, void *stream)
{
    if (a.n_reads <= 0) {
        return LFQ_OK;
    }
    hipLaunchKernelGGL(lfq_pileup_scatter_kernel, dim3((unsigned)((a.n_reads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}
