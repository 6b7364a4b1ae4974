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
#include "lfq_device.h"

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


/* ---- indel fields of the pileup (plp.c:1019-1192) ------------------------------------------------------------
 * Every pileup entry of a read (a base of an M/=/X operation, or a position inside a D/N operation) carries the
 * read's BI / BD quality at the entry's query position and, at the last position of an operation, the indel
 * that follows (htslib resolve_cigar2: +len for I, -len for D).  Entries passing min_plp_idq are counted
 * (num_non_indels / num_ins / num_dels, the strand counts of the reads without an insertion resp. deletion) and
 * their (quality, MAPQ) go to the column's ins_quals / del_quals arrays -- those are only materialised for
 * positions that have an event (the host knows them from the CIGARs), the only ones call_indels reads. */
template <bool SCATTER>
__global__ __launch_bounds__(256) void lfq_plp_indel_kernel(LfqPlpIndelArgs A)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.n_reads) {
        return;
    }
    const uint32_t *cg = A.cigar + A.cigar_off[r];
    const int n_cigar = (int)(A.cigar_off[r + 1] - A.cigar_off[r]);
    const int64_t s0 = A.seq_off[r];
    const int l_qseq = (int)(A.seq_off[r + 1] - s0);
    const uint32_t fl = A.tag_flags ? A.tag_flags[r] : 3u;
    const uint8_t *bi = (A.bi && (fl & 1u)) ? A.bi + s0 : nullptr, *bd = (A.bd && (fl & 2u)) ? A.bd + s0 : nullptr;
    const int rev = A.reverse[r] ? 1 : 0;
    const int16_t mq = (int16_t)A.mapq[r];
    int64_t end = A.pos[r];                                 /* bam_endpos - 1: is_tail */
    for (int k = 0; k < n_cigar; ++k) {
        const int op = cg[k] & 0xf;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) {
            end += cg[k] >> 4;
        }
    }
    end -= 1;
    int64_t x = A.pos[r];
    int y = 0;
    for (int k = 0; k < n_cigar; ++k) {
        const int op = cg[k] & 0xf, l = cg[k] >> 4;
        if (op == 0 || op == 7 || op == 8 || op == 2 || op == 3) {
            const bool is_del = op == 2 || op == 3;
            int indel_last = 0;                             /* the peek of resolve_cigar2 at the operation's last position */
            if (k + 1 < n_cigar) {
                const int op2 = cg[k + 1] & 0xf, l2 = cg[k + 1] >> 4;
                if (op2 == 2) {
                    indel_last = -l2;
                } else if (op2 == 1) {
                    indel_last = l2;
                } else if (op2 == 6 && k + 2 < n_cigar) {
                    int l3 = 0;
                    for (int kk = k + 2; kk < n_cigar; ++kk) {
                        const int o3 = cg[kk] & 0xf;
                        if (o3 == 1) {
                            l3 += cg[kk] >> 4;
                        } else if (o3 == 2 || o3 == 0 || o3 == 3 || o3 == 7 || o3 == 8) {
                            break;
                        }
                    }
                    if (l3 > 0) {
                        indel_last = l3;
                    }
                }
            }
            for (int j = 0; j < l; j++) {
                const int64_t c = x + j - A.begin;
                if (c < 0 || c >= A.width) {
                    continue;
                }
                int qpos = is_del ? y : y + j;
                qpos = qpos < l_qseq ? qpos : l_qseq - 1;
                const int iq = (bi && qpos >= 0) ? (int)bi[qpos] - 33 : 0, dq = (bd && qpos >= 0) ? (int)bd[qpos] - 33 : 0;   /* plp.c:1023-1059 */
                const int indel = (j == l - 1) ? indel_last : 0;
                const bool pass = !(iq < A.min_plp_idq || dq < A.min_plp_idq);                    /* :1062 */
                if (!SCATTER) {
                    atomicAdd(&A.cov[c], 1);
                    if (!is_del && x + j == end) {
                        atomicAdd(&A.tails[c], 1);                                              /* :920-922 */
                    }
                    if (pass) {
                        if (indel > 0) {
                            atomicAdd(&A.n_ins[c], 1);
                            atomicAdd(&A.non_del_fw[c], rev ? 0 : 1);                           /* :1100-1104 */
                        } else if (indel < 0) {
                            atomicAdd(&A.n_dels[c], 1);
                            atomicAdd(&A.non_ins_fw[c], rev ? 0 : 1);                           /* :1151-1155 */
                        } else {
                            atomicAdd(&A.non_indels[c], 1);
                            atomicAdd(&A.non_ins_fw[c], rev ? 0 : 1);
                            atomicAdd(&A.non_del_fw[c], rev ? 0 : 1);
                        }
                    }
                } else if (pass) {
                    if (indel <= 0 && A.ne_off[0][c] >= 0) {            /* no insertion here: ins_quals (:1147, 1162) */
                        const int64_t slot = A.ne_off[0][c] + atomicAdd(&A.cursor[0][c], 1);
                        A.ne_q[0][slot] = (int16_t)iq;
                        A.ne_mq[0][slot] = mq;
                    }
                    if (indel >= 0 && A.ne_off[1][c] >= 0) {            /* no deletion here: del_quals (:1096, 1173) */
                        const int64_t slot = A.ne_off[1][c] + atomicAdd(&A.cursor[1][c], 1);
                        A.ne_q[1][slot] = (int16_t)dq;
                        A.ne_mq[1][slot] = mq;
                    }
                }
            }
            x += l;
            if (!is_del) {
                y += l;
            }
        } else if (op == 1 || op == 4) {
            y += l;
        }
    }
}

int lfq_launch_plp_indel(const LfqPlpIndelArgs &a, int scatter, void *stream)
{
    if (a.n_reads <= 0) {
        return LFQ_OK;
    }
    const dim3 grid((unsigned)((a.n_reads + 255) / 256)), block(256);
    if (scatter) {
        hipLaunchKernelGGL(lfq_plp_indel_kernel<true>, grid, block, 0, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL(lfq_plp_indel_kernel<false>, grid, block, 0, (hipStream_t)stream, a);
    }
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}


/* two byte arrays at a list of positions (the ai / ad qualities of the reads that carry an indel event) */
__global__ __launch_bounds__(256) void lfq_gather2_kernel(const uint8_t *a, const uint8_t *b, const int64_t *idx, int64_t n,
                                                          uint8_t *oa, uint8_t *ob)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        oa[i] = a[idx[i]];
        ob[i] = b[idx[i]];
    }
}

int lfq_launch_gather2(const uint8_t *a, const uint8_t *b, const int64_t *idx, int64_t n, uint8_t *oa, uint8_t *ob, void *stream)
{
    if (n <= 0) {
        return LFQ_OK;
    }
    hipLaunchKernelGGL(lfq_gather2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, idx, n, oa, ob);
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}


/* ---- column-major pileup for position-sorted reads (what mpileup requires anyway) ----------------------------
 * One wavefront per reference position: the reads that can overlap it are a window of the sorted read list --
 * from the first read whose running maximum of end coordinates exceeds the position to the last read starting at
 * or before it (two binary searches).  The lanes walk the window 64 reads at a time, each lane resolving its read's
 * CIGAR at the position (resolve_cigar2: a base, a deleted / skipped position, or no overlap).  Pass 0 counts
 * coverage_plp and num_bases with wave reductions; pass 1 writes the kept bases to the column's slice at
 * base + rank (ballot + mbcnt): no atomics, coalesced stores, and the observations of a column come out in pileup
 * order.  The read bytes are fetched from L2: neighbouring positions share their window. */
__device__ __forceinline__ int lfq_plp_locate(const uint32_t *cg, int n_cigar, int64_t x, int64_t p, int *qpos)
{
    int y = 0;                                   /* -> 1: base at *qpos, 2: deleted / skipped, 0: no overlap */
    for (int k = 0; k < n_cigar; ++k) {
        const int op = cg[k] & 0xf, l = cg[k] >> 4;
        if (op == 0 || op == 7 || op == 8) {
            if (p < x + l) {
                *qpos = y + (int)(p - x);
                return p >= x ? 1 : 0;
            }
            x += l; y += l;
        } else if (op == 2 || op == 3) {
            if (p < x + l) {
                return p >= x ? 2 : 0;
            }
            x += l;
        } else if (op == 1 || op == 4) {
            y += l;
        }
    }
    return 0;
}

/* First index in [lo, hi) whose value exceeds p, hi if there is none; `a` is non-decreasing.  The wavefront searches
 * together: 64 probes cut the range into 65 parts per round, so 2 M reads take 4 dependent loads instead of the 21 of
 * a bisection -- the two searches were 42 of the ~70 dependent memory round trips a column costs this kernel, and
 * those round trips are all it waits for.  Wave-uniform arguments and result. */
__device__ __forceinline__ int64_t lfq_wave_first_above(const int32_t *__restrict__ a, int64_t lo, int64_t hi, int64_t p,
                                                        int lane)
{
    for (;;) {
        const int64_t n = hi - lo;
        if (n <= 64) {
            const int64_t i = lo + lane;
            const uint64_t m = __ballot(i < hi && (int64_t)a[i] > p);
            return m ? lo + (int64_t)__builtin_ctzll(m) : hi;
        }
        /* probe j sits at lo + n (j + 1) / 65: strictly increasing for n >= 65, inside (lo, hi) */
        const uint64_t m = __ballot((int64_t)a[lo + n * (lane + 1) / 65] > p);
        const int k = m ? (int)__builtin_ctzll(m) : 64;           /* first probe above p: the answer is in (probe k-1, probe k] */
        const int64_t nlo = (k == 0) ? lo : lo + n * k / 65 + 1;
        const int64_t nhi = (k == 64) ? hi : lo + n * (k + 1) / 65;
        lo = nlo;
        hi = nhi;
    }
}

template <bool SCATTER>
__global__ __launch_bounds__(256) void lfq_pileup_columns_kernel(LfqPileupArgs A)
{
    const int lane = (int)(threadIdx.x & 63u);
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= A.width) {
        return;
    }
    const int64_t p = A.begin + c;
    int ci = 0;
    uint64_t base = 0;
    if (SCATTER) {
        ci = A.col_index[c];
        if (ci < 0) {
            return;
        }
        base = A.col_off[ci];
    }
    /* window [lo, hi): lo = first read with pmax_end > p, hi = first read with pos > p (wave-uniform searches) */
    const int64_t lo = lfq_wave_first_above(A.pmax_end, 0, A.n_reads, p, lane);
    const int64_t hi = lfq_wave_first_above(A.pos, lo, A.n_reads, p, lane);
    uint32_t n_cov = 0, n_kept = 0;
    for (int64_t r0 = lo; r0 < hi; r0 += 64) {
        const int64_t r = r0 + lane;
        int kind = 0, qpos = 0;
        if (r < hi) {
            const int64_t co = A.cigar_off[r];
            kind = lfq_plp_locate(A.cigar + co, (int)(A.cigar_off[r + 1] - co), A.pos[r], p, &qpos);
        }
        int bq = 0;
        int64_t s0 = 0;
        if (kind == 1) {
            s0 = A.seq_off[r];
            bq = A.qual[s0 + qpos];
        }
        const bool kept = kind == 1 && bq >= A.min_plp_bq;
        const uint64_t mk = __ballot(kept);
        if (!SCATTER) {
            n_cov += (uint32_t)__popcll(__ballot(kind != 0));
            n_kept += (uint32_t)__popcll(mk);
        } else {
            if (kept) {
                const uint64_t slot = base + n_kept + (uint64_t)__popcll(mk & ((1ull << lane) - 1ull));
                const uint32_t code = A.seq[s0 + qpos];
                A.t_nt[slot] = (uint8_t)((code > 4 ? 4u : code) | (A.reverse[r] ? 8u : 0u));
                A.t_bq[slot] = (uint8_t)(bq > 93 ? 93 : bq);                                   /* plp.c:948-952 */
                const uint32_t lb = A.baq ? A.baq[s0 + qpos] : 0u;
                A.t_baq[slot] = A.baq ? (uint8_t)(lb >= 33 ? lb - 33 : 255) : (uint8_t)255;
                A.t_mq[slot] = A.mapq[r];
                if (A.t_sq) {
                    A.t_sq[slot] = A.sq[r];
                }
            }
            n_kept += (uint32_t)__popcll(mk);
        }
    }
    if (!SCATTER && lane == 0) {
        A.cov[c] = (int32_t)n_cov;
        A.nb[c] = (int32_t)n_kept;
    }
}

/* ---- the same two passes, a TILE of 64 positions per workgroup -----------------------------------------------------
 * The column-major kernel above fetches the three per-base bytes of every (read, position) pair with byte gathers: 64
 * lanes = 64 reads = 64 cache lines per load instruction, and the texture path handles a line per cycle -- for the 3 x 10^8
 * pairs of a 1 Mb x 300x region that alone is 1.5 ms per CU, and the CIGAR of a read is resolved once per position it
 * covers.  Here a workgroup owns 64 consecutive positions.  Its reads -- the window from the first read that can reach the
 * tile to the last one that starts inside it, 256 at a time -- are resolved ONCE against the tile (phase A, a read per
 * thread): which of the 64 positions it covers with a base, which with a deleted / skipped position (two 64-bit masks),
 * and the stretch of the query those positions map to as aligned 16-byte words into LDS -- up to six loads per track
 * instead of 64 byte loads.  One indel inside the tile is part of this (two stretches: the row holds the query range from
 * the first base to the last, and the positions behind the indel use a second offset); two or more, or an insertion too
 * long for the row, are resolved per position as before (one pair in 10^4).  Phase B is the column-major kernel's inner
 * loop (a wavefront per position, lanes = reads in read order, ballot + rank), reading its bytes from LDS: no load is
 * issued between its stores, so nothing waits for them (vmcnt counts loads and stores in one order).  Same order of the
 * observations, same bytes. */
#define LFQ_TILE 64
#define LFQ_TILE_ROW 100        /* bytes of LDS per read and track: 6 aligned 16-byte words + 4 (25 dwords: conflict-free rows) */

/* what phase A finds out about one read against one tile */
struct LfqTileRead {
    unsigned long long cm, dm;  /* positions covered with a base / with a deleted or skipped position */
    int nseg;                   /* stretches of the query the covered positions map to */
    int off0, off1, split;      /* row byte of position p0 + dp: off0 + dp below `split`, off1 + dp from there on */
    int nw;                     /* aligned 16-byte words to fetch per track (0: none, or position by position) */
    int64_t a0;                 /* index of the first of them in the per-base arrays */
    bool slow;                  /* two indels inside the tile, or an insertion too long for the row */
};

__device__ __forceinline__ LfqTileRead lfq_tile_resolve(const uint32_t *cg, int nc, int64_t x, int64_t s0, int64_t p0, int tile)
{
    LfqTileRead R;
    R.cm = R.dm = 0;
    R.off0 = R.off1 = 0;
    R.split = 64;
    R.nw = 0;
    R.a0 = 0;
    R.slow = false;
    int y = 0, nseg = 0, q1 = 0, p1 = 0, q2 = 0, p2 = 0, n2 = 0, n1 = 0;
    for (int k = 0; k < nc && x < p0 + tile; ++k) {
        const int op = cg[k] & 0xf, l = cg[k] >> 4;
        if (op == 0 || op == 7 || op == 8 || op == 2 || op == 3) {
            const int64_t a = x > p0 ? x : p0, b = (x + l < p0 + tile) ? x + l : p0 + tile;
            if (a < b) {
                const int n = (int)(b - a), sh = (int)(a - p0);
                const unsigned long long m = (n >= 64 ? ~0ull : ((1ull << n) - 1ull)) << sh;
                if (op == 2 || op == 3) {
                    R.dm |= m;
                } else {
                    const int q = y + (int)(a - x);
                    R.cm |= m;
                    if (nseg == 0) {
                        q1 = q; p1 = sh; n1 = n;
                        nseg = 1;
                    } else if (nseg == 1 && q == q1 + n1 && sh == p1 + n1) {
                        n1 += n;                                /* M next to = / X: the same stretch goes on */
                    } else if (nseg == 1) {
                        q2 = q; p2 = sh; n2 = n;
                        nseg = 2;
                    } else if (nseg == 2 && q == q2 + n2 && sh == p2 + n2) {
                        n2 += n;
                    } else {
                        nseg++;
                    }
                }
            }
            x += l;
            if (op != 2 && op != 3) {
                y += l;
            }
        } else if (op == 1 || op == 4) {
            y += l;
        }
    }
    R.nseg = nseg;
    if (nseg == 1 || nseg == 2) {
        const int64_t g0 = s0 + q1;
        R.a0 = g0 & ~(int64_t)15;
        const int span = nseg == 1 ? n1 : (q2 + n2 - q1);       /* query bytes from the first base to the last */
        R.nw = (int)((g0 - R.a0) + span + 15) >> 4;
        R.off0 = (int)(g0 - R.a0) - p1;
        R.off1 = nseg == 1 ? R.off0 : (int)(g0 - R.a0) + (q2 - q1) - p2;
        R.split = nseg == 1 ? 64 : p2;
    }
    if (nseg > 2 || R.nw > 6) {
        R.slow = true;
        R.nw = 0;
        R.off0 = R.off1 = 0;                                    /* the row is indexed by the position itself */
        R.split = 64;
    }
    return R;
}

/* six aligned words of one per-base array into a read's row (words past the stretch repeat its last word: no branch
 * around a load, and every load is issued before the first LDS write waits for one) */
__device__ __forceinline__ void lfq_tile_fetch(uint32_t *row, const uint8_t *src, int64_t a0, int nw)
{
    uint4 w[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        w[j] = *reinterpret_cast<const uint4 *>(src + a0 + 16 * (j < nw ? j : nw - 1));
    }
#pragma unroll
    for (int j = 0; j < 6; j++) {
        row[4 * j + 0] = w[j].x; row[4 * j + 1] = w[j].y; row[4 * j + 2] = w[j].z; row[4 * j + 3] = w[j].w;
    }
}

/* the same for two arrays at once */
__device__ __forceinline__ void lfq_tile_fetch2(uint32_t *row_a, const uint8_t *src_a, uint32_t *row_b, const uint8_t *src_b,
                                                int64_t a0, int nw)
{
    uint4 wa[6], wb[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const int64_t o = a0 + 16 * (j < nw ? j : nw - 1);
        wa[j] = *reinterpret_cast<const uint4 *>(src_a + o);
        wb[j] = *reinterpret_cast<const uint4 *>(src_b + o);
    }
#pragma unroll
    for (int j = 0; j < 6; j++) {
        row_a[4 * j + 0] = wa[j].x; row_a[4 * j + 1] = wa[j].y; row_a[4 * j + 2] = wa[j].z; row_a[4 * j + 3] = wa[j].w;
        row_b[4 * j + 0] = wb[j].x; row_b[4 * j + 1] = wb[j].y; row_b[4 * j + 2] = wb[j].z; row_b[4 * j + 3] = wb[j].w;
    }
}

/* Count pass: 256 reads per round, a read per thread.  Scatter pass: three tracks per read, so 128 reads per round (38 KB of
 * LDS: four workgroups per CU) with two threads per read -- one fetches the qualities and publishes the masks, the other the
 * bases and the BAQ bytes. */
template <bool SCATTER, int RC>
__global__ __launch_bounds__(256) void lfq_pileup_tiles_kernel(LfqPileupArgs A)
{
    constexpr int NTR = SCATTER ? 3 : 1;                            /* qual (+ seq, baq) */
    __shared__ uint32_t s_raw[NTR][RC][LFQ_TILE_ROW / 4];
    __shared__ unsigned long long s_cm[RC], s_dm[RC];
    __shared__ int16_t s_off0[RC], s_off1[RC];
    __shared__ uint8_t s_split[RC];
    __shared__ uint8_t s_mq[RC], s_rev[RC], s_sq[RC];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t c0 = (int64_t)blockIdx.x * LFQ_TILE;
    if (c0 >= A.width) {
        return;
    }
    const int64_t p0 = A.begin + c0;
    const int tile = (int)((A.width - c0 < LFQ_TILE) ? (A.width - c0) : LFQ_TILE);     /* positions of this tile */
    /* window [lo, hi): first read with pmax_end > p0 ... first read with pos > the tile's last position */
    const int64_t lo = lfq_wave_first_above(A.pmax_end, 0, A.n_reads, p0, lane);
    const int64_t hi = lfq_wave_first_above(A.pos, lo, A.n_reads, p0 + tile - 1, lane);
    /* the 16 positions of this wavefront: running counts, and for the scatter pass the column's slice */
    uint32_t n_cov[16], n_kept[16];
    uint64_t base[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        n_cov[i] = n_kept[i] = 0;
        base[i] = ~0ull;
    }
    if (SCATTER) {
        int ci = -1;
        unsigned long long b = ~0ull;
        if (lane < 16 && wave * 16 + lane < tile) {
            ci = A.col_index[c0 + wave * 16 + lane];
            if (ci >= 0) {
                b = A.col_off[ci];
            }
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            base[i] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(b >> 32), i) << 32)
                      | (uint32_t)__builtin_amdgcn_readlane((int)(b & 0xffffffffull), i);
        }
    }
    const int rt = tid & (RC - 1);                                  /* this thread's read of the round */
    const bool first = tid < RC;                                    /* fetches the qualities and publishes the masks; the other: bases and BAQ bytes */
    for (int64_t r0 = lo; r0 < hi; r0 += RC) {
        /* ---- phase A: every read of the round against the tile ---- */
        {
            const int64_t r = r0 + rt;
            LfqTileRead R;
            R.cm = R.dm = 0;
            R.off0 = R.off1 = 0;
            R.split = 64;
            if (r < hi) {
                const int64_t co = A.cigar_off[r], s0 = A.seq_off[r];
                const int nc = (int)(A.cigar_off[r + 1] - co);
                const uint32_t *cg = A.cigar + co;
                R = lfq_tile_resolve(cg, nc, A.pos[r], s0, p0, tile);
                if (R.nw > 0) {
                    if (first) {
                        lfq_tile_fetch(&s_raw[0][rt][0], A.qual, R.a0, R.nw);
                    } else if (A.baq) {
                        lfq_tile_fetch2(&s_raw[NTR > 1 ? 1 : 0][rt][0], A.seq, &s_raw[NTR > 2 ? 2 : 0][rt][0], A.baq, R.a0, R.nw);
                    } else {
                        lfq_tile_fetch(&s_raw[NTR > 1 ? 1 : 0][rt][0], A.seq, R.a0, R.nw);
                    }
                } else if (R.slow) {
                    /* position by position, here: phase B must not issue a load (see above) */
                    uint8_t *wq = reinterpret_cast<uint8_t *>(&s_raw[0][rt][0]);
                    uint8_t *ws = reinterpret_cast<uint8_t *>(&s_raw[NTR > 1 ? 1 : 0][rt][0]);
                    uint8_t *wb = reinterpret_cast<uint8_t *>(&s_raw[NTR > 2 ? 2 : 0][rt][0]);
                    for (int dp = 0; dp < tile; dp++) {
                        if ((R.cm >> dp) & 1ull) {
                            int qpos = 0;
                            (void)lfq_plp_locate(cg, nc, A.pos[r], p0 + dp, &qpos);
                            if (first) {
                                wq[dp] = A.qual[s0 + qpos];
                            } else {
                                ws[dp] = A.seq[s0 + qpos];
                                if (A.baq) {
                                    wb[dp] = A.baq[s0 + qpos];
                                }
                            }
                        }
                    }
                }
                if (SCATTER && first && R.cm) {
                    s_mq[rt] = A.mapq[r];
                    s_rev[rt] = A.reverse[r] ? 8 : 0;
                    s_sq[rt] = A.sq ? A.sq[r] : 0;
                }
            }
            if (first) {
                s_cm[rt] = R.cm;
                s_dm[rt] = R.dm;
                s_off0[rt] = (int16_t)R.off0;
                s_off1[rt] = (int16_t)R.off1;
                s_split[rt] = (uint8_t)R.split;
            }
        }
        __syncthreads();
        /* ---- phase B: a wavefront per position, lanes = the reads of a 64-read slice in read order ---- */
        const int n_here = (int)((hi - r0 < RC) ? (hi - r0) : RC);
        for (int sc = 0; sc * 64 < n_here; sc++) {
            const int t = sc * 64 + lane;
            const unsigned long long cm = s_cm[t], dm = s_dm[t];
            if (__ballot((cm | dm) != 0) == 0) {
                continue;
            }
            const int off0 = s_off0[t], off1 = s_off1[t], split = s_split[t];
            const uint8_t *rq = reinterpret_cast<const uint8_t *>(&s_raw[0][t][0]);
            const uint8_t *rs = reinterpret_cast<const uint8_t *>(&s_raw[NTR > 1 ? 1 : 0][t][0]);
            const uint8_t *rb = reinterpret_cast<const uint8_t *>(&s_raw[NTR > 2 ? 2 : 0][t][0]);
            uint32_t mq = 0, rev = 0, sq = 0;
            if (SCATTER) {
                mq = s_mq[t]; rev = s_rev[t]; sq = s_sq[t];
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int dp = wave * 16 + i;
                const bool covered = (cm >> dp) & 1ull;
                const bool any = covered || ((dm >> dp) & 1ull);
                const int off = dp < split ? off0 : off1;
                int bq = 0;
                if (covered) {
                    bq = rq[off + dp];
                }
                const bool kept = covered && bq >= A.min_plp_bq;
                const uint64_t mk = __ballot(kept);
                if (!SCATTER) {
                    n_cov[i] += (uint32_t)__popcll(__ballot(any));
                } else if (kept) {
                    const uint64_t slot = base[i] + n_kept[i] + (uint64_t)__popcll(mk & ((1ull << lane) - 1ull));
                    const uint32_t code = rs[off + dp];
                    const uint32_t lb = A.baq ? rb[off + dp] : 0u;
                    A.t_nt[slot] = (uint8_t)((code > 4 ? 4u : code) | rev);
                    A.t_bq[slot] = (uint8_t)(bq > 93 ? 93 : bq);                                   /* plp.c:948-952 */
                    A.t_baq[slot] = A.baq ? (uint8_t)(lb >= 33 ? lb - 33 : 255) : (uint8_t)255;
                    A.t_mq[slot] = (uint8_t)mq;
                    if (A.t_sq) {
                        A.t_sq[slot] = (uint8_t)sq;
                    }
                }
                n_kept[i] += (uint32_t)__popcll(mk);
            }
        }
        __syncthreads();
    }
    if (!SCATTER && lane < 16 && wave * 16 + lane < tile) {
        uint32_t cv = 0, kp = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            cv = (lane == i) ? n_cov[i] : cv;
            kp = (lane == i) ? n_kept[i] : kp;
        }
        A.cov[c0 + wave * 16 + lane] = (int32_t)cv;
        A.nb[c0 + wave * 16 + lane] = (int32_t)kp;
    }
}

int lfq_launch_pileup_columns(const LfqPileupArgs &a, int scatter, void *stream)
{
    if (a.n_reads > 0 && a.width > 0 && lfq_knobs().pileup_tiles) {
        const dim3 grid((unsigned)((a.width + LFQ_TILE - 1) / LFQ_TILE)), block(256);
        if (scatter) {
            hipLaunchKernelGGL((lfq_pileup_tiles_kernel<true, 128>), grid, block, 0, (hipStream_t)stream, a);
        } else {
            hipLaunchKernelGGL((lfq_pileup_tiles_kernel<false, 256>), grid, block, 0, (hipStream_t)stream, a);
        }
        return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
    }
    if (a.n_reads <= 0 || a.width <= 0) {
        return LFQ_OK;
    }
    const dim3 grid((unsigned)((a.width + 3) / 4)), block(256);
    if (scatter) {
        hipLaunchKernelGGL(lfq_pileup_columns_kernel<true>, grid, block, 0, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL(lfq_pileup_columns_kernel<false>, grid, block, 0, (hipStream_t)stream, a);
    }
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}


/* ---- the indel fields, column-major (position-sorted reads): same window walk as lfq_pileup_columns_kernel ------
 * Per read and position: the pileup entry (base or deleted / skipped position), its query position (htslib: the
 * next base for an entry inside a deletion), the indel that follows if this is the last position of its CIGAR
 * operation, whether it is the read's last aligned position (is_tail).  Pass 0: the seven per-position counts by
 * ballots; pass 1: the (quality, MAPQ) of the reads without an insertion resp. deletion, in pileup order. */
__device__ __forceinline__ int lfq_plp_locate_indel(const uint32_t *cg, int n_cigar, int64_t x, int64_t p, int l_qseq,
                                                    int *qpos, int *indel, bool *tail)
{
    int y = 0, kind = 0;
    *indel = 0;
    for (int k = 0; k < n_cigar; ++k) {
        const int op = cg[k] & 0xf, l = cg[k] >> 4;
        const bool m = op == 0 || op == 7 || op == 8, d = op == 2 || op == 3;
        if ((m || d) && kind == 0 && p >= x && p < x + l) {
            kind = m ? 1 : 2;
            int q = m ? y + (int)(p - x) : y;
            *qpos = q < l_qseq ? q : l_qseq - 1;
            if (p == x + l - 1 && k + 1 < n_cigar) {            /* resolve_cigar2: peek at the next operation */
                const int op2 = cg[k + 1] & 0xf, l2 = cg[k + 1] >> 4;
                if (op2 == 2) {
                    *indel = -l2;
                } else if (op2 == 1) {
                    *indel = l2;
                } else if (op2 == 6 && k + 2 < n_cigar) {
                    int l3 = 0;
                    for (int kk = k + 2; kk < n_cigar; ++kk) {
                        const int o3 = cg[kk] & 0xf;
                        if (o3 == 1) {
                            l3 += cg[kk] >> 4;
                        } else if (o3 == 2 || o3 == 0 || o3 == 3 || o3 == 7 || o3 == 8) {
                            break;
                        }
                    }
                    *indel = l3 > 0 ? l3 : 0;
                }
            }
        }
        if (m) {
            x += l; y += l;
        } else if (d) {
            x += l;
        } else if (op == 1 || op == 4) {
            y += l;
        }
    }
    *tail = (p == x - 1);                        /* x is now bam_endpos */
    return kind;
}

template <bool SCATTER>
__global__ __launch_bounds__(256) void lfq_plp_indel_columns_kernel(LfqPlpIndelArgs A)
{
    const int lane = (int)(threadIdx.x & 63u);
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= A.width) {
        return;
    }
    const int64_t p = A.begin + c;
    int64_t off0 = -1, off1 = -1;
    if (SCATTER) {
        off0 = A.ne_off[0][c];
        off1 = A.ne_off[1][c];
        if (off0 < 0 && off1 < 0) {
            return;
        }
    }
    const int64_t lo = lfq_wave_first_above(A.pmax_end, 0, A.n_reads, p, lane);
    const int64_t hi = lfq_wave_first_above(A.pos, lo, A.n_reads, p, lane);
    uint32_t cnt[7] = {0, 0, 0, 0, 0, 0, 0};      /* cov, tails, non_indels, n_ins, n_dels, non_ins_fw, non_del_fw */
    uint32_t qs0 = 0, qs1 = 0;                    /* per lane: indel qualities of the reads without an insertion / deletion */
    uint32_t w0 = 0, w1 = 0;                      /* scatter cursors of the two sides */
    for (int64_t r0 = lo; r0 < hi; r0 += 64) {
        const int64_t r = r0 + lane;
        int kind = 0, qpos = 0, indel = 0, iq = 0, dq = 0, rev = 0;
        bool tail = false;
        int16_t mq = 0;
        if (r < hi) {
            const int64_t co = A.cigar_off[r], s0 = A.seq_off[r];
            const int l_qseq = (int)(A.seq_off[r + 1] - s0);
            kind = lfq_plp_locate_indel(A.cigar + co, (int)(A.cigar_off[r + 1] - co), A.pos[r], p, l_qseq, &qpos, &indel, &tail);
            if (kind) {
                const uint32_t fl = A.tag_flags ? A.tag_flags[r] : 3u;
                iq = (A.bi && (fl & 1u) && qpos >= 0) ? (int)A.bi[s0 + qpos] - 33 : 0;           /* plp.c:1023-1059 */
                dq = (A.bd && (fl & 2u) && qpos >= 0) ? (int)A.bd[s0 + qpos] - 33 : 0;
                rev = A.reverse[r] ? 1 : 0;
                mq = (int16_t)A.mapq[r];
            }
        }
        const bool pass = kind != 0 && !(iq < A.min_plp_idq || dq < A.min_plp_idq);              /* :1062 */
        const bool no_ins = pass && indel <= 0, no_del = pass && indel >= 0;
        if (!SCATTER) {
            cnt[0] += (uint32_t)__popcll(__ballot(kind != 0));
            cnt[1] += (uint32_t)__popcll(__ballot(kind == 1 && tail));                           /* :920-922 */
            cnt[2] += (uint32_t)__popcll(__ballot(pass && indel == 0));
            cnt[3] += (uint32_t)__popcll(__ballot(pass && indel > 0));
            cnt[4] += (uint32_t)__popcll(__ballot(pass && indel < 0));
            cnt[5] += (uint32_t)__popcll(__ballot(no_ins && !rev));
            cnt[6] += (uint32_t)__popcll(__ballot(no_del && !rev));
            qs0 += no_ins ? (uint32_t)iq : 0u;
            qs1 += no_del ? (uint32_t)dq : 0u;
        } else {
            const uint64_t m0 = __ballot(no_ins), m1 = __ballot(no_del), below = (1ull << lane) - 1ull;
            if (no_ins && off0 >= 0) {
                const int64_t slot = off0 + w0 + __popcll(m0 & below);
                A.ne_q[0][slot] = (int16_t)iq;
                A.ne_mq[0][slot] = mq;
            }
            if (no_del && off1 >= 0) {
                const int64_t slot = off1 + w1 + __popcll(m1 & below);
                A.ne_q[1][slot] = (int16_t)dq;
                A.ne_mq[1][slot] = mq;
            }
            w0 += (uint32_t)__popcll(m0);
            w1 += (uint32_t)__popcll(m1);
        }
    }
    if (!SCATTER) {
        qs0 = lfq_wave_sum_u32(qs0);
        qs1 = lfq_wave_sum_u32(qs1);
    }
    if (!SCATTER && lane == 0) {
        A.ne_qsum[0][c] = (int32_t)qs0;
        A.ne_qsum[1][c] = (int32_t)qs1;
        A.cov[c] = (int32_t)cnt[0];
        A.tails[c] = (int32_t)cnt[1];
        A.non_indels[c] = (int32_t)cnt[2];
        A.n_ins[c] = (int32_t)cnt[3];
        A.n_dels[c] = (int32_t)cnt[4];
        A.non_ins_fw[c] = (int32_t)cnt[5];
        A.non_del_fw[c] = (int32_t)cnt[6];
    }
}

/* ---- the counter pass of the indel fields by tiles of 64 positions (see lfq_pileup_tiles_kernel) ---------------------
 * Per read and tile, phase A (a read per thread pair) keeps: the two masks; the BI / BD bytes of the query stretch the tile's
 * positions map to (one indel inside the tile included) as aligned words in LDS; the positions at which an insertion or
 * deletion FOLLOWS (the last position of a CIGAR operation, resolve_cigar2's peek: at most three per tile here) with their
 * lengths; the read's last aligned position if it lies in the tile.  A deleted / skipped position takes the qualities of the
 * next query base (htslib's qpos inside a deletion): the base of the next covered position when the deletion is followed
 * directly by a match inside the tile.  Everything else -- three stretches, more than three events, a tile that ends inside
 * a deletion, a deletion followed by an insertion, an insertion too long for the row -- is flagged and resolved per
 * position in phase B exactly as the column-major kernel does (there is no store in this pass that such a load could
 * stall).  Phase B: a wavefront per position, lanes = reads in read order; the seven counts by ballots, the two quality
 * sums per lane, added to the tile's accumulators in LDS once per round. */
struct LfqTileIndel {
    unsigned long long cm, dm;
    int off0, off1, split, nw;
    int64_t a0;
    int ev_dp[3], ev_val[3];
    int tail_dp;
    bool slow;
};

__device__ __forceinline__ int lfq_cigar_peek_indel(const uint32_t *cg, int nc, int k)
{
    int indel = 0;                                      /* resolve_cigar2: what follows operation k */
    if (k + 1 < nc) {
        const int op2 = cg[k + 1] & 0xf, l2 = cg[k + 1] >> 4;
        if (op2 == 2) {
            indel = -l2;
        } else if (op2 == 1) {
            indel = l2;
        } else if (op2 == 6 && k + 2 < nc) {
            int l3 = 0;
            for (int kk = k + 2; kk < nc; ++kk) {
                const int o3 = cg[kk] & 0xf;
                if (o3 == 1) {
                    l3 += cg[kk] >> 4;
                } else if (o3 == 2 || o3 == 0 || o3 == 3 || o3 == 7 || o3 == 8) {
                    break;
                }
            }
            indel = l3 > 0 ? l3 : 0;
        }
    }
    return indel;
}

__device__ __forceinline__ LfqTileIndel lfq_tile_resolve_indel(const uint32_t *cg, int nc, int64_t x, int64_t s0, int64_t p0, int tile)
{
    LfqTileIndel R;
    R.cm = R.dm = 0;
    R.off0 = R.off1 = 0;
    R.split = 64;
    R.nw = 0;
    R.a0 = 0;
    R.slow = false;
    R.tail_dp = 255;
    R.ev_dp[0] = R.ev_dp[1] = R.ev_dp[2] = 255;
    R.ev_val[0] = R.ev_val[1] = R.ev_val[2] = 0;
    int y = 0, nseg = 0, nev = 0, q1 = 0, p1 = 0, q2 = 0, p2 = 0, n2 = 0, n1 = 0;
    for (int k = 0; k < nc; ++k) {
        const int op = cg[k] & 0xf, l = cg[k] >> 4;
        const bool m = op == 0 || op == 7 || op == 8, d = op == 2 || op == 3;
        if (m || d) {
            const int64_t a = x > p0 ? x : p0, b = (x + l < p0 + tile) ? x + l : p0 + tile;
            if (a < b) {
                const int n = (int)(b - a), sh = (int)(a - p0);
                const unsigned long long mk = (n >= 64 ? ~0ull : ((1ull << n) - 1ull)) << sh;
                if (d) {
                    R.dm |= mk;
                    /* its positions take the qualities of the next query base: that is the next covered position's base
                     * only if a match follows directly, inside the tile */
                    const int opn = k + 1 < nc ? (int)(cg[k + 1] & 0xf) : -1;
                    if (!(opn == 0 || opn == 7 || opn == 8) || x + l >= p0 + tile) {
                        R.slow = true;
                    }
                } else {
                    const int q = y + (int)(a - x);
                    R.cm |= mk;
                    if (nseg == 0) {
                        q1 = q; p1 = sh; n1 = n;
                        nseg = 1;
                    } else if (nseg == 1 && q == q1 + n1 && sh == p1 + n1) {
                        n1 += n;
                    } else if (nseg == 1) {
                        q2 = q; p2 = sh; n2 = n;
                        nseg = 2;
                    } else if (nseg == 2 && q == q2 + n2 && sh == p2 + n2) {
                        n2 += n;
                    } else {
                        nseg++;
                    }
                }
            }
            const int64_t pl = x + l - 1;               /* the operation's last position: what follows it? */
            if (l > 0 && pl >= p0 && pl < p0 + tile) {
                const int v = lfq_cigar_peek_indel(cg, nc, k);
                if (v != 0) {
                    const int e = (int)(pl - p0);       /* (no indexing by nev: the arrays would go to scratch memory) */
                    if (nev == 0) {
                        R.ev_dp[0] = e; R.ev_val[0] = v;
                    } else if (nev == 1) {
                        R.ev_dp[1] = e; R.ev_val[1] = v;
                    } else if (nev == 2) {
                        R.ev_dp[2] = e; R.ev_val[2] = v;
                    }
                    nev++;
                }
            }
            x += l;
            if (m) {
                y += l;
            }
        } else if (op == 1 || op == 4) {
            y += l;
        }
    }
    /* x is bam_endpos now: is_tail (plp.c:920-922) */
    if (x - 1 >= p0 && x - 1 < p0 + tile && ((R.cm >> (int)(x - 1 - p0)) & 1ull)) {
        R.tail_dp = (int)(x - 1 - p0);
    }
    if (nseg == 1 || nseg == 2) {
        const int64_t g0 = s0 + q1;
        R.a0 = g0 & ~(int64_t)15;
        const int span = nseg == 1 ? n1 : (q2 + n2 - q1);
        R.nw = (int)((g0 - R.a0) + span + 15) >> 4;
        R.off0 = (int)(g0 - R.a0) - p1;
        R.off1 = nseg == 1 ? R.off0 : (int)(g0 - R.a0) + (q2 - q1) - p2;
        R.split = nseg == 1 ? 64 : p2;
    }
    if (nseg > 2 || R.nw > 6 || nev > 3 || (nseg == 0 && R.dm)) {
        R.slow = true;
    }
    if (R.slow) {
        R.nw = 0;
    }
    return R;
}

__global__ __launch_bounds__(256) void lfq_plp_indel_tiles_kernel(LfqPlpIndelArgs A)
{
    constexpr int RC = 128;                                         /* reads per round, two threads each: BI row / BD row */
    __shared__ uint32_t s_row[2][RC][LFQ_TILE_ROW / 4];
    __shared__ unsigned long long s_cm[RC], s_dm[RC];
    __shared__ int16_t s_off0[RC], s_off1[RC];
    __shared__ uint8_t s_split[RC], s_tail[RC], s_fl[RC];           /* s_fl: bit 0 BI, 1 BD, 2 reverse strand, 3 resolved per position */
    __shared__ uint32_t s_evdp[RC];                                 /* byte j: position of event j, 255 = none */
    __shared__ int32_t s_evv[3][RC];
    __shared__ uint32_t s_acc[LFQ_TILE][9];                         /* the tile's nine outputs per position */
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t c0 = (int64_t)blockIdx.x * LFQ_TILE;
    if (c0 >= A.width) {
        return;
    }
    const int64_t p0 = A.begin + c0;
    const int tile = (int)((A.width - c0 < LFQ_TILE) ? (A.width - c0) : LFQ_TILE);
    for (int i = tid; i < LFQ_TILE * 9; i += 256) {
        (&s_acc[0][0])[i] = 0;
    }
    __syncthreads();
    const int64_t lo = lfq_wave_first_above(A.pmax_end, 0, A.n_reads, p0, lane);
    const int64_t hi = lfq_wave_first_above(A.pos, lo, A.n_reads, p0 + tile - 1, lane);
    const int rt = tid & (RC - 1);
    const bool first = tid < RC;
    for (int64_t r0 = lo; r0 < hi; r0 += RC) {
        /* ---- phase A ---- */
        {
            const int64_t r = r0 + rt;
            LfqTileIndel R;
            R.cm = R.dm = 0;
            R.off0 = R.off1 = 0;
            R.split = 64;
            R.tail_dp = 255;
            R.ev_dp[0] = R.ev_dp[1] = R.ev_dp[2] = 255;
            R.ev_val[0] = R.ev_val[1] = R.ev_val[2] = 0;
            R.slow = false;
            uint32_t fl = 0;
            if (r < hi) {
                const int64_t co = A.cigar_off[r], s0 = A.seq_off[r];
                R = lfq_tile_resolve_indel(A.cigar + co, (int)(A.cigar_off[r + 1] - co), A.pos[r], s0, p0, tile);
                const uint32_t tf = A.tag_flags ? A.tag_flags[r] : 3u;
                fl = ((A.bi && (tf & 1u)) ? 1u : 0u) | ((A.bd && (tf & 2u)) ? 2u : 0u) | (A.reverse[r] ? 4u : 0u) | (R.slow ? 8u : 0u);
                if (R.nw > 0) {
                    if (first && (fl & 1u)) {
                        lfq_tile_fetch(&s_row[0][rt][0], A.bi, R.a0, R.nw);
                    } else if (!first && (fl & 2u)) {
                        lfq_tile_fetch(&s_row[1][rt][0], A.bd, R.a0, R.nw);
                    }
                }
            }
            if (first) {
                s_cm[rt] = R.cm;
                s_dm[rt] = R.dm;
                s_off0[rt] = (int16_t)R.off0;
                s_off1[rt] = (int16_t)R.off1;
                s_split[rt] = (uint8_t)R.split;
                s_tail[rt] = (uint8_t)R.tail_dp;
                s_fl[rt] = (uint8_t)fl;
                s_evdp[rt] = (uint32_t)R.ev_dp[0] | ((uint32_t)R.ev_dp[1] << 8) | ((uint32_t)R.ev_dp[2] << 16);
                s_evv[0][rt] = R.ev_val[0];
                s_evv[1][rt] = R.ev_val[1];
                s_evv[2][rt] = R.ev_val[2];
            }
        }
        __syncthreads();
        /* ---- phase B: positions wave * 16 .. + 15; both 64-read slices of the round in registers ---- */
        unsigned long long cm[2], dm[2];
        int off0[2], off1[2], split[2], tail[2], evv0[2], evv1[2], evv2[2];
        uint32_t fl[2], evdp[2];
#pragma unroll
        for (int sc = 0; sc < 2; sc++) {
            const int t = sc * 64 + lane;
            cm[sc] = s_cm[t]; dm[sc] = s_dm[t];
            off0[sc] = s_off0[t]; off1[sc] = s_off1[t]; split[sc] = s_split[t]; tail[sc] = s_tail[t];
            fl[sc] = s_fl[t]; evdp[sc] = s_evdp[t];
            evv0[sc] = s_evv[0][t]; evv1[sc] = s_evv[1][t]; evv2[sc] = s_evv[2][t];
        }
        for (int i = 0; i < 16; i++) {
            const int dp = wave * 16 + i;
            uint32_t cnt[7] = {0, 0, 0, 0, 0, 0, 0}, qs0 = 0, qs1 = 0;
#pragma unroll
            for (int sc = 0; sc < 2; sc++) {
                const int t = sc * 64 + lane;
                bool covered = (cm[sc] >> dp) & 1ull;
                bool any = covered || ((dm[sc] >> dp) & 1ull);
                int iq = 0, dq = 0, indel = 0;
                bool is_tail = covered && dp == tail[sc];
                if (any && !(fl[sc] & 8u)) {
                    int dpn = dp;
                    if (!covered) {
                        dpn = dp + __builtin_ctzll(cm[sc] >> dp);           /* the next covered position: there is one (phase A) */
                    }
                    const int idx = (dpn < split[sc] ? off0[sc] : off1[sc]) + dpn;
                    if (fl[sc] & 1u) {
                        iq = (int)reinterpret_cast<const uint8_t *>(&s_row[0][t][0])[idx] - 33;
                    }
                    if (fl[sc] & 2u) {
                        dq = (int)reinterpret_cast<const uint8_t *>(&s_row[1][t][0])[idx] - 33;
                    }
                    indel = (dp == (int)(evdp[sc] & 255u)) ? evv0[sc]
                            : (dp == (int)((evdp[sc] >> 8) & 255u)) ? evv1[sc]
                            : (dp == (int)((evdp[sc] >> 16) & 255u)) ? evv2[sc] : 0;
                } else if (any) {
                    /* resolved per position, as the column-major kernel does */
                    const int64_t r = r0 + t, co = A.cigar_off[r], s0 = A.seq_off[r];
                    int qpos = 0;
                    bool tl = false;
                    const int kind = lfq_plp_locate_indel(A.cigar + co, (int)(A.cigar_off[r + 1] - co), A.pos[r], p0 + dp,
                                                          (int)(A.seq_off[r + 1] - s0), &qpos, &indel, &tl);
                    covered = kind == 1;
                    any = kind != 0;
                    is_tail = covered && tl;
                    iq = ((fl[sc] & 1u) && qpos >= 0) ? (int)A.bi[s0 + qpos] - 33 : 0;
                    dq = ((fl[sc] & 2u) && qpos >= 0) ? (int)A.bd[s0 + qpos] - 33 : 0;
                }
                const bool rev = (fl[sc] & 4u) != 0;
                const bool pass = any && !(iq < A.min_plp_idq || dq < A.min_plp_idq);              /* plp.c:1062 */
                const bool no_ins = pass && indel <= 0, no_del = pass && indel >= 0;
                cnt[0] += (uint32_t)__popcll(__ballot(any));
                cnt[1] += (uint32_t)__popcll(__ballot(is_tail));                                 /* :920-922 */
                cnt[2] += (uint32_t)__popcll(__ballot(pass && indel == 0));
                cnt[3] += (uint32_t)__popcll(__ballot(pass && indel > 0));
                cnt[4] += (uint32_t)__popcll(__ballot(pass && indel < 0));
                cnt[5] += (uint32_t)__popcll(__ballot(no_ins && !rev));
                cnt[6] += (uint32_t)__popcll(__ballot(no_del && !rev));
                qs0 += no_ins ? (uint32_t)iq : 0u;
                qs1 += no_del ? (uint32_t)dq : 0u;
            }
            qs0 = lfq_wave_sum_u32(qs0);
            qs1 = lfq_wave_sum_u32(qs1);
            if (lane < 9) {
                uint32_t v = lane == 0 ? cnt[0] : lane == 1 ? cnt[1] : lane == 2 ? cnt[2] : lane == 3 ? cnt[3] : lane == 4 ? cnt[4]
                             : lane == 5 ? cnt[5] : lane == 6 ? cnt[6] : lane == 7 ? qs0 : qs1;
                s_acc[dp][lane] += v;
            }
        }
        __syncthreads();
    }
    if (tid < tile) {
        const int64_t c = c0 + tid;
        A.cov[c] = (int32_t)s_acc[tid][0];
        A.tails[c] = (int32_t)s_acc[tid][1];
        A.non_indels[c] = (int32_t)s_acc[tid][2];
        A.n_ins[c] = (int32_t)s_acc[tid][3];
        A.n_dels[c] = (int32_t)s_acc[tid][4];
        A.non_ins_fw[c] = (int32_t)s_acc[tid][5];
        A.non_del_fw[c] = (int32_t)s_acc[tid][6];
        A.ne_qsum[0][c] = (int32_t)s_acc[tid][7];
        A.ne_qsum[1][c] = (int32_t)s_acc[tid][8];
    }
}

int lfq_launch_plp_indel_columns(const LfqPlpIndelArgs &a, int scatter, void *stream)
{
    if (!scatter && a.n_reads > 0 && a.width > 0 && lfq_knobs().pileup_tiles) {
        const dim3 grid((unsigned)((a.width + LFQ_TILE - 1) / LFQ_TILE)), block(256);
        hipLaunchKernelGGL(lfq_plp_indel_tiles_kernel, grid, block, 0, (hipStream_t)stream, a);
        return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
    }
    if (a.n_reads <= 0 || a.width <= 0) {
        return LFQ_OK;
    }
    const dim3 grid((unsigned)((a.width + 3) / 4)), block(256);
    if (scatter) {
        hipLaunchKernelGGL(lfq_plp_indel_columns_kernel<true>, grid, block, 0, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL(lfq_plp_indel_columns_kernel<false>, grid, block, 0, (hipStream_t)stream, a);
    }
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}


/* ---- indel pseudo-columns on the device ----------------------------------------------------------------------
 * One wavefront per tested event: the column's reads without an event of that side (indel quality, MAPQ) followed
 * by the reads of all its events, the tested one's marked as the alt allele and carrying their alignment quality
 * (snpcaller.c:502-623) -- what pack_indel_test does on the host, reading the quality arrays where
 * lfq_readset_pileup_indels left them. */
__device__ __forceinline__ uint8_t lfq_q8(int q)
{
    return q < 0 ? (uint8_t)255 : (uint8_t)(q > 254 ? 254 : q);
}

__global__ __launch_bounds__(256) void lfq_indel_pack_kernel(LfqIndelPackArgs A)
{
    const int lane = (int)(threadIdx.x & 63u);
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= A.n_tests) {
        return;
    }
    const LfqIndelTestDesc D = A.tests[t];
    const int sd = D.side;
    const int16_t *nq = A.ne_q[sd] + D.ne_off, *nm = A.ne_mq[sd] + D.ne_off;
    for (int i = lane; i < D.ne_len; i += 64) {
        const int64_t o = D.out_off + i;
        const int q = nq[i];
        A.nt[o] = 0;
        A.bq[o] = (uint8_t)(q < 0 ? 0 : (q > 254 ? 254 : q));
        A.baq[o] = 255;
        A.mq[o] = A.use_mq ? lfq_q8(nm[i]) : (uint8_t)255;
        A.sq[o] = 255;
    }
    for (int i = lane; i < D.rd_len; i += 64) {
        const int64_t o = D.out_off + D.ne_len + i, g = D.rd_begin + i;
        const bool me = i >= D.me_begin && i < D.me_begin + D.me_len;
        const int q = A.rd_q[sd][g];
        A.nt[o] = me ? 1 : 0;
        A.bq[o] = (uint8_t)(q < 0 ? 0 : (q > 254 ? 254 : q));
        A.baq[o] = (me && A.use_aq) ? lfq_q8(A.rd_aq[sd][g]) : (uint8_t)255;
        A.mq[o] = A.use_mq ? lfq_q8(A.rd_mq[sd][g]) : (uint8_t)255;
        A.sq[o] = A.use_sq ? lfq_q8(A.rd_sq[sd][g]) : (uint8_t)255;
    }
}

int lfq_launch_indel_pack(const LfqIndelPackArgs &a, void *stream)
{
    if (a.n_tests <= 0) {
        return LFQ_OK;
    }
    hipLaunchKernelGGL(lfq_indel_pack_kernel, dim3((unsigned)((a.n_tests + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}

/* per read: the resident tag flags (bit 0 BI, 1 BD) take the ai / ad bits the BAQ kernels left in `tag` (bit 0 ai, 1 ad) */
__global__ __launch_bounds__(256) void lfq_flag_merge_kernel(uint8_t *fl, const uint8_t *tag, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        fl[i] = (uint8_t)((fl[i] & 3u) | ((tag[i] & 3u) << 2));
    }
}

int lfq_launch_flag_merge(uint8_t *fl, const uint8_t *tag, int64_t n, void *stream)
{
    if (n <= 0) {
        return LFQ_OK;
    }
    hipLaunchKernelGGL(lfq_flag_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fl, tag, n);
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}

/* columns the caller takes out of the SNV pass (lofreq_call.c:1049-1053: a column whose consensus is an indel is
 * not tested for substitutions): their base count goes to 0, which is what the count kernel's depth test reads */
__global__ __launch_bounds__(256) void lfq_skip_columns_kernel(int32_t *nb, const uint8_t *skip, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && skip[i]) {
        nb[i] = 0;
    }
}

/* byte-per-observation nt track -> LFQ_TRACKS_NT_PACKED (include/lofreq_amd.h): observations in groups of 8 by their index in
 * the track, byte k of a group's four bytes = observation k (low nibble) | observation 4 + k (high nibble).  One thread
 * per group: 8 bytes in, 4 bytes out; the track is zero-padded past its last observation. */
__global__ void lfq_pack_nt_kernel(const uint2 *__restrict__ nt, uint32_t *__restrict__ out, int64_t n_groups)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n_groups) {
        const uint2 v = nt[g];
        out[g] = (v.x & 0x0F0F0F0Fu) | ((v.y & 0x0F0F0F0Fu) << 4);
    }
}

int lfq_launch_pack_nt(const uint8_t *nt_bytes, uint8_t *nt_packed, int64_t n_obs, void *stream)
{
    const int64_t n_groups = (n_obs + 7) / 8;
    if (n_groups <= 0) {
        return LFQ_OK;
    }
    hipLaunchKernelGGL(lfq_pack_nt_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint2 *)nt_bytes, (uint32_t *)nt_packed, n_groups);
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}

int lfq_launch_skip_columns(int32_t *nb, const uint8_t *skip, int64_t n, void *stream)
{
    if (n <= 0) {
        return LFQ_OK;
    }
    hipLaunchKernelGGL(lfq_skip_columns_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, nb, skip, n);
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}

/* ------------------------------------------------------------------------------------------ */
/* columns of the SNV pileup from the per-position counters, on the device                     */
/* ------------------------------------------------------------------------------------------ */
/* The count pass leaves coverage and kept bases per reference position; the tracks need, per COVERED position, a column
 * index, the offset of its first observation, its reference base and the two counts -- an exclusive scan over the
 * positions and a compaction.  That was a round trip: 8 MB of counters to the host, two threaded passes over the
 * positions, 21 MB back, two stream synchronisations, with the GPU idle in between (2 - 4 ms per 1 Mb region).  Three small
 * kernels instead (tiles of 4096 positions: sums per tile, scan of the tile sums + the totals, apply); the host waits
 * once, for three numbers (it sizes the tracks from them). */
#define LFQ_PC_THREADS 256
#define LFQ_PC_ITEMS 16
#define LFQ_PC_TILE (LFQ_PC_THREADS * LFQ_PC_ITEMS)

__device__ __forceinline__ void lfq_pc_block_scan(int32_t &n, unsigned long long &o, int32_t *s_n, unsigned long long *s_o)
{
    /* inclusive scan of (n, o) over the threads of the block (Hillis-Steele in LDS: 8 steps for 256 threads) */
    const int t = (int)threadIdx.x;
    s_n[t] = n;
    s_o[t] = o;
    __syncthreads();
    for (int d = 1; d < LFQ_PC_THREADS; d <<= 1) {
        const int32_t pn = t >= d ? s_n[t - d] : 0;
        const unsigned long long po = t >= d ? s_o[t - d] : 0ull;
        __syncthreads();
        s_n[t] += pn;
        s_o[t] += po;
        __syncthreads();
    }
    n = s_n[t];
    o = s_o[t];
}

__global__ __launch_bounds__(LFQ_PC_THREADS) void lfq_plp_compact_tiles_kernel(const int32_t *__restrict__ cov,
                                                                              const int32_t *__restrict__ nb, int64_t width,
                                                                              int64_t *__restrict__ tile_cols,
                                                                              unsigned long long *__restrict__ tile_obs,
                                                                              int32_t *__restrict__ tile_max)
{
    __shared__ int32_t s_n[LFQ_PC_THREADS];
    __shared__ unsigned long long s_o[LFQ_PC_THREADS];
    __shared__ int32_t s_m[LFQ_PC_THREADS];
    const int64_t p0 = (int64_t)blockIdx.x * LFQ_PC_TILE + (int64_t)threadIdx.x * LFQ_PC_ITEMS;
    int32_t n = 0, mx = 0;
    unsigned long long o = 0;
    for (int i = 0; i < LFQ_PC_ITEMS; i++) {
        const int64_t p = p0 + i;
        if (p < width && cov[p] > 0) {
            const int32_t b = nb[p];
            n++;
            o += (unsigned long long)b;
            mx = max(mx, b);
        }
    }
    s_m[threadIdx.x] = mx;
    lfq_pc_block_scan(n, o, s_n, s_o);
    for (int d = LFQ_PC_THREADS / 2; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            s_m[threadIdx.x] = max(s_m[threadIdx.x], s_m[threadIdx.x + d]);
        }
        __syncthreads();
    }
    if (threadIdx.x == LFQ_PC_THREADS - 1) {
        tile_cols[blockIdx.x] = n;
        tile_obs[blockIdx.x] = o;
    }
    if (threadIdx.x == 0) {
        tile_max[blockIdx.x] = s_m[0];
    }
}

/* one block: exclusive scan of the tile sums in place; totals[0] = columns, [1] = observations, [2] = deepest column */
__global__ __launch_bounds__(LFQ_PC_THREADS) void lfq_plp_compact_sums_kernel(int64_t ntiles, int64_t *__restrict__ tile_cols,
                                                                             unsigned long long *__restrict__ tile_obs,
                                                                             const int32_t *__restrict__ tile_max,
                                                                             int64_t *__restrict__ totals)
{
    __shared__ int32_t s_n[LFQ_PC_THREADS];
    __shared__ unsigned long long s_o[LFQ_PC_THREADS];
    __shared__ int32_t s_m[LFQ_PC_THREADS];
    int64_t base_n = 0;
    unsigned long long base_o = 0;
    int32_t mx = 0;
    for (int64_t t0 = 0; t0 < ntiles; t0 += LFQ_PC_THREADS) {
        const int64_t t = t0 + threadIdx.x;
        const int32_t n_in = t < ntiles ? (int32_t)tile_cols[t] : 0;
        const unsigned long long o_in = t < ntiles ? tile_obs[t] : 0ull;
        if (t < ntiles) {
            mx = max(mx, tile_max[t]);
        }
        int32_t n = n_in;
        unsigned long long o = o_in;
        lfq_pc_block_scan(n, o, s_n, s_o);
        if (t < ntiles) {
            tile_cols[t] = base_n + n - n_in;
            tile_obs[t] = base_o + o - o_in;
        }
        base_n += s_n[LFQ_PC_THREADS - 1];
        base_o += s_o[LFQ_PC_THREADS - 1];
        __syncthreads();
    }
    s_m[threadIdx.x] = mx;
    __syncthreads();
    for (int d = LFQ_PC_THREADS / 2; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            s_m[threadIdx.x] = max(s_m[threadIdx.x], s_m[threadIdx.x + d]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        totals[0] = base_n;
        totals[1] = (int64_t)base_o;
        totals[2] = s_m[0];
    }
}

__global__ __launch_bounds__(LFQ_PC_THREADS) void lfq_plp_compact_apply_kernel(const int32_t *__restrict__ cov,
                                                                              const int32_t *__restrict__ nb, int64_t width,
                                                                              int64_t begin, const uint8_t *__restrict__ ref,
                                                                              int64_t ref_len,
                                                                              const int64_t *__restrict__ tile_cols,
                                                                              const unsigned long long *__restrict__ tile_obs,
                                                                              const int64_t *__restrict__ totals,
                                                                              int32_t *__restrict__ col_index,
                                                                              uint64_t *__restrict__ col_off,
                                                                              uint8_t *__restrict__ ref_base,
                                                                              int32_t *__restrict__ cov_c,
                                                                              int32_t *__restrict__ nb_c,
                                                                              int64_t *__restrict__ col_pos)
{
    __shared__ int32_t s_n[LFQ_PC_THREADS];
    __shared__ unsigned long long s_o[LFQ_PC_THREADS];
    const int64_t p0 = (int64_t)blockIdx.x * LFQ_PC_TILE + (int64_t)threadIdx.x * LFQ_PC_ITEMS;
    int32_t n = 0;
    unsigned long long o = 0;
    for (int i = 0; i < LFQ_PC_ITEMS; i++) {
        const int64_t p = p0 + i;
        if (p < width && cov[p] > 0) {
            n++;
            o += (unsigned long long)nb[p];
        }
    }
    const int32_t n_own = n;
    const unsigned long long o_own = o;
    lfq_pc_block_scan(n, o, s_n, s_o);
    int64_t ci = tile_cols[blockIdx.x] + (n - n_own);
    unsigned long long run = tile_obs[blockIdx.x] + (o - o_own);
    for (int i = 0; i < LFQ_PC_ITEMS; i++) {
        const int64_t p = p0 + i;
        if (p >= width) {
            break;
        }
        const int32_t cv = cov[p];
        if (cv <= 0) {
            col_index[p] = -1;
            continue;
        }
        const int32_t b = nb[p];
        const int64_t gp = begin + p;
        uint8_t rb = gp < ref_len ? ref[gp] : (uint8_t)'N';                 /* plp.c:818-823 */
        if (!(rb == 'A' || rb == 'C' || rb == 'T' || rb == 'G' || rb == 'N')) {
            rb = 'N';
        }
        col_index[p] = (int32_t)ci;
        col_off[ci] = run;
        ref_base[ci] = rb;
        cov_c[ci] = cv;
        nb_c[ci] = b;
        col_pos[ci] = gp;
        run += (unsigned long long)b;
        ci++;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        col_off[totals[0]] = (uint64_t)totals[1];
    }
}

int lfq_launch_plp_compact_sums(const int32_t *cov, const int32_t *nb, int64_t width, int64_t *tile_cols, uint64_t *tile_obs,
                                int32_t *tile_max, int64_t *totals, void *stream)
{
    const int64_t ntiles = (width + LFQ_PC_TILE - 1) / LFQ_PC_TILE;
    hipLaunchKernelGGL(lfq_plp_compact_tiles_kernel, dim3((unsigned)ntiles), dim3(LFQ_PC_THREADS), 0, (hipStream_t)stream, cov, nb,
                       width, tile_cols, (unsigned long long *)tile_obs, tile_max);
    hipLaunchKernelGGL(lfq_plp_compact_sums_kernel, dim3(1), dim3(LFQ_PC_THREADS), 0, (hipStream_t)stream, ntiles, tile_cols,
                       (unsigned long long *)tile_obs, tile_max, totals);
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}

int lfq_launch_plp_compact_apply(const int32_t *cov, const int32_t *nb, int64_t width, int64_t begin, const uint8_t *ref,
                                 int64_t ref_len, const int64_t *tile_cols, const uint64_t *tile_obs, const int64_t *totals,
                                 int32_t *col_index, uint64_t *col_off, uint8_t *ref_base, int32_t *cov_c, int32_t *nb_c,
                                 int64_t *col_pos, void *stream)
{
    const int64_t ntiles = (width + LFQ_PC_TILE - 1) / LFQ_PC_TILE;
    hipLaunchKernelGGL(lfq_plp_compact_apply_kernel, dim3((unsigned)ntiles), dim3(LFQ_PC_THREADS), 0, (hipStream_t)stream, cov, nb,
                       width, begin, ref, ref_len, tile_cols, (const unsigned long long *)tile_obs, totals, col_index, col_off,
                       ref_base, cov_c, nb_c, col_pos);
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}
