/*
 * lfq_internal.h -- device-side parameter blocks shared by the HIP kernels and the C-ABI layer.
 * Not part of the public ABI.
 */
#ifndef LFQ_INTERNAL_H
#define LFQ_INTERNAL_H

#include <stdint.h>

#include "lofreq_amd.h"

/* phred -> probability tables, computed on the HOST with the same pow() call as
 * PHREDQUAL_TO_PROB (utils.h:42) so that the quality merge is bit-identical.  Index 255 of the
 * baq/mq/sq tables is the "missing" code (probability 0, snpcaller.c:307-331); mq[0] = 0.5
 * (snpcaller.c:64, 315-316).  A track that is absent or switched off by `flag` is looked up at
 * index 255 for every observation. */
struct LfqLuts {
    double bq[256];
    double baq[256];
    double mq[256];
    double sq[256];
};

struct LfqParams {
    int32_t min_bq4;          /* clamp(min_bq, 0, 128) */
    int32_t min_alt_bq4;      /* clamp(max(min_bq, min_alt_bq), 0, 128): an alt base must pass both */
    int32_t def_alt_bq;       /* 0 keep, >0 override, -1 median of reference-base BQs */
    int32_t general;          /* 1: merged-quality filters / median override active -> slow count path */
    double jq_reject_above;   /* observation dropped if merged prob > this (min_jq);  +inf = off */
    double alt_jq_reject_above; /* same for alt observations (min_alt_jq) */
    double def_alt_jp;        /* < 0: keep merged prob; else overrides it for alt observations */
    int32_t min_cov;
    int32_t use_baq, use_mq, use_sq;   /* track present AND enabled by conf->flag */
    int32_t bonf_dynamic;
    int64_t bonf_base;        /* conf->bonf_subst (or bonf_indel) before this batch */
    int32_t bonf_step;        /* tests added per tested column: 3 for SNVs (lofreq_call.c:798), 1 for indel tests (:694) */
    int32_t bonf_reset_first; /* SNVs: the first tested column SETS the factor to 3 instead of adding (lofreq_call.c:795-796) */
    double sig;               /* (double)(float)conf->sig */
    double prune_slack;       /* prune only if P*bonf > sig*(1+slack); host applies the exact test */
    int32_t seg_max;          /* row segments per split column of the big class, 2..LFQ_SEG_MAX */
    int32_t seg_max_mid;      /* ... of the mid class, its first stretch included (2 = the rest of the column in one piece) */
    int32_t seg_budget_mid, seg_budget_big;   /* segments per column = min(seg_max, budget / columns of the class): a few
                                               * thousand wavefronts of row segments fill the chip, more only cost folds */
    int32_t phase1_chunks;    /* mid class: 64-row chunks run unsplit before a surviving column is cut up */
    /* `lofreq uniq --use-det-lim` (lofreq_uniq.c:274-333): per column the assumed allele frequency; the first alt
     * count becomes (int)(af * n_err_probs), the others 0, and only the 'N' reference gate applies.  null = off */
    const float *detlim_af;
    int32_t lazy_strand;      /* 1: the count kernel skips the strand planes; lfq_strand_* fill them where a record is emitted */
    int32_t approx_n;         /* > 0: columns with more error probabilities than this pass the Poisson gate first (snpcaller.c:1131) */
    int32_t sparse_counts;    /* 1: the shared-wavefront count kernel writes the dense entry of a column only if it is tested (nothing
                               * on the device reads the others: lfq_set_dense_counts) */
    int32_t pad_;
};

struct LfqTracksDev {
    const uint8_t *nt, *bq, *baq, *mq, *sq;
    const uint64_t *col_off;
    const uint8_t *ref_base;
    const int32_t *coverage_plp, *num_bases;
    int64_t ncols;
    int32_t nt_packed;        /* LFQ_TRACKS_NT_PACKED: two observations per byte in the nt track (lofreq_amd.h) */
    int32_t pad_;
};

/* one tested column, as the DP kernels consume it: everything a wavefront needs to start the column
 * in a single 32-byte record, laid out in work-list order so that it can be prefetched */
struct LfqEntry {
    uint64_t off0;            /* first observation */
    int32_t n_obs;
    int32_t col;
    int32_t prefix;           /* inclusive count of tested columns up to this one (running Bonferroni) */
    int32_t kmax;
    int16_t median_ref_bq;
    uint8_t ref_code;         /* 0..3 */
    uint8_t pad_;
    int32_t pad2_;
};

/* ---- row-split ("long") columns ------------------------------------------------------------------
 * The rows of a column are independent Bernoulli trials, so the count distribution of the whole column is
 * the convolution of the distributions of disjoint row ranges.  A column that is still alive after a
 * first stretch of rows (or is known to be long: the big class) is cut into up to LFQ_SEG_MAX row
 * segments that run concurrently from the identity distribution; a combine step folds the segment
 * distributions (cells 0..K-1 plus the absorbing tail cell K) back together.  This turns one
 * N-row latency chain into N/R rows + (R-1) K^2/2-term convolutions. */
struct LfqSegCell {
    double v;                 /* mantissa in [0.5, 1) or 0 */
    int32_t e;                /* binary exponent: value = v * 2^e */
    int32_t pad_;
};

struct LfqLong {              /* 128 bytes */
    uint64_t off0;            /* the column, as in its LfqEntry */
    int64_t bonf;             /* running Bonferroni factor at this column */
    int64_t cell0;            /* segment r, cell k lives at pool[cell0 + r * (K + 1) + k] */
    int32_t n_obs, col;
    int32_t K;                /* recurrence size (largest allele count that is not an 80-bit underflow) */
    int32_t n_seg;            /* segments incl. segment 0 */
    int32_t ch_begin;         /* first chunk of the split range */
    int32_t phase1;           /* 1: segment 0 is the state the wave kernel had reached at ch_begin */
    uint32_t uf_mask;
    int32_t force_fe;
    int32_t pruned;           /* set by any segment whose own tail already exceeds the pruning threshold */
    int32_t rows;             /* kept rows processed so far (diagnostic, lfq_col_pvals.dp_rows) */
    int16_t median_ref_bq;
    uint8_t ref_code;
    uint8_t pad0_;
    int32_t pad1_;
    double uf_bound[3];
    int64_t pad_[4];
};

#define LFQ_SEG_MAX 8
/* a row segment always runs on ONE wavefront, with as many cells per lane as its K needs: the classes are
 * K <= 63 (1 cell per lane), <= 252 (4), <= 504 (8), <= 1008 (16), <= 2016 (32) -- (K + C) / C <= 64 lanes
 * incl. alignment.  More cells per lane = fewer instructions
 * per cell (the per-row overhead is shared), which is what counts once the segments provide the parallelism. */
#define LFQ_SEG_CLASSES 5
#define LFQ_SEG_MIN_CHUNKS 16     /* no segment shorter than this many 64-row chunks */
#ifndef LFQ_SEG_MIN_CHUNKS_SHORT
#define LFQ_SEG_MIN_CHUNKS_SHORT 4  /* ... of a column of less than LFQ_SEG_SHORT_BELOW chunks (lfq_split_plan) */
#endif
#define LFQ_SEG_SHORT_BELOW 48
#define LFQ_PHASE1_CHUNKS 8       /* mid class: rows run unsplit before a surviving column is cut up (measured: 4..8 best) */
#define LFQ_SPLIT_MAX_K 2016      /* 63 * 32: one wavefront at 32 cells per lane; the combine kernel keeps two
                                   * (K+1)-cell distributions in LDS */

/* largest K a screen-kernel variant with lfq_khist_thr(f) + 1 cells per lane serves (lfq_dp_screen_kernel<KREG>) */
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline int lfq_khist_thr(int f)
{
    return f == 0 ? 5 : f == 1 ? 7 : f == 2 ? 9 : f == 3 ? 11 : f == 4 ? 15 : f == 5 ? 23 : 31;
}

/* work lists and counters produced by the scan kernels, consumed by the DP kernel */
struct LfqWork {
    int32_t *tested_prefix;   /* [ncols] inclusive count of tested columns up to and incl. c */
    LfqEntry *entries;        /* this segment's work list: [light | mid | big] = [kmax < LFQ_MID_K | < LFQ_BIG_K | rest] */
    int32_t *gcounters;       /* batch-wide counters shared by all segments, see LFQ_GC_* */
    int32_t *counters;        /* [16] of this segment, see LFQ_CNT_* */
    int32_t *block_sums;      /* scan scratch */
    LfqLong *longs;           /* [long_cap]: LFQ_SEG_CLASSES lists of long_cap / LFQ_SEG_CLASSES records each */
    int32_t *unsplit;         /* big-list indices of the big columns that run unsplit (lfq_dp_big_kernel) */
    LfqSegCell *pool;         /* [pool_cells] segment distributions, bump-allocated (LFQ_CNT_POOL) */
    int32_t long_cap;
    int32_t pool_cells;
};

#define LFQ_MID_K 64          /* K+1 cells no longer fit one cell per lane */
#define LFQ_BIG_K 250         /* K+1 (+alignment) cells no longer fit one 64x4 strip: strip pipeline */
#define LFQ_NCOUNTERS 320
#define LFQ_MAX_SEGMENTS 1      /* launch sequences per batch (per-segment counters and events are laid out for this many) */
#define LFQ_GC_PVALS 0         /* records appended to the sparse output */
#define LFQ_GC_OVERFLOW 1
#define LFQ_GC_TESTED 2        /* running total of tested columns (carry between segments) */
#define LFQ_GC_MAXDEPTH 3
#define LFQ_GC_CELLS 8         /* and 9: uint64, DP cells processed = sum over the kept rows n of min(n, K) (SURVEY 8d secondary) */
#define LFQ_GC_ROWS 10         /* and 11: uint64, kept rows the DP kernels processed */
#define LFQ_GC_APPROX_PRUNED 13 /* tested columns the Poisson gate (-t) gave up */
#define LFQ_GC_SCREEN_RETRY 12 /* light columns the screen kernel handed to the one-column-per-wavefront kernel */
#define LFQ_CNT_TESTED 0
#define LFQ_CNT_BIG 1
#define LFQ_CNT_MID 2
#define LFQ_CNT_LIGHT 3
#define LFQ_CNT_HEAD 4        /* dequeue head of the big-column list */
#define LFQ_CNT_CARRY_IN 15    /* tested columns in earlier segments of the batch */
#define LFQ_CNT_HEAD_LIGHT 12  /* dynamic work distribution of the wave-per-column kernels */
#define LFQ_CNT_HEAD_MID 13
#define LFQ_CNT_UNSPLIT 5       /* big columns left to lfq_dp_big_kernel */
#define LFQ_CNT_HEAD_PREP 6
#define LFQ_CNT_POOL 7         /* cells handed out from LfqWork::pool */
#define LFQ_CNT_HEAD_COMB 24        /* and 25: one head per fold kernel mode */
#define LFQ_CNT_HEAD_FOLD 48        /* and 49: the wavefront-per-column fold kernel's heads */
#define LFQ_CNT_KLE7 26             /* light columns with K <= 7 / <= 15 / <= 31 (coarse histogram of the scan) */
#define LFQ_CNT_KLE15 27
#define LFQ_CNT_KLE31 28
#define LFQ_CNT_XHEAD 64       /* screen kernel: dequeue heads of the eight XCD slices of the light list, one per 128-byte
                                * line (head x at counters[LFQ_CNT_XHEAD + 32 x]): atomics on one line serialise */
#define LFQ_CNT_KHIST 40            /* .. 46: light columns with K <= lfq_khist_thr(f): the screen kernel's register variants */
#define LFQ_NKHIST 7
#define LFQ_CNT_LONG0 16       /* +class: row-split columns per cells-per-lane class (LFQ_SEG_CLASSES) */

/* ---- experiment / debugging knobs ----------------------------------------------------------------
 * Environment variables, read ONCE per process (first lfq_create / first use) into this struct; no entry point
 * calls getenv() on its own.  All optional; DESIGN.md "Environment knobs" documents them. */
/* Read bases are codes: 0..3 = A, C, G, T, 4 = N (seq_nt16_int of the BAM base; everything the HMM, the pileups and the tests
 * look at), and -- round 6 -- 5..15 = the other letters of htslib's seq_nt16_str in its order, "=MRSVWYHKDB".  Wherever the
 * reference compares the LETTER of a read base with a reference letter (idaq's repeat scan bam_md_ext.c:197, count_cigar_ops
 * samutils.c:486-489) or prints it (the key of an insertion, plp.c:1046-1060) an ambiguity code now is its own letter instead of
 * N; everywhere else a code above 3 behaves like N, as the reference's seq_nt16_int / bam_nt16_nt4_table do. */
#define LFQ_SEQ_LETTERS "ACGTN=MRSVWYHKDB"
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline char lfq_seq_letter(unsigned code)
{
    return code > 15u ? 'N' : LFQ_SEQ_LETTERS[code];
}

struct LfqKnobs {
    int timing;                /* LFQ_TIMING: host phases of the layer-2 calls to stderr */
    int single_stream;         /* LFQ_SINGLE_STREAM: every kernel on the caller's stream (rocprofv3 --pmc passes) */
    int no_sb_precompute;      /* LFQ_NO_SB_PRECOMPUTE */
    int debug_sync;            /* LFQ_DEBUG_SYNC: serialise and name the stages */
    int skip_light, skip_mid, skip_big;   /* LFQ_DEBUG_SKIP=light,mid,big: run the DP classes in isolation */
    int light_kernel;          /* LFQ_LIGHT_KERNEL=wave: 2 = one light column per wavefront instead of the screen kernel (the
                                * kernel that serves K >= 32 anyway); 0 = screen (one light column per lane) */
    int screen_waves_per_cu;   /* LFQ_SCREEN_WAVES_PER_CU (-1 = auto: 4, and for a context that queues its batches without a gate 1 for deep / 2 for
                                * shallow batches): the screen is latency-bound per column, more wavefronts only crowd the other chains */
    int screen_rounds;         /* LFQ_SCREEN_ROUNDS (24): 16-row windows before a light column goes to the retry kernel */
    int phase1_chunks;         /* LFQ_PHASE1_CHUNKS */
    int seg_max;               /* LFQ_SEG_MAX: both classes; LFQ_SEG_MAX_BIG / LFQ_SEG_MAX_MID: one of them (-1 = not given) */
    int seg_max_mid;
    int seg_budget_mid, seg_budget_big;   /* LFQ_SEG_BUDGET_MID (4096), LFQ_SEG_BUDGET_BIG (4096) */
    int split_pool_cells;      /* LFQ_SPLIT_POOL_CELLS (8 Mi; 0 disables the row split) */
    long count_multi_below;    /* LFQ_COUNT_MULTI_BELOW (4096) */
    int count_waves_per_wg;    /* LFQ_COUNT_WAVES_PER_WG (16; 4, 8): columns per workgroup of the one-column-per-wavefront count kernel */
    int count_ahead_deep;      /* LFQ_COUNT_AHEAD_DEEP (2; 3, 4): chunks of 16 observations in flight per lane of the lean count kernel */
    int count_cols_per_wave;   /* LFQ_COUNT_COLS_PER_WAVE (1; 2, 4): columns a wavefront of the lean count kernel takes one after the other,
                                * all their headers requested when it starts */
    int big_on_side;           /* LFQ_BIG_ON_SIDE: the unsplit big columns behind the big chain instead of on the count kernel's stream
                                * (a context with LFQ_GATE_NONE runs that way by itself) */
    int baq_one_variant;       /* LFQ_BAQ_ONE_VARIANT: every wavefront of the plain narrow-band BAQ launches through the instantiation with the N case */
    int pileup_tiles;          /* LFQ_PILEUP_TILES (1): SNV pileup of sorted reads by tiles of 64 positions; 0 = a wavefront per position */
    long host_loop_threads;    /* LFQ_HOST_LOOP_THREADS (8): threads (caller included) a host loop over reads / positions / events is cut for, at most 16 */
    long sb_par_min_cost;      /* LFQ_SB_PAR_MIN_COST (4000; 20000 until the end of round 5: a 200x batch's 700 tables of 10-30 alt bases were 0.22 ms of a host-paced 0.85 ms step on one thread): summed alt counts of the strand-bias tests of a batch from which they go to the host pool */
    long count_lpg4_below, count_lpg8_below;   /* LFQ_COUNT_LPG4_BELOW (320), LFQ_COUNT_LPG8_BELOW (900): deepest column up to which 4 / 8 lanes share a column */
    long host_spin_us;         /* LFQ_HOST_SPIN_US (2000): how long the helper threads of the host loops spin for the next loop; -1 = no pool */
    int sync_upload;           /* LFQ_SYNC_UPLOAD: 1 = lfq_readset_create waits for its copies itself (no helper thread), 2 = helper thread whatever the size */
    int host_threads;          /* LFQ_HOST_THREADS: -1 = from the core count */
    long host_par_min;         /* LFQ_HOST_PAR_MIN (200000): reads / positions from which the host loops of the read-set steps split over threads */
    int local_world_size;      /* LOCAL_WORLD_SIZE (torchrun): processes sharing this host's cores, >= 1 */
    int indel_host_pack;       /* LFQ_INDEL_HOST_PACK */
    int pileup_atomic;         /* LFQ_PILEUP_ATOMIC: read-major pileup kernels even for sorted reads */
    int baq_lds;               /* LFQ_BAQ_LDS (1) */
    int baq_idaq_beside;       /* LFQ_BAQ_IDAQ_BESIDE (0): the narrow-band reads with indels on a side stream beside the plain launches (a lone BAQ + IDAQ call 6.2 -> 5.3 ms per 400 K reads, the reads -> VCF chain 37.7 -> 39.0 ms per region: off) */
    long baq_scratch_mb;       /* LFQ_BAQ_SCRATCH_MB: -1 = from free HBM */
    int tail_light;            /* LFQ_TAIL_LIGHT (1): where the light chain records the device's tail event (what the next batch's count
                                * kernel waits for under LFQ_GATE_TAIL): 0 = behind the retry kernel, 1 = behind the screen kernel (the
                                * retry kernel is a few hundred latency-bound wavefronts: beside the count kernel like the folds), 2 = behind the scan */
    int count_shallow_wgs_none; /* LFQ_COUNT_SHALLOW_WGS_NONE (2): workgroups per CU of the shared-wavefront count kernel of a context whose batches are
                                * queued without a gate (0 = as many as fit: four) -- the other half of every SIMD is what another batch's DP kernels run in */
    int count_lean_lds_pad;    /* LFQ_COUNT_LEAN_LDS_PAD (0): bytes of unused dynamic LDS per workgroup of the lean count kernel */
    int count_shallow_lds_pad; /* LFQ_COUNT_SHALLOW_LDS_PAD (0): bytes of unused dynamic LDS per workgroup of the shared-wavefront count kernel */
    int join_on_side;          /* LFQ_JOIN_ON_SIDE (1): a batch's join (strand counts of its records, counters to the host) on the big chain's stream instead of the light chain's */
    int private_stream;        /* LFQ_PRIVATE_STREAM (0): lfq_create gives every context a launch stream of its own (lfq_set_private_stream) */
    int heavy_after_screen;    /* LFQ_HEAVY_AFTER_SCREEN (1): the strand counts of the heavy columns (host Fisher precompute) behind the screen
                                * kernel instead of in front of it: one launch less between the scan and the light chain */
};
const LfqKnobs &lfq_knobs(void);
/* CPUs this process may actually use: the affinity mask and the cgroup's cpu.max quota, not the machine's core count
 * (a container that is granted 16 of 256 cores must not start 63 helper threads); >= 1 */
unsigned lfq_cpu_budget(void);

/* ---- strand-bias precompute (host, lfq_host.cpp) ---------------------------------------------------
 * report_var's Fisher test (lofreq_call.c:117-129) depends only on the DP4 counts, which are final after
 * the count kernel.  While the DP kernels run, a host thread of the context computes it for the columns with many alt
 * bases (the expensive ones) into a process-wide content-addressed cache that lfq_finalize_pvals consults.
 * Purely an optimisation: a miss is computed on the spot, bit-identically.  Synchronisation is per context
 * (lfq_ctx::sb_pending): a context waits for its own precompute in lfq_call_snvs_collect, never for another's. */
/* tuples: n x {ref_fw, ref_rv, alt_fw, alt_rv}; all-zero tuples are skipped. */
void lfq_sb_precompute(const int32_t *tuples, int64_t n);

/* ---- BAQ (lfq_baq.hip) ----------------------------------------------------------------------------------- */
struct LfqBaqGeom {            /* 12 bytes per read, built on the host from the CIGAR (bam_md_ext.c:312-380) and uploaded */
    int32_t xb, l_ref;         /* reference window [xb, xb + l_ref) */
    int32_t bw;                /* band width handed to the HMM */
};
struct LfqBaqRead {            /* what a kernel works with: the geometry + the read's entries of the resident arrays */
    int32_t pos;               /* bam1_core_t.pos */
    int32_t l_qseq;
    int32_t xb, l_ref;
    int32_t bw;
    int32_t n_cigar;
    int64_t cigar_off;
};

struct LfqBaqArgs {
    const LfqBaqGeom *geom;    /* [n] */
    const int32_t *pos;        /* [n] */
    const int64_t *cigar_off;  /* [n+1] */
    const int64_t *seq_off;    /* [n+1] */
    const uint32_t *cigar;
    const uint8_t *seq, *qual; /* 0..4 / phred */
    const uint8_t *ref;        /* contig, ASCII */
    uint8_t *lb_out;           /* [seq_off[n]] */
    double *scratch;           /* per wavefront: F[rows][W][64], B[2][W][64], S[rows + 2][64], 1/S[rows + 2][64] */
    int32_t *expect;           /* per wavefront: [rows][64] expected reference offset of a matched base, INT32_MIN otherwise */
    uint8_t *tmp8;             /* per wavefront: [2][rows][64] running maxima of the extended BAQ */
    const float *qual2prob;    /* [256] pow(10, -q/10.) as float, computed on the host (kprobaln_ext.c:121-123) */
    int64_t n_reads;
    int32_t rows, W;           /* scratch geometry: rows >= max l_qseq + 1, W >= max (2 bw + 1) * 3 + 6 */
    int32_t baq_extended;
    float par_d, par_e;        /* kpa_ext_par_t.d / .e (kprobaln_ext.c:48-51): gap open, gap extension */
    int32_t first_read;        /* reads [first_read, first_read + n_launch) of the arrays (of `order`, if given) */
    const int32_t *order;      /* read indices, narrow-band reads first; null = identity */
    int32_t max_lref;          /* longest reference window among the narrow-band reads (LDS sizing) */
    int32_t lds_rows;          /* longest narrow-band read + 1: rows of the per-row scalars the register kernel keeps in LDS */
    /* indel alignment qualities (idaq, bam_md_ext.c:73-248); all null / 0 = not requested */
    uint8_t *ai_out, *ad_out;  /* [seq_off[n]] bytes of the ai / ad tags ('~' = nothing) */
    uint8_t *tag_flags;        /* [n] bit 0: the read gets an ai tag, bit 1: an ad tag */
    int32_t *itab;             /* per wavefront: [LFQ_BAQ_MAX_INDELS][4][64] kept indels: type|qpos, k0, rep, term offset */
    double *terms;             /* per wavefront: [LFQ_BAQ_MAX_TERMS][64] posterior terms, summed in the reference's order */
    uint8_t *nflag;            /* per wavefront of the launch: it meets an N (lfq_baq_nflag_kernel); null = one instantiation for all */
};
#define LFQ_BAQ_MAX_INDELS 64
#define LFQ_BAQ_MAX_TERMS 1024
#define LFQ_BAQ_LDS_CELLS 51    /* row width (cells) up to which a read runs in the LDS variant of the kernel */
#define LFQ_BAQ_BAND8_CELLS 57  /* row width (cells) of band 8: (2 * 8 + 1) * 3 + 6 */
#define LFQ_BAQ_LDS_BAND 15      /* cells (reference positions) per row of such a read: 2 * 7 + 1 */
#define LFQ_BAQ_LDS_MAX_LREF 300 /* ... and whose reference window is this short: codes (32 B per base pair and wavefront) + row
                                  * scalars (128 B per query base) + 1 KiB stay below 64 KiB of LDS per wavefront */
/* nmode (launches with a.nflag): 0 = flag kernel and both instantiations; 1 = flag kernel and the instantiation without the
 * N case; 2 = only the one with it (the flagged wavefronts of an earlier nmode 1 call with the same arguments) */
int lfq_launch_baq(const LfqBaqArgs &a, int64_t n_launch, int lds, void *stream, int nmode = 0);

/* ---- device-side pileup (lfq_pileup.hip) ------------------------------------------------------------------ */
struct LfqPileupArgs {
    int64_t n_reads;
    const int32_t *pos;
    const int64_t *cigar_off, *seq_off;
    const uint32_t *cigar;
    const uint8_t *seq, *qual, *baq;      /* baq: lb tag bytes (BAQ + 33) or null */
    const uint8_t *mapq, *reverse;        /* [n] */
    int64_t begin, width;                 /* region [begin, begin + width) */
    int32_t min_plp_bq;
    int32_t *cov, *nb, *cursor;           /* [width] per reference position */
    const int32_t *col_index;             /* [width] position -> column (covered positions only) */
    const uint64_t *col_off;              /* [ncols + 1] */
    uint8_t *t_nt, *t_bq, *t_baq, *t_mq;  /* packed tracks */
    const uint8_t *sq;                    /* [n] per-read source quality byte or null */
    uint8_t *t_sq;
    const int32_t *pmax_end;              /* [n] sorted reads: max over reads 0..r of the end coordinate (exclusive); null = unsorted input */
};
int lfq_launch_pileup_count(const LfqPileupArgs &a, void *stream);
int lfq_launch_pileup_columns(const LfqPileupArgs &a, int scatter, void *stream);

/* the indel fields of compile_plp_col (plp.c:1019-1192), dense part on the device */
struct LfqPlpIndelArgs {
    int64_t n_reads;
    const int32_t *pos;
    const int64_t *cigar_off, *seq_off;
    const uint32_t *cigar;
    const uint8_t *bi, *bd;               /* per base tag bytes (quality + 33) or null */
    const uint8_t *tag_flags;             /* [n] bit 0: read has BI, bit 1: BD; null = wherever the array exists */
    const uint8_t *mapq, *reverse;        /* [n] */
    int64_t begin, width;
    int32_t min_plp_idq;
    /* per reference position of the region */
    int32_t *cov, *tails, *non_indels, *n_ins, *n_dels, *non_ins_fw, *non_del_fw;
    int32_t *ne_qsum[2];                  /* column-major kernel: sum of the indel qualities of the reads without an event (consensus, plp.c:1236) */
    /* scatter pass: positions with at least one event get the (quality, MAPQ) of their non-event reads */
    const int64_t *ne_off[2];             /* [width] start of the position's slice per side, -1 = not wanted */
    int32_t *cursor[2];                   /* [width] */
    int16_t *ne_q[2], *ne_mq[2];
    const int32_t *pmax_end;              /* sorted reads: see LfqPileupArgs; selects the column-major kernel */
};
int lfq_launch_plp_indel(const LfqPlpIndelArgs &a, int scatter, void *stream);
int lfq_launch_plp_indel_columns(const LfqPlpIndelArgs &a, int scatter, void *stream);
int lfq_launch_flag_merge(uint8_t *fl, const uint8_t *tag, int64_t n, void *stream);
int lfq_launch_skip_columns(int32_t *nb, const uint8_t *skip, int64_t n, void *stream);
int lfq_launch_pack_nt(const uint8_t *nt_bytes, uint8_t *nt_packed, int64_t n_obs, void *stream);
/* columns of the SNV pileup from the per-position counters (lfq_pileup.hip): tile sums + their scan + the totals
 * (totals[0] columns, [1] observations, [2] deepest column), then the per-column arrays */
#define LFQ_PLP_COMPACT_TILE 4096
int lfq_launch_plp_compact_sums(const int32_t *cov, const int32_t *nb, int64_t width, int64_t *tile_cols, uint64_t *tile_obs,
                                int32_t *tile_max, int64_t *totals, void *stream);
int lfq_launch_plp_compact_apply(const int32_t *cov, const int32_t *nb, int64_t width, int64_t begin, const uint8_t *ref,
                                 int64_t ref_len, const int64_t *tile_cols, const uint64_t *tile_obs, const int64_t *totals,
                                 int32_t *col_index, uint64_t *col_off, uint8_t *ref_base, int32_t *cov_c, int32_t *nb_c,
                                 int64_t *col_pos, void *stream);
int lfq_launch_gather2(const uint8_t *a, const uint8_t *b, const int64_t *idx, int64_t n, uint8_t *oa, uint8_t *ob, void *stream);

/* indel pseudo-columns built on the device from the resident quality arrays of lfq_readset_pileup_indels */
struct LfqIndelTestDesc {       /* 48 bytes: one tested event (plp_to_ins_errprobs / plp_to_del_errprobs, snpcaller.c:502-623) */
    int64_t out_off;            /* first observation of the pseudo-column in the output tracks */
    int64_t ne_off;             /* the column's reads without an event of this side: slice of ne_q / ne_mq */
    int64_t rd_begin;           /* all event reads of the column (every event of this side) */
    int32_t ne_len, rd_len;
    int32_t me_begin, me_len;   /* the tested event's reads, relative to rd_begin */
    int32_t side, pad_;
};
struct LfqIndelPackArgs {
    const LfqIndelTestDesc *tests;
    int64_t n_tests;
    const int16_t *ne_q[2], *ne_mq[2];                  /* device, per side */
    const int16_t *rd_q[2], *rd_aq[2], *rd_mq[2], *rd_sq[2];
    int32_t use_mq, use_sq, use_aq, pad_;
    uint8_t *nt, *bq, *baq, *mq, *sq;                   /* output tracks */
};
int lfq_launch_indel_pack(const LfqIndelPackArgs &a, void *stream);

/* source quality (lfq_srcq.hip) */
#define LFQ_DBL_EPSILON 2.220446049250313e-16
#define LFQ_SRCQ_LDS_CELLS 768      /* K below this: the DP cells of a read live in LDS, else in its scratch slice */
#define LFQ_SRCQ_NA 0               /* count_cigar_ops found nothing: source_qual returns -1 */
#define LFQ_SRCQ_PERFECT 1          /* at most one non-match: PROB_TO_PHREDQUAL(LDBL_MIN) */
#define LFQ_SRCQ_VALUE 2            /* prob[r] = P(X = K - 1) of the row poissbin stopped at */
struct LfqSrcqArgs {
    int64_t n_reads;
    const int32_t *pos;
    const int64_t *cigar_off, *seq_off;
    const uint32_t *cigar;
    const uint8_t *seq, *qual;
    const char *ref;
    int64_t ref_len;
    const uint8_t *ign;                 /* [ref_len] or null */
    int32_t nonmatch_qual, min_bq;
    double *scratch;                    /* 2 * scratch_cells doubles per wavefront of the launch, or null */
    int64_t scratch_cells;
    double *prob;                       /* [n] */
    uint8_t *status;                    /* [n] */
};
int lfq_launch_srcq(const LfqSrcqArgs &a, const LfqLuts *d_luts, int n_blocks, void *stream);
int lfq_launch_pileup_scatter(const LfqPileupArgs &a, void *stream);

/* kernel launchers (lfq_kernels.hip); all asynchronous on `stream` */
int lfq_launch_strand_heavy(const LfqTracksDev &t, const LfqWork &w, lfq_col_counts *d_counts, int32_t *tuples_mapped,
                            int32_t *n_mapped, int cap_entries, int min_alt, void *stream);
int lfq_launch_strand_pvals(const LfqTracksDev &t, lfq_col_pvals *d_pvals, const int32_t *d_n_pvals, int64_t cap, int n_blocks,
                            void *stream);
int lfq_launch_gather_heavy(const LfqWork &w, const lfq_col_counts *d_counts, int32_t *tuples_mapped,
                            int32_t *n_mapped, int cap_entries, int min_alt, void *stream);
int lfq_launch_maxdepth(const LfqTracksDev &t, int32_t *d_gcounters, void *stream);
int lfq_launch_ntcount(const LfqTracksDev &t, int32_t *d_out, void *stream);
bool lfq_count_is_shallow(const LfqTracksDev &t, const LfqParams &p, int64_t max_col_obs);
int lfq_launch_count(const LfqTracksDev &t, int64_t c0, int64_t c1, const LfqParams &p, const LfqLuts *d_luts,
                     lfq_col_counts *d_counts, uint8_t *d_flags, int64_t max_col_obs, void *stream, int shallow_wgs_per_cu = 0);
/* (shallow_wgs_per_cu: workgroups per CU the shared-wavefront count kernel may have resident, 0 = as many as fit) */
int lfq_launch_scan(const LfqTracksDev &t, int64_t c0, int64_t c1, const uint8_t *d_flags,
                    const lfq_col_counts *d_counts, const LfqWork &w, void *stream, bool relist = false);
/* -t / --approx-threshold (snpcaller.c:1128-1142): clears the flag byte of the listed columns the Poisson gate gives up;
 * lfq_launch_scan(..., relist = true) then rebuilds the work list without them.  d_mu: one double per column of the segment */
int lfq_launch_approx_gate(const LfqTracksDev &t, const LfqParams &p, const LfqLuts *d_luts, const lfq_col_counts *d_counts,
                           const LfqWork &w, int64_t ncols_seg, double *d_mu, uint8_t *d_flags, void *stream);
int lfq_launch_dp_light(const LfqTracksDev &t, const LfqParams &p, const LfqLuts *d_luts,
                        const lfq_col_counts *d_counts, const LfqWork &w, lfq_col_pvals *d_pvals,
                        int64_t pvals_capacity, int n_waves, void *stream);
/* kreg_hint: cells per lane of the screen kernel variant to launch (from the K histogram of the context's previous
 * batch; any value is correct, a poor one only sends more columns to the retry kernel); 64 = the light class needs
 * whole wavefronts */
int lfq_launch_dp_quad(const LfqTracksDev &t, const LfqParams &p, const LfqLuts *d_luts,
                       const lfq_col_counts *d_counts, const LfqWork &w, uint8_t *d_retry, lfq_col_pvals *d_pvals,
                       int64_t pvals_capacity, int n_waves, int kreg_hint, void *stream, int phase = 0);
/* (phase: 0 = screen + retry, 1 = the screen kernel only, 2 = the retry kernel only) */
int lfq_launch_dp_mid(const LfqTracksDev &t, const LfqParams &p, const LfqLuts *d_luts,
                      const lfq_col_counts *d_counts, const LfqWork &w, lfq_col_pvals *d_pvals,
                      int64_t pvals_capacity, int n_waves, void *stream);
int lfq_launch_dp_big(const LfqTracksDev &t, const LfqParams &p, const LfqLuts *d_luts,
                      const lfq_col_counts *d_counts, const LfqWork &w, lfq_col_pvals *d_pvals,
                      int64_t pvals_capacity, double *d_scratch, int64_t scratch_doubles_per_block,
                      int n_blocks, void *stream);
int lfq_launch_dp_big_prep(const LfqTracksDev &t, const LfqParams &p, const LfqLuts *d_luts,
                           const lfq_col_counts *d_counts, const LfqWork &w, lfq_col_pvals *d_pvals,
                           int64_t pvals_capacity, int n_blocks, void *stream);
int lfq_launch_dp_seg(int mode, const LfqTracksDev &t, const LfqParams &p, const LfqLuts *d_luts,
                      const LfqWork &w, int n_waves, void *stream);
int lfq_launch_dp_combine(int mode, const LfqParams &p, const lfq_col_counts *d_counts, const LfqWork &w,
                          lfq_col_pvals *d_pvals, int64_t pvals_capacity, int n_blocks, void *stream);
int lfq_launch_synth(const struct lfq_synth_spec *d_spec_host, int64_t col_begin, int64_t ncols,
                     uint8_t *d_nt, uint8_t *d_bq, uint8_t *d_baq, uint8_t *d_mq, uint64_t *d_col_off,
                     uint8_t *d_ref_base, int nt_packed, void *stream);

#endif
