/*
 * lofreq_amd.h -- C ABI of the MI355X-native LoFreq per-column SNV calling path.
 *
 * This is the drop-in boundary for the path
 *     plp_to_errprobs -> qsort -> snpcaller/poissbin/pruned_calc_prob_dist -> Bonferroni emit test
 * of `lofreq call` (reference: src/lofreq/lofreq_call.c:735-879 call_snvs,
 * src/lofreq/snpcaller.c:346, 831, 1020, 1075).  Plain pointers and sizes only; no C++ or
 * torch types.  INTEGRATION.md shows the binding a LoFreq maintainer adds behind the
 * `void (*plp_proc_func)(const plp_col_t*, void*)` callback of mpileup() (plp.h:159-163).
 *
 * Layers:
 *   (1) lfq_snv_batch_device  -- kernels only, device pointers, asynchronous.  Replaces, for a
 *       batch of columns, the reference's inner numeric API plp_to_errprobs()+qsort()+snpcaller()
 *       (snpcaller.h:72-75, 97-102).
 *   (2) lfq_call_snvs_batch   -- the call_snvs() loop (lofreq_call.c:735-879) over a batch:
 *       runs (1), then does the 80-bit p-value conversion with the reference's errno/fenv clamp
 *       (snpcaller.c:1047-1059, 1169-1188), the running-Bonferroni emit test (lofreq_call.c:832),
 *       AF/DP4/HQA/QUAL (lofreq_call.c:835-863) and strand bias (lofreq_call.c:117-129) on the host,
 *       and returns one record per reported variant, in column order.
 *   (3) lfq_format_snv_record / lfq_filter_records / lfq_snvqual_thresh -- VCF text (vcf.c:469-497,
 *       608-629) and the final `lofreq filter` step that `lofreq call` runs on its own output
 *       (lofreq_call.c:1506-1538; lofreq_filter.c).
 *
 * All functions return 0 on success or a negative lfq_status.  Errors never fall back to a CPU
 * implementation: if no HIP device / kernel image is available the call fails.
 */
#ifndef LOFREQ_AMD_H
#define LOFREQ_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: lfq_conf grew to 80 bytes (approx_threshold_n), lfq_dp_work gained n_approx_pruned, lfq_filter_records_ex.
 * A caller compiled against another version must not run: compare lfq_abi_version() with this value once, as the
 * bindings in integration/ and the Python loader do. */
/* 4: lfq_set_batch_gate, lfq_last_baq_times; lfq_call_snvs_collect refuses h_counts for a batch whose dense entries are sparse. */
/* 5: lfq_set_private_stream. */
#define LFQ_ABI_VERSION 6

typedef enum lfq_status {
    LFQ_OK = 0,
    LFQ_ERR_INVALID = -1,      /* bad argument */
    LFQ_ERR_NO_DEVICE = -2,    /* no HIP device or HIP runtime error */
    LFQ_ERR_NOMEM = -3,
    LFQ_ERR_CAPACITY = -4,     /* caller-provided output capacity too small */
    LFQ_ERR_UNSUPPORTED = -5,  /* option the reference rejects too (def_alt_jq == -1) */
    LFQ_ERR_HIP = -6
} lfq_status;

/* varcall_conf_t flag bits (defaults.h:71-76) */
#define LFQ_USE_BAQ 1
#define LFQ_USE_MQ 2
#define LFQ_USE_SQ 4
#define LFQ_USE_IDAQ 8

/* ---- packed column batch ------------------------------------------------------------------
 * Struct-of-arrays, one byte per observation per track, columns concatenated (CSR offsets).
 * Built from plp_col_t's per-nucleotide int_varray_t arrays (plp.h:88-91):
 *   nt   bits 0..2 = nt4 code (0..3 = A,C,G,T; 4 = N, skipped like snpcaller.c:386),
 *        bit 3 = read on reverse strand (feeds fw_counts/rv_counts, plp.c:1007-1011)
 *   bq   base quality, 0..93 (plp.c:937-953)
 *   baq  base alignment quality 0..93; 255 = missing (-1, plp.c:956-962).  NULL track = BAQ off.
 *   mq   mapping quality as in the BAM record, 0..255 (255 = NA is handled as snpcaller.c:451 does)
 *   sq   source quality 0..253; 254 = anything larger (source_qual's 49314: probability 0.0); 255 = missing (-1).
 *        NULL track = no source quality.
 * Track base pointers must be 16-byte aligned and readable up to the next multiple of 16 bytes
 * past col_off[ncols]; col_off itself may be arbitrary (columns need not be aligned).
 *
 * LFQ_TRACKS_NT_PACKED (flags): the nt track holds two observations per byte ((n_obs + 7) / 8 * 4 bytes).  What the device
 * pileup (lfq_pileup_snv_tracks / lfq_readset_pileup_snv) hands out by default, what the plp_proc_func shim
 * (integration/lofreq_amd_shim.c) builds as the columns arrive, and what lfq_pack_nt_track makes of a byte track.
 * Observations are taken in groups of 8 (by their index in the track); byte k (k = 0..3) of a group's 4 bytes
 * carries observation k in its low nibble and observation 4 + k in its high nibble -- the even / odd nibbles of a
 * dword then line up with the group's two bq dwords.  The count kernel, the dominant one and HBM-bound, reads
 * 1.5 instead of 2 bytes per observation.
 */
#define LFQ_TRACKS_NT_PACKED 1
#define LFQ_Q_MISSING 255

typedef struct lfq_tracks {
    const uint8_t *nt;
    const uint8_t *bq;
    const uint8_t *baq;        /* may be NULL */
    const uint8_t *mq;
    const uint8_t *sq;         /* may be NULL */
    const uint64_t *col_off;   /* ncols + 1 entries */
    const uint8_t *ref_base;   /* ncols ASCII reference bases (plp_col_t.ref_base); no alignment required (a count kernel reads the aligned
                                * 32-bit word around a column's byte: same page, never past a mapping) */
    const int32_t *coverage_plp; /* ncols, or NULL: = observation count (plp_col_t.coverage_plp) */
    const int32_t *num_bases;    /* ncols, or NULL: = observation count (plp_col_t.num_bases) */
    int64_t ncols;
    int64_t max_col_obs;         /* deepest column of the batch, or 0 = unknown (costs one sync) */
    int64_t flags;               /* LFQ_TRACKS_NT_PACKED or 0 */
} lfq_tracks;

/* the SNV-path fields of varcall_conf_t (snpcaller.h:38-63); defaults: lfq_conf_init */
typedef struct lfq_conf {
    int32_t min_bq, min_alt_bq, def_alt_bq;
    int32_t min_jq, min_alt_jq, def_alt_jq;
    int32_t bonf_dynamic;
    int32_t min_cov;
    int64_t bonf_subst;        /* running Bonferroni factor; mutated by lfq_call_snvs_batch */
    float sig;                 /* float, like the reference */
    int32_t flag;
    int64_t num_snv_tests;     /* the global of lofreq_call.c:84; mutated */
    int64_t bonf_indel;        /* running indel Bonferroni factor (snpcaller.h:52); mutated by the indel calls */
    int64_t num_indel_tests;   /* the global of lofreq_call.c:85; mutated */
    /* -t / --approx-threshold (snpcaller.h:62, snpcaller.c:1128-1142): a column or indel test with more error
     * probabilities than this is given up without the exact test when the Poisson tail with the same mean, times the
     * Bonferroni factor, exceeds sig.  <= 0 (default -1, snpcaller.c:650): off.  The reference needs libgsl for it (a
     * build without aborts, :1118-1125); here the definition gsl_cdf_poisson_P implements is evaluated -- parity
     * unpinned, see DESIGN.md.  It never adds a call and, as in the reference, does not change the Bonferroni counts. */
    int32_t approx_threshold_n;
    int32_t pad_;
} lfq_conf;

/* dense per-column output of the counting kernel == plp_to_errprobs()'s integer outputs
 * (snpcaller.c:346-498) plus the strand counts report_var() needs.  64 bytes. */
typedef struct lfq_col_counts {
    int32_t n_err_probs;       /* #probabilities the reference would hand to snpcaller */
    int32_t alt_counts[3];     /* filtered alt counts, alleles in A,C,G,T order minus ref */
    int32_t alt_raw_counts[3]; /* unfiltered alt counts (AF numerator, lofreq_call.c:835) */
    int32_t alt_fw[3];         /* forward-strand part of alt_raw_counts */
    int32_t ref_fw, ref_rv;    /* strand counts of the reference base */
    int32_t kmax;              /* max(alt_counts) = num_failures handed to poissbin */
    uint8_t tested;            /* column reaches the Bonferroni bump (lofreq_call.c:794-801) */
    uint8_t gated;             /* 1 = skipped by the ref-N / coverage gates (lofreq_call.c:747,754,930) */
    uint8_t pad_[2];
    int32_t median_ref_bq;     /* median reference-base BQ when def_alt_bq == -1, else -1 */
    int32_t coverage;          /* coverage_plp used for this column (DP / AF denominator) */
} lfq_col_counts;

/* status of one allele's p-value */
enum {
    LFQ_PV_NONE = 0,        /* count 0 or column pruned: reference returns LDBL_MAX */
    LFQ_PV_LOG = 1,         /* logp valid: pvalue = expl(logp) (+ the reference's clamp on expl) */
    LFQ_PV_LOG_FECLAMP = 2, /* logp valid but the reference's tail-sum exp() chain underflows
                               (snpcaller.c:1169-1188): pvalue clamps to LDBL_MIN / LDBL_MAX */
    LFQ_PV_UNDERFLOW = 3    /* p-value proven below the 80-bit range (logp = an upper bound < -12200):
                               the reference's expl() underflows and it reports LDBL_MIN */
};

/* sparse output: one record per column whose main allele survives the pruning test
 * P(X>=K)*bonf > sig (snpcaller.c:950, 1155).  128 bytes. */
typedef struct lfq_col_pvals {
    int64_t col;               /* column index within the batch */
    int64_t bonf;              /* running Bonferroni factor at this column */
    double logp[3];            /* natural-log p-values, same allele order as the counts */
    uint8_t status[3];
    uint8_t ref_base;          /* 'A','C','G','T': the column's reference base, so that the host needs no track */
    uint8_t pad_[4];
    lfq_col_counts counts;     /* copy of the dense entry, so the host needs nothing else */
    int32_t dp_rows;           /* DP rows processed (diagnostic) */
    int32_t pad2_;
    int64_t reserved_;
} lfq_col_pvals;

typedef struct lfq_batch_stats {
    int64_t n_tested;          /* columns that passed the gates with >= 1 filtered alt base */
    int64_t n_pvals;           /* records written to the sparse output */
    int64_t n_obs;             /* observations in the batch */
} lfq_batch_stats;

/* one reported SNV == the arguments of report_var() (lofreq_call.c:862-864) */
typedef struct lfq_snv_record {
    int64_t col;
    int32_t qual;              /* PROB_TO_PHREDQUAL(pvalue) */
    int32_t dp;                /* coverage_plp */
    int32_t alt_raw_count;     /* AF = alt_raw_count / (float) dp */
    int32_t sb;                /* strand-bias phred (INT_MAX special case included) */
    int32_t ref_fw, ref_rv, alt_fw, alt_rv;
    int32_t hqa;               /* filtered alt count */
    char ref, alt;
    uint8_t pad_[2];
    long double pvalue;
} lfq_snv_record;

typedef struct lfq_ctx lfq_ctx;

/* --- lifetime --- */
int lfq_abi_version(void);
const char *lfq_strerror(int status);
void lfq_conf_init(lfq_conf *conf);                 /* init_varcall_conf, snpcaller.c:627-651 */
int lfq_create(lfq_ctx **ctx, int device_ordinal);  /* one context per GPU / stream */
/* Which GPU a worker PROCESS takes.  The reference's parallel wrapper forks one `lofreq call -r <bin>` per worker
 * (lofreq2_call_pparallel.py:640-667); with this every such process lands on a GPU of its own without the wrapper
 * knowing about GPUs:  LFQ_DEVICE (an ordinal, taken literally)  >  LOCAL_RANK (torchrun-style launchers, modulo the
 * device count)  >  the first free worker slot k of the node (a lock file <LFQ_SLOT_DIR or /tmp>/lofreq_amd.<uid>.slot<k>
 * held for the life of the process: device k mod n, so W concurrent workers spread over the n GPUs and a slot is
 * reused when its worker exits)  >  getpid() mod n.  n_devices <= 0: ask the HIP runtime.  Returns the ordinal
 * (>= 0) for lfq_create, or a negative lfq_status; *slot_out_or_null gets the slot (or -1). */
int lfq_device_count(void);
/* Pinned host memory for a producer's track buffers (hipHostMalloc): with it the uploads of lfq_call_snvs_submit(...,
 * tracks_on_device = 0) are DMA transfers at the link's rate that return at once -- pageable memory is staged.  NULL when
 * there is no device / no memory. */
void *lfq_host_alloc(size_t bytes);
void lfq_host_free(void *p);
int lfq_pick_device(int n_devices, int *slot_out_or_null);
void lfq_destroy(lfq_ctx *ctx);
int lfq_synchronize(lfq_ctx *ctx);

/* --- layer 1: kernels --- */
/* All pointers in `tracks` and the outputs are DEVICE pointers.  `d_counts` holds ncols entries;
 * `d_pvals` holds pvals_capacity entries.  `bonf_base` is conf->bonf_subst before this batch.
 * `stream` is a hipStream_t (NULL = the context's own stream).  Asynchronous; `stats` is filled
 * by lfq_batch_finish(), which waits for the batch.
 * ONE batch in flight per context: a batch ends on an internal stream (its DP kernels), and what the library itself
 * queues for the context afterwards (the next batch, a pileup or generator call that rewrites context-owned tracks) is
 * ordered behind the batch's last event -- but the caller's own writes to d_counts / d_pvals / the tracks are not:
 * call lfq_batch_finish() (or pass your stream, into which the batch is then joined) before touching them. */
int lfq_snv_batch_device(lfq_ctx *ctx, const lfq_conf *conf, const lfq_tracks *tracks,
                         lfq_col_counts *d_counts, lfq_col_pvals *d_pvals, int64_t pvals_capacity,
                         void *stream);
int lfq_batch_finish(lfq_ctx *ctx, lfq_batch_stats *stats);

/* --- layer 2: the call_snvs loop --- */
/* tracks_on_device != 0: `tracks` holds device pointers (data resident in HBM);
 * otherwise host pointers, copied to the device first.  Writes at most records_capacity
 * records (column order) and sets *n_records.  Updates conf->bonf_subst / num_snv_tests exactly
 * like the per-column loop of the reference.  `h_counts_or_null`: optional host copy of the dense
 * per-column counts (ncols entries). */
int lfq_call_snvs_batch(lfq_ctx *ctx, lfq_conf *conf, const lfq_tracks *tracks, int tracks_on_device,
                        lfq_snv_record *records, int64_t records_capacity, int64_t *n_records,
                        lfq_col_counts *h_counts_or_null, lfq_batch_stats *stats);

/* --- indel tests (call_indels, lofreq_call.c:619-726) on the same kernels ------------------------------
 * One "test" = one indel event of one column = snpcaller() on all reads of the column with
 * counts = {event count, 0, 0} (lofreq_call.c:306-426).  The caller packs each test as a
 * pseudo-column in the track format: bq = indel quality, baq = indel alignment quality of the tested
 * event's reads (255 elsewhere or with IDAQ off), mq, sq; nt = 1 ('C') for the tested event's reads, 0 for
 * every other read; ref_base = 'A'.  No base filters apply; the running factor is conf->bonf_indel, +1 per
 * test (lofreq_call.c:693-696).  Returns the significant tests (p * bonf_indel < sig) in test order. */
typedef struct lfq_indel_call {
    int64_t test;              /* pseudo-column index within the batch */
    int64_t bonf;              /* bonf_indel at this test */
    long double pvalue;
    int32_t qual;              /* PROB_TO_PHREDQUAL(pvalue) */
    int32_t count;             /* event count (AF numerator, lofreq_call.c:334) */
} lfq_indel_call;
int lfq_indel_batch_device(lfq_ctx *ctx, const lfq_conf *conf, const lfq_tracks *tracks,
                           lfq_col_counts *d_counts, lfq_col_pvals *d_pvals, int64_t pvals_capacity,
                           void *stream);
int lfq_call_indel_tests_batch(lfq_ctx *ctx, lfq_conf *conf, const lfq_tracks *tracks, int tracks_on_device,
                               lfq_indel_call *calls, int64_t calls_capacity, int64_t *n_calls,
                               lfq_batch_stats *stats);
/* --- call_indels drop-in (lofreq_call.c:619-726): the indel fields of a batch of plp_col_t, flattened ----
 * side[0] = insertions, side[1] = deletions.  Events of a column are listed in the order the reference
 * iterates its uthash (insertion order = first occurrence in the pileup, utils.h:101-135).  "Non-event reads"
 * are ins_quals/ins_map_quals (del_quals/del_map_quals): reads of the column carrying no event of that side
 * (plp.c compile_plp_col).  Qualities are phred ints as in the reference; -1 = not available. */
typedef struct lfq_indel_side {
    const int32_t *non_fw, *non_rv;    /* [ncols]     non_ins_fw_rv / non_del_fw_rv (plp.h:127-128) */
    const int64_t *ne_off;             /* [ncols+1]   non-event reads of each column */
    const int16_t *ne_q, *ne_mq;       /*             ins_quals / ins_map_quals (plp.h:113-116) */
    const int64_t *ev_off;             /* [ncols+1]   events of each column */
    const int64_t *key_off;            /* [nevents+1] event key (inserted / deleted sequence) in key_chars */
    const char *key_chars;
    const int32_t *ev_fw, *ev_rv;      /* [nevents]   ins_event.fw_rv (utils.h:108) */
    const int64_t *rd_off;             /* [nevents+1] reads of each event; count = rd_off[e+1]-rd_off[e] */
    const int16_t *rd_q, *rd_aq, *rd_mq, *rd_sq;   /* ins_quals / ins_aln_quals / ins_map_quals / ins_source_quals */
} lfq_indel_side;

typedef struct lfq_indel_columns {
    int64_t ncols;
    const uint8_t *ref_base;
    const int32_t *coverage_plp, *num_tails, *num_non_indels, *num_ins, *num_dels, *hrun;
    lfq_indel_side side[2];
    /* optional (NULL when unknown; not read by lfq_call_indels_batch): 1 where the column's consensus is an
     * insertion or deletion, cons_base[0] == '+' / '-' (plp.c:1236-1270) -- call_vars does not call SNVs there
     * (lofreq_call.c:928-931), see lfq_pileup_skip_snv_columns */
    const uint8_t *cons_indel;
} lfq_indel_columns;

typedef struct lfq_indel_record {
    int64_t col;
    int32_t side;                      /* 0 insertion, 1 deletion */
    int32_t event;                     /* index into side[side]'s event arrays */
    int32_t qual, dp, sb;              /* report_var (lofreq_call.c:95-155) */
    int32_t ref_fw, ref_rv, alt_fw, alt_rv;
    int32_t hrun;
    float af;
    int32_t count;
    int64_t bonf;
    long double pvalue;
} lfq_indel_record;

/* call_indels over a batch of columns, in column/side/event order: gates (min_cov on non_indels+ins+dels,
 * 'N' reference, the poly-AT 1-bp A/T rule of :649-681), packs every surviving event as a pseudo-column,
 * runs them through lfq_call_indel_tests_batch and assembles the report_var fields.  conf->bonf_indel and
 * conf->num_indel_tests are advanced exactly as the reference does. */
int lfq_call_indels_batch(lfq_ctx *ctx, lfq_conf *conf, const lfq_indel_columns *cols,
                          lfq_indel_record *records, int64_t records_capacity, int64_t *n_records,
                          int64_t *n_tests);

/* `lofreq filter` as `lofreq call` invokes it, for indel records (lofreq_filter.c:325-335, 210-236, 599-601):
 * QUAL >= indelqual_thresh (= lfq_snvqual_thresh(sig, bonf_indel), lofreq_call.c:1529-1534; 0 = off), and with
 * the defaults on, DP >= 10.  The strand-bias filter skips indels by default. keep[i] = 1 for PASS. */
int lfq_filter_indel_records(const lfq_indel_record *records, int64_t n, int indelqual_thresh, int apply_defaults,
                             int32_t *keep);

/* vcf_write_var + vcf_var_sprintf_info for an indel record (vcf.c:469-497, 608-629);
 * af = count / ((float)coverage_plp - num_tails), dp = coverage_plp - num_tails (lofreq_call.c:132, 334) */
int lfq_format_indel_record(char *buf, int buflen, const char *chrom, int64_t pos0, const char *ref,
                            const char *alt, int qual, int dp, float af, int sb, int ref_fw, int ref_rv,
                            int alt_fw, int alt_rv, int hrun, const char *filter_or_null);

/* --- base alignment quality: the per-read pre-step -----------------------------------
 * bam_prob_realn_core_ext (bam_md_ext.c:260-491) with baq_flag = 1 for reads without a pre-existing `lb` tag:
 * alignment window and band from the CIGAR, kpa_ext_glocal (kprobaln_ext.c:80-270, kpa_ext_par_lofreq_illumina),
 * then the plain or extended BAQ of every base.  One call = a batch of reads of ONE contig.  The caller
 *  skips unmapped / zero-length reads like the reference (:287-289) and appends
 * lb_out[seq_off[r] .. seq_off[r+1]) as the read's `lb:Z` tag (bytes are BAQ + 33, capped at '~').
 * lfq_baq_idaq_batch additionally computes the indel alignment qualities (idaq, :73-248): ai_out / ad_out get the
 * bytes of the `ai` / `ad` tags ('~' where there is nothing), tag_flags[r] bit 0 / bit 1 say whether read r gets an
 * ai / ad tag at all (n_ins / n_del > 0, :238-243).  More than 64 indels or 1024 repeat cells per read are not tracked.
 * BASE CODES (every `seq` array of this header): 0..3 = A, C, G, T, 4 = N -- seq_nt16_int of the BAM base -- and 5..15 = the
 * other letters of htslib's seq_nt16_str in its order, "=MRSVWYHKDB" (LFQ_SEQ_LETTERS "ACGTN=MRSVWYHKDB"[code]).  A code
 * above 3 behaves like N in the HMM, the pileups and the tests, as it does in the reference; where the reference compares or
 * prints the LETTER of a read base -- the repeat scan of idaq (:197), count_cigar_ops (samutils.c:486-489), the key of an
 * insertion (plp.c:1092-1093) -- an ambiguity code is its own letter.  A caller that only knows "not A, C, G, T" passes 4. */
typedef struct lfq_baq_reads {
    int64_t n_reads;
    const int32_t *pos;        /* [n]   bam1_core_t.pos (0-based leftmost reference coordinate) */
    const int64_t *cigar_off;  /* [n+1] into cigar */
    const uint32_t *cigar;     /*       BAM encoding: len << 4 | op (M0 I1 D2 N3 S4 H5 P6 =7 X8) */
    const int64_t *seq_off;    /* [n+1] into seq / qual / lb_out */
    const uint8_t *seq;        /*       0..3 = A,C,G,T, 4 = anything else (seq_nt16_int of the BAM base) */
    const uint8_t *qual;       /*       phred base qualities */
    const char *ref;           /*       the contig's sequence (faidx_fetch_seq), ASCII */
    int64_t ref_len;
} lfq_baq_reads;
int lfq_baq_batch(lfq_ctx *ctx, const lfq_baq_reads *reads, int baq_extended, uint8_t *lb_out);
int lfq_baq_idaq_batch(lfq_ctx *ctx, const lfq_baq_reads *reads, int baq_extended, uint8_t *lb_out,
                       uint8_t *ai_out, uint8_t *ad_out, uint8_t *tag_flags);

/* --- device-side pileup: reads -> the packed SNV tracks of a region -------------------------
 * What compile_plp_col (plp.c:797-1017) builds per column, for all columns of [region_begin, region_end) at once,
 * directly in HBM in the lfq_tracks layout.  The host has done what mplp_func does per read (plp.c:600-700): flag /
 * MAPQ filtering, and BAQ (lfq_baq_batch) if wanted.  Columns = the covered positions in order (mpileup yields no
 * column for a position without alignments); col_pos_out[c] is the reference position of column c.
 * The returned tracks point into memory owned by the context, valid until the next lfq_pileup_snv_tracks call;
 * they can be handed straight to lfq_call_snvs_batch(..., tracks_on_device = 1, ...). */
typedef struct lfq_pileup_reads {
    int64_t n_reads;
    const int32_t *pos;        /* [n]   bam1_core_t.pos */
    const int64_t *cigar_off;  /* [n+1] */
    const uint32_t *cigar;     /*       BAM encoding */
    const int64_t *seq_off;    /* [n+1] into seq / qual / baq */
    const uint8_t *seq;        /*       0..4 */
    const uint8_t *qual;       /*       phred */
    const uint8_t *baq;        /*       lb tag bytes (BAQ + 33), or NULL (every BAQ missing) */
    const uint8_t *mapq;       /* [n]   bam1_core_t.qual */
    const uint8_t *reverse;    /* [n]   bam_is_rev */
    const char *ref;           /*       the contig */
    int64_t ref_len;
    const uint8_t *sq;         /* [n]   source quality byte of the read (lfq_source_qual_batch), or NULL: no sq track */
} lfq_pileup_reads;
int lfq_pileup_snv_tracks(lfq_ctx *ctx, const lfq_pileup_reads *reads, int64_t region_begin, int64_t region_end,
                          int min_plp_bq, lfq_tracks *tracks_out, int64_t *col_pos_out);

/* The indel fields of compile_plp_col (plp.c:1019-1192) for the same reads: per column (= covered position, the
 * same column numbering as lfq_pileup_snv_tracks) coverage_plp, num_tails, num_non_indels / num_ins / num_dels,
 * hrun (get_hrun, plp.c:744-787), per side the strand counts and -- at columns with at least one event, the only
 * ones call_indels reads them for -- the ins_quals / del_quals arrays of the reads without such an event, and
 * the event tables (key, strands, per-read qualities) in the reference's order (first appearance; reads in pileup
 * order).  Pileup entries follow htslib's resolve_cigar2: an entry inside a deletion / reference skip takes the
 * qualities at the query position of the next base; the indel of an entry is the I / D operation following the
 * last position of its CIGAR operation.  Entries with BI or BD below min_plp_idq are ignored (plp.c:1062).
 * *cols_out points into memory owned by the context, valid until the next lfq_pileup_indel_columns call; it goes
 * straight into lfq_call_indels_batch. */
/* on = 0: lfq_pileup_indel_columns / lfq_readset_pileup_indels keep the ins_quals / del_quals arrays (ne_q, ne_mq: the
 * bulk of an lfq_indel_columns) on the device only -- the pointers in the struct are NULL, ne_off stays valid -- and
 * lfq_call_indels_batch, given that struct while it is still the context's current one, builds its pseudo-columns
 * from the resident copy.  Saves the largest transfer of the reads -> VCF chain.  Default: on = 1. */
int lfq_set_indel_arrays_on_host(lfq_ctx *ctx, int on);

typedef struct lfq_pileup_indel_tags {
    const uint8_t *bi, *bd;    /* per base (seq_off layout): BI / BD tag bytes (quality + 33); NULL = tag absent (quality 0) */
    const uint8_t *ai, *ad;    /* per base: ai / ad tag bytes (lfq_baq_idaq_batch); NULL = absent (-1) */
    const uint8_t *tag_flags;  /* [n] bit 0..3: read has BI / BD / ai / ad; NULL = every read has every non-NULL array */
    const int32_t *sq;         /* [n] source quality of the read or NULL (-1) */
} lfq_pileup_indel_tags;
int lfq_pileup_indel_columns(lfq_ctx *ctx, const lfq_pileup_reads *reads, const lfq_pileup_indel_tags *tags_or_null,
                             int64_t region_begin, int64_t region_end, int min_plp_idq,
                             const lfq_indel_columns **cols_out, int64_t *col_pos_out);

/* call_vars' gate for SNVs (lofreq_call.c:928-931): columns with skip[col] != 0 (e.g. lfq_indel_columns.cons_indel)
 * of the tracks last returned by lfq_pileup_snv_tracks are taken out of the SNV path -- their num_bases becomes
 * 0, which is the other half of the same gate (num_bases * 2 < coverage_plp): no test, no Bonferroni step. */
int lfq_pileup_skip_snv_columns(lfq_ctx *ctx, const uint8_t *skip, int64_t ncols);

/* --- resident read set: the device-side chain of the steps above ---------------------------------------------
 * lfq_readset_create uploads the reads of one contig region once (reads->baq / reads->sq and the BI / BD bytes of
 * `tags` go along when given); the steps below then work on the device copy and leave their per-base results in
 * HBM for the next one -- BAQ / IDAQ -> source quality -> both pileups -> calls -- so that only VCF-sized data
 * comes back.  The host arrays handed to lfq_readset_create must outlive the read set (the sparse host-side
 * steps -- geometry from the CIGARs, the indel event tables -- read them in place).  The host-buffer entry points
 * above (lfq_baq_idaq_batch, lfq_source_qual_batch, lfq_pileup_snv_tracks, lfq_pileup_indel_columns) are these
 * functions around a temporary read set.
 * Completion: lfq_readset_create may return while its copies are still in flight (a helper thread feeds them to an
 * upload stream) and lfq_readset_baq returns when its kernels are queued; every later call on the read set waits for
 * what it needs, lfq_readset_destroy for everything.  The caller sees no difference as long as the host arrays stay
 * as they are until the read set is destroyed (they must outlive it anyway).
 * Pinned arrays (lfq_host_alloc, hipHostMalloc, hipHostRegister): when the per-base arrays -- seq, qual, BI, BD -- are
 * pinned, lfq_readset_create queues every copy as a DMA transfer itself and returns, and what waits for the data is the
 * stream of the kernels that read it, never the calling thread.  Same results either way.
 * A read set belongs to its context: destroy it before lfq_destroy(ctx) (its device allocations go back to the context,
 * which hands them to the next read set -- keep one context per worker from region to region). */
typedef struct lfq_readset lfq_readset;
int lfq_readset_create(lfq_ctx *ctx, const lfq_pileup_reads *reads, const lfq_pileup_indel_tags *tags_or_null,
                       lfq_readset **out);
void lfq_readset_destroy(lfq_readset *rs);
/* lb (and, with want_idaq, ai / ad) of every read, kept on the device; needs reads->qual */
int lfq_readset_baq(lfq_ctx *ctx, lfq_readset *rs, int baq_extended, int want_idaq);
/* source quality; the per-read byte for the sq track stays on the device, sq_out_or_null gets source_qual()'s value */
int lfq_readset_source_qual(lfq_ctx *ctx, lfq_readset *rs, int def_nm_q, int min_bq, const uint8_t *ign_or_null,
                            int32_t *sq_out_or_null);
/* Returns when the scatter pass is QUEUED: ncols, col_pos_out and max_col_obs are final, the tracks themselves are
 * complete in stream order -- lfq_call_snvs_batch(..., tracks_on_device = 1), lfq_pileup_skip_snv_columns and the uniq
 * calls are queued behind them; call lfq_synchronize(ctx) before reading the device memory yourself.  A region worker
 * calls it between the indel pileup and the indel tests: the scatter pass then runs under the host part of the tests. */
int lfq_readset_pileup_snv(lfq_ctx *ctx, lfq_readset *rs, int64_t region_begin, int64_t region_end, int min_plp_bq,
                           lfq_tracks *tracks_out, int64_t *col_pos_out);
int lfq_readset_pileup_indels(lfq_ctx *ctx, lfq_readset *rs, int64_t region_begin, int64_t region_end, int min_plp_idq,
                              const lfq_indel_columns **cols_out, int64_t *col_pos_out);
/* copies of the resident tags for writing them back to the BAM; NULL = not wanted.  tag_flags as lfq_baq_idaq_batch */
int lfq_readset_fetch_tags(lfq_ctx *ctx, lfq_readset *rs, uint8_t *lb_out, uint8_t *ai_out, uint8_t *ad_out,
                           uint8_t *tag_flags);

/* --- source quality: the per-read pre-step of `lofreq call -s` -------------------------
 * source_qual (plp.c:427-593) over count_cigar_ops (samutils.c:437-614) for a batch of reads of one contig (same
 * read layout as lfq_baq_batch; the reference letters are compared as given, the caller upper-cases the contig
 * like mplp_func does, plp.c:652).  def_nm_q: -T/--def-nm-q (>= 0 replaces EVERY operation's quality, :500-504;
 * -1 = off); min_bq: the reference passes DEFAULT_MIN_BQ = 6 (plp.c:728); ign_or_null: one byte per reference
 * position, != 0 where the -S/--ign-vcf list has a variant (var_in_ign_list, plp.c:305-323).
 * sq_out[r] = what source_qual returns: -1 (nothing to count), 49314 (at most one non-match) or
 * (int)(-10 log10l(1 - P)) (INT32_MIN when P rounds to 1, as the x86-64 build yields); mplp_func stores
 * max(sq, 0) in the read's `sq` tag (plp.c:731-734).  sq_byte_or_null[r] gets the byte the packed sq track takes:
 * min(max(sq, 0), 254) -- every value above 3240 is the probability 0.0 in PHREDQUAL_TO_PROB, and source_qual
 * yields nothing between 160 and 49314. */
int lfq_source_qual_batch(lfq_ctx *ctx, const lfq_baq_reads *reads, int def_nm_q, int min_bq,
                          const uint8_t *ign_or_null, int32_t *sq_out, uint8_t *sq_byte_or_null);

/* Strand counts (ref_fw / ref_rv / alt_fw) and the unfiltered alt counts (alt_raw_counts, the AF numerator) are only ever
 * read for reported variants (DP4, SB and AF: lofreq_call.c:117-129, 835, 853-857), and counting them for every column
 * costs the dominant kernel half of its instructions.  With on = 0, lfq_snv_batch_device and lfq_call_snvs_submit
 * count them only for the columns of the sparse output (lfq_col_pvals.counts is complete either way, and so are the
 * records); in the dense lfq_col_counts entries the strand fields are then 0 and alt_raw_counts is either the true
 * value or 0 (which kernel ran decides; the decision fields n_err_probs, alt_counts, kmax, tested, gated, coverage are
 * always there).  Default: on = 1.  lfq_call_snvs_batch decides by itself: complete dense entries exactly when
 * h_counts_or_null is given. */
int lfq_set_dense_strand_counts(lfq_ctx *ctx, int on);

/* One step further for a caller of lfq_snv_batch_device that reads only the sparse output: with on = 0 the dense entry of a
 * column that is not tested (tested == 0: nothing downstream looks at it) need not be written at all -- d_counts then keeps
 * whatever it held there (the shared-wavefront count kernel, i.e. batches of the packed nt layout whose deepest column has at
 * most a few thousand observations, skips those stores; at 200x the dense entries are a fifth of the bytes it moves, at the
 * price HBM asks for writes among reads).  Takes effect only together with lfq_set_dense_strand_counts(ctx, 0).
 * Default: on = 1.  (lfq_call_snvs_batch without h_counts runs this way by itself on the context's own array;
 * lfq_call_snvs_submit only after lfq_set_dense_counts(ctx, 0), and lfq_call_snvs_collect then refuses h_counts.) */
int lfq_set_dense_counts(lfq_ctx *ctx, int on);

/* The profile HMM's gap-open and gap-extension probabilities, kpa_ext_par_t.d / .e (kprobaln_ext.h:31-34), for every
 * BAQ call of the context after this one.  Default: kpa_ext_par_lofreq_illumina = { 1e-5, 0.4 } (kprobaln_ext.c:50), what
 * bam_prob_realn_core_ext uses (bam_md_ext.c:275); a reference built with -DPACBIO_REALN uses kpa_ext_par_lofreq_pacbio =
 * { 0.1, 0.4 } (kprobaln_ext.c:51, bam_md_ext.c:268-273).  kpa_ext_par_t.bw is not a parameter: the caller overwrites it
 * per read (bam_md_ext.c:376-379).  0 < gap_open < 0.5, 0 < gap_ext < 1, else LFQ_ERR_INVALID. */
int lfq_set_baq_hmm_params(lfq_ctx *ctx, float gap_open, float gap_ext);

/* nt layout of the tracks the device pileup returns: on = 1 (default) LFQ_TRACKS_NT_PACKED, on = 0 one byte per
 * observation.  lfq_pack_nt_track: the same packing for a host byte track (packed_out: (n_obs + 7) / 8 * 4 bytes). */
int lfq_set_pileup_nt_packed(lfq_ctx *ctx, int on);
/* Reads that are NOT sorted by position: off (default) = the pileup calls return LFQ_ERR_INVALID, as mpileup stops at a file
 * that is not coordinate-sorted (bam_mplp_auto, plp.c:1406-1447); on = they are taken by the read-major kernels (one thread per
 * read, an atomic cursor per column): the same columns and the same observations per column, but in no fixed ORDER within a
 * column, so that the last bits of a p-value can differ from run to run (QUAL and every integer output do not). */
int lfq_set_pileup_unsorted(lfq_ctx *ctx, int on);
int lfq_pack_nt_track(const uint8_t *nt_bytes, int64_t n_obs, uint8_t *packed_out);

/* lfq_call_snvs_batch in two halves, for callers that keep more than one batch in flight (one context per batch in
 * flight): submit launches the kernels and returns; collect waits, fetches the sparse records and finishes them on
 * the host.  While batch k is collected, the kernels of batch k+1 (submitted on another context) run.  conf's
 * running Bonferroni factor is read at submit and advanced at collect, so batches in flight at the same time
 * need confs of their own -- independent regions, as call-parallel's bins are.  Host tracks handed to submit
 * (tracks_on_device = 0) must stay valid until collect.  A second submit on a context whose batch has not been
 * collected returns LFQ_ERR_INVALID. */
int lfq_call_snvs_submit(lfq_ctx *ctx, const lfq_conf *conf, const lfq_tracks *tracks, int tracks_on_device);
/* blocks until the KERNELS of the submitted batch are done.  The pattern that keeps one GPU busy with two contexts:
 * wait(A); submit(B, next batch); collect(A) -- the host finish of batch k runs under the kernels of batch k + 1,
 * and the two batches' kernels never compete for the same CUs. */
int lfq_call_snvs_wait(lfq_ctx *ctx);
/* What the count kernel of this context's NEXT batch waits for on the device when the previous batch of the same GPU (any
 * context) is still running -- only matters to a caller that submits batch k + 1 before it waits for batch k:
 *   LFQ_GATE_TAIL  (default) that batch is past its row-bound DP kernels: the count kernel runs beside its folds, combines
 *                  and join (pays where the DP tail is long next to a short count kernel: 1000x);
 *   LFQ_GATE_END   all of that batch's kernels are done: batch after batch on the device with no host round trip between
 *                  them -- `submit(k + 1); wait(k); collect(k)` then hides every host latency without putting two batches'
 *                  kernels on the machine at once (the loop of lofreq_call.c:735-879 has no such gap to hide: it is serial);
 *   LFQ_GATE_NONE  nothing: it starts as soon as its stream is free.  The context then shapes its DP work for running BESIDE
 *                  a count kernel (a long column is cut into two row segments at most instead of eight: less work and
 *                  residency at the price of a longer chain, which is hidden there).
 * Results do not depend on the choice (p-values within the tolerance the parity tests hold every decomposition to).  LFQ_ERR_INVALID for another value. */
#define LFQ_GATE_TAIL 0
#define LFQ_GATE_END 1
#define LFQ_GATE_NONE 2
int lfq_set_batch_gate(lfq_ctx *ctx, int gate);
/* The stream this context's own launches go to (count kernels, BAQ, pileups; the DP chains always run on the device's three
 * shared DP streams).  By default every context of a GPU shares ONE: batches submitted through several contexts then run in
 * submission order, which is what a single thread that pipelines batches wants (and what the batch gates order).  A caller
 * that drives several contexts from several HOST THREADS (one region / bin per thread, as call-parallel's workers do) sets
 * on = 1 per context: the context gets a stream of its own, a thread's waits then cover its own work only and one thread's
 * BAQ kernels run beside another's pileups.  The batch gates do not apply to such a context.  Call it while the context is
 * idle (it synchronizes).  Results do not depend on the choice. */
int lfq_set_private_stream(lfq_ctx *ctx, int on);
/* h_counts_or_null: the dense entries, ncols of them.  Needs a batch whose dense entries are complete: LFQ_ERR_INVALID when
 * the batch was submitted after lfq_set_dense_counts(ctx, 0) (the entries of its untested columns were never written). */
int lfq_call_snvs_collect(lfq_ctx *ctx, lfq_conf *conf, lfq_snv_record *records, int64_t records_capacity,
                          int64_t *n_records, lfq_col_counts *h_counts_or_null, lfq_batch_stats *stats_out);

/* second half for a SHARDED run (N processes, below): the batch's sparse records as the device wrote them -- shard-local
 * running Bonferroni factors, no emit test yet.  The caller exchanges its test counts (lfq_shard_exchange_counts),
 * rebases the factors (lfq_shard_rebase_bonferroni) and then runs lfq_finalize_pvals.  conf is not advanced
 * (lfq_shard_advance_conf does that with the batch's stats->n_tested).  LFQ_ERR_CAPACITY: *n_pvals says how many. */
int lfq_call_snvs_collect_pvals(lfq_ctx *ctx, lfq_col_pvals *pvals, int64_t pvals_capacity, int64_t *n_pvals,
                                lfq_batch_stats *stats_out);

/* host finishing step of layer 2, exposed for tests: sparse device records -> reported SNVs */
int lfq_finalize_pvals(const lfq_conf *conf, const lfq_col_pvals *pvals, int64_t n_pvals,
                       const int32_t *coverage_plp_or_null, const uint8_t *ref_base,
                       lfq_snv_record *records, int64_t records_capacity, int64_t *n_records);
/* expl() + the reference's clamp; exposed for tests */
long double lfq_pvalue_from_log(double logp, int status);

/* --- layer 3: output and final filter --- */
int lfq_format_snv_record(char *buf, int buflen, const char *chrom, int64_t pos0,
                          const lfq_snv_record *rec, const char *filter_or_null);
/* all (kept) records of a batch; pos0_or_null == NULL uses rec.col as the 0-based position.
 * Returns the byte count of the full text (written only if it fits into buflen). */
int64_t lfq_format_vcf(char *buf, int64_t buflen, const char *chrom, const int64_t *pos0_or_null,
                       const lfq_snv_record *recs, int64_t n, const uint8_t *keep_or_null,
                       const char *filter_or_null);
int lfq_snvqual_thresh(float sig, int64_t bonf_subst);             /* lofreq_call.c:1523-1527 */
int lfq_sb_phred(int ref_fw, int ref_rv, int alt_fw, int alt_rv);  /* lofreq_call.c:117-129 */
double lfq_fisher_exact(int n11, int n12, int n21, int n22, double *left, double *right, double *two);
int64_t lfq_fdr(const double *pvals, int64_t n, double alpha, int64_t num_tests, int64_t *rejected_idx);
void lfq_bonf_corr(double *pvals, int64_t n, int64_t num_tests);
void lfq_holm_bonf_corr(double *pvals, int64_t n, double alpha, int64_t num_tests);
/* `lofreq filter` as `lofreq call` invokes it: keep[i] = 1 iff record i ends up PASS */
int lfq_filter_records(const lfq_snv_record *records, int64_t n, int snvqual_thresh,
                       int apply_defaults, uint8_t *keep);

/* --- `lofreq filter`, every mode (lofreq_filter.c:861-1331): what lfq_filter_records / lfq_filter_indel_records do for
 * the one configuration `lofreq call` passes, for any configuration of the command.  The reference works on the variants
 * of a VCF file in file order, SNVs and indels mixed; here a variant is the handful of fields the filters look at. */
typedef enum lfq_mtc_type {        /* mtc_type_t, multtest.h:33-39 */
    LFQ_MTC_NONE = 0,
    LFQ_MTC_BONF = 1,
    LFQ_MTC_HOLMBONF = 2,
    LFQ_MTC_FDR = 3
} lfq_mtc_type;

typedef struct lfq_filter_conf {    /* filter_conf_t, lofreq_filter.c:59-112 */
    int32_t only_snvs, only_indels;          /* --only-snvs / --only-indels: the other kind is dropped, not filtered */
    int32_t dp_min, dp_max;                  /* -v / -V; < 1 = off */
    float af_min, af_max;                    /* -a / -A; <= 0 = off */
    int32_t sb_thresh;                       /* -B: filter if SB > thresh (and the compound rule); 0 = off */
    int32_t sb_mtc_type;                     /* -b: lfq_mtc_type; conflicts with sb_thresh */
    double sb_alpha;                         /* -c */
    int64_t sb_ntests;                       /* 0 = the number of variants tested; set on return */
    int32_t sb_no_compound;                  /* --sb-no-compound: without "85 % of the alt bases on one strand" (:57, 214-236) */
    int32_t sb_incl_indels;                  /* --sb-incl-indels */
    int32_t snvqual_thresh;                  /* -Q: filter SNVs with -1 < QUAL < thresh; 0 = off */
    int32_t snvqual_mtc_type;                /* -q */
    double snvqual_alpha;                    /* -r */
    int64_t snvqual_ntests;                  /* -s; 0 = the number of SNVs; set on return */
    int32_t indelqual_thresh;                /* -K */
    int32_t indelqual_mtc_type;              /* -k */
    double indelqual_alpha;                  /* -l */
    int64_t indelqual_ntests;                /* -m */
} lfq_filter_conf;

/* main_filter's initial values (:1091-1097): everything off, the three alphas at DEFAULT_SIG 0.01 */
void lfq_filter_conf_init(lfq_filter_conf *conf);
/* what main_filter does after its options are parsed unless --no-defaults is given (:1184-1197): strand bias by FDR at
 * alpha 0.001 if neither -B nor -b was set, DP >= 10 if -v was not set.  Call it after setting the options. */
void lfq_filter_conf_defaults(lfq_filter_conf *conf);

typedef struct lfq_filter_var {     /* one variant as mtc_quals_from_vcf_file (:790-858) and the apply_* functions read it */
    int32_t is_indel;
    int32_t qual;                   /* QUAL; -1 = missing (counts as INT_MAX in the corrections, :824-831) */
    int32_t dp;                     /* INFO DP */
    int32_t sb;                     /* INFO SB (0 if absent, :845) */
    int32_t alt_fw, alt_rv;         /* DP4[2], DP4[3] */
    float af;                       /* INFO AF as strtof of its text (%f: six decimals) -- lfq_filter_var_from_* do that */
    int32_t pad_;
} lfq_filter_var;

/* bits of a variant's result */
#define LFQ_FILT_AF_MIN 1u
#define LFQ_FILT_AF_MAX 2u
#define LFQ_FILT_DP_MIN 4u
#define LFQ_FILT_DP_MAX 8u
#define LFQ_FILT_SNVQUAL 16u
#define LFQ_FILT_INDELQUAL 32u
#define LFQ_FILT_SB 64u
#define LFQ_FILT_DROPPED 128u       /* --only-snvs / --only-indels: not written at all */

/* fail[i] = the filters variant i fails, in the bit order the reference appends their ids to FILTER (:1255-1297);
 * 0 = PASS.  conf's *_ntests are filled in as the reference fills them in (:407-415).  Returns LFQ_ERR_INVALID for the
 * option conflicts main_filter rejects (:1205-1227). */
int lfq_filter_vars(lfq_filter_conf *conf, const lfq_filter_var *vars, int64_t n, uint32_t *fail);
/* the id one bit stands for (min_dp_10, sb_fdr, snvqual_bonf, min_snvqual_53, ...; cfg_filter_to_vcf_header,
 * :682-786); returns its length, 0 if that filter is off */
int lfq_filter_id(const lfq_filter_conf *conf, uint32_t bit, char *buf, int buflen);
/* the FILTER column of a variant: ids joined by ';' (vcf_var_add_to_filter, vcf.c:524-563), "PASS" for 0 */
int lfq_filter_string(const lfq_filter_conf *conf, uint32_t fail_bits, char *buf, int buflen);
/* the ##FILTER header lines the configuration adds, in the reference's order; returns the text's length */
int lfq_filter_header_lines(const lfq_filter_conf *conf, char *buf, int buflen);
void lfq_filter_var_from_snv(const lfq_snv_record *rec, lfq_filter_var *out);
void lfq_filter_var_from_indel(const lfq_indel_record *rec, lfq_filter_var *out);

/* --- `lofreq uniq --use-det-lim` (uniq_snv, lofreq_uniq.c:222-333): the second consumer of the
 * column boundary.  For every column (the pileup of the OTHER sample at a variant's position, built with uniq's
 * own mpileup settings: no BAQ, MAPQ >= 1, lofreq_uniq.c:461-465) and the variant's allele frequency af[col]:
 * would the variant have been detectable here?  The default varcall_conf (init_varcall_conf), the first alt count
 * replaced by (int)(af * n_err_probs) (float product, truncated), snpcaller(bonf 1, alpha 0.01f);
 * detectable[col] = pvalue * 1 < 0.01f, the condition under which uniq_snv adds the UNIQ flag.  Only the 'N'
 * reference gate applies (uniq_snv does not go through call_snvs).  pvalue_or_null[col] gets snpcaller's value for
 * the columns that were emitted (LDBL_MAX elsewhere: K = 0, or pruned as not significant).
 * An AF outside [0, 1] is reset like the reference does (af < 0 -> 0.01, af > 1 -> 1.0, lofreq_uniq.c:262-268), here
 * and in lfq_uniq_binom_batch; a NaN is LFQ_ERR_INVALID. */
int lfq_uniq_detlim_batch(lfq_ctx *ctx, const lfq_tracks *tracks, int tracks_on_device, const float *af,
                          uint8_t *detectable, long double *pvalue_or_null);

/* --- `lofreq uniq`, default mode (uniq_snv's binomial branch, lofreq_uniq.c:335-393, + apply_uniq_filter_mtc, :140-206).
 * Per column (the other sample's pileup at a variant's position, uniq's own mpileup settings as above): coverage =
 * coverage_plp (tracks->coverage_plp, or the observation count), alt_count = the bases of nucleotide alt_base[col] in
 * the column whatever their quality (base_count, plp.c:128-132; counted on the device), pvalue =
 * binom(coverage, alt_count, af) = P(X <= alt_count), X ~ Binomial(coverage, af) (binom.c:52-69 -> cdflib90's cdfbin),
 * uq_out[col] = PROB_TO_PHREDQUAL_SAFE(pvalue), the value of the UQ= INFO tag; -1 where the reference adds none
 * (coverage < 1, or cdfbin rejects its arguments).  SNVs only (indel variants take their count from the event table,
 * which is host data: lfq_indel_columns).  lfq_uniq_mtc then decides PASS / uq_<mtc> like apply_uniq_filter_mtc:
 * mtc_type 1 bonf, 2 holm, 3 fdr (multtest.h; `lofreq uniq` defaults: fdr, alpha 0.001, ntests 0 = the number of
 * variants).  lfq_binom_cdf is the scalar test itself (status: cdfbin's code, 0 = ok). */
int lfq_uniq_binom_batch(lfq_ctx *ctx, const lfq_tracks *tracks, int tracks_on_device, const float *af,
                         const char *alt_base, int32_t *uq_out, double *pvalue_or_null);
int lfq_uniq_mtc(const int32_t *uq, int64_t n, int mtc_type, double alpha, int64_t ntests, uint8_t *pass);
double lfq_binom_cdf(int n, int k, double pr, int *status_or_null);

/* --- synthetic workload (bench / tests): fills device tracks per include/lofreq_synth.h --- */
int lfq_synth_fill_device(lfq_ctx *ctx, uint64_t seed, uint32_t depth, uint32_t plant_period,
                          int64_t col_begin, int64_t ncols, uint8_t *d_nt, uint8_t *d_bq,
                          uint8_t *d_baq, uint8_t *d_mq, uint64_t *d_col_off, uint8_t *d_ref_base,
                          void *stream);
/* same, with the nt track in the LFQ_TRACKS_NT_PACKED layout if nt_packed != 0 (d_nt then needs half the bytes) */
int lfq_synth_fill_device_layout(lfq_ctx *ctx, uint64_t seed, uint32_t depth, uint32_t plant_period,
                                 int64_t col_begin, int64_t ncols, uint8_t *d_nt, uint8_t *d_bq,
                                 uint8_t *d_baq, uint8_t *d_mq, uint64_t *d_col_off, uint8_t *d_ref_base,
                                 int nt_packed, void *stream);

/* --- device timing of the last batch (HIP events on the stream the kernels ran on) --- */
typedef struct lfq_kernel_times {
    float ms_count;     /* sum over the batch's segments of the count kernel launches */
    float ms_scan;      /* sum of the scan (prefix + work-list) kernels */
    float ms_dp;        /* DP time not hidden under a count kernel: last count kernel end -> all done */
    float ms_total;     /* first kernel start -> all done */
    float ms_dp_light, ms_dp_mid, ms_dp_big;    /* sums of the concurrent DP kernels' own durations */
    int32_t n_segments; /* count-kernel launches in this batch */
} lfq_kernel_times;
int lfq_last_kernel_times(lfq_ctx *ctx, lfq_kernel_times *t);
/* the BAQ kernels of the context's last lfq_readset_baq call (waits for them): first narrow-band launch -> everything done,
 * on the device's clock; the main-stream launches in between; the call's reads and bases.  All 0 before the first call. */
typedef struct lfq_baq_times {
    float ms_kernels;
    int32_t n_launches;
    int64_t n_reads, n_bases;
} lfq_baq_times;
int lfq_last_baq_times(lfq_ctx *ctx, lfq_baq_times *t);

/* --- DP work of the last batch (device counters; SURVEY 8d "ALGORITHMIC DP work", the secondary roofline) ---
 * cells = sum over the tested columns of sum_{n = 1..N*} min(n, K): N* = the kept row at which this implementation's
 * pruning test fired (track order; the reference sorts its probabilities ascending and therefore prunes LATER, so
 * its own cell count for the same columns is larger), or the column's end.  Columns finished by the underflow
 * shortcut count only the rows their remaining recurrence ran.  A light column that the screen kernel hands to the
 * retry kernel is counted in both (the work was done twice). */
typedef struct lfq_dp_work {
    int64_t cells;             /* recurrence cells processed */
    int64_t rows;              /* kept rows processed */
    int64_t n_light, n_mid, n_big;   /* tested columns per scheduling class */
    int64_t n_light_retry;     /* light columns finished by the one-column-per-wavefront kernel */
    int64_t bytes_read_count;  /* track + header bytes the count kernel instantiation of this batch reads (layout bytes) */
    int64_t bytes_written_count; /* dense records (of the tested columns only where lfq_set_dense_counts(0) applies) + class flags it writes */
    int64_t n_approx_pruned;   /* tested columns the Poisson gate (lfq_conf.approx_threshold_n) gave up: not in the classes above */
} lfq_dp_work;
int lfq_last_dp_work(lfq_ctx *ctx, lfq_dp_work *w);

/* --- N processes, one per GPU: the exchange of a sharded run, from C ---
 * The reference's `call-parallel` wrapper runs one `lofreq call -r <bin>` per worker and merges afterwards: it sums the
 * per-worker test counts it parses from the logs and concatenates the VCFs (lofreq2_call_pparallel.py:131-185, 685-707).
 * Here every process calls its shard with the SAME starting conf (layer 1 or 2), then:
 *     lfq_shard_exchange_counts   one all-gather of a few int64 per rank (tested columns, indel tests) -> every rank's
 *                                 counts and this rank's exclusive prefix
 *     lfq_shard_rebase_bonferroni the shard-local running Bonferroni factors of the sparse p-value records become the
 *                                 single-process ones (3 tests per tested column of the earlier shards,
 *                                 lofreq_call.c:794-801); lfq_finalize_pvals then applies the exact emit test
 *     lfq_shard_gather_records    every rank's reported variants, in shard order, `col` made global by col_offset;
 *                                 every rank takes part whatever its `capacity` (0 on the ranks that do not want the
 *                                 records): too little room returns LFQ_ERR_CAPACITY with *n_out = the total
 *     lfq_shard_advance_conf      conf->bonf_subst / num_snv_tests as after the single-process loop over all shards
 * `comm` is an ncclComm_t of RCCL (one rank per process, created by the caller: ncclCommInitRank) or NULL when
 * world == 1.  RCCL is looked up at run time (dlopen of librccl): the library has no link-time dependency on it.
 * lofreq_amd/shard.py calls these same functions (round 6: ONE implementation of the exchange; torch.distributed only
 * carries the ncclUniqueId to the ranks and, as a gloo group, serves as the host transport for the counts). */
/* A launcher that has no RCCL communicator (MPI, a shared directory, a test double) supplies the one collective the
 * exchange needs: all-gather of `bytes` bytes per rank, host buffers, recv = world * bytes in rank order; 0 = ok.
 * While set (process-wide; NULL restores RCCL) `comm` is ignored by the lfq_shard_* calls. */
typedef int (*lfq_host_allgather_fn)(void *user, int world, int rank, const void *send, void *recv, size_t bytes);
int lfq_shard_set_host_allgather(lfq_host_allgather_fn fn, void *user);
/* The library's own host transport for the ranks of ONE node: an all-gather through a POSIX shared-memory segment (a slot
 * per rank, double-buffered, sequence numbers with release / acquire order; microseconds where a loopback TCP ring takes
 * milliseconds at eight ranks).  lfq_shard_shm_open maps /dev/shm<name> (created by whoever comes first; `name` = "/..." and
 * NEW PER RUN -- a nonce every rank of the run knows) and installs it as the host all-gather; once every rank has opened
 * it (any barrier, e.g. a first all-gather through it), one rank calls lfq_shard_shm_unlink so that nothing is left behind
 * if the run dies; lfq_shard_shm_close unmaps and uninstalls. */
int lfq_shard_shm_open(const char *name, int world, int rank);
int lfq_shard_shm_unlink(void);
int lfq_shard_shm_close(void);
/* the collective itself: `bytes` bytes of every rank, in rank order (all = world * bytes) */
int lfq_shard_allgather(lfq_ctx *ctx, void *comm, int world, int rank, const void *mine, int64_t bytes, void *all);
int lfq_shard_exchange_counts(lfq_ctx *ctx, void *comm, int world, int rank, const int64_t *local, int n,
                              int64_t *all_out /* [world][n] */, int64_t *prefix_out /* [n] */);
int lfq_shard_rebase_bonferroni(lfq_col_pvals *pvals, int64_t n, int64_t prefix_tested);
int lfq_shard_gather_records(lfq_ctx *ctx, void *comm, int world, int rank, const lfq_snv_record *recs, int64_t n,
                             int64_t col_offset, lfq_snv_record *out, int64_t capacity, int64_t *n_out);
int lfq_shard_advance_conf(lfq_conf *conf, int64_t total_tested);
/* The gather in two halves, for a caller that pipelines steps (bench.py's sharded step, lofreq_amd/shard.py): every rank hands
 * over ONE piece of `piece_bytes` bytes (the same size on every rank; what is inside -- a record count in front of a fixed
 * number of record slots -- is the caller's business).  _start returns at once: with a communicator the piece goes up from
 * a pinned copy, the collective (ncclAllGather) and -- where want_all is set -- the copy of all pieces back to pinned memory
 * are queued on a high-priority stream of the handle's own, and nothing waits for the device; without one (a host transport
 * set by lfq_shard_set_host_allgather, or world == 1) the blocking all-gather runs inside _start.  _wait blocks until this
 * rank's side is done and copies the world * piece_bytes bytes in rank order into `all` (NULL, or a rank that did not set
 * want_all: nothing is copied).  Unlike the blocking calls above, _start uses the communicator when one is given even
 * while a host transport is set: the test counts are host integers and take the host road, the records the device's.
 * Every rank must call _start and _wait for every gather, in the same order. */
typedef struct lfq_shard_gather lfq_shard_gather;
int lfq_shard_gather_start(lfq_ctx *ctx, void *comm, int world, int rank, const void *piece, int64_t piece_bytes,
                           int want_all, lfq_shard_gather **handle_out);
int lfq_shard_gather_wait(lfq_shard_gather *handle, void *all_or_null, int64_t all_bytes);

#ifdef __cplusplus
}
#endif
#endif
