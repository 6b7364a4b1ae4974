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

#include <algorithm>

#include "lfq_device.h"
#include "lofreq_synth.h"

/* ------------------------------------------------------------------------------------------ */
/* count kernel                                                                                */
/* ------------------------------------------------------------------------------------------ */

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

/* Per-lane partial counts as bit-plane popcounts.  With the nt4 code bits b0,b1 (b2 set = N) of the
 * observations that are in range and not N ("valid"):
 *     n[0] = #valid, n[1] = #(b0), n[2] = #(b1), n[3] = #(b0 & b1)
 * so that  #T = n3, #C = n1 - n3, #G = n2 - n3, #A = n0 - n1 - n2 + n3.  The same four planes are
 * counted again restricted to forward-strand reads (fw), to bq >= min_bq (ge) and, when the alt
 * threshold differs, to bq >= max(min_bq, min_alt_bq) (ga). */
struct LfqAcc {
    uint32_t raw[4], fw[4], ge[4], ga[4];
};

/* popcount + accumulate as the ONE instruction the hardware has for it (left to itself the compiler counts into a fresh
 * register and folds pairs of counts with v_add3_u32: 1.5 instructions per count) */
__device__ __forceinline__ void lfq_bcnt_acc(uint32_t &acc, uint32_t x)
{
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x));
}

template <bool SAME_THR, bool STRAND = true>
__device__ __forceinline__ void lfq_count_dword(LfqAcc &a, uint32_t ntw, uint32_t bqw, uint32_t vm,
                                                uint32_t minbq4, uint32_t minalt4)
{
    /* everything lives in bit 7 of each byte; shifted words are "dirty" outside bit 7 and only ever
     * ANDed with clean masks */
    const uint32_t s0 = ntw << 7, s1 = ntw << 6, s2 = ntw << 5, s3 = ntw << 4;
    const uint32_t p0 = vm & ~s2;                  /* valid: in range and not N */
    const uint32_t p1 = p0 & s0;
    const uint32_t p2 = p0 & s1;
    const uint32_t p3 = p1 & s1;
    lfq_bcnt_acc(a.raw[0], p0);
    lfq_bcnt_acc(a.raw[1], p1);
    lfq_bcnt_acc(a.raw[2], p2);
    lfq_bcnt_acc(a.raw[3], p3);
    if (STRAND) {                                  /* lazy-strand mode: only the columns that emit get these (lfq_strand_*) */
        lfq_bcnt_acc(a.fw[0], p0 & ~s3);
        lfq_bcnt_acc(a.fw[1], p1 & ~s3);
        lfq_bcnt_acc(a.fw[2], p2 & ~s3);
        lfq_bcnt_acc(a.fw[3], p3 & ~s3);
    }
    const uint32_t hi = bqw | 0x80808080u;
    const uint32_t g = hi - minbq4;                /* bit 7: bq >= min_bq (bq < 128) */
    lfq_bcnt_acc(a.ge[0], p0 & g);
    lfq_bcnt_acc(a.ge[1], p1 & g);
    lfq_bcnt_acc(a.ge[2], p2 & g);
    lfq_bcnt_acc(a.ge[3], p3 & g);
    if (!SAME_THR) {
        const uint32_t g2 = (hi - minalt4) & g;    /* ... and >= min_alt_bq */
        lfq_bcnt_acc(a.ga[0], p0 & g2);
        lfq_bcnt_acc(a.ga[1], p1 & g2);
        lfq_bcnt_acc(a.ga[2], p2 & g2);
        lfq_bcnt_acc(a.ga[3], p3 & g2);
    }
}

/* The same counts for the packed nt layout, EIGHT observations per operation: a packed nt dword holds the codes of the
 * observations of two bq dwords (even nibbles <-> bytes of `bqa`, odd nibbles <-> bytes of `bqb`, lfq_nt_at), so the
 * planes are formed in the nibble domain (everything lives in bit 3 of each nibble, `vm` = 0x8 per observation in
 * range) and only the quality gates come from the byte domain: bit 7 of a byte of `bqb`'s gate IS bit 3 of its odd
 * nibble, `bqa`'s gate moves down by 4.  26 instead of 46 instructions per 8 observations. */
template <bool SAME_THR, bool STRAND = true>
__device__ __forceinline__ void lfq_count_nib8(LfqAcc &a, uint32_t ntw, uint32_t bqa, uint32_t bqb, uint32_t vm,
                                               uint32_t minbq4, uint32_t minalt4)
{
    const uint32_t s0 = ntw << 3, s1 = ntw << 2, s2 = ntw << 1;
    const uint32_t p0 = vm & ~s2;                  /* valid: in range and not N */
    const uint32_t p1 = p0 & s0;
    const uint32_t p2 = p0 & s1;
    const uint32_t p3 = p1 & s1;
    lfq_bcnt_acc(a.raw[0], p0);
    lfq_bcnt_acc(a.raw[1], p1);
    lfq_bcnt_acc(a.raw[2], p2);
    lfq_bcnt_acc(a.raw[3], p3);
    if (STRAND) {                                  /* the strand flag is bit 3 of the nibble itself */
        lfq_bcnt_acc(a.fw[0], p0 & ~ntw);
        lfq_bcnt_acc(a.fw[1], p1 & ~ntw);
        lfq_bcnt_acc(a.fw[2], p2 & ~ntw);
        lfq_bcnt_acc(a.fw[3], p3 & ~ntw);
    }
    const uint32_t ha = bqa | 0x80808080u, hb = bqb | 0x80808080u;
    const uint32_t ga = ha - minbq4, gb = hb - minbq4;             /* bit 7 of a byte: bq >= min_bq (bq < 128) */
    const uint32_t g = ((ga >> 4) & 0x08080808u) | (gb & 0x80808080u);
    lfq_bcnt_acc(a.ge[0], p0 & g);
    lfq_bcnt_acc(a.ge[1], p1 & g);
    lfq_bcnt_acc(a.ge[2], p2 & g);
    lfq_bcnt_acc(a.ge[3], p3 & g);
    if (!SAME_THR) {
        const uint32_t ga2 = ha - minalt4, gb2 = hb - minalt4;
        const uint32_t g2 = (((ga2 >> 4) & 0x08080808u) | (gb2 & 0x80808080u)) & g;       /* ... and >= min_alt_bq */
        lfq_bcnt_acc(a.ga[0], p0 & g2);
        lfq_bcnt_acc(a.ga[1], p1 & g2);
        lfq_bcnt_acc(a.ga[2], p2 & g2);
        lfq_bcnt_acc(a.ga[3], p3 & g2);
    }
}

/* The lean form of lfq_count_nib8 for the default filters when only the DECISION counts are wanted of every column
 * (n_err_probs, alt_counts, kmax, tested: the planes restricted to bq >= min_bq) -- the raw and the forward-strand counts
 * reach nothing but the records of the ~1e-3 of the columns that emit one, and lfq_strand_* count them there.
 *   - the gate by ONE add per bq dword: byte + (128 - min) has bit 7 set exactly when byte >= min; bytes are <= 127 (the
 *     contract: 0..93, include/lofreq_amd.h), so no byte carries into its neighbour; kge / kga = 0x01010101 * (128 - min)
 *   - both dwords' gates into the nibble domain by a shift and a bit-field insert, the dirt outside bit 3 of each nibble
 *     removed by the one three-input operation that also applies "in range" (vm) and "not N"
 *   - every plane = one AND, every count = one v_bcnt with its accumulator
 * 15 instructions per 8 observations with one threshold (the general form above compiles to 34), 27 with two. */
template <bool SAME_THR>
__device__ __forceinline__ void lfq_count_nib8_lean(uint32_t (&ge)[4], uint32_t (&ga)[4], uint32_t ntw, uint32_t bqa,
                                                    uint32_t bqb, uint32_t vm, uint32_t kge, uint32_t kga)
{
    const uint32_t s0 = ntw << 3, s1 = ntw << 2, s2 = ntw << 1;
    const uint32_t a = bqa + kge, b = bqb + kge;
    /* (v_bitop3_b32 spelled out: the compiler's own choice for these is three instructions longer per call) */
    const uint32_t g = __builtin_amdgcn_bitop3_b32(0x80808080u, b, a >> 4, 0xCA);        /* bit 7 of each byte from b, the rest from a >> 4 */
    const uint32_t p0 = __builtin_amdgcn_bitop3_b32(g, s2, vm, 0x20);                    /* g & ~s2 & vm: passes, not N, in range */
    const uint32_t p1 = p0 & s0;
    const uint32_t p2 = p0 & s1;
    const uint32_t p3 = __builtin_amdgcn_bitop3_b32(p0, s0, s1, 0x80);
    lfq_bcnt_acc(ge[0], p0);
    lfq_bcnt_acc(ge[1], p1);
    lfq_bcnt_acc(ge[2], p2);
    lfq_bcnt_acc(ge[3], p3);
    if (!SAME_THR) {                               /* kga: the larger of the two thresholds, so these are subsets */
        const uint32_t a2 = bqa + kga, b2 = bqb + kga;
        const uint32_t g2 = __builtin_amdgcn_bitop3_b32(0x80808080u, b2, a2 >> 4, 0xCA);
        lfq_bcnt_acc(ga[0], p0 & g2);
        lfq_bcnt_acc(ga[1], p1 & g2);
        lfq_bcnt_acc(ga[2], p2 & g2);
        lfq_bcnt_acc(ga[3], p3 & g2);
    }
}

template <bool SAME_THR, bool PACKED, bool STRAND = true>
__device__ __forceinline__ void lfq_count_chunks(LfqAcc &a, const LfqTracksDev &T, uint64_t off0, uint64_t off1,
                                                 uint32_t minbq4, uint32_t minalt4, int lane = lfq_lane(),
                                                 int lanes = LFQ_WAVE)
{
    /* `lanes` lanes (a wavefront, or a 16-lane group of lfq_count_multi_kernel) stride over the column */
    const uint4 *nt16 = reinterpret_cast<const uint4 *>(T.nt);
    const uint4 *bq16 = reinterpret_cast<const uint4 *>(T.bq);
    const int64_t c0 = (int64_t)(off0 >> 4), c1 = (int64_t)((off1 + 15) >> 4);
    if (PACKED) {
        /* 16 observations per step like the byte layout, but the nt half is 8 bytes of nibbles: the even nibbles of an
         * nt dword are the observations of the first bq dword of its group of 8, the odd nibbles those of the second
         * (lfq_nt_at), so the bit-plane counts run on the packed dword itself, eight observations at a time
         * (lfq_count_nib8) -- 1.5 instead of 2 bytes of HBM traffic per observation, every load of a
         * wavefront contiguous.  (Two steps' loads in flight per lane were measured: slower, the registers cost a
         * wavefront of residency.) */
        const uint2 *nt8 = reinterpret_cast<const uint2 *>(T.nt);
        for (int64_t ch = c0 + lane; ch < c1; ch += lanes) {
            const uint2 n2 = nt8[ch];
            const uint4 b4 = bq16[ch];
            const int64_t base = ch << 4;
            const int lo = (int64_t)off0 > base ? (int)((int64_t)off0 - base) : 0;
            const int hi = (int64_t)off1 < base + 16 ? (int)((int64_t)off1 - base) : 16;
            if (lo == 0 && hi == 16) {
                lfq_count_nib8<SAME_THR, STRAND>(a, n2.x, b4.x, b4.y, 0x88888888u, minbq4, minalt4);
                lfq_count_nib8<SAME_THR, STRAND>(a, n2.y, b4.z, b4.w, 0x88888888u, minbq4, minalt4);
            } else {
                lfq_count_nib8<SAME_THR, STRAND>(a, n2.x, b4.x, b4.y,
                                                 (lfq_bytes_mask(lo, hi, 0) >> 4) | lfq_bytes_mask(lo, hi, 1), minbq4, minalt4);
                lfq_count_nib8<SAME_THR, STRAND>(a, n2.y, b4.z, b4.w,
                                                 (lfq_bytes_mask(lo, hi, 2) >> 4) | lfq_bytes_mask(lo, hi, 3), minbq4, minalt4);
            }
        }
        return;
    }
    for (int64_t ch = c0 + lane; ch < c1; ch += lanes) {
        const uint4 n4 = nt16[ch];      /* (non-temporal loads were measured: no difference at 6.0 TB/s) */
        const uint4 b4 = bq16[ch];
        const int64_t base = ch << 4;
        const int lo = (int64_t)off0 > base ? (int)((int64_t)off0 - base) : 0;
        const int hi = (int64_t)off1 < base + 16 ? (int)((int64_t)off1 - base) : 16;
        if (lo == 0 && hi == 16) {
            lfq_count_dword<SAME_THR, STRAND>(a, n4.x, b4.x, 0x80808080u, minbq4, minalt4);
            lfq_count_dword<SAME_THR, STRAND>(a, n4.y, b4.y, 0x80808080u, minbq4, minalt4);
            lfq_count_dword<SAME_THR, STRAND>(a, n4.z, b4.z, 0x80808080u, minbq4, minalt4);
            lfq_count_dword<SAME_THR, STRAND>(a, n4.w, b4.w, 0x80808080u, minbq4, minalt4);
        } else {
            lfq_count_dword<SAME_THR, STRAND>(a, n4.x, b4.x, lfq_bytes_mask(lo, hi, 0), minbq4, minalt4);
            lfq_count_dword<SAME_THR, STRAND>(a, n4.y, b4.y, lfq_bytes_mask(lo, hi, 1), minbq4, minalt4);
            lfq_count_dword<SAME_THR, STRAND>(a, n4.z, b4.z, lfq_bytes_mask(lo, hi, 2), minbq4, minalt4);
            lfq_count_dword<SAME_THR, STRAND>(a, n4.w, b4.w, lfq_bytes_mask(lo, hi, 3), minbq4, minalt4);
        }
    }
}

/* bit-plane counts -> per-nucleotide counts (A,C,G,T) */
__device__ __forceinline__ void lfq_planes_to_classes(const uint32_t n[4], uint32_t c[4])
{
    c[3] = n[3];
    c[1] = n[1] - n[3];
    c[2] = n[2] - n[3];
    c[0] = n[0] - n[1] - n[2] + n[3];
}

/* counts per nucleotide -> the column record + class flag (one lane) */
__device__ __forceinline__ void lfq_count_emit(bool strand, lfq_col_counts &r, const uint32_t raw[4], const uint32_t fw[4],
                                               const uint32_t filt[4], int ref_code,
                                               lfq_col_counts *__restrict__ out, uint8_t *__restrict__ flags,
                                               int64_t col)
{
    {
        uint8_t flag = 0;
        if (!r.gated) {
            /* the three non-reference nucleotides in A,C,G,T order (snpcaller.c:391-397) */
            const int x0 = (ref_code == 0) ? 1 : 0;
            const int x1 = (ref_code <= 1) ? 2 : 1;
            const int x2 = (ref_code <= 2) ? 3 : 2;
#define LFQ_PICK(arr, x) ((x) == 0 ? arr[0] : (x) == 1 ? arr[1] : (x) == 2 ? arr[2] : arr[3])
            r.ref_fw = (int)LFQ_PICK(fw, ref_code);
            r.ref_rv = strand ? (int)(LFQ_PICK(raw, ref_code) - LFQ_PICK(fw, ref_code)) : 0;
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
            /* scheduling class.  Columns whose alt count is far above what sequencing errors explain
             * (~ n/1000 at Q30) almost surely run the full recurrence: they go to the long-column
             * kernel even when K < 64, so that the light kernel only sees quick exits. */
            const int suspicious = max(12, r.n_err_probs / 512 + 8);
            flag = (uint8_t)((r.tested ? 1 : 0)
                             | ((kmax >= LFQ_BIG_K) ? 4 : (kmax >= LFQ_MID_K || kmax >= suspicious) ? 2 : 0));
        }
        out[col] = r;               /* (out: the dense records, or the wavefront's staging rows in LDS with col = the group) */
        flags[col] = flag;
    }
}

/* The chunks of one column of lfq_count_shallow_kernel (packed nt layout): lane l of the column's LPG lanes takes the chunks
 * l, l + LPG, ...  What the general loop (lfq_count_chunks) spends around the counting proper -- 64-bit chunk addresses, the
 * range of every chunk from 64-bit offsets, a second pass with byte masks whenever any lane of the wavefront is at a
 * column's first or last chunk: 75 of 125 VALU instructions per chunk -- is done once per column here: chunk indices are
 * 32-bit and relative to the column, the addresses are a wavefront-uniform base (buffer loads: the base in scalar
 * registers) plus a 32-bit lane offset, the masks of the first and the last chunk come out of a table in LDS (s_tab[h]:
 * the nibbles of observations [0, h) of a chunk), and every chunk takes the same path.
 * Two halves:
 *   lfq_group_issue    the loads of AHEAD chunks of the lane, i0, i0 + LPG, ... (a chunk index past the column's end reads on
 *                      into the next column -- never past `pass_last`, the last chunk of the wavefront's 64 columns -- and
 *                      counts nothing);
 *   lfq_group_consume  counts them.
 * A batch whose deepest column has more than AHEAD * LPG chunks takes several such rounds per group (`rounds`, the same for
 * every wavefront: a kernel argument). */
#ifndef LFQ_COUNT_AHEAD
#define LFQ_COUNT_AHEAD 4
#endif
typedef uint32_t lfq_v2u __attribute__((ext_vector_type(2)));
typedef uint32_t lfq_v4u __attribute__((ext_vector_type(4)));
template <int AHEAD>
struct LfqGroupLoads {
    lfq_v2u n2[AHEAD];
    lfq_v4u b4[AHEAD];
};

template <bool SAME_THR, bool STRAND>
__device__ __forceinline__ void lfq_count_chunk_packed(LfqAcc &a, lfq_v2u n2, lfq_v4u b4, uint32_t vx, uint32_t vy,
                                                       uint32_t minbq4, uint32_t minalt4)
{
    if (!STRAND) {              /* lazy record counts: the decision planes only (a.raw stays 0: alt_raw_counts = 0 in the entry) */
        const uint32_t kge = 0x80808080u - minbq4, kga = 0x80808080u - minalt4;      /* 0x01010101 * (128 - min) */
        lfq_count_nib8_lean<SAME_THR>(a.ge, a.ga, n2.x, b4.x, b4.y, vx, kge, kga);
        lfq_count_nib8_lean<SAME_THR>(a.ge, a.ga, n2.y, b4.z, b4.w, vy, kge, kga);
        return;
    }
    lfq_count_nib8<SAME_THR, STRAND>(a, n2.x, b4.x, b4.y, vx, minbq4, minalt4);
    lfq_count_nib8<SAME_THR, STRAND>(a, n2.y, b4.z, b4.w, vy, minbq4, minalt4);
}

/* h: the column's header out of LDS -- start (64 bit), length (0: gated or past the end), - */
template <int LPG, int AHEAD, bool ALWAYS = false>
__device__ __forceinline__ void lfq_group_issue(LfqGroupLoads<AHEAD> &L, const LfqTracksDev &T, uint4 h, int i0, uint64_t pass_last)
{
    const uint32_t c0_lo = (h.x >> 4) | (h.y << 28);                  /* chunk of the column's start, low half */
    /* lane 0's column starts lowest (columns ascend with the group) */
    const uint32_t f_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)h.x), f_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)h.y);
    const uint64_t c_head = ((uint64_t)f_lo | ((uint64_t)f_hi << 32)) >> 4;
    const uint64_t c_first = c_head < pass_last ? c_head : pass_last;   /* (empty columns at the very end start one past it) */
    const uint32_t rel = c0_lo - (uint32_t)c_first + (uint32_t)i0;    /* (the columns of a wavefront are neighbours: small) */
    const uint32_t rel_last = (uint32_t)(pass_last - c_first);
    const __amdgpu_buffer_rsrc_t nt_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(T.nt + (c_first << 3)), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t bq_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(T.bq + (c_first << 4)), 0, -1, 0x00020000);
    const int n_ch = (int)(((h.x & 15u) + h.z + 15u) >> 4);
#pragma unroll
    for (int k = 0; k < AHEAD; k++) {
        /* (a step no column of the wavefront's group reaches is skipped as a whole: wave-uniform, the same test as in
         * lfq_group_consume) */
        /* (ALWAYS: the caller keeps two steps' loads in flight and wants their number known when it waits for the older ones) */
        if (ALWAYS || k == 0 || __any(i0 + k * LPG < n_ch)) {
            const uint32_t at = min(rel + (uint32_t)(k * LPG), rel_last);
            L.n2[k] = __builtin_amdgcn_raw_buffer_load_b64(nt_rsrc, (int)(at << 3), 0, 0);
            L.b4[k] = __builtin_amdgcn_raw_buffer_load_b128(bq_rsrc, (int)(at << 4), 0, 0);
        }
    }
}

template <bool SAME_THR, bool STRAND, int LPG, int AHEAD>
__device__ __forceinline__ void lfq_group_consume(LfqAcc &a, const LfqGroupLoads<AHEAD> &L, const LfqTracksDev &T, uint4 h,
                                                  uint32_t minbq4, uint32_t minalt4, int i0, const uint2 *s_tab)
{
    constexpr uint32_t FULL = 0x88888888u;
    const int lo = (int)(h.x & 15u);
    const int n_ch = (int)(((uint32_t)lo + h.z + 15u) >> 4);         /* chunks the column touches */
    const int last = n_ch - 1;
    const uint2 t_lo = s_tab[lo];
    const uint2 t_hi = s_tab[n_ch > 0 ? lo + (int)h.z - 16 * last : 0];
    const uint32_t lo_x = ~t_lo.x & FULL, lo_y = ~t_lo.y & FULL;    /* observations [lo, 16) of the first chunk */
#pragma unroll
    for (int k = 0; k < AHEAD; k++) {
        const int idx = i0 + k * LPG;
        uint32_t vx = idx < last ? FULL : (idx == last ? t_hi.x : 0u);
        uint32_t vy = idx < last ? FULL : (idx == last ? t_hi.y : 0u);
        if (k == 0) {                                /* idx == 0: lane 0's first chunk, first round */
            vx &= idx == 0 ? lo_x : FULL;
            vy &= idx == 0 ? lo_y : FULL;
        }
        if (k == 0 || __any(idx < n_ch)) {
            lfq_count_chunk_packed<SAME_THR, STRAND>(a, L.n2[k], L.b4[k], vx, vy, minbq4, minalt4);
        }
    }
}

/* Shallow columns (depth up to a few thousand): the per-column epilogue (12 reductions + the record) costs more
 * than the loads, so several columns share a wavefront, LPG lanes each: 64 / LPG records built at once, reductions of
 * log2(LPG) steps inside the DPP row.  LPG = 16 (four columns) up to a few thousand observations per column, 8 below a
 * thousand, 4 for exome-like depths: the instructions of one pass over the epilogue are the same for any LPG, so the
 * cost per column falls with the number of columns that share it until the loads dominate (at 200x: 0.65 -> see
 * DESIGN.md 3.1).  Fast path only (nt + bq tracks); everything else runs lfq_count_kernel. */
#ifndef LFQ_COUNT_AHEAD
#define LFQ_COUNT_AHEAD 4
#endif
#ifndef LFQ_COUNT_WAVES
#define LFQ_COUNT_WAVES 4
#endif
template <bool PACKED, bool STRAND, int LPG>
__global__ __launch_bounds__(256) void lfq_count_multi_kernel(LfqTracksDev T, LfqParams P,
                                                              lfq_col_counts *__restrict__ out,
                                                              uint8_t *__restrict__ flags, int64_t c0, int64_t c1)
{
    constexpr int G = 64 / LPG;                      /* columns per wavefront */
    __shared__ __attribute__((aligned(16))) lfq_col_counts s_rec[4][G];
    __shared__ uint8_t s_flag[4][G];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = lfq_lane();
    const int g = lane / LPG, l = lane % LPG;
    const int64_t stride = (int64_t)gridDim.x * 4 * G;
    const bool same_thr = (P.min_alt_bq4 == P.min_bq4);
    const uint32_t minbq4 = 0x01010101u * (uint32_t)P.min_bq4;
    const uint32_t minalt4 = 0x01010101u * (uint32_t)P.min_alt_bq4;
    for (int64_t colb = c0 + ((int64_t)blockIdx.x * 4 + wave) * G; colb < c1; colb += stride) {
        const int64_t col = colb + g;
        const bool valid = col < c1;
        /* the column's header in one round trip: every load is issued before the first one is waited for (the lanes past
         * the end read the last column's: no branch around a load) */
        const int64_t colc = valid ? col : c1 - 1;
        const uint64_t o0 = T.col_off[colc], o1 = T.col_off[colc + 1];
        const int cov_h = T.coverage_plp ? T.coverage_plp[colc] : 0;
        const int nb_h = T.num_bases ? T.num_bases[colc] : 0;
        const uint32_t rb_h = T.ref_base[colc];
        const uint64_t off0 = valid ? o0 : 0, off1 = valid ? o1 : 0;
        const int64_t n_obs = (int64_t)(off1 - off0);
        const int cov = (valid && T.coverage_plp) ? cov_h : (int)n_obs;
        const int nb = (valid && T.num_bases) ? nb_h : (int)n_obs;
        const uint32_t rb = valid ? rb_h : 'N';
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
        r.gated = (ref_code < 0) || (!P.detlim_af && (((int64_t)nb * 2 < (int64_t)cov) || (nb < P.min_cov)));
        LfqAcc a;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            a.raw[x] = a.fw[x] = a.ge[x] = a.ga[x] = 0;
        }
        if (valid && !r.gated) {
            if (same_thr) {
                lfq_count_chunks<true, PACKED, STRAND>(a, T, off0, off1, minbq4, minalt4, l, LPG);
            } else {
                lfq_count_chunks<false, PACKED, STRAND>(a, T, off0, off1, minbq4, minalt4, l, LPG);
            }
        }
        /* sums over the LPG lanes of the group (every lane of the wavefront takes part) */
        uint32_t n_raw[4], n_fw[4], n_ge[4], n_ga[4], raw[4], fw[4], c_ge[4], c_ga[4], filt[4];
#pragma unroll
        for (int x = 0; x < 4; x++) {
            n_raw[x] = a.raw[x];
            n_fw[x] = a.fw[x];
            n_ge[x] = a.ge[x];
            n_ga[x] = a.ga[x];
            n_raw[x] = lfq_group_sum_u32<LPG>(n_raw[x]);
            if (STRAND) {
                n_fw[x] = lfq_group_sum_u32<LPG>(n_fw[x]);
            }
            n_ge[x] = lfq_group_sum_u32<LPG>(n_ge[x]);
            n_ga[x] = same_thr ? n_ge[x] : lfq_group_sum_u32<LPG>(n_ga[x]);
        }
        lfq_planes_to_classes(n_raw, raw);
        lfq_planes_to_classes(n_fw, fw);
        lfq_planes_to_classes(n_ge, c_ge);
        lfq_planes_to_classes(n_ga, c_ga);
#pragma unroll
        for (int x = 0; x < 4; x++) {
            filt[x] = (x == ref_code) ? c_ge[x] : c_ga[x];
        }
        /* The records of the wavefront's G columns are G x 64 contiguous bytes: lane 0 of every group builds its record in
         * LDS, then the wavefront writes them out 16 bytes per lane -- one coalesced store instead of five per group that
         * touch a cache line each (and the class flags as one run of bytes). */
        if (l == 0) {
            lfq_count_emit(STRAND, r, raw, fw, filt, ref_code, &s_rec[wave][0], &s_flag[wave][0], g);
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        if (lane < G * 4 && colb + (lane >> 2) < c1) {
            const uint4 v = reinterpret_cast<const uint4 *>(&s_rec[wave][0])[lane];
            reinterpret_cast<uint4 *>(out + colb)[lane] = v;
        }
        if (lane < G && colb + lane < c1) {
            flags[colb + lane] = s_flag[wave][lane];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

/* The shared-wavefront kernel for the packed nt layout: a wavefront takes 64 consecutive columns per pass and works on
 * them in three phases, each with the lane <-> work mapping that suits it.
 *   header    lane j <-> column j: offsets, reference base, coverage gates -- coalesced loads, once; the column's start and
 *             length into LDS.
 *   counting  64 / LPG columns at a time, LPG lanes each (lfq_count_group_packed); a group gets its column's range out of
 *             LDS (no global round trip per group), and its plane sums go back there.
 *   records   lane j <-> column j again: ONE pass over the epilogue (classes, alt counts, K, class flag: ~150
 *             instructions, as many as the counting of a 1000x column costs a lane) builds 64 records instead of 64 / LPG,
 *             and the wavefront writes them as 4 KiB in a row.
 * lfq_count_multi_kernel ran header and epilogue once per 64 / LPG columns: at 1000x (LPG 16) half of its instructions,
 * in a kernel bound by what it issues. */
template <bool STRAND, int LPG>
__global__ __launch_bounds__(256, STRAND ? 3 : LFQ_COUNT_WAVES) void lfq_count_shallow_kernel(LfqTracksDev T, LfqParams P,
                                                                                lfq_col_counts *__restrict__ out,
                                                                                uint8_t *__restrict__ flags, int64_t c0, int64_t c1,
                                                                                int rounds)
{
    constexpr int G = 64 / LPG;                      /* columns counted at a time */
    constexpr int NSUM = STRAND ? 4 : 3;             /* uint4 plane sums per column: ge, ga, raw (, fw) */
    __shared__ __attribute__((aligned(16))) uint4 s_hdr[4][64];          /* start (64 bit), length, - */
    __shared__ __attribute__((aligned(16))) uint4 s_sum[4][64][NSUM];
    __shared__ __attribute__((aligned(16))) lfq_col_counts s_rec[4][64];
    __shared__ uint2 s_tab[17];                      /* nibbles of the observations [0, h) of a chunk (lfq_count_group_packed) */
    if (threadIdx.x < 17) {
        const int h = (int)threadIdx.x;
        s_tab[h] = make_uint2((lfq_bytes_mask(0, h, 0) >> 4) | lfq_bytes_mask(0, h, 1),
                              (lfq_bytes_mask(0, h, 2) >> 4) | lfq_bytes_mask(0, h, 3));
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = lfq_lane();
    const int g = lane / LPG, l = lane % LPG;
    const int64_t stride = (int64_t)gridDim.x * 4 * 64;
    const bool same_thr = (P.min_alt_bq4 == P.min_bq4);
    const uint32_t minbq4 = 0x01010101u * (uint32_t)P.min_bq4;
    const uint32_t minalt4 = 0x01010101u * (uint32_t)P.min_alt_bq4;
    for (int64_t colb = c0 + ((int64_t)blockIdx.x * 4 + wave) * 64; colb < c1; colb += stride) {
        /* ---- header: lane <-> column ---- */
        const int64_t col = colb + lane;
        const bool valid = col < c1;
        const int64_t colc = valid ? col : c1 - 1;
        const uint64_t off0 = T.col_off[colc], off1 = T.col_off[colc + 1];
        const int cov_h = T.coverage_plp ? T.coverage_plp[colc] : 0;
        const int nb_h = T.num_bases ? T.num_bases[colc] : 0;
        const uint32_t rb = T.ref_base[colc];
        const int64_t n_obs = (int64_t)(off1 - off0);
        const int cov = T.coverage_plp ? cov_h : (int)n_obs;
        const int nb = T.num_bases ? nb_h : (int)n_obs;
        const int ref_code = (rb == 'A') ? 0 : (rb == 'C') ? 1 : (rb == 'G') ? 2 : (rb == 'T') ? 3 : -1;
        const bool gated = (ref_code < 0) || (!P.detlim_af && (((int64_t)nb * 2 < (int64_t)cov) || (nb < P.min_cov)));
        s_hdr[wave][lane] = make_uint4((uint32_t)off0, (uint32_t)(off0 >> 32), (valid && !gated) ? (uint32_t)n_obs : 0u, 0u);
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        /* ---- counting: LPG lanes <-> column ---- */
        const int n_here = (int)((c1 - colb) < 64 ? (c1 - colb) : 64);
        const int n_sub = (n_here + G - 1) / G;
        /* the last chunk the wavefront's columns touch (a load may run past its own column, not past this one) */
        const uint64_t end_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)off1, n_here - 1);
        const uint64_t end_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(off1 >> 32), n_here - 1);
        const uint64_t beg_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)off0);
        const uint64_t beg_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(off0 >> 32));
        const uint64_t pass_beg = beg_lo | (beg_hi << 32), pass_end = end_lo | (end_hi << 32);
        if (pass_end > pass_beg) {                   /* (nothing to read otherwise: not even a chunk that is surely there) */
            const uint64_t pass_last = (pass_end - 1) >> 4;
#define LFQ_SHALLOW_ZERO(a_)                                                                                            \
            _Pragma("unroll")                                                                                           \
            for (int x = 0; x < 4; x++) {                                                                               \
                a_.raw[x] = a_.fw[x] = a_.ge[x] = a_.ga[x] = 0;                                                         \
            }
#define LFQ_SHALLOW_CONSUME(a_, L_, h_, i0_)                                                                            \
            do {                                                                                                        \
                if (same_thr) {                                                                                         \
                    lfq_group_consume<true, STRAND, LPG, LFQ_COUNT_AHEAD>(a_, L_, T, h_, minbq4, minalt4, i0_, s_tab);  \
                } else {                                                                                                \
                    lfq_group_consume<false, STRAND, LPG, LFQ_COUNT_AHEAD>(a_, L_, T, h_, minbq4, minalt4, i0_, s_tab); \
                }                                                                                                       \
            } while (0)
#define LFQ_SHALLOW_SUMS(a_, cw_)                                                                                       \
            do {                                                                                                        \
                uint32_t n_ge[4], n_ga[4], n_raw[4], n_fw[4];                                                           \
                _Pragma("unroll")                                                                                       \
                for (int x = 0; x < 4; x++) {                                                                           \
                    n_ge[x] = lfq_group_sum_u32<LPG>(a_.ge[x]);                                                         \
                    n_ga[x] = same_thr ? n_ge[x] : lfq_group_sum_u32<LPG>(a_.ga[x]);                                    \
                    n_raw[x] = lfq_group_sum_u32<LPG>(a_.raw[x]);                                                       \
                    if (STRAND) {                                                                                       \
                        n_fw[x] = lfq_group_sum_u32<LPG>(a_.fw[x]);                                                     \
                    }                                                                                                   \
                }                                                                                                       \
                if (l == 0) {                                                                                           \
                    s_sum[wave][cw_][0] = make_uint4(n_ge[0], n_ge[1], n_ge[2], n_ge[3]);                               \
                    s_sum[wave][cw_][1] = make_uint4(n_ga[0], n_ga[1], n_ga[2], n_ga[3]);                               \
                    s_sum[wave][cw_][2] = make_uint4(n_raw[0], n_raw[1], n_raw[2], n_raw[3]);                           \
                    if (STRAND) {                                                                                       \
                        s_sum[wave][cw_][3] = make_uint4(n_fw[0], n_fw[1], n_fw[2], n_fw[3]);                           \
                    }                                                                                                   \
                }                                                                                                       \
            } while (0)
            {
                /* The steps of a pass -- the G columns `sub` at a time, AHEAD chunks per lane and round `rd` -- as ONE sequence
                 * with the loads of step s + 1 requested before step s is counted: a wavefront alone has nothing in flight
                 * while it counts, and four of them per SIMD do not cover that (per wavefront and pass: 16 x (latency +
                 * counting) one after the other).  Two sets of load registers, the loop unrolled by two so that each set is
                 * a fixed set of registers; the step after (sub, rd) is the next round of the same columns if any of them
                 * reaches into it, else the first round of the next G columns. */
                int sub = 0, rd = 0;
                uint4 hA = s_hdr[wave][g], hB = hA;
                int i0A = l, i0B = l;
                LfqGroupLoads<LFQ_COUNT_AHEAD> LA, LB;
                LfqAcc a;
                LFQ_SHALLOW_ZERO(a);
                lfq_group_issue<LPG, LFQ_COUNT_AHEAD, true>(LA, T, hA, i0A, pass_last);
#define LFQ_SHALLOW_STEP(Lc_, hc_, i0c_, Ln_, hn_, i0n_)                                                                \
                {                                                                                                       \
                    int sub_n = sub, rd_n = rd + 1;                                                                     \
                    hn_ = hc_;                                                                                          \
                    i0n_ = l + rd_n * (LFQ_COUNT_AHEAD * LPG);                                                          \
                    if (rd_n >= rounds || !__any(i0n_ < (int)(((hc_.x & 15u) + hc_.z + 15u) >> 4))) {                   \
                        sub_n = sub + 1;                                                                                \
                        rd_n = 0;                                                                                       \
                        i0n_ = l;                                                                                       \
                        hn_ = s_hdr[wave][(sub_n < n_sub ? sub_n : sub) * G + g];                                       \
                    }                                                                                                   \
                    const bool have_n = sub_n < n_sub;                                                                  \
                    /* (behind the pass's last step too: the last columns' chunks once more, counted by nobody -- a branch   \
                     * around the loads would cost the compiler its count of what is in flight, and with it the prefetch) */ \
                    lfq_group_issue<LPG, LFQ_COUNT_AHEAD, true>(Ln_, T, hn_, i0n_, pass_last);                          \
                    LFQ_SHALLOW_CONSUME(a, Lc_, hc_, i0c_);                                                             \
                    if (!have_n || rd_n == 0) {                                                                         \
                        LFQ_SHALLOW_SUMS(a, sub * G + g);                                                               \
                        LFQ_SHALLOW_ZERO(a);                                                                            \
                    }                                                                                                   \
                    if (!have_n) {                                                                                      \
                        break;                                                                                          \
                    }                                                                                                   \
                    sub = sub_n;                                                                                        \
                    rd = rd_n;                                                                                          \
                }
#pragma unroll 1
                for (;;) {
                    LFQ_SHALLOW_STEP(LA, hA, i0A, LB, hB, i0B)
                    LFQ_SHALLOW_STEP(LB, hB, i0B, LA, hA, i0A)
                }
#undef LFQ_SHALLOW_STEP
            }
#undef LFQ_SHALLOW_ZERO
#undef LFQ_SHALLOW_CONSUME
#undef LFQ_SHALLOW_SUMS
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        /* ---- records: lane <-> column ---- */
        bool is_tested = false;
        if (valid) {
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
            r.gated = gated;
            uint32_t n_ge[4] = {0, 0, 0, 0}, n_ga[4] = {0, 0, 0, 0}, n_raw[4] = {0, 0, 0, 0}, n_fw[4] = {0, 0, 0, 0};
            if (!gated && pass_end > pass_beg) {
                const uint4 v0 = s_sum[wave][lane][0], v1 = s_sum[wave][lane][1];
                n_ge[0] = v0.x; n_ge[1] = v0.y; n_ge[2] = v0.z; n_ge[3] = v0.w;
                n_ga[0] = v1.x; n_ga[1] = v1.y; n_ga[2] = v1.z; n_ga[3] = v1.w;
                const uint4 v2 = s_sum[wave][lane][2];
                n_raw[0] = v2.x; n_raw[1] = v2.y; n_raw[2] = v2.z; n_raw[3] = v2.w;
                if (STRAND) {
                    const uint4 v3 = s_sum[wave][lane][3];
                    n_fw[0] = v3.x; n_fw[1] = v3.y; n_fw[2] = v3.z; n_fw[3] = v3.w;
                }
            }
            uint32_t raw[4], fw[4], c_ge[4], c_ga[4], filt[4];
            lfq_planes_to_classes(n_raw, raw);
            lfq_planes_to_classes(n_fw, fw);
            lfq_planes_to_classes(n_ge, c_ge);
            lfq_planes_to_classes(n_ga, c_ga);
#pragma unroll
            for (int x = 0; x < 4; x++) {
                filt[x] = (x == ref_code) ? c_ge[x] : c_ga[x];
            }
            lfq_count_emit(STRAND, r, raw, fw, filt, ref_code, &s_rec[wave][0], flags + colb, lane);
            is_tested = r.tested != 0;
        }
        /* (sparse mode: only the tested columns' entries are ever read again -- work lists, DP kernels, records --, and the
         * entries are a fifth of what this kernel moves at 200x, at the price HBM asks for writes among reads) */
        const unsigned long long keep = P.sparse_counts ? __ballot(is_tested) : ~0ull;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        /* 64 records = 4 KiB in a row: four stores of 1 KiB per wavefront */
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int q = t * 64 + lane;             /* 16-byte piece q of the wavefront's records: column q / 4 */
            if (colb + (q >> 2) < c1 && ((keep >> (q >> 2)) & 1ull) != 0) {
                reinterpret_cast<uint4 *>(out + colb)[q] = reinterpret_cast<const uint4 *>(&s_rec[wave][0])[q];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

/* One column on one wavefront: the integer outputs of plp_to_errprobs (snpcaller.c:346-498) + the gates of call_snvs /
 * call_vars in front of it.  s_hist: the wavefront's 128 bins for the median-BQ override of the general path. */
template <bool PACKED, bool STRAND>
__device__ __forceinline__ void lfq_count_column(const LfqTracksDev &T, const LfqParams &P,
                                                 const LfqLuts *__restrict__ luts,
                                                 lfq_col_counts *__restrict__ out,
                                                 uint8_t *__restrict__ flags, int64_t col, int lane,
                                                 uint32_t *s_hist)
{
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
    r.gated = (ref_code < 0) || (!P.detlim_af && (((int64_t)nb * 2 < (int64_t)cov) || (nb < P.min_cov)));

    LfqAcc a;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        a.raw[x] = a.fw[x] = a.ge[x] = a.ga[x] = 0;
    }
    /* general path accumulates per-class counts directly */
    uint32_t g_raw[4] = {0, 0, 0, 0}, g_fw[4] = {0, 0, 0, 0}, g_filt[4] = {0, 0, 0, 0};
    const bool same_thr = (P.min_alt_bq4 == P.min_bq4);

    if (!r.gated && !P.general) {
        /* fast path: only the nt and bq tracks decide the counts */
        const uint32_t minbq4 = 0x01010101u * (uint32_t)P.min_bq4;
        const uint32_t minalt4 = 0x01010101u * (uint32_t)P.min_alt_bq4;
        if (same_thr) {
            lfq_count_chunks<true, PACKED, STRAND>(a, T, off0, off1, minbq4, minalt4);
        } else {
            lfq_count_chunks<false, PACKED, STRAND>(a, T, off0, off1, minbq4, minalt4);
        }
    } else if (!r.gated) {
        /* general path: merged-quality filters and/or the median-of-reference-BQ override
         * (snpcaller.c:363-379) need the full per-observation evaluation */
        int median = -1;
        if (P.def_alt_bq == -1) {
            for (int i = lane; i < 128; i += LFQ_WAVE) {
                s_hist[i] = 0;
            }
            __builtin_amdgcn_wave_barrier();
            for (int64_t i = lane; i < n_obs; i += LFQ_WAVE) {
                const uint32_t ntb = lfq_nt_at(T, off0 + (uint64_t)i);
                if ((int)(ntb & 7u) == ref_code) {
                    atomicAdd(&s_hist[T.bq[off0 + i] & 127u], 1u);
                }
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            /* int_median (utils.c:436-457) from the histogram: lane-serial, <= 128 bins */
            uint32_t total = 0;
            for (int i = 0; i < 128; i++) {
                total += s_hist[i];
            }
            if (total) {
                const uint32_t r_hi = total / 2, r_lo = (total & 1u) ? r_hi : r_hi - 1;
                int q_lo = -1, q_hi = -1;
                uint32_t run = 0;
                for (int i = 0; i < 128; i++) {
                    run += s_hist[i];
                    if (q_lo < 0 && run > r_lo) q_lo = i;
                    if (q_hi < 0 && run > r_hi) q_hi = i;
                }
                median = (q_lo + q_hi) / 2;     /* (a+b)/2.0 truncated; odd size: q_lo == q_hi */
            }
        }
        r.median_ref_bq = median;
        for (int64_t i = lane; i < n_obs; i += LFQ_WAVE) {
            const uint32_t ntb = lfq_nt_at(T, off0 + (uint64_t)i);
            const uint32_t code = ntb & 7u;
            if (code > 3u) {
                continue;
            }
            const LfqObs o = lfq_eval_obs(ntb, T.bq[off0 + i], T.baq ? T.baq[off0 + i] : 255u,
                                          T.mq[off0 + i], T.sq ? T.sq[off0 + i] : 255u, ref_code, median,
                                          P, luts);
#pragma unroll
            for (int x = 0; x < 4; x++) {
                g_raw[x] += (code == (uint32_t)x) ? 1u : 0u;
                g_fw[x] += (code == (uint32_t)x && !(ntb & 8u)) ? 1u : 0u;
                g_filt[x] += (code == (uint32_t)x && o.keep) ? 1u : 0u;
            }
        }
    }

    uint32_t raw[4], fw[4], filt[4];
    if (!P.general) {
        uint32_t n_raw[4], n_fw[4], n_ge[4], n_ga[4], c_ge[4], c_ga[4];
#pragma unroll
        for (int x = 0; x < 4; x++) {
            n_raw[x] = lfq_wave_sum_u32(a.raw[x]);
            n_fw[x] = lfq_wave_sum_u32(a.fw[x]);
            n_ge[x] = lfq_wave_sum_u32(a.ge[x]);
            n_ga[x] = same_thr ? n_ge[x] : lfq_wave_sum_u32(a.ga[x]);
        }
        lfq_planes_to_classes(n_raw, raw);
        lfq_planes_to_classes(n_fw, fw);
        lfq_planes_to_classes(n_ge, c_ge);
        lfq_planes_to_classes(n_ga, c_ga);
#pragma unroll
        for (int x = 0; x < 4; x++) {
            filt[x] = (x == ref_code) ? c_ge[x] : c_ga[x];     /* alt bases must pass both thresholds */
        }
    } else {
#pragma unroll
        for (int x = 0; x < 4; x++) {
            raw[x] = lfq_wave_sum_u32(g_raw[x]);
            fw[x] = lfq_wave_sum_u32(g_fw[x]);
            filt[x] = lfq_wave_sum_u32(g_filt[x]);
        }
    }

    if (lane == 0) {        /* (same block as lfq_count_emit; a call with array arguments costs this kernel registers and LDS) */
        uint8_t flag = 0;
        if (!r.gated) {
            /* the three non-reference nucleotides in A,C,G,T order (snpcaller.c:391-397) */
            const int x0 = (ref_code == 0) ? 1 : 0;
            const int x1 = (ref_code <= 1) ? 2 : 1;
            const int x2 = (ref_code <= 2) ? 3 : 2;
#define LFQ_PICK(arr, x) ((x) == 0 ? arr[0] : (x) == 1 ? arr[1] : (x) == 2 ? arr[2] : arr[3])
            r.ref_fw = (int)LFQ_PICK(fw, ref_code);
            r.ref_rv = STRAND ? (int)(LFQ_PICK(raw, ref_code) - LFQ_PICK(fw, ref_code)) : 0;   /* lazy: lfq_strand_* */
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
            if (P.detlim_af) {                       /* lofreq_uniq.c:297-301: float product, truncated */
                r.alt_counts[0] = (int)(P.detlim_af[col] * (float)r.n_err_probs);
                r.alt_counts[1] = r.alt_counts[2] = 0;
            }
            const int kmax = max(r.alt_counts[0], max(r.alt_counts[1], r.alt_counts[2]));
            r.kmax = kmax;
            r.tested = kmax > 0;                     /* lofreq_call.c:768-780 */
            /* scheduling class.  Columns whose alt count is far above what sequencing errors explain
             * (~ n/1000 at Q30) almost surely run the full recurrence: they go to the long-column
             * kernel even when K < 64, so that the light kernel only sees quick exits. */
            const int suspicious = max(12, r.n_err_probs / 512 + 8);
            flag = (uint8_t)((r.tested ? 1 : 0)
                             | ((kmax >= LFQ_BIG_K) ? 4 : (kmax >= LFQ_MID_K || kmax >= suspicious) ? 2 : 0));
        }
        out[col] = r;
        flags[col] = flag;
    }
}

/* one column per wavefront: every configuration (merged-quality filters, the median-BQ override, lofreq uniq's detection
 * limit).  The default filters take lfq_count_fast_kernel below. */
template <bool PACKED, bool STRAND>
__global__ __launch_bounds__(256) void lfq_count_kernel(LfqTracksDev T, LfqParams P,
                                                        const LfqLuts *__restrict__ luts,
                                                        lfq_col_counts *__restrict__ out,
                                                        uint8_t *__restrict__ flags, int64_t c0, int64_t c1)
{
    __shared__ uint32_t s_hist[4][128];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t col = c0 + (int64_t)blockIdx.x * 4 + wave;
    if (col >= c1) {
        return;
    }
    lfq_count_column<PACKED, STRAND>(T, P, luts, out, flags, col, lfq_lane(), s_hist[wave]);
}

/* ---- the default filters (min_jq = min_alt_jq = 0, def_alt_bq != -1): only the nt and bq tracks decide the counts ----
 * The same column on a wavefront as lfq_count_column's fast branch, as a function of its own: no quality tables, no
 * per-observation evaluation, no LDS, the two thresholds' relation a template parameter, chunk indices 32-bit and relative
 * to the column, the byte masks only for a column's first and last chunk, two chunks' loads in flight per lane.  What it
 * buys is registers (36-44 instead of 58-72): half of a SIMD's register file stays free beside eight of these wavefronts,
 * which is what lets the DP kernels of the previous batch run beside the count kernel of the next one. */
struct LfqCountArgs {           /* what the default filters need of LfqTracksDev and LfqParams: a third of the scalar registers */
    const uint8_t *nt, *bq;
    const uint64_t *col_off;
    const uint8_t *ref_base;
    const int32_t *coverage_plp, *num_bases;
    int32_t min_bq4, min_alt_bq4, min_cov, pad_;
};

template <bool PACKED, bool STRAND, bool SAME_THR>
__device__ __forceinline__ void lfq_count_column_fast(const LfqCountArgs &T, lfq_col_counts *__restrict__ out,
                                                      uint8_t *__restrict__ flags, int64_t col, int lane)
{
    const LfqCountArgs &P = T;
    const uint64_t off0 = T.col_off[col], off1 = T.col_off[col + 1];
    const int64_t n_obs = (int64_t)(off1 - off0);
    const int cov = T.coverage_plp ? T.coverage_plp[col] : (int)n_obs;
    const int nb = T.num_bases ? T.num_bases[col] : (int)n_obs;
    const uint32_t rb = T.ref_base[col];
    const int ref_code = (rb == 'A') ? 0 : (rb == 'C') ? 1 : (rb == 'G') ? 2 : (rb == 'T') ? 3 : -1;
    /* gates: lofreq_call.c:892/754 (ref N; non-ACGT refs are N, plp.c:819-823), :930, :747 */
    const bool gated = (ref_code < 0) || ((int64_t)nb * 2 < (int64_t)cov) || (nb < P.min_cov);

    LfqAcc a;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        a.raw[x] = a.fw[x] = a.ge[x] = a.ga[x] = 0;
    }
    if (!gated && n_obs > 0) {
        const uint32_t minbq4 = 0x01010101u * (uint32_t)P.min_bq4;
        const uint32_t minalt4 = 0x01010101u * (uint32_t)P.min_alt_bq4;
        const int64_t cbeg = (int64_t)(off0 >> 4);
        const int n_ch = (int)((int64_t)((off1 + 15) >> 4) - cbeg);          /* chunks of 16 observations the column touches */
        const int lo = (int)(off0 & 15u);                                     /* first observation inside chunk 0 */
        const int hi_last = (int)((int64_t)off1 - ((cbeg + n_ch - 1) << 4));   /* observations of the last chunk: 1..16 */
        constexpr int UNROLL = 2;                   /* chunks in flight per lane (4 was measured: 72 registers, one workgroup
                                                     * per CU -- 1.5 % faster alone, slower beside another batch's DP kernels) */
        const uint4 *bq16 = reinterpret_cast<const uint4 *>(T.bq) + cbeg;
        /* one chunk: full masks inside the column, byte masks at its two ends (a wavefront meets them in its first and in
         * its last trip only) */
#define LFQ_FAST_CHUNK(I, NTV, B4)                                                                                   \
        do {                                                                                                         \
            const int i_ = (I);                                                                                      \
            if (__builtin_expect(i_ != 0 && i_ != n_ch - 1, 1)) {                                                    \
                if (PACKED) {                                                                                        \
                    lfq_count_nib8<SAME_THR, STRAND>(a, (NTV).x, (B4).x, (B4).y, 0x88888888u, minbq4, minalt4);      \
                    lfq_count_nib8<SAME_THR, STRAND>(a, (NTV).y, (B4).z, (B4).w, 0x88888888u, minbq4, minalt4);      \
                } else {                                                                                             \
                    lfq_count_dword<SAME_THR, STRAND>(a, (NTV).x, (B4).x, 0x80808080u, minbq4, minalt4);             \
                    lfq_count_dword<SAME_THR, STRAND>(a, (NTV).y, (B4).y, 0x80808080u, minbq4, minalt4);             \
                    lfq_count_dword<SAME_THR, STRAND>(a, (NTV).z, (B4).z, 0x80808080u, minbq4, minalt4);             \
                    lfq_count_dword<SAME_THR, STRAND>(a, (NTV).w, (B4).w, 0x80808080u, minbq4, minalt4);             \
                }                                                                                                    \
            } else {                                                                                                 \
                const int l_ = i_ == 0 ? lo : 0, h_ = i_ == n_ch - 1 ? hi_last : 16;                                 \
                if (PACKED) {                                                                                        \
                    lfq_count_nib8<SAME_THR, STRAND>(a, (NTV).x, (B4).x, (B4).y,                                     \
                                                     (lfq_bytes_mask(l_, h_, 0) >> 4) | lfq_bytes_mask(l_, h_, 1), minbq4, minalt4); \
                    lfq_count_nib8<SAME_THR, STRAND>(a, (NTV).y, (B4).z, (B4).w,                                     \
                                                     (lfq_bytes_mask(l_, h_, 2) >> 4) | lfq_bytes_mask(l_, h_, 3), minbq4, minalt4); \
                } else {                                                                                             \
                    lfq_count_dword<SAME_THR, STRAND>(a, (NTV).x, (B4).x, lfq_bytes_mask(l_, h_, 0), minbq4, minalt4); \
                    lfq_count_dword<SAME_THR, STRAND>(a, (NTV).y, (B4).y, lfq_bytes_mask(l_, h_, 1), minbq4, minalt4); \
                    lfq_count_dword<SAME_THR, STRAND>(a, (NTV).z, (B4).z, lfq_bytes_mask(l_, h_, 2), minbq4, minalt4); \
                    lfq_count_dword<SAME_THR, STRAND>(a, (NTV).w, (B4).w, lfq_bytes_mask(l_, h_, 3), minbq4, minalt4); \
                }                                                                                                    \
            }                                                                                                        \
        } while (0)
        int i = lane;
        if (PACKED) {
            const uint2 *nt8 = reinterpret_cast<const uint2 *>(T.nt) + cbeg;
            for (; i + (UNROLL - 1) * LFQ_WAVE < n_ch; i += UNROLL * LFQ_WAVE) {
                uint2 nv[UNROLL];
                uint4 bv[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; u++) {          /* UNROLL chunks' loads in flight per lane */
                    nv[u] = nt8[i + u * LFQ_WAVE];
                    bv[u] = bq16[i + u * LFQ_WAVE];
                }
#pragma unroll
                for (int u = 0; u < UNROLL; u++) {
                    const uint4 n4 = make_uint4(nv[u].x, nv[u].y, 0u, 0u);
                    LFQ_FAST_CHUNK(i + u * LFQ_WAVE, n4, bv[u]);
                }
            }
            for (; i < n_ch; i += LFQ_WAVE) {
                const uint2 na = nt8[i];
                const uint4 ba = bq16[i];
                const uint4 na4 = make_uint4(na.x, na.y, 0u, 0u);
                LFQ_FAST_CHUNK(i, na4, ba);
            }
        } else {
            const uint4 *nt16 = reinterpret_cast<const uint4 *>(T.nt) + cbeg;
            for (; i + (UNROLL - 1) * LFQ_WAVE < n_ch; i += UNROLL * LFQ_WAVE) {
                uint4 nv[UNROLL], bv[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; u++) {
                    nv[u] = nt16[i + u * LFQ_WAVE];
                    bv[u] = bq16[i + u * LFQ_WAVE];
                }
#pragma unroll
                for (int u = 0; u < UNROLL; u++) {
                    LFQ_FAST_CHUNK(i + u * LFQ_WAVE, nv[u], bv[u]);
                }
            }
            for (; i < n_ch; i += LFQ_WAVE) {
                const uint4 na = nt16[i];
                const uint4 ba = bq16[i];
                LFQ_FAST_CHUNK(i, na, ba);
            }
        }
#undef LFQ_FAST_CHUNK
    }

    /* the column's sums end up in lane 63, which writes the record */
    uint32_t n_raw[4], n_fw[4], n_ge[4], n_ga[4];
#pragma unroll
    for (int x = 0; x < 4; x++) {
        n_raw[x] = lfq_wave_sum_lane63_u32(a.raw[x]);
        n_fw[x] = STRAND ? lfq_wave_sum_lane63_u32(a.fw[x]) : 0u;
        n_ge[x] = lfq_wave_sum_lane63_u32(a.ge[x]);
        n_ga[x] = SAME_THR ? n_ge[x] : lfq_wave_sum_lane63_u32(a.ga[x]);
    }
    if (lane == LFQ_WAVE - 1) {
        uint32_t raw[4], fw[4], c_ge[4], c_ga[4], filt[4];
        lfq_planes_to_classes(n_raw, raw);
        lfq_planes_to_classes(n_fw, fw);
        lfq_planes_to_classes(n_ge, c_ge);
        lfq_planes_to_classes(n_ga, c_ga);
#pragma unroll
        for (int x = 0; x < 4; x++) {
            filt[x] = (x == ref_code) ? c_ge[x] : c_ga[x];     /* alt bases must pass both thresholds */
        }
        lfq_col_counts r;
        r.n_err_probs = 0;
        for (int k = 0; k < 3; k++) {
            r.alt_counts[k] = r.alt_raw_counts[k] = r.alt_fw[k] = 0;
        }
        r.ref_fw = r.ref_rv = 0;
        r.kmax = 0;
        r.tested = 0;
        r.pad_[0] = r.pad_[1] = 0;
        r.median_ref_bq = -1;
        r.coverage = cov;
        r.gated = gated;
        uint8_t flag = 0;
        if (!gated) {
            /* the three non-reference nucleotides in A,C,G,T order (snpcaller.c:391-397) */
            const int x0 = (ref_code == 0) ? 1 : 0;
            const int x1 = (ref_code <= 1) ? 2 : 1;
            const int x2 = (ref_code <= 2) ? 3 : 2;
#define LFQ_PICK(arr, x) ((x) == 0 ? arr[0] : (x) == 1 ? arr[1] : (x) == 2 ? arr[2] : arr[3])
            r.ref_fw = (int)LFQ_PICK(fw, ref_code);
            r.ref_rv = STRAND ? (int)(LFQ_PICK(raw, ref_code) - LFQ_PICK(fw, ref_code)) : 0;   /* lazy: lfq_strand_* */
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
            /* scheduling class: as in lfq_count_column */
            const int suspicious = max(12, r.n_err_probs / 512 + 8);
            flag = (uint8_t)((r.tested ? 1 : 0)
                             | ((kmax >= LFQ_BIG_K) ? 4 : (kmax >= LFQ_MID_K || kmax >= suspicious) ? 2 : 0));
        }
        out[col] = r;
        flags[col] = flag;
    }
}

/* ---- default filters + packed nt + lazy record counts: only the decision counts of every column ----
 * What the batches of lfq_call_vars / lfq_call_snvs_batch run when no dense strand counts are asked for (P.lazy_strand): the
 * planes restricted to bq >= min_bq are all that n_err_probs, alt_counts, kmax and `tested` need; alt_raw_counts and the
 * strand fields are 0 in the dense entry and counted by lfq_strand_heavy_kernel / lfq_strand_pvals_kernel for the columns
 * whose records can use them.  The interior chunks of the column run through a loop without a test for the column's two
 * ragged ends (lfq_count_nib8_lean, 15 instructions per 8 observations); the two end chunks go to lanes 63 and 62, which
 * have the fewest interior ones. */
/* a column's header: all of it through the scalar unit (the column index is the wavefront's), so that the header of a
 * wavefront's NEXT column can be requested before the current one is counted and costs no vector register */
struct LfqColHdr {
    uint64_t off0, off1;
    int32_t cov, nb;                 /* valid when the tracks carry them */
    uint32_t rb_word;                /* the aligned dword that holds the reference base */
};

__device__ __forceinline__ LfqColHdr lfq_load_col_hdr(const LfqCountArgs &T, int64_t col)
{
    LfqColHdr h;
    h.off0 = T.col_off[col];
    h.off1 = T.col_off[col + 1];
    h.cov = T.coverage_plp ? T.coverage_plp[col] : 0;
    h.nb = T.num_bases ? T.num_bases[col] : 0;
    /* the aligned 32-bit word AROUND the column's byte: up to three bytes in front of ref_base[0] / behind ref_base[ncols - 1] are
     * read (never used).  That cannot fault -- an aligned word that holds a valid byte lies in that byte's page -- and puts no
     * alignment requirement on a caller's device array; a memory checker may report it as a read past the array. */
    const uint8_t *p = T.ref_base + col;                 /* (pointer arithmetic, no integer round trip: the load stays a scalar one) */
    h.rb_word = *reinterpret_cast<const uint32_t *>(p - (reinterpret_cast<uintptr_t>(p) & 3u));
    return h;
}

template <bool SAME_THR, int UNROLL>
__device__ __forceinline__ void lfq_count_column_lean(const LfqCountArgs &T, lfq_col_counts *__restrict__ out,
                                                      uint8_t *__restrict__ flags, int64_t col, int lane, const LfqColHdr &H)
{
#ifdef LFQ_COUNT_STAMP      /* profiling build (profiles/wave_stamps.py): when a wavefront started, got its header, left its loop */
    const uint64_t st0 = wall_clock64();
#endif
    const uint64_t off0 = H.off0, off1 = H.off1;
    const int64_t n_obs = (int64_t)(off1 - off0);
    const int cov = T.coverage_plp ? H.cov : (int)n_obs;
    const int nb = T.num_bases ? H.nb : (int)n_obs;
    const uint32_t rb = (H.rb_word >> (8u * (uint32_t)(reinterpret_cast<uintptr_t>(T.ref_base + col) & 3u))) & 0xFFu;
    const int ref_code = (rb == 'A') ? 0 : (rb == 'C') ? 1 : (rb == 'G') ? 2 : (rb == 'T') ? 3 : -1;
    /* gates: lofreq_call.c:892/754 (ref N; non-ACGT refs are N, plp.c:819-823), :930, :747 */
    const bool gated = (ref_code < 0) || ((int64_t)nb * 2 < (int64_t)cov) || (nb < T.min_cov);

    uint32_t ge[4] = {0u, 0u, 0u, 0u}, ga[4] = {0u, 0u, 0u, 0u};
#ifdef LFQ_COUNT_STAMP
    uint64_t st1 = st0, st2 = st0;
#endif
    if (!gated && n_obs > 0) {
        const uint32_t kge = 0x01010101u * (uint32_t)(128 - T.min_bq4);          /* thresholds are clamped to 0..128 */
        const uint32_t kga = 0x01010101u * (uint32_t)(128 - T.min_alt_bq4);
        const int64_t cbeg = (int64_t)(off0 >> 4);
        const int n_ch = (int)((int64_t)((off1 + 15) >> 4) - cbeg);          /* chunks of 16 observations the column touches */
        const uint2 *nt8 = reinterpret_cast<const uint2 *>(T.nt) + cbeg;
        const uint4 *bq16 = reinterpret_cast<const uint4 *>(T.bq) + cbeg;
        const int n_in = n_ch - 1;                                            /* interior chunks: 1 .. n_ch - 2 */
#ifdef LFQ_COUNT_STAMP
        st1 = wall_clock64() + (uint64_t)(n_ch & 0);
#endif
        int i = 1 + lane;
        for (; i + (UNROLL - 1) * LFQ_WAVE < n_in; i += UNROLL * LFQ_WAVE) {  /* UNROLL chunks' loads in flight per lane */
            uint2 nv[UNROLL];
            uint4 bv[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                nv[u] = nt8[i + u * LFQ_WAVE];
                bv[u] = bq16[i + u * LFQ_WAVE];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                lfq_count_nib8_lean<SAME_THR>(ge, ga, nv[u].x, bv[u].x, bv[u].y, 0x88888888u, kge, kga);
                lfq_count_nib8_lean<SAME_THR>(ge, ga, nv[u].y, bv[u].z, bv[u].w, 0x88888888u, kge, kga);
            }
        }
        {                                                                     /* what is left: up to UNROLL - 1 chunks per lane */
            uint2 nv[UNROLL - 1];
            uint4 bv[UNROLL - 1];
#pragma unroll
            for (int u = 0; u < UNROLL - 1; u++) {
                const int at = i + u * LFQ_WAVE < n_in ? i + u * LFQ_WAVE : 0;    /* (chunk 0: there, and masked out below) */
                nv[u] = nt8[at];
                bv[u] = bq16[at];
            }
#pragma unroll
            for (int u = 0; u < UNROLL - 1; u++) {
                const uint32_t vm = i + u * LFQ_WAVE < n_in ? 0x88888888u : 0u;
                lfq_count_nib8_lean<SAME_THR>(ge, ga, nv[u].x, bv[u].x, bv[u].y, vm, kge, kga);
                lfq_count_nib8_lean<SAME_THR>(ge, ga, nv[u].y, bv[u].z, bv[u].w, vm, kge, kga);
            }
        }
        if (lane >= LFQ_WAVE - 2) {                                           /* the ends: byte masks */
            const bool first = lane == LFQ_WAVE - 1;
            const int ci = first ? 0 : n_ch - 1;
            if (first || n_ch > 1) {
                const int l_ = first ? (int)(off0 & 15u) : 0;
                const int h_ = ci == n_ch - 1 ? (int)((int64_t)off1 - ((cbeg + ci) << 4)) : 16;     /* 1..16 */
                const uint2 n0 = nt8[ci];
                const uint4 b0 = bq16[ci];
                lfq_count_nib8_lean<SAME_THR>(ge, ga, n0.x, b0.x, b0.y,
                                              (lfq_bytes_mask(l_, h_, 0) >> 4) | lfq_bytes_mask(l_, h_, 1), kge, kga);
                lfq_count_nib8_lean<SAME_THR>(ge, ga, n0.y, b0.z, b0.w,
                                              (lfq_bytes_mask(l_, h_, 2) >> 4) | lfq_bytes_mask(l_, h_, 3), kge, kga);
            }
        }
    }

#ifdef LFQ_COUNT_STAMP
    st2 = wall_clock64() + (uint64_t)((ge[0] + ge[1] + ge[2] + ge[3]) & 0u);
#endif
    /* the column's sums end up in lane 63, which writes the record */
    uint32_t n_ge[4], n_ga[4];
#pragma unroll
    for (int x = 0; x < 4; x++) {
        n_ge[x] = lfq_wave_sum_lane63_u32(ge[x]);
        n_ga[x] = SAME_THR ? n_ge[x] : lfq_wave_sum_lane63_u32(ga[x]);
    }
    if (lane == LFQ_WAVE - 1) {
        uint32_t c_ge[4], c_ga[4], filt[4];
        lfq_planes_to_classes(n_ge, c_ge);
        lfq_planes_to_classes(n_ga, c_ga);
#pragma unroll
        for (int x = 0; x < 4; x++) {
            filt[x] = (x == ref_code) ? c_ge[x] : c_ga[x];     /* alt bases must pass both thresholds */
        }
        lfq_col_counts r;
        r.n_err_probs = 0;
        for (int k = 0; k < 3; k++) {
            r.alt_counts[k] = r.alt_raw_counts[k] = r.alt_fw[k] = 0;
        }
        r.ref_fw = r.ref_rv = 0;
        r.kmax = 0;
        r.tested = 0;
        r.pad_[0] = r.pad_[1] = 0;
        r.median_ref_bq = -1;
        r.coverage = cov;
        r.gated = gated;
        uint8_t flag = 0;
        if (!gated) {
            /* the three non-reference nucleotides in A,C,G,T order (snpcaller.c:391-397) */
            const int x0 = (ref_code == 0) ? 1 : 0;
            const int x1 = (ref_code <= 1) ? 2 : 1;
            const int x2 = (ref_code <= 2) ? 3 : 2;
#define LFQ_PICK(arr, x) ((x) == 0 ? arr[0] : (x) == 1 ? arr[1] : (x) == 2 ? arr[2] : arr[3])
            r.alt_counts[0] = (int)LFQ_PICK(filt, x0);
            r.alt_counts[1] = (int)LFQ_PICK(filt, x1);
            r.alt_counts[2] = (int)LFQ_PICK(filt, x2);
#undef LFQ_PICK
            r.n_err_probs = (int)(filt[0] + filt[1] + filt[2] + filt[3]);
            const int kmax = max(r.alt_counts[0], max(r.alt_counts[1], r.alt_counts[2]));
            r.kmax = kmax;
            r.tested = kmax > 0;                     /* lofreq_call.c:768-780 */
            /* scheduling class: as in lfq_count_column */
            const int suspicious = max(12, r.n_err_probs / 512 + 8);
            flag = (uint8_t)((r.tested ? 1 : 0)
                             | ((kmax >= LFQ_BIG_K) ? 4 : (kmax >= LFQ_MID_K || kmax >= suspicious) ? 2 : 0));
        }
#ifdef LFQ_COUNT_STAMP
        {
            const uint64_t st3 = wall_clock64() + (uint64_t)(flag & 0);
            r.alt_raw_counts[0] = (int)(uint32_t)st0;
            r.alt_raw_counts[1] = (int)(uint32_t)(st0 >> 32);
            r.alt_raw_counts[2] = (int)(uint32_t)(st1 - st0);
            r.alt_fw[0] = (int)(uint32_t)(st2 - st0);
            r.alt_fw[1] = (int)(uint32_t)(st3 - st0);
            r.alt_fw[2] = (int)__builtin_amdgcn_s_getreg((4 /* HW_ID */) | (0 << 6) | (31 << 11));
            r.ref_fw = (int)__builtin_amdgcn_s_getreg((20 /* XCC_ID */) | (0 << 6) | (31 << 11));
        }
#endif
        out[col] = r;
        flags[col] = flag;
    }
}

/* CPW columns per wavefront, one after the other (a workgroup: CPW runs of WAVES neighbouring columns).  The headers of all
 * of them are requested at once when the wavefront starts -- scalar loads, they cost no vector register --, so that only the first
 * column waits for its header: two dependent scalar round trips, 3.3 of the 14.9 us a wavefront of one column lives.  A wavefront
 * that lives on round trips moves its bytes in proportion to how few of them it needs, and beside another batch's DP kernels the
 * count kernel has fewer wavefronts resident, not slower ones (profiles/wave_stamps.py).  The loop over the columns is unrolled:
 * as a loop it kept lane-dependent values and constants in vector registers across the columns (46 instead of 30 registers). */
template <bool SAME_THR, int WAVES, int UNROLL, int CPW>
__global__ __launch_bounds__(64 * WAVES, 8) void lfq_count_lean_kernel(LfqCountArgs T, lfq_col_counts *__restrict__ out,
                                                                    uint8_t *__restrict__ flags, int64_t c0, int64_t c1)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t first = c0 + (int64_t)blockIdx.x * (WAVES * CPW) + wave;
    if (first >= c1) {
        return;
    }
    LfqColHdr h[CPW];
#pragma unroll
    for (int j = 0; j < CPW; j++) {
        const int64_t col = first + (int64_t)j * WAVES;
        h[j] = lfq_load_col_hdr(T, col < c1 ? col : first);
    }
#pragma unroll
    for (int j = 0; j < CPW; j++) {
        const int64_t col = first + (int64_t)j * WAVES;
        if (col < c1) {
            /* (the lane index afresh for every column, from the execution mask: nothing lane-dependent, not even the index
             * itself, stays in a vector register from one column to the next -- the kernel has to stay within 32) */
            int lane_j;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_j));
            lfq_count_column_lean<SAME_THR, UNROLL>(T, out, flags, col, lane_j, h[j]);
        }
    }
}

/* one column per wavefront, WAVES columns per workgroup.  16 is the default (LFQ_COUNT_WAVES_PER_WG: 4, 8, 16): a 1024-thread
 * workgroup retires sixteen columns at once -- a quarter of the workgroups to dispatch (C3: 2.42 against 2.46 ms per
 * launch) and, for a caller that keeps several batches queued without a gate, four wave slots per SIMD freed at a time,
 * room for any workgroup of the other batch's DP kernels (a 256-thread workgroup's single slots starve the 512-thread
 * ones: profiles/NOTES.md). */
template <bool PACKED, bool STRAND, bool SAME_THR, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void lfq_count_fast_kernel(LfqCountArgs T,
                                                                    lfq_col_counts *__restrict__ out,
                                                                    uint8_t *__restrict__ flags, int64_t c0, int64_t c1)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t col = c0 + (int64_t)blockIdx.x * WAVES + wave;
    if (col >= c1) {
        return;
    }
    lfq_count_column_fast<PACKED, STRAND, SAME_THR>(T, out, flags, col, lfq_lane());
}

/* base_count() (plp.c:128-132) for every column: the bases of each nucleotide, whatever their quality -- what
 * `lofreq uniq`'s binomial test counts (lofreq_uniq.c:373).  One wavefront per column over the nt track with the
 * counting loop of lfq_count_kernel (bit planes of the raw counts); out[4 * col + x], x = A, C, G, T. */
template <bool PACKED>
__global__ __launch_bounds__(256) void lfq_ntcount_kernel(LfqTracksDev T, int32_t *__restrict__ out)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t col = (int64_t)blockIdx.x * 4 + wave;
    if (col >= T.ncols) {
        return;
    }
    const uint64_t off0 = T.col_off[col], off1 = T.col_off[col + 1];
    LfqAcc a;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        a.raw[x] = a.fw[x] = a.ge[x] = a.ga[x] = 0;
    }
    lfq_count_chunks<true, PACKED, false>(a, T, off0, off1, 0u, 0u);
    uint32_t n[4], cls[4];
#pragma unroll
    for (int x = 0; x < 4; x++) {
        n[x] = lfq_wave_sum_u32(a.raw[x]);
    }
    lfq_planes_to_classes(n, cls);
    if (lfq_lane() < 4) {
        const int l = lfq_lane();
        out[4 * col + l] = (int32_t)(l == 0 ? cls[0] : l == 1 ? cls[1] : l == 2 ? cls[2] : cls[3]);
    }
}

int lfq_launch_ntcount(const LfqTracksDev &t, int32_t *d_out, void *stream)
{
    if (t.ncols <= 0) {
        return LFQ_OK;
    }
    const unsigned blocks = (unsigned)((t.ncols + 3) / 4);
    if (t.nt_packed) {
        hipLaunchKernelGGL(lfq_ntcount_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, d_out);
    } else {
        hipLaunchKernelGGL(lfq_ntcount_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, d_out);
    }
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}

/* ------------------------------------------------------------------------------------------ */
/* scan kernels: running Bonferroni prefix + work-list compaction                              */
/* ------------------------------------------------------------------------------------------ */

#ifndef LFQ_SCAN_THREADS
#define LFQ_SCAN_THREADS 1024      /* (256 .. 1024: the workspace holds a tile sum per 1024 columns, lfq_api.hip) */
#endif
#define LFQ_SCAN_ITEMS 4
#define LFQ_SCAN_TILE (LFQ_SCAN_THREADS * LFQ_SCAN_ITEMS)

/* three counters scanned together: tested columns, mid columns, big columns */
struct LfqTriple {
    uint32_t t, m, b;
};

__device__ __forceinline__ LfqTriple lfq_triple_of_flag(uint32_t f)
{
    LfqTriple x;
    x.t = f & 1u;
    x.m = (f >> 1) & 1u;
    x.b = (f >> 2) & 1u;
    return x;
}

__device__ __forceinline__ void lfq_triple_add(LfqTriple &a, const LfqTriple &b)
{
    a.t += b.t;
    a.m += b.m;
    a.b += b.b;
}

/* block-wide exclusive scan; s_wave holds one triple per wave */
__device__ LfqTriple lfq_block_excl_scan(LfqTriple x, LfqTriple *total, LfqTriple *s_wave /*[16]*/)
{
    const int lane = lfq_lane(), wave = (int)(threadIdx.x >> 6);
    LfqTriple incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        LfqTriple y;
        y.t = (uint32_t)__shfl_up((int)incl.t, d, 64);
        y.m = (uint32_t)__shfl_up((int)incl.m, d, 64);
        y.b = (uint32_t)__shfl_up((int)incl.b, d, 64);
        if (lane >= d) {
            lfq_triple_add(incl, y);
        }
    }
    if (lane == 63) {
        s_wave[wave] = incl;
    }
    __syncthreads();
    LfqTriple wave_off = {0, 0, 0}, tot = {0, 0, 0};
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) {
        const LfqTriple v = s_wave[w];
        if (w < wave) {
            lfq_triple_add(wave_off, v);
        }
        lfq_triple_add(tot, v);
    }
    __syncthreads();
    *total = tot;
    LfqTriple ex;
    ex.t = wave_off.t + incl.t - x.t;
    ex.m = wave_off.m + incl.m - x.m;
    ex.b = wave_off.b + incl.b - x.b;
    return ex;
}

__global__ __launch_bounds__(LFQ_SCAN_THREADS) void lfq_scan_tiles_kernel(int64_t ncols,
                                                                         const uint8_t *__restrict__ flags,
                                                                         LfqTriple *__restrict__ tile_sums)
{
    __shared__ LfqTriple s_wave[16];
    const int64_t base = (int64_t)blockIdx.x * LFQ_SCAN_TILE + (int64_t)threadIdx.x * LFQ_SCAN_ITEMS;
    LfqTriple x = {0, 0, 0};
    for (int i = 0; i < LFQ_SCAN_ITEMS; i++) {
        if (base + i < ncols) {
            lfq_triple_add(x, lfq_triple_of_flag(flags[base + i]));
        }
    }
    LfqTriple total;
    (void)lfq_block_excl_scan(x, &total, s_wave);
    if (threadIdx.x == 0) {
        tile_sums[blockIdx.x] = total;
    }
}

/* single block: exclusive scan over the tile sums, totals into the counters.  RELIST: the second pass of a batch with the
 * Poisson gate (lfq_launch_approx_gate cleared the flag bytes of the columns it gave up): only the list sizes change --
 * the tested count and the Bonferroni carry are those of the first pass */
template <bool RELIST>
__global__ __launch_bounds__(LFQ_SCAN_THREADS) void lfq_scan_sums_kernel(int64_t ntiles,
                                                                        LfqTriple *__restrict__ tile_sums,
                                                                        int32_t *__restrict__ counters,
                                                                        int32_t *__restrict__ gcounters)
{
    __shared__ LfqTriple s_wave[16];
    LfqTriple carry = {0, 0, 0};
    for (int64_t b = 0; b < ntiles; b += LFQ_SCAN_THREADS) {
        const int64_t i = b + threadIdx.x;
        LfqTriple x = {0, 0, 0};
        if (i < ntiles) {
            x = tile_sums[i];
        }
        LfqTriple total;
        LfqTriple ex = lfq_block_excl_scan(x, &total, s_wave);
        if (i < ntiles) {
            lfq_triple_add(ex, carry);
            tile_sums[i] = ex;
        }
        lfq_triple_add(carry, total);
    }
    if (RELIST) {
        if (threadIdx.x == 0) {
            counters[LFQ_CNT_MID] = (int32_t)carry.m;
            counters[LFQ_CNT_BIG] = (int32_t)carry.b;
            counters[LFQ_CNT_LIGHT] = (int32_t)(carry.t - carry.m - carry.b);
        }
        if (threadIdx.x < 3) {                      /* the K histograms of the light class are counted again */
            counters[LFQ_CNT_KLE7 + (int)threadIdx.x] = 0;
        }
        if (threadIdx.x < LFQ_NKHIST) {
            counters[LFQ_CNT_KHIST + (int)threadIdx.x] = 0;
        }
        return;
    }
    if (threadIdx.x == 0) {
        counters[LFQ_CNT_TESTED] = (int32_t)carry.t;
        counters[LFQ_CNT_MID] = (int32_t)carry.m;
        counters[LFQ_CNT_BIG] = (int32_t)carry.b;
        counters[LFQ_CNT_LIGHT] = (int32_t)(carry.t - carry.m - carry.b);
        /* running Bonferroni carry between the segments of a batch (stream-ordered) */
        const int32_t before = gcounters[LFQ_GC_TESTED];
        counters[LFQ_CNT_CARRY_IN] = before;
        gcounters[LFQ_GC_TESTED] = before + (int32_t)carry.t;
    }
}

template <bool RELIST>
__global__ __launch_bounds__(LFQ_SCAN_THREADS) void lfq_scan_apply_kernel(LfqTracksDev T, int64_t c0, int64_t c1,
                                                                         const uint8_t *__restrict__ flags,
                                                                         const lfq_col_counts *__restrict__ counts,
                                                                         const LfqTriple *__restrict__ tile_sums,
                                                                         LfqWork W)
{
    __shared__ LfqTriple s_wave[16];
    const int64_t ncols = c1;
    const int64_t base = c0 + (int64_t)blockIdx.x * LFQ_SCAN_TILE + (int64_t)threadIdx.x * LFQ_SCAN_ITEMS;
    uint32_t f[LFQ_SCAN_ITEMS];
    LfqTriple x = {0, 0, 0};
    for (int i = 0; i < LFQ_SCAN_ITEMS; i++) {
        f[i] = (base + i < ncols) ? flags[base + i] : 0u;
        lfq_triple_add(x, lfq_triple_of_flag(f[i]));
    }
    LfqTriple total;
    LfqTriple ex = lfq_block_excl_scan(x, &total, s_wave);
    lfq_triple_add(ex, tile_sums[blockIdx.x]);
    /* list layout [light | mid | big]; the class totals were published by lfq_scan_sums_kernel */
    const uint32_t base_mid = (uint32_t)W.counters[LFQ_CNT_LIGHT];
    const uint32_t base_big = base_mid + (uint32_t)W.counters[LFQ_CNT_MID];
    const uint32_t carry_in = (uint32_t)W.counters[LFQ_CNT_CARRY_IN];
    uint32_t kle7 = 0, kle15 = 0, kle31 = 0;       /* K histogram of the light class (lfq_light_group_lanes) */
    unsigned long long khist = 0;
    for (int i = 0; i < LFQ_SCAN_ITEMS; i++) {
        const int64_t c = base + i;
        if (c >= ncols) {
            break;
        }
        if (f[i] & 1u) {
            uint32_t pos;
            const lfq_col_counts *cn = &counts[c];
            if (f[i] & 4u) {
                pos = base_big + ex.b++;
            } else if (f[i] & 2u) {
                pos = base_mid + ex.m++;
            } else {
                pos = ex.t - ex.m - ex.b;
                kle7 += cn->kmax <= 7;
                kle15 += cn->kmax <= 15;
                kle31 += cn->kmax <= 31;
                /* finer histogram for the screen kernel's register variants: 9-bit fields, <= 4 per thread */
#pragma unroll
                for (int f = 0; f < LFQ_NKHIST; f++) {
                    khist += (unsigned long long)(cn->kmax <= lfq_khist_thr(f)) << (9 * f);
                }
            }
            ex.t++;
            const uint32_t rb = T.ref_base[c];
            LfqEntry e;
            e.off0 = T.col_off[c];
            e.n_obs = (int32_t)(T.col_off[c + 1] - e.off0);
            e.col = (int32_t)c;
            /* inclusive, batch-wide (RELIST: as the first pass left it -- columns the gate gave up still count) */
            e.prefix = RELIST ? W.tested_prefix[c] : (int32_t)(carry_in + ex.t);
            e.kmax = cn->kmax;
            e.median_ref_bq = (int16_t)cn->median_ref_bq;
            e.ref_code = (uint8_t)((rb == 'A') ? 0 : (rb == 'C') ? 1 : (rb == 'G') ? 2 : 3);
            e.pad_ = 0;
            e.pad2_ = 0;
            W.entries[pos] = e;
        }
        if (!RELIST) {
            W.tested_prefix[c] = (int32_t)(carry_in + ex.t);   /* inclusive, batch-wide */
        }
    }
    /* one set of global atomics per workgroup (per wavefront they contend: +0.1 ms on a 1 M column batch) */
    __shared__ uint32_t s_k[3 + LFQ_NKHIST];
    if (threadIdx.x < 3 + LFQ_NKHIST) {
        s_k[threadIdx.x] = 0;
    }
    __syncthreads();
    kle7 = lfq_wave_sum_u32(kle7);
    kle15 = lfq_wave_sum_u32(kle15);
    kle31 = lfq_wave_sum_u32(kle31);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        khist += __shfl_xor(khist, d, 64);          /* 64 lanes x <= 4: every field stays below 2^9 */
    }
    if (lfq_lane() == 0 && kle31) {
        atomicAdd(&s_k[0], kle7);
        atomicAdd(&s_k[1], kle15);
        atomicAdd(&s_k[2], kle31);
#pragma unroll
        for (int f = 0; f < LFQ_NKHIST; f++) {
            atomicAdd(&s_k[3 + f], (uint32_t)((khist >> (9 * f)) & 511ull));
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 && s_k[2]) {
        atomicAdd(&W.counters[LFQ_CNT_KLE7 + (int)threadIdx.x], (int)s_k[threadIdx.x]);
    }
    if (threadIdx.x >= 3 && threadIdx.x < 3 + LFQ_NKHIST && s_k[2]) {
        atomicAdd(&W.counters[LFQ_CNT_KHIST + (int)threadIdx.x - 3], (int)s_k[threadIdx.x]);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* synthetic workload generator (include/lofreq_synth.h)                                       */
/* ------------------------------------------------------------------------------------------ */

__global__ __launch_bounds__(256) void lfq_synth_kernel(lfq_synth_spec S, int64_t col_begin, int64_t ncols,
                                                        uint8_t *__restrict__ nt, uint8_t *__restrict__ bq,
                                                        uint8_t *__restrict__ baq, uint8_t *__restrict__ mq,
                                                        uint64_t *__restrict__ col_off,
                                                        uint8_t *__restrict__ ref_base, int nt_packed)
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
        if (nt_packed) {
            /* two groups of 8 observations -> 2 x 4 bytes: byte k = observation k | observation 4 + k << 4 (lfq_nt_at) */
            const uint32_t g0 = (wn[0] & 0x0F0F0F0Fu) | ((wn[1] & 0x0F0F0F0Fu) << 4);
            const uint32_t g1 = (wn[2] & 0x0F0F0F0Fu) | ((wn[3] & 0x0F0F0F0Fu) << 4);
            reinterpret_cast<uint2 *>(nt)[t] = make_uint2(g0, g1);
        } else {
            reinterpret_cast<uint4 *>(nt)[t] = make_uint4(wn[0], wn[1], wn[2], wn[3]);
        }
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

__global__ __launch_bounds__(256) void lfq_maxdepth_kernel(const uint64_t *__restrict__ col_off, int64_t ncols,
                                                           int32_t *__restrict__ gcounters)
{
    int best = 0;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t d = col_off[c + 1] - col_off[c];
        best = max(best, (int)min(d, (uint64_t)0x7fffffff));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        best = max(best, __shfl_xor(best, d, 64));
    }
    if (lfq_lane() == 0) {
        atomicMax(&gcounters[LFQ_GC_MAXDEPTH], best);
    }
}

int lfq_launch_maxdepth(const LfqTracksDev &t, int32_t *d_gcounters, void *stream)
{
    if (t.ncols <= 0) {
        return LFQ_OK;
    }
    const unsigned blocks = (unsigned)std::min<int64_t>((t.ncols + 255) / 256, 1024);
    hipLaunchKernelGGL(lfq_maxdepth_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t.col_off, t.ncols,
                       d_gcounters);
    LFQ_HIP_TRY(hipGetLastError());
    return LFQ_OK;
}

/* does a batch with this deepest column take lfq_count_shallow_kernel (the only one that honours LfqParams::sparse_counts) */
bool lfq_count_is_shallow(const LfqTracksDev &t, const LfqParams &p, int64_t max_col_obs)
{
    return !p.general && !p.detlim_af && max_col_obs > 0 && max_col_obs < lfq_knobs().count_multi_below && t.nt_packed;
}

int lfq_launch_count(const LfqTracksDev &t, int64_t c0, int64_t c1, const LfqParams &p, const LfqLuts *d_luts,
                     lfq_col_counts *d_counts, uint8_t *d_flags, int64_t max_col_obs, void *stream, int shallow_wgs_per_cu)
{
    if (c1 <= c0) {
        return LFQ_OK;
    }
    const int64_t multi_below = lfq_knobs().count_multi_below;   /* deepest column of the batch below this: four columns per wavefront */
    if (!p.general && !p.detlim_af && max_col_obs > 0 && max_col_obs < multi_below) {
        /* lanes per column by the deepest column of the batch: each lane takes chunks of 16 observations */
        const LfqKnobs &kn = lfq_knobs();
        const int lpg = max_col_obs <= kn.count_lpg4_below ? 4 : max_col_obs <= kn.count_lpg8_below ? 8 : 16;
        if (t.nt_packed) {
            /* 64 columns per wavefront and pass; enough blocks for every SIMD's wavefronts, several passes each */
            const int64_t blocks64 = (c1 - c0 + 255) / 256;
            const unsigned nb = (unsigned)std::min<int64_t>(blocks64, (int64_t)256 * 8 * 4);
            /* chunks the deepest column can touch (an unaligned start adds one), in rounds of AHEAD per lane */
            const int rounds = (int)((max_col_obs / 16 + 2 + LFQ_COUNT_AHEAD * lpg - 1) / (LFQ_COUNT_AHEAD * lpg));
            /* Unused dynamic LDS per workgroup = fewer workgroups per CU (33 / 37 KB each: four fit).  A context whose batches are
             * queued without a gate asks for two: its count kernel then leaves half of every SIMD's registers and wave slots to
             * the DP kernels of the batch before -- which otherwise start only where a count workgroup retires -- and is itself
             * no slower (C2, four batches queued: 0.66 -> 0.56-0.59 ms per step, count kernel 0.33-0.38 ms either way).
             * LFQ_COUNT_SHALLOW_LDS_PAD: the bytes directly (experiments). */
            unsigned lds_pad = (unsigned)kn.count_shallow_lds_pad;
            /* (not where four lanes share a column -- depth <= 320 --: 3.75 M x 200 0.86 ms per step with four workgroups, 1.14 with two) */
            if (lds_pad == 0 && shallow_wgs_per_cu >= 1 && shallow_wgs_per_cu <= 3 && lpg > 4) {
                lds_pad = shallow_wgs_per_cu == 1 ? 60000u : shallow_wgs_per_cu == 2 ? 44000u : 17000u;
            }
#define LFQ_LAUNCH_SHALLOW(ST)                                                                                       \
            do {                                                                                                     \
                if (lpg == 4) hipLaunchKernelGGL((lfq_count_shallow_kernel<ST, 4>), dim3(nb), dim3(256), lds_pad, (hipStream_t)stream, t, p, d_counts, d_flags, c0, c1, rounds); \
                else if (lpg == 8) hipLaunchKernelGGL((lfq_count_shallow_kernel<ST, 8>), dim3(nb), dim3(256), lds_pad, (hipStream_t)stream, t, p, d_counts, d_flags, c0, c1, rounds); \
                else hipLaunchKernelGGL((lfq_count_shallow_kernel<ST, 16>), dim3(nb), dim3(256), lds_pad, (hipStream_t)stream, t, p, d_counts, d_flags, c0, c1, rounds); \
            } while (0)
            if (!p.lazy_strand) LFQ_LAUNCH_SHALLOW(true); else LFQ_LAUNCH_SHALLOW(false);
#undef LFQ_LAUNCH_SHALLOW
            LFQ_HIP_TRY(hipGetLastError());
            return LFQ_OK;
        }
        const int64_t per_block = 4 * (64 / lpg);
        const unsigned blocks = (unsigned)((c1 - c0 + per_block - 1) / per_block);
#define LFQ_LAUNCH_MULTI_L(PK, ST, L)                                                                                \
        hipLaunchKernelGGL((lfq_count_multi_kernel<PK, ST, L>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, p,  \
                           d_counts, d_flags, c0, c1)
#define LFQ_LAUNCH_MULTI(PK, ST)                                                                                     \
        do {                                                                                                         \
            if (lpg == 4) LFQ_LAUNCH_MULTI_L(PK, ST, 4);                                                             \
            else if (lpg == 8) LFQ_LAUNCH_MULTI_L(PK, ST, 8);                                                        \
            else LFQ_LAUNCH_MULTI_L(PK, ST, 16);                                                                     \
        } while (0)
        const bool strand_m = !p.lazy_strand;
        if (t.nt_packed) {
            if (strand_m) LFQ_LAUNCH_MULTI(true, true); else LFQ_LAUNCH_MULTI(true, false);
        } else {
            if (strand_m) LFQ_LAUNCH_MULTI(false, true); else LFQ_LAUNCH_MULTI(false, false);
        }
#undef LFQ_LAUNCH_MULTI_L
#undef LFQ_LAUNCH_MULTI
        LFQ_HIP_TRY(hipGetLastError());
        return LFQ_OK;
    }
    const bool strand = !p.lazy_strand || p.general;       /* the general path evaluates every observation anyway */
    const LfqKnobs &kn = lfq_knobs();
    if (!p.general && !p.detlim_af) {
        /* default filters: the lean kernel, by layout x strand counts x "one threshold" */
        const bool same_thr = p.min_alt_bq4 == p.min_bq4;
        LfqCountArgs ca;
        ca.nt = t.nt;
        ca.bq = t.bq;
        ca.col_off = t.col_off;
        ca.ref_base = t.ref_base;
        ca.coverage_plp = t.coverage_plp;
        ca.num_bases = t.num_bases;
        ca.min_bq4 = p.min_bq4;
        ca.min_alt_bq4 = p.min_alt_bq4;
        ca.min_cov = p.min_cov;
        ca.pad_ = 0;
        const int variant = (t.nt_packed ? 4 : 0) | (strand ? 2 : 0) | (same_thr ? 1 : 0);
        const int wpw = kn.count_waves_per_wg;
        const int cpw = kn.count_cols_per_wave >= 4 ? 4 : kn.count_cols_per_wave >= 2 ? 2 : 1;
        const unsigned blocks = (unsigned)((c1 - c0 + (int64_t)wpw * cpw - 1) / ((int64_t)wpw * cpw));
        if (t.nt_packed && !strand) {
            /* lazy record counts on the packed layout: the decision counts only */
            /* (LFQ_COUNT_LEAN_LDS_PAD: unused dynamic LDS per workgroup -- the kernel has no LDS of its own, so this alone says
             * how many workgroups a CU holds and how many wave slots stay free for another batch's DP kernels) */
            const unsigned lean_pad = (unsigned)kn.count_lean_lds_pad;
#define LFQ_LAUNCH_LC(SM, W, U, CP)                                                                                  \
    hipLaunchKernelGGL((lfq_count_lean_kernel<SM, W, U, CP>), dim3(blocks), dim3(64 * W), lean_pad, (hipStream_t)stream, ca, d_counts, \
                       d_flags, c0, c1)
#define LFQ_LAUNCH_L(SM, W, U)                                                                                       \
    do {                                                                                                             \
        if (cpw == 4) LFQ_LAUNCH_LC(SM, W, U, 4);                                                                    \
        else if (cpw == 2) LFQ_LAUNCH_LC(SM, W, U, 2);                                                               \
        else LFQ_LAUNCH_LC(SM, W, U, 1);                                                                             \
    } while (0)
#define LFQ_LAUNCH_LU(SM, W)                                                                                         \
    do {                                                                                                             \
        if (kn.count_ahead_deep == 4) LFQ_LAUNCH_L(SM, W, 4);                                                        \
        else if (kn.count_ahead_deep == 3) LFQ_LAUNCH_L(SM, W, 3);                                                   \
        else LFQ_LAUNCH_L(SM, W, 2);                                                                                 \
    } while (0)
#define LFQ_LAUNCH_LW(SM)                                                                                            \
    do {                                                                                                             \
        if (wpw == 16) LFQ_LAUNCH_LU(SM, 16);                                                                        \
        else if (wpw == 12) LFQ_LAUNCH_LU(SM, 12);                                                                   \
        else if (wpw == 8) LFQ_LAUNCH_LU(SM, 8);                                                                     \
        else LFQ_LAUNCH_LU(SM, 4);                                                                                   \
    } while (0)
            if (same_thr) LFQ_LAUNCH_LW(true); else LFQ_LAUNCH_LW(false);
#undef LFQ_LAUNCH_LW
#undef LFQ_LAUNCH_LU
#undef LFQ_LAUNCH_L
#undef LFQ_LAUNCH_LC
            LFQ_HIP_TRY(hipGetLastError());
            return LFQ_OK;
        }
#define LFQ_LAUNCH_F(PK, ST, SM, W)                                                                                  \
    hipLaunchKernelGGL((lfq_count_fast_kernel<PK, ST, SM, W>), dim3((unsigned)((c1 - c0 + W - 1) / W)), dim3(64 * W), 0,     \
                       (hipStream_t)stream, ca, d_counts, d_flags, c0, c1)
#define LFQ_LAUNCH_FW(PK, ST, SM)                                                                                    \
    do {                                                                                                             \
        if (wpw >= 12) LFQ_LAUNCH_F(PK, ST, SM, 16);                                                                 \
        else if (wpw == 8) LFQ_LAUNCH_F(PK, ST, SM, 8);                                                              \
        else LFQ_LAUNCH_F(PK, ST, SM, 4);                                                                            \
    } while (0)
        switch (variant) {
        case 0: LFQ_LAUNCH_FW(false, false, false); break;
        case 1: LFQ_LAUNCH_FW(false, false, true); break;
        case 2: LFQ_LAUNCH_FW(false, true, false); break;
        case 3: LFQ_LAUNCH_FW(false, true, true); break;
        case 6: LFQ_LAUNCH_FW(true, true, false); break;
        default: LFQ_LAUNCH_FW(true, true, true); break;
        }
#undef LFQ_LAUNCH_FW
#undef LFQ_LAUNCH_F
        LFQ_HIP_TRY(hipGetLastError());
        return LFQ_OK;
    }
    const unsigned blocks = (unsigned)((c1 - c0 + 3) / 4);
#define LFQ_LAUNCH_COUNT(PK, ST)                                                                                     \
    hipLaunchKernelGGL((lfq_count_kernel<PK, ST>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, p, d_luts,       \
                       d_counts, d_flags, c0, c1)
    if (t.nt_packed) {
        if (strand) LFQ_LAUNCH_COUNT(true, true); else LFQ_LAUNCH_COUNT(true, false);
    } else {
        if (strand) LFQ_LAUNCH_COUNT(false, true); else LFQ_LAUNCH_COUNT(false, false);
    }
#undef LFQ_LAUNCH_COUNT
    LFQ_HIP_TRY(hipGetLastError());
    return LFQ_OK;
}

int lfq_launch_scan(const LfqTracksDev &t, int64_t c0, int64_t c1, const uint8_t *d_flags,
                    const lfq_col_counts *d_counts, const LfqWork &w, void *stream, bool relist)
{
    const int64_t ncols = c1 - c0;
    if (ncols <= 0) {
        return LFQ_OK;
    }
    const int64_t ntiles = (ncols + LFQ_SCAN_TILE - 1) / LFQ_SCAN_TILE;
    LfqTriple *tile_sums = reinterpret_cast<LfqTriple *>(w.block_sums);
    hipLaunchKernelGGL(lfq_scan_tiles_kernel, dim3((unsigned)ntiles), dim3(LFQ_SCAN_THREADS), 0,
                       (hipStream_t)stream, ncols, d_flags + c0, tile_sums);
    if (relist) {
        hipLaunchKernelGGL(lfq_scan_sums_kernel<true>, dim3(1), dim3(LFQ_SCAN_THREADS), 0, (hipStream_t)stream, ntiles,
                           tile_sums, w.counters, w.gcounters);
        hipLaunchKernelGGL(lfq_scan_apply_kernel<true>, dim3((unsigned)ntiles), dim3(LFQ_SCAN_THREADS), 0,
                           (hipStream_t)stream, t, c0, c1, d_flags, d_counts, (const LfqTriple *)tile_sums, w);
    } else {
        hipLaunchKernelGGL(lfq_scan_sums_kernel<false>, dim3(1), dim3(LFQ_SCAN_THREADS), 0, (hipStream_t)stream, ntiles,
                           tile_sums, w.counters, w.gcounters);
        hipLaunchKernelGGL(lfq_scan_apply_kernel<false>, dim3((unsigned)ntiles), dim3(LFQ_SCAN_THREADS), 0,
                           (hipStream_t)stream, t, c0, c1, d_flags, d_counts, (const LfqTriple *)tile_sums, w);
    }
    LFQ_HIP_TRY(hipGetLastError());
    return LFQ_OK;
}

int lfq_launch_synth(const lfq_synth_spec *spec, int64_t col_begin, int64_t ncols, uint8_t *d_nt,
                     uint8_t *d_bq, uint8_t *d_baq, uint8_t *d_mq, uint64_t *d_col_off, uint8_t *d_ref_base,
                     int nt_packed, void *stream)
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
                       col_begin, ncols, d_nt, d_bq, d_baq, d_mq, d_col_off, d_ref_base, nt_packed);
    LFQ_HIP_TRY(hipGetLastError());
    return LFQ_OK;
}


/* ------------------------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------------------------ */
/* lazy strand counts: DP4 only where something is reported                                    */
/* ------------------------------------------------------------------------------------------ */
/* The forward-strand planes are 8 of the ~30 instructions the count kernel spends per 4 observations, and the
 * kernel is as much VALU- as HBM-bound -- but ref_fw / ref_rv / alt_fw only ever reach report_var (DP4, SB,
 * lofreq_call.c:117-129, 853-857), i.e. the ~1e-3 of the columns that emit a record.  In lazy mode the count kernel
 * skips them; these two kernels count them afterwards, one wavefront per column: for the heavy columns right after
 * the scan (their DP4 tuples feed the host's strand-bias precompute while the DP runs), and for every record of the
 * sparse output once the DP kernels are done. */
template <bool PACKED>
__device__ __forceinline__ void lfq_strand_of_column(const LfqTracksDev &T, int64_t col, int *ref_fw, int *ref_rv,
                                                     int alt_fw[3], int alt_raw[3])
{
    const uint64_t off0 = T.col_off[col], off1 = T.col_off[col + 1];
    LfqAcc a;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a.raw[i] = a.fw[i] = a.ge[i] = a.ga[i] = 0;
    }
    lfq_count_chunks<true, PACKED, true>(a, T, off0, off1, 0x80808080u, 0x80808080u);
    uint32_t raw[4], fw[4], rp[4], fp[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        rp[i] = lfq_wave_sum_u32(a.raw[i]);
        fp[i] = lfq_wave_sum_u32(a.fw[i]);
    }
    lfq_planes_to_classes(rp, raw);
    lfq_planes_to_classes(fp, fw);
    const uint32_t rb = T.ref_base[col];
    const int ref_code = (rb == 'A') ? 0 : (rb == 'C') ? 1 : (rb == 'G') ? 2 : (rb == 'T') ? 3 : -1;
    *ref_fw = *ref_rv = 0;
    alt_fw[0] = alt_fw[1] = alt_fw[2] = 0;
    alt_raw[0] = alt_raw[1] = alt_raw[2] = 0;
    if (ref_code < 0) {
        return;
    }
    int ai = 0;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        if (x == ref_code) {
            *ref_fw = (int)fw[x];
            *ref_rv = (int)(raw[x] - fw[x]);
        } else {
            alt_fw[ai] = (int)fw[x];
            alt_raw[ai] = (int)raw[x];
            ai++;
        }
    }
}

template <bool PACKED>
__global__ __launch_bounds__(256) void lfq_strand_heavy_kernel(LfqTracksDev T, LfqWork W, lfq_col_counts *__restrict__ counts,
                                                               int32_t *__restrict__ tuples, int32_t *__restrict__ n_out,
                                                               int cap_entries, int min_alt)
{
    const int n_light = W.counters[LFQ_CNT_LIGHT];
    const int n_heavy = min(W.counters[LFQ_CNT_MID] + W.counters[LFQ_CNT_BIG], cap_entries);
    const int i = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        __hip_atomic_store(n_out, n_heavy, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (i >= n_heavy) {
        return;
    }
    const int64_t col = W.entries[n_light + i].col;
    int ref_fw, ref_rv, alt_fw[3], alt_raw[3];
    lfq_strand_of_column<PACKED>(T, col, &ref_fw, &ref_rv, alt_fw, alt_raw);
    if (lfq_lane() == 0) {
        counts[col].ref_fw = ref_fw;                 /* the DP kernels copy the dense entry into their records */
        counts[col].ref_rv = ref_rv;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            counts[col].alt_fw[a] = alt_fw[a];
            counts[col].alt_raw_counts[a] = alt_raw[a];
            int4 t = make_int4(0, 0, 0, 0);
            if (alt_raw[a] >= min_alt) {
                t = make_int4(ref_fw, ref_rv, alt_fw[a], alt_raw[a] - alt_fw[a]);
            }
            reinterpret_cast<int4 *>(tuples)[3 * i + a] = t;
        }
    }
}

template <bool PACKED>
__global__ __launch_bounds__(256) void lfq_strand_pvals_kernel(LfqTracksDev T, lfq_col_pvals *__restrict__ pvals,
                                                               const int32_t *__restrict__ n_pvals_ptr, int64_t cap)
{
    const int64_t n = min((int64_t)*n_pvals_ptr, cap);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += n_waves) {
        int ref_fw, ref_rv, alt_fw[3], alt_raw[3];
        lfq_strand_of_column<PACKED>(T, pvals[i].col, &ref_fw, &ref_rv, alt_fw, alt_raw);
        if (lfq_lane() == 0) {
            pvals[i].counts.ref_fw = ref_fw;
            pvals[i].counts.ref_rv = ref_rv;
            pvals[i].counts.alt_fw[0] = alt_fw[0];
            pvals[i].counts.alt_fw[1] = alt_fw[1];
            pvals[i].counts.alt_fw[2] = alt_fw[2];
            pvals[i].counts.alt_raw_counts[0] = alt_raw[0];
            pvals[i].counts.alt_raw_counts[1] = alt_raw[1];
            pvals[i].counts.alt_raw_counts[2] = alt_raw[2];
        }
    }
}

int lfq_launch_strand_heavy(const LfqTracksDev &t, const LfqWork &w, lfq_col_counts *d_counts, int32_t *tuples_mapped,
                            int32_t *n_mapped, int cap_entries, int min_alt, void *stream)
{
    const unsigned blocks = (unsigned)((cap_entries + 3) / 4);
    if (t.nt_packed) {
        hipLaunchKernelGGL(lfq_strand_heavy_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, w, d_counts,
                           tuples_mapped, n_mapped, cap_entries, min_alt);
    } else {
        hipLaunchKernelGGL(lfq_strand_heavy_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, w, d_counts,
                           tuples_mapped, n_mapped, cap_entries, min_alt);
    }
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}

int lfq_launch_strand_pvals(const LfqTracksDev &t, lfq_col_pvals *d_pvals, const int32_t *d_n_pvals, int64_t cap, int n_blocks,
                            void *stream)
{
    if (t.nt_packed) {
        hipLaunchKernelGGL(lfq_strand_pvals_kernel<true>, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, t, d_pvals,
                           d_n_pvals, cap);
    } else {
        hipLaunchKernelGGL(lfq_strand_pvals_kernel<false>, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, t, d_pvals,
                           d_n_pvals, cap);
    }
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}

/* ------------------------------------------------------------------------------------------ */
/* DP4 tuples of the columns with many alt bases -> host-mapped memory (strand-bias precompute) */

/* ------------------------------------------------------------------------------------------ */

__global__ __launch_bounds__(256) void lfq_gather_heavy_kernel(LfqWork W, const lfq_col_counts *__restrict__ counts,
                                                               int32_t *__restrict__ tuples, int32_t *__restrict__ n_out,
                                                               int cap_entries, int min_alt)
{
    const int n_light = W.counters[LFQ_CNT_LIGHT];
    const int n_heavy = min(W.counters[LFQ_CNT_MID] + W.counters[LFQ_CNT_BIG], cap_entries);
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i == 0) {
        __hip_atomic_store(n_out, n_heavy, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (i >= n_heavy) {
        return;
    }
    const LfqEntry en = W.entries[n_light + i];
    const lfq_col_counts c = counts[en.col];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        int4 t = make_int4(0, 0, 0, 0);
        if (c.alt_raw_counts[a] >= min_alt) {
            t = make_int4(c.ref_fw, c.ref_rv, c.alt_fw[a], c.alt_raw_counts[a] - c.alt_fw[a]);
        }
        reinterpret_cast<int4 *>(tuples)[3 * i + a] = t;
    }
}

int lfq_launch_gather_heavy(const LfqWork &w, const lfq_col_counts *d_counts, int32_t *tuples_mapped,
                            int32_t *n_mapped, int cap_entries, int min_alt, void *stream)
{
    const unsigned blocks = (unsigned)((cap_entries + 255) / 256);
    hipLaunchKernelGGL(lfq_gather_heavy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, d_counts,
                       tuples_mapped, n_mapped, cap_entries, min_alt);
    return hipGetLastError() == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
}
