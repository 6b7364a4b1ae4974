/*
 * lfq_api.hip -- C-ABI entry points that own device state: context, workspace, the batch driver
 * (layer 1) and the call_snvs batch loop (layer 2).  See include/lofreq_amd.h.
 *
 * There is deliberately no CPU fallback in this file: every compute entry point needs a HIP
 * device and fails with LFQ_ERR_NO_DEVICE / LFQ_ERR_HIP otherwise.
 */
#include <hip/hip_runtime.h>

#include <ctype.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <atomic>
#include <functional>
#include <unistd.h>
#include <fcntl.h>
#include <sys/file.h>
#include <thread>
#include <vector>

#include "lfq_internal.h"
#include "lofreq_synth.h"

#define LFQ_TRY_HIP(expr)           \
    do {                            \
        hipError_t e_ = (expr);     \
        if (e_ != hipSuccess) {     \
            return LFQ_ERR_HIP;     \
        }                           \
    } while (0)

#define LFQ_TRY(expr)               \
    do {                            \
        int rc_ = (expr);           \
        if (rc_ != LFQ_OK) {        \
            return rc_;             \
        }                           \
    } while (0)

/* the columns handed out by lfq_pileup_indel_columns live here until the next call */
static inline double lfq_now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define lfq_timing_on (lfq_knobs().timing != 0)

/* The helper threads of the host loops below.  A region's steps run a dozen such loops milliseconds apart; threads
 * created (or woken from a condition variable) per loop start on idle cores and the same loop took anything between
 * 0.7 and 4.5 ms (BAQ geometry of 400 K reads, 4 threads; one thread: 2.4 ms).  These seven stay: after a loop they spin
 * for LFQ_HOST_SPIN_US (2000) microseconds waiting for the next one before they go to sleep.  One loop at a time
 * (try_run fails when another thread is using the pool, and in a forked child: the caller then creates threads). */
#if defined(__x86_64__) || defined(__i386__)
#define LFQ_CPU_PAUSE() __builtin_ia32_pause()
#elif defined(__aarch64__)
#define LFQ_CPU_PAUSE() __asm__ __volatile__("yield")
#else
#define LFQ_CPU_PAUSE() std::this_thread::yield()
#endif

class LfqLoopPool {
public:
    static LfqLoopPool &instance()
    {
        static LfqLoopPool p;
        return p;
    }
    bool try_run(int parts, const std::function<void(int)> &task)       /* task(1 .. parts - 1) here, part 0 by the caller */
    {
        if (parts - 1 > (int)th_.size() || getpid() != pid_ || !call_m_.try_lock()) {
            return false;
        }
        /* every helper acknowledges every generation (those beyond `parts` without running anything): none of them can
         * still be looking at this generation's task when the next one is written */
        job_ = &task;
        want_ = parts - 1;
        pending_.store((int)th_.size(), std::memory_order_relaxed);
        /* W gen_ then R sleepers_ here, W sleepers_ then R gen_ in the helper: sequentially consistent on both sides,
         * so that at least one of them sees the other's write and no wake-up is lost on a weaker memory model */
        gen_.fetch_add(1, std::memory_order_seq_cst);
        if (sleepers_.load(std::memory_order_seq_cst) > 0) {
            { std::lock_guard<std::mutex> lk(m_); }
            cv_.notify_all();
        }
        return true;
    }
    void finish()                                                        /* after the caller's own part */
    {
        while (pending_.load(std::memory_order_acquire) > 0) {
            LFQ_CPU_PAUSE();
        }
        job_ = nullptr;
        call_m_.unlock();
    }

private:
    LfqLoopPool() : pid_(getpid())
    {
        spin_us_ = lfq_knobs().host_spin_us;
        /* the helpers spin between loops: a node's ranks share its cores (LOCAL_WORLD_SIZE processes) */
        const unsigned hw = std::max(1u, lfq_cpu_budget() / (unsigned)std::max(lfq_knobs().local_world_size, 1));
        const int n = spin_us_ < 0 ? 0 : (int)std::min(7u, hw - 1u);
        for (int i = 0; i < n; i++) {
            th_.emplace_back([this, i] { loop(i); });
        }
    }
    ~LfqLoopPool()
    {
        if (getpid() != pid_) {                 /* a forked child: the threads stayed with the parent */
            for (auto &t : th_) {
                t.detach();
            }
            return;
        }
        stop_.store(true);
        gen_.fetch_add(1, std::memory_order_seq_cst);
        { std::lock_guard<std::mutex> lk(m_); }
        cv_.notify_all();
        for (auto &t : th_) {
            t.join();
        }
    }
    void loop(int idx)
    {
        uint64_t seen = 0;
        for (;;) {
            const double t0 = lfq_now_ms();
            int polls = 0;
            while (gen_.load(std::memory_order_acquire) == seen) {
                LFQ_CPU_PAUSE();
                if ((++polls & 255) == 0 && (lfq_now_ms() - t0) * 1e3 > (double)spin_us_) {
                    std::unique_lock<std::mutex> lk(m_);
                    sleepers_.fetch_add(1, std::memory_order_seq_cst);
                    cv_.wait(lk, [&] { return gen_.load(std::memory_order_seq_cst) != seen; });
                    sleepers_.fetch_sub(1, std::memory_order_seq_cst);
                }
            }
            if (stop_.load()) {
                return;
            }
            seen = gen_.load(std::memory_order_acquire);
            if (idx < want_ && job_) {
                (*job_)(idx + 1);
            }
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> th_;
    std::mutex call_m_, m_;
    std::condition_variable cv_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<int> pending_{0}, sleepers_{0};
    std::atomic<bool> stop_{false};
    const std::function<void(int)> *job_ = nullptr;
    int want_ = 0;
    long spin_us_ = 2000;
    pid_t pid_;
};

/* per-read host loops of the read-set steps (geometry from the CIGARs, event candidates): independent reads, split
 * over a few threads when there are enough of them.  f(begin, end, part) */
template <typename F>
static void lfq_for_reads(int64_t n, F f, int *parts_out = nullptr)
{
    int parts = 1;
    const int64_t par_min = lfq_knobs().host_par_min;        /* LFQ_HOST_PAR_MIN (200000): below it one thread does it */
    if (n >= par_min) {
        unsigned hw = lfq_cpu_budget();
        hw = std::max(1u, hw / (unsigned)lfq_knobs().local_world_size);
        parts = (int)std::min<int64_t>(std::min<unsigned>(hw ? hw : 1u, 8u), n / std::max<int64_t>(par_min / 2, 1));
        parts = std::max(parts, 1);
    }
    if (parts_out) {
        *parts_out = parts;
    }
    if (parts == 1) {
        f((int64_t)0, n, 0);
        return;
    }
    const std::function<void(int)> task = [&](int p) { f(n * p / parts, n * (p + 1) / parts, p); };
    LfqLoopPool &pool = LfqLoopPool::instance();
    if (pool.try_run(parts, task)) {
        f((int64_t)0, n / parts, 0);
        pool.finish();
        return;
    }
    std::vector<std::thread> th;
    for (int p = 1; p < parts; p++) {
        th.emplace_back([&, p] { f(n * p / parts, n * (p + 1) / parts, p); });
    }
    f((int64_t)0, n / parts, 0);
    for (auto &t : th) {
        t.join();
    }
}

struct LfqIndelColsOwned {
    lfq_indel_columns cols;
    std::vector<uint8_t> ref_base, cons_indel;
    std::vector<int32_t> cov, tails, non_indels, n_ins, n_dels, hrun;
    struct Side {
        std::vector<int32_t> non_fw, non_rv, ev_fw, ev_rv;
        std::vector<int64_t> ne_off, ev_off, key_off, rd_off;
        std::vector<int16_t> ne_q, ne_mq, rd_q, rd_aq, rd_mq, rd_sq;
        std::vector<char> key_chars;
    } side[2];
    void reset()
    {
        memset(&cols, 0, sizeof(cols));
        ref_base.clear(); cons_indel.clear();
        cov.clear(); tails.clear(); non_indels.clear(); n_ins.clear(); n_dels.clear(); hrun.clear();
        for (Side &s : side) {
            s.non_fw.clear(); s.non_rv.clear(); s.ev_fw.clear(); s.ev_rv.clear();
            s.ne_off.clear(); s.ev_off.clear(); s.key_off.clear(); s.rd_off.clear();
            s.ne_q.clear(); s.ne_mq.clear(); s.rd_q.clear(); s.rd_aq.clear(); s.rd_mq.clear(); s.rd_sq.clear();
            s.key_chars.clear();
        }
    }
};

#define LFQ_PIN_SLOTS 40

struct lfq_ctx {
    int device;
    hipStream_t stream;
    LfqLuts *d_luts;
    /* per-batch workspace, grown on demand */
    int64_t ws_cols;
    uint8_t *d_flags;
    uint8_t *d_approx_mu;      /* -t: one double per column of a segment (lfq_launch_approx_gate), grow-only */
    int64_t approx_mu_bytes;
    int32_t *d_prefix, *d_counters;
    LfqEntry *d_entries;
    hipStream_t dps;           /* scan + light DP of a segment, beside the next segment's count kernel */
    hipStream_t side[2];       /* big / mid DP kernels run beside the light one */
    hipEvent_t ev_mid;
    hipEvent_t ev_cnt[LFQ_MAX_SEGMENTS][2];    /* count kernel of segment s: start, stop (main stream) */
    hipEvent_t ev_scan[LFQ_MAX_SEGMENTS];      /* work lists of segment s ready (dps) */
    hipEvent_t ev_light[LFQ_MAX_SEGMENTS][2];  /* light kernel (dps) */
    hipEvent_t ev_side[2][LFQ_MAX_SEGMENTS][2];/* big / mid kernels (side streams) */
    hipEvent_t ev_join[3];
    int cur_segments;
    uint64_t *d_tiles;
    double *d_scratch;
    int64_t scratch_doubles;
    LfqLong *d_longs;             /* row-split columns (lfq_internal.h) */
    LfqSegCell *d_pool;
    int32_t long_cap, pool_cells;
    hipEvent_t ev_segw, ev_prep;
    int32_t *d_unsplit;
    uint8_t *d_retry;          /* light columns the quad kernel hands to the one-column-per-wave kernel */
    int32_t *h_counters;   /* pinned */
    /* layer-2 owned outputs / staging */
    lfq_col_counts *d_counts;
    int64_t counts_cap;
    lfq_col_pvals *d_pvals;
    int64_t pvals_cap;
    uint8_t *d_stage;
    int64_t stage_bytes;
    /* state of the batch in flight */
    hipStream_t cur_stream;
    int64_t cur_pvals_cap;
    int64_t cur_ncols;
    hipEvent_t ev[4];
    lfq_kernel_times times;
    lfq_dp_work work;
    int64_t cur_count_read, cur_count_written;   /* layout bytes of this batch's count kernel (lfq_dp_work) */
    const uint64_t *cur_col_off;                 /* device: CSR offsets of the batch in flight */
    int cur_obs_bytes_x2, cur_col_bytes;
    int n_cu;
    /* strand-bias precompute (lfq_internal.h): DP4 tuples land in host-mapped memory right after the scan;
     * a leader thread waits for that and runs the Fisher tests on the host pool while the DP kernels run */
    int32_t *h_tuples, *d_tuples_mapped;      /* [3 * heavy_cap][4] */
    int32_t *h_nheavy, *d_nheavy_mapped;
    int heavy_cap;
    hipEvent_t ev_heavy;
    uint8_t *d_plp_in, *d_plp_out;   /* device-side pileup: inputs + counters, and the tracks handed out (grow-only:
                                      * hipMalloc / hipFree of gigabytes per region cost milliseconds each) */
    int64_t plp_in_bytes, plp_out_bytes;
    uint8_t *d_tmp[5];               /* grow-only temporaries: BAQ geometry, indel counters, gathers, and the event-read
                                      * arrays + pseudo-column tracks of lfq_call_indels_batch */
    /* the allocations of the read set destroyed last (reads, tags, read ends, tag flags, pinned flags): the next
     * lfq_readset_create / _baq takes them over when they are large enough -- a worker goes from region to region, and
     * hipMalloc + hipFree of 2 GB per region are milliseconds and a device synchronisation each */
    struct { void *p; size_t cap; } rs_cache[5];
    hipStream_t up_stream;           /* lfq_readset_create's uploads and the staging copies of host tracks (created on first use) */
    hipEvent_t ev_up;                /* end of the staging copies of a batch of host tracks */
    uint8_t *h_pin;                  /* pinned host staging of the BAQ geometry + launch order (grow-only) */
    int64_t pin_bytes;
    uint8_t *h_pin2;                 /* pinned landing area of the indel pileup's per-position counters (grow-only) */
    int64_t pin2_bytes;
    int64_t tmp_bytes[5];
    int64_t plp_ne_cap;              /* capacity of d_plp_ne in int16 elements */
    LfqIndelColsOwned *plp_indel;
    int indel_host_arrays;           /* lfq_set_indel_arrays_on_host */
    int16_t *d_plp_ne;               /* quality arrays of the columns above, resident: [q0 | mq0 | q1 | mq1] */
    int64_t plp_ne_total[2];
    int dense_strand;                /* lfq_set_dense_strand_counts: layer 1 / async layer 2 fill the strand fields of every dense entry */
    int lazy_forced;                 /* set by lfq_call_snvs_batch around its submit */
    int lazy_now;                    /* this batch: strand counts only for the columns of the sparse output */
    int64_t sub_ncols;               /* batch submitted with lfq_call_snvs_submit and not collected yet: its columns, else -1 */
    int batch_recorded;              /* ev[3] has been recorded: a batch of this context may still be running */
    float baq_par_d, baq_par_e;      /* lfq_set_baq_hmm_params; kpa_ext_par_lofreq_illumina (kprobaln_ext.c:50) by default */
    int plp_nt_bytes;                /* lfq_set_pileup_nt_packed(ctx, 0): the device pileup hands out one nt byte per observation */
    const uint8_t *sub_ref_host;
    double sub_t0, sub_t1;
    const float *detlim_af;          /* device: per-column allele frequency while lfq_uniq_detlim_batch runs, else null */
    float *d_detlim;
    int64_t detlim_cap;
    int32_t *d_plp_nb;               /* num_bases of the tracks last handed out, and their column count */
    int64_t plp_ncols;
    /* BAQ scratch (lfq_baq_batch), kept between calls */
    double *d_baq_scr;
    int32_t *d_baq_expect;
    uint8_t *d_baq_tmp8;
    int32_t *d_baq_itab;
    double *d_baq_terms;
    int64_t baq_scr_bytes, baq_expect_bytes, baq_tmp8_bytes, baq_itab_bytes, baq_terms_bytes;
    std::thread *leader;
    std::mutex *lm;
    std::condition_variable *lcv;
    int leader_go, leader_stop;
    int own_streams;                 /* holds a reference on the device's shared streams */
    int kreg_hint, kreg_hint_indel;  /* screen-kernel variant for the next SNV / indel batch: from the last batch's K histogram */
    int cur_indel_mode;
    int sb_pending;                  /* strand-bias precomputes of this context not finished yet (under lm) */
    struct { void *p; size_t cap; int used; } pin_pool[LFQ_PIN_SLOTS];   /* LfqPin: pinned host temporaries */
};

namespace {

/* The four streams of a device (= the four hardware queues HIP multiplexes a process's streams onto) are shared by
 * every context on that device.  Two contexts with four streams each would have their streams doubled up on the
 * same queues in an order nobody chose -- measured: the light chain of one batch started only after the big chain of
 * the same batch had finished -- while sharing them keeps the stream plan of lfq_snv_batch_device intact and simply
 * queues a second context's batch behind the first one's, which is what a caller that pipelines batches
 * (lfq_call_snvs_wait) wants anyway.  Completion is tracked per context with events, never by draining a stream. */
struct LfqDeviceStreams {
    hipStream_t stream = nullptr, dps = nullptr, side[2] = {nullptr, nullptr};
    int refs = 0;
};
std::mutex g_streams_m;
LfqDeviceStreams g_streams[64];

bool acquire_streams(int device, lfq_ctx *c);
void release_streams(int device);

template <typename T>
int grow(T **ptr, int64_t *cap, int64_t need)
{
    if (need <= *cap && *ptr) {
        return LFQ_OK;
    }
    if (*ptr) {
        (void)hipFree(*ptr);
        *ptr = nullptr;
    }
    int64_t n = std::max<int64_t>(need, 16);
    if (hipMalloc((void **)ptr, (size_t)n * sizeof(T)) != hipSuccess) {
        *cap = 0;
        return LFQ_ERR_NOMEM;
    }
    *cap = n;
    return LFQ_OK;
}

/* Host temporaries a DMA reads or writes come from a grow-only pool of pinned blocks owned by the context, never from
 * a std::vector.  Functionally pageable memory would do -- but the runtime registers a pageable range with the driver
 * for the copy, and when a multi-megabyte vector later goes back to the OS (free -> munmap) the driver's MMU notifier
 * evicts the process's hardware queues: the device sat idle for 20-30 ms before the next launch (measured on the
 * read-set chain: the SNV call after lfq_readset_pileup_snv, 3 processes in 4 -- whenever glibc served the vectors
 * from mmap; never with MALLOC_MMAP_THRESHOLD_ raised).  Blocks are handed back on scope exit and reused. */
void *pin_acquire(lfq_ctx *c, size_t bytes, int *slot)
{
    int best = -1, empty = -1, smallest = -1;
    for (int i = 0; i < LFQ_PIN_SLOTS; i++) {
        auto &b = c->pin_pool[i];
        if (b.used) {
            continue;
        }
        if (!b.p) {
            empty = (empty < 0) ? i : empty;
        } else if (b.cap >= bytes) {
            best = (best < 0 || b.cap < c->pin_pool[best].cap) ? i : best;
        } else {
            smallest = (smallest < 0 || b.cap < c->pin_pool[smallest].cap) ? i : smallest;
        }
    }
    if (best < 0) {
        best = (empty >= 0) ? empty : smallest;     /* no free block fits: a new one, in place of the smallest if full */
        if (best < 0) {
            return nullptr;
        }
        auto &b = c->pin_pool[best];
        if (b.p) {
            (void)hipHostFree(b.p);
            b.p = nullptr;
            b.cap = 0;
        }
        const size_t cap = std::max<size_t>(bytes + bytes / 4, (size_t)1 << 16);
        if (hipHostMalloc(&b.p, cap, hipHostMallocDefault) != hipSuccess) {
            b.p = nullptr;
            return nullptr;
        }
        b.cap = cap;
    }
    c->pin_pool[best].used = 1;
    *slot = best;
    return c->pin_pool[best].p;
}

template <typename T>
struct LfqPin {
    lfq_ctx *c;
    T *p = nullptr;
    size_t n = 0;
    int slot = -1;
    LfqPin(lfq_ctx *ctx, size_t count) : c(ctx), n(count)
    {
        p = (T *)pin_acquire(c, std::max<size_t>(count, 1) * sizeof(T), &slot);
    }
    LfqPin(lfq_ctx *ctx, size_t count, T v) : LfqPin(ctx, count)
    {
        if (p) {
            std::fill(p, p + n, v);
        }
    }
    LfqPin(const LfqPin &) = delete;
    LfqPin &operator=(const LfqPin &) = delete;
    ~LfqPin()
    {
        if (slot >= 0) {
            c->pin_pool[slot].used = 0;
        }
    }
    bool ok() const { return p != nullptr; }
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    T &back() { return p[n - 1]; }
};
#define LFQ_PIN_OK(v)                                                                                                  \
    do {                                                                                                               \
        if (!(v).ok()) {                                                                                               \
            return LFQ_ERR_NOMEM;                                                                                      \
        }                                                                                                              \
    } while (0)

/* PROB_TO_PHREDQUAL_SAFE (utils.h:46) */
int phred_safe(double p) { return (p <= 0.0) ? INT32_MAX : (int)(-10.0 * log10l(p)); }

/* Largest double x with PROB_TO_PHREDQUAL_SAFE(x) >= m: the device-side form of the integer
 * filter `merged_qual < min_jq` (snpcaller.c:466-481) is then `prob > x`.  Found by bisection on
 * the bit pattern with the reference's own expression, so the integer decision is identical. */
double jq_threshold(int m)
{
    if (m <= 0) {
        return INFINITY;
    }
    uint64_t lo = 1, hi;                /* lo: smallest positive double, phred huge */
    double one = 2.0;
    memcpy(&hi, &one, 8);               /* phred_safe(2.0) < 1 <= m */
    if (phred_safe(4.9406564584124654e-324) < m) {
        return 0.0;                     /* nothing positive passes */
    }
    while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        double x;
        memcpy(&x, &mid, 8);
        if (phred_safe(x) >= m) {
            lo = mid;
        } else {
            hi = mid;
        }
    }
    double x;
    memcpy(&x, &lo, 8);
    return x;
}

int make_params(const lfq_conf *conf, const lfq_tracks *tr, LfqParams *P, bool indel_mode)
{
    if (conf->def_alt_jq == -1) {
        return LFQ_ERR_UNSUPPORTED;     /* reference: LOG_FATAL + exit (snpcaller.c:482-484) */
    }
    memset(P, 0, sizeof(*P));
    P->min_bq4 = std::min(std::max(conf->min_bq, 0), 128);
    P->min_alt_bq4 = std::min(std::max(std::max(conf->min_bq, conf->min_alt_bq), 0), 128);
    P->def_alt_bq = conf->def_alt_bq;
    P->jq_reject_above = jq_threshold(conf->min_jq);
    P->alt_jq_reject_above = jq_threshold(conf->min_alt_jq);
    P->def_alt_jp = (conf->def_alt_jq != 0) ? pow(10.0, -1.0 * conf->def_alt_jq / 10.0) : -1.0;
    P->general = (conf->min_jq > 0) || (conf->min_alt_jq > 0) || (conf->def_alt_bq == -1);
    P->min_cov = conf->min_cov;
    P->use_baq = (conf->flag & LFQ_USE_BAQ) && tr->baq;
    P->use_mq = (conf->flag & LFQ_USE_MQ) != 0;
    P->use_sq = (conf->flag & LFQ_USE_SQ) && tr->sq;
    P->bonf_dynamic = conf->bonf_dynamic;
    P->bonf_base = conf->bonf_subst;
    P->sig = (double)conf->sig;
    P->prune_slack = 1e-6;
    P->bonf_step = 3;
    P->bonf_reset_first = 1;
    P->phase1_chunks = lfq_knobs().phase1_chunks;
    P->seg_budget_mid = lfq_knobs().seg_budget_mid;
    P->seg_budget_big = lfq_knobs().seg_budget_big;
    P->seg_max = lfq_knobs().seg_max;                            /* experiments: fewer, longer row segments */
    if (indel_mode) {
        /* call_indels: no base / merged-quality filters, every event is a test (lofreq_call.c:684-725);
         * the alignment-quality track is "used" wherever the packer filled it in */
        P->min_bq4 = P->min_alt_bq4 = 0;
        P->def_alt_bq = 0;
        P->jq_reject_above = P->alt_jq_reject_above = INFINITY;
        P->def_alt_jp = -1.0;
        P->general = 0;
        P->min_cov = 0;
        P->use_baq = tr->baq != nullptr;
        P->bonf_base = conf->bonf_indel;
        P->bonf_step = 1;
        P->bonf_reset_first = 0;
    }
    return LFQ_OK;
}

bool acquire_streams(int device, lfq_ctx *c)
{
    std::lock_guard<std::mutex> lk(g_streams_m);
    if (device < 0 || device >= 64) {
        return false;
    }
    LfqDeviceStreams &d = g_streams[device];
    if (d.refs == 0) {
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        bool ok = true;
        const int split = lfq_knobs().cu_split;
        hipDeviceProp_t prop;
        int n_cu = 256;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
            n_cu = prop.multiProcessorCount;
        }
        if (split > 0 && split < n_cu) {
            /* Spatial partition (LFQ_CU_SPLIT): the three DP streams own `split` CUs, the main stream the rest, so that
             * with two batches in flight the HBM-bound count kernel of batch k + 1 and the latency / issue-bound DP
             * chains of batch k do not take wave slots and issue cycles from each other.  Bit i of a queue's CU mask is
             * CU i / n_xcc of XCC i mod n_xcc on this part (the mask is dealt round-robin over the XCDs), so a prefix of
             * the mask is the same number of CUs on every XCD. */
            const int words = (n_cu + 31) / 32;
            std::vector<uint32_t> m_dp((size_t)words, 0u), m_main((size_t)words, 0u);
            for (int i = 0; i < n_cu; i++) {
                (i < split ? m_dp : m_main)[(size_t)(i >> 5)] |= 1u << (i & 31);
            }
            ok = hipExtStreamCreateWithCUMask(&d.stream, (uint32_t)words, m_main.data()) == hipSuccess;
            ok = ok && hipExtStreamCreateWithCUMask(&d.dps, (uint32_t)words, m_dp.data()) == hipSuccess;
            for (int i = 0; ok && i < 2; i++) {
                ok = hipExtStreamCreateWithCUMask(&d.side[i], (uint32_t)words, m_dp.data()) == hipSuccess;
            }
        } else {
            ok = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) == hipSuccess;
            ok = ok && hipStreamCreateWithPriority(&d.dps, hipStreamNonBlocking, prio_hi) == hipSuccess;
            for (int i = 0; ok && i < 2; i++) {
                ok = hipStreamCreateWithPriority(&d.side[i], hipStreamNonBlocking, prio_hi) == hipSuccess;
            }
        }
        if (!ok) {
            return false;
        }
    }
    d.refs++;
    c->stream = d.stream;
    c->dps = d.dps;
    c->side[0] = d.side[0];
    c->side[1] = d.side[1];
    return true;
}

void release_streams(int device)
{
    std::lock_guard<std::mutex> lk(g_streams_m);
    LfqDeviceStreams &d = g_streams[device];
    if (d.refs > 0 && --d.refs == 0) {
        if (d.stream) (void)hipStreamDestroy(d.stream);
        if (d.dps) (void)hipStreamDestroy(d.dps);
        for (int i = 0; i < 2; i++) {
            if (d.side[i]) (void)hipStreamDestroy(d.side[i]);
        }
        d = LfqDeviceStreams();
    }
}

void fill_luts(LfqLuts *L)
{
    for (int q = 0; q < 256; q++) {
        const double p = pow(10.0, -1.0 * q / 10.0);    /* PHREDQUAL_TO_PROB, utils.h:42 */
        L->bq[q] = p;
        L->baq[q] = p;
        L->mq[q] = p;
        L->sq[q] = p;
    }
    L->baq[255] = 0.0;      /* -1: missing (snpcaller.c:321-322) */
    L->sq[255] = 0.0;       /* snpcaller.c:307-308 */
    L->sq[254] = 0.0;       /* source_qual's PROB_TO_PHREDQUAL(LDBL_MIN) = 49314 (plp.c:521): pow(10, -4931.4) == 0.0 */
    L->mq[255] = 0.0;       /* MQ 255 = NA -> -1 (snpcaller.c:451-453, 313-314) */
    L->mq[0] = 0.5;         /* MQ0_ERRPROB (snpcaller.c:64, 315-316) */
}

void leader_main(lfq_ctx *c)
{
    (void)hipSetDevice(c->device);
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(*c->lm);
            c->lcv->wait(lk, [&] { return c->leader_go > 0 || c->leader_stop; });
            if (c->leader_stop) {
                return;
            }
            c->leader_go--;
        }
        int64_t n = 0;
        if (hipEventSynchronize(c->ev_heavy) == hipSuccess) {
            n = std::min<int64_t>(std::max<int64_t>(*(volatile int32_t *)c->h_nheavy, 0), c->heavy_cap);
        }
        lfq_sb_precompute(c->h_tuples, 3 * n);
        {
            std::lock_guard<std::mutex> lk(*c->lm);
            c->sb_pending--;
        }
        c->lcv->notify_all();
    }
}

int ensure_workspace(lfq_ctx *c, int64_t ncols)
{
    if (ncols > c->ws_cols) {
        int64_t cap = 0;
        int64_t want = ncols + ncols / 8 + 1024;
        if (c->d_flags) (void)hipFree(c->d_flags);
        if (c->d_prefix) (void)hipFree(c->d_prefix);
        if (c->d_entries) (void)hipFree(c->d_entries);
        if (c->d_tiles) (void)hipFree(c->d_tiles);
        if (c->d_unsplit) (void)hipFree(c->d_unsplit);
        if (c->d_retry) (void)hipFree(c->d_retry);
        c->d_unsplit = nullptr;
        c->d_retry = nullptr;
        c->d_flags = nullptr;
        c->d_prefix = nullptr;
        c->d_entries = nullptr;
        c->d_tiles = nullptr;
        c->ws_cols = 0;
        LFQ_TRY(grow(&c->d_flags, &cap, want));
        cap = 0;
        LFQ_TRY(grow(&c->d_prefix, &cap, want));
        cap = 0;
        LFQ_TRY(grow(&c->d_entries, &cap, want));
        cap = 0;
        LFQ_TRY(grow(&c->d_tiles, &cap, 2 * (want / 4096 + 8 * (LFQ_MAX_SEGMENTS + 1))));
        cap = 0;
        LFQ_TRY(grow(&c->d_unsplit, &cap, want));
        cap = 0;
        LFQ_TRY(grow(&c->d_retry, &cap, want));
        c->ws_cols = want;
    }
    return LFQ_OK;
}

}  // namespace

extern "C" {

/* pinned host memory for a producer's track buffers: the copies of lfq_call_snvs_submit(tracks_on_device = 0) are then
 * DMA transfers that return at once and run at the link's rate (pageable memory is staged by the runtime) */
void *lfq_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void lfq_host_free(void *p)
{
    if (p) {
        (void)hipHostFree(p);
    }
}

int lfq_device_count(void)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 0) {
        return 0;
    }
    return ndev;
}

/* Which GPU a worker process takes (include/lofreq_amd.h).  The slot files are held (flock) for the life of the process:
 * the descriptor is deliberately never closed. */
int lfq_pick_device(int n_devices, int *slot_out)
{
    if (slot_out) {
        *slot_out = -1;
    }
    if (n_devices <= 0) {
        n_devices = lfq_device_count();
    }
    if (n_devices <= 0) {
        return LFQ_ERR_NO_DEVICE;
    }
    auto env_int = [](const char *name, long *v) {
        const char *e = getenv(name);
        if (!e || !*e) {
            return false;
        }
        char *end = nullptr;
        const long x = strtol(e, &end, 10);
        if (end == e || *end != 0 || x < 0) {
            return false;
        }
        *v = x;
        return true;
    };
    long v = 0;
    if (env_int("LFQ_DEVICE", &v)) {
        return v < n_devices ? (int)v : LFQ_ERR_INVALID;        /* an explicit ordinal is taken literally */
    }
    if (env_int("LOCAL_RANK", &v)) {
        return (int)(v % n_devices);                            /* torchrun / mpirun style launchers */
    }
    static int held_slot = -1;                                  /* this process already holds a slot */
    if (held_slot >= 0) {
        if (slot_out) {
            *slot_out = held_slot;
        }
        return held_slot % n_devices;
    }
    const char *dir = getenv("LFQ_SLOT_DIR");
    if (!dir || !*dir) {
        dir = "/tmp";
    }
    for (int k = 0; k < 64 * n_devices; k++) {
        char path[512];
        snprintf(path, sizeof(path), "%s/lofreq_amd.%ld.slot%d", dir, (long)getuid(), k);
        const int fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC, 0600);
        if (fd < 0) {
            break;                                              /* no usable directory: fall through to the pid rule */
        }
        if (flock(fd, LOCK_EX | LOCK_NB) == 0) {
            held_slot = k;
            if (slot_out) {
                *slot_out = k;
            }
            return k % n_devices;
        }
        close(fd);
    }
    return (int)((long)getpid() % n_devices);
}

int lfq_create(lfq_ctx **out, int device_ordinal)
{
    if (!out) {
        return LFQ_ERR_INVALID;
    }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_ordinal < 0 || device_ordinal >= ndev) {
        return LFQ_ERR_NO_DEVICE;
    }
    if (hipSetDevice(device_ordinal) != hipSuccess) {
        return LFQ_ERR_NO_DEVICE;
    }
    lfq_ctx *c = (lfq_ctx *)calloc(1, sizeof(lfq_ctx));
    if (!c) {
        return LFQ_ERR_NOMEM;
    }
    c->device = device_ordinal;
    c->sub_ncols = -1;
    c->dense_strand = 1;
    c->indel_host_arrays = 1;
    c->baq_par_d = 0.00001f;            /* kpa_ext_par_lofreq_illumina (kprobaln_ext.c:50) */
    c->baq_par_e = 0.4f;
    hipDeviceProp_t prop;
    c->n_cu = 256;
    if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0) {
        c->n_cu = prop.multiProcessorCount;
    }
    bool ok = acquire_streams(device_ordinal, c);
    c->own_streams = ok ? 1 : 0;
    ok = ok && hipMalloc((void **)&c->d_luts, sizeof(LfqLuts)) == hipSuccess;
    /* counter blocks: one per segment + one batch-wide */
    ok = ok && hipMalloc((void **)&c->d_counters, (LFQ_MAX_SEGMENTS + 1) * LFQ_NCOUNTERS * sizeof(int32_t)) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&c->h_counters, (LFQ_MAX_SEGMENTS + 1) * LFQ_NCOUNTERS * sizeof(int32_t) + 16,
                             hipHostMallocDefault) == hipSuccess;      /* + first / last CSR offset (lfq_batch_finish) */
    for (int i = 0; ok && i < 4; i++) {
        ok = hipEventCreate(&c->ev[i]) == hipSuccess;
    }
    for (int i = 0; ok && i < 3; i++) {
        ok = hipEventCreate(&c->ev_join[i]) == hipSuccess;
    }
    for (int s = 0; ok && s < LFQ_MAX_SEGMENTS; s++) {
        ok = hipEventCreate(&c->ev_cnt[s][0]) == hipSuccess && hipEventCreate(&c->ev_cnt[s][1]) == hipSuccess;
        ok = ok && hipEventCreate(&c->ev_scan[s]) == hipSuccess;
        ok = ok && hipEventCreate(&c->ev_light[s][0]) == hipSuccess && hipEventCreate(&c->ev_light[s][1]) == hipSuccess;
        for (int i = 0; ok && i < 2; i++) {
            ok = hipEventCreate(&c->ev_side[i][s][0]) == hipSuccess && hipEventCreate(&c->ev_side[i][s][1]) == hipSuccess;
        }
    }
    if (ok && !lfq_knobs().no_sb_precompute) {
        c->heavy_cap = 1 << 16;
        ok = hipHostMalloc((void **)&c->h_tuples, (size_t)c->heavy_cap * 3 * 4 * sizeof(int32_t), hipHostMallocMapped) == hipSuccess
             && hipHostMalloc((void **)&c->h_nheavy, 64, hipHostMallocMapped) == hipSuccess
             && hipHostGetDevicePointer((void **)&c->d_tuples_mapped, c->h_tuples, 0) == hipSuccess
             && hipHostGetDevicePointer((void **)&c->d_nheavy_mapped, c->h_nheavy, 0) == hipSuccess
             && hipEventCreateWithFlags(&c->ev_heavy, hipEventDisableTiming) == hipSuccess;
        if (ok) {
            *c->h_nheavy = 0;
            c->lm = new std::mutex();
            c->lcv = new std::condition_variable();
            c->leader = new std::thread(leader_main, c);
        }
    }
    if (ok) {
        LfqLuts h;
        fill_luts(&h);
        ok = hipMemcpy(c->d_luts, &h, sizeof(h), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) {
        lfq_destroy(c);
        return LFQ_ERR_HIP;
    }
    *out = c;
    return LFQ_OK;
}

void lfq_destroy(lfq_ctx *c)
{
    if (c && c->leader) {
        {
            std::lock_guard<std::mutex> lk(*c->lm);
            c->leader_stop = 1;
        }
        c->lcv->notify_all();
        c->leader->join();
        delete c->leader;
        delete c->lm;
        delete c->lcv;
        c->leader = nullptr;
    }
    if (c) {
        if (c->d_plp_in) (void)hipFree(c->d_plp_in);
        if (c->d_plp_out) (void)hipFree(c->d_plp_out);
        delete c->plp_indel;
        if (c->d_plp_ne) (void)hipFree(c->d_plp_ne);
        for (int i = 0; i < 5; i++) {
            if (c->d_tmp[i]) (void)hipFree(c->d_tmp[i]);
        }
        for (int k = 0; k < 5; k++) {                   /* 4 = LFQ_RSC_PINFL: pinned host memory */
            if (c->rs_cache[k].p) (void)(k == 4 ? hipHostFree(c->rs_cache[k].p) : hipFree(c->rs_cache[k].p));
        }
        if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
        if (c->ev_up) (void)hipEventDestroy(c->ev_up);
        if (c->h_pin) (void)hipHostFree(c->h_pin);
        for (int i = 0; i < LFQ_PIN_SLOTS; i++) {
            if (c->pin_pool[i].p) (void)hipHostFree(c->pin_pool[i].p);
        }
        if (c->h_pin2) (void)hipHostFree(c->h_pin2);
        if (c->d_detlim) (void)hipFree(c->d_detlim);
        if (c->d_baq_scr) (void)hipFree(c->d_baq_scr);
        if (c->d_baq_expect) (void)hipFree(c->d_baq_expect);
        if (c->d_baq_tmp8) (void)hipFree(c->d_baq_tmp8);
        if (c->d_baq_itab) (void)hipFree(c->d_baq_itab);
        if (c->d_baq_terms) (void)hipFree(c->d_baq_terms);
        if (c->h_tuples) (void)hipHostFree(c->h_tuples);
        if (c->h_nheavy) (void)hipHostFree(c->h_nheavy);
        if (c->ev_heavy) (void)hipEventDestroy(c->ev_heavy);
    }
    if (!c) {
        return;
    }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void *bufs[] = {c->d_luts, c->d_flags, c->d_prefix, c->d_entries, c->d_counters, c->d_tiles, c->d_longs, c->d_pool, c->d_unsplit, c->d_retry,
                    c->d_scratch, c->d_counts, c->d_pvals, c->d_stage, c->d_approx_mu};
    for (void *b : bufs) {
        if (b) (void)hipFree(b);
    }
    if (c->h_counters) (void)hipHostFree(c->h_counters);
    for (int i = 0; i < 4; i++) {
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    }
    for (int i = 0; i < 3; i++) {
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
        if (i == 0 && c->ev_segw) (void)hipEventDestroy(c->ev_segw);
        if (i == 0 && c->ev_prep) (void)hipEventDestroy(c->ev_prep);
        if (i == 0 && c->ev_mid) (void)hipEventDestroy(c->ev_mid);
    }
    for (int s = 0; s < LFQ_MAX_SEGMENTS; s++) {
        hipEvent_t evs[] = {c->ev_cnt[s][0], c->ev_cnt[s][1], c->ev_scan[s], c->ev_light[s][0], c->ev_light[s][1],
                            c->ev_side[0][s][0], c->ev_side[0][s][1], c->ev_side[1][s][0], c->ev_side[1][s][1]};
        for (hipEvent_t e : evs) {
            if (e) (void)hipEventDestroy(e);
        }
    }
    if (c->own_streams) {
        release_streams(c->device);
    }
    free(c);
}

int lfq_synchronize(lfq_ctx *c)
{
    if (!c) {
        return LFQ_ERR_INVALID;
    }
    LFQ_TRY_HIP(hipStreamSynchronize(c->stream));
    LFQ_TRY_HIP(hipEventSynchronize(c->ev[3]));    /* a batch ends on the dps stream (batch_device_impl), not on c->stream */
    return LFQ_OK;
}

/* A batch ends on the dps stream (the join in batch_device_impl), not on the stream it was launched on.  Everything
 * the library itself queues afterwards that rewrites what the batch's DP kernels still read or write -- the next
 * batch's counter / retry memsets, the staging copies of host tracks, the pileup's output tracks and num_bases, a
 * generator fill -- waits for the batch's last event first.  One batch in flight per context (layer 1 too). */
static int order_after_batch(lfq_ctx *c, hipStream_t st)
{
    if (c->batch_recorded) {
        LFQ_TRY_HIP(hipStreamWaitEvent(st, c->ev[3], 0));
    }
    return LFQ_OK;
}

static int batch_device_impl(lfq_ctx *c, const lfq_conf *conf, const lfq_tracks *tr, lfq_col_counts *d_counts,
                             lfq_col_pvals *d_pvals, int64_t pvals_capacity, void *stream_or_null,
                             bool indel_mode)
{
    if (!c || !conf || !tr || tr->ncols < 0 || tr->ncols > 0x7ffffff0LL) {
        return LFQ_ERR_INVALID;
    }
    if (tr->ncols > 0 && (!tr->nt || !tr->bq || !tr->mq || !tr->col_off || !tr->ref_base || !d_counts
                          || (!d_pvals && pvals_capacity > 0))) {
        return LFQ_ERR_INVALID;
    }
    if ((((uintptr_t)tr->nt) | ((uintptr_t)tr->bq)) & 15u) {
        return LFQ_ERR_INVALID;         /* 16-byte alignment contract of the track base pointers */
    }
    LFQ_TRY_HIP(hipSetDevice(c->device));
    hipStream_t st = stream_or_null ? (hipStream_t)stream_or_null : c->stream;
    /* LFQ_SINGLE_STREAM: every kernel on the caller's stream, in dependency order (counter collection with
     * rocprofv3 --pmc serialises dispatches and does not get along with the cross-stream waits) */
    const LfqKnobs &kn = lfq_knobs();
    const bool single_stream = kn.single_stream != 0;
    hipStream_t dps = single_stream ? st : c->dps;
    hipStream_t side0 = single_stream ? st : c->side[0], side1 = single_stream ? st : c->side[1];
    hipStream_t side_i[2] = {side0, side1};
    LfqParams P;
    LFQ_TRY(make_params(conf, tr, &P, indel_mode));
    P.detlim_af = indel_mode ? nullptr : c->detlim_af;      /* set only inside lfq_uniq_detlim_batch */
    P.lazy_strand = (c->lazy_now && !indel_mode && !P.general && !P.detlim_af) ? 1 : 0;
    /* -t (snpcaller.c:1131); lofreq uniq hands snpcaller -1 (lofreq_uniq.c:311-312) */
    P.approx_n = (!P.detlim_af && conf->approx_threshold_n > 0) ? conf->approx_threshold_n : 0;
    LFQ_TRY(ensure_workspace(c, tr->ncols));

    LfqTracksDev T;
    T.nt = tr->nt;
    T.nt_packed = (tr->flags & LFQ_TRACKS_NT_PACKED) ? 1 : 0;
    T.pad_ = 0;
    T.bq = tr->bq;
    T.baq = tr->baq;
    T.mq = tr->mq;
    T.sq = tr->sq;
    T.col_off = tr->col_off;
    T.ref_base = tr->ref_base;
    T.coverage_plp = tr->coverage_plp;
    T.num_bases = tr->num_bases;
    T.ncols = tr->ncols;

    const int64_t ncols = tr->ncols;
    int32_t *gcounters = c->d_counters + LFQ_MAX_SEGMENTS * LFQ_NCOUNTERS;
    c->cur_stream = st;
    c->cur_pvals_cap = pvals_capacity;
    c->cur_ncols = ncols;
    c->cur_segments = 0;
    c->cur_col_off = tr->col_off;
    c->cur_indel_mode = indel_mode ? 1 : 0;
    /* layout bytes per observation / per column of the count kernel instantiation this batch runs (lfq_dp_work) */
    c->cur_obs_bytes_x2 = (T.nt_packed ? 1 : 2) + 2;                       /* nt + bq, in half bytes */
    if (P.general) {
        c->cur_obs_bytes_x2 += 2 * ((T.baq ? 1 : 0) + 1 + (T.sq ? 1 : 0)); /* + baq, mq, sq: full evaluation */
        if (P.def_alt_bq == -1) {
            c->cur_obs_bytes_x2 += (T.nt_packed ? 1 : 2) + 2;              /* median pass reads nt + bq once more */
        }
    }
    c->cur_col_bytes = 8 + 1 + (T.coverage_plp ? 4 : 0) + (T.num_bases ? 4 : 0) + (P.detlim_af ? 4 : 0);
    LFQ_TRY(order_after_batch(c, st));
    LFQ_TRY_HIP(hipMemsetAsync(c->d_counters, 0, (LFQ_MAX_SEGMENTS + 1) * LFQ_NCOUNTERS * sizeof(int32_t), st));
    if (ncols > 0) {
        LFQ_TRY_HIP(hipMemsetAsync(c->d_retry, 0, (size_t)ncols, st));
    }
    LFQ_TRY_HIP(hipEventRecord(c->ev[0], st));
    if (ncols == 0) {
        LFQ_TRY_HIP(hipMemcpyAsync(c->h_counters, c->d_counters,
                                   (LFQ_MAX_SEGMENTS + 1) * LFQ_NCOUNTERS * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        LFQ_TRY_HIP(hipEventRecord(c->ev[3], st));
        c->batch_recorded = 1;
        return LFQ_OK;
    }

    /* big-column scratch: 2 doubles per observation (pass boundary) + K+1 log-probabilities per resident
     * workgroup; needs the deepest column of the batch */
    int64_t max_depth = tr->max_col_obs;
    if (max_depth <= 0) {
        LFQ_TRY(lfq_launch_maxdepth(T, gcounters, st));
        LFQ_TRY_HIP(hipMemcpyAsync(c->h_counters, gcounters, LFQ_NCOUNTERS * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        LFQ_TRY_HIP(hipStreamSynchronize(st));
        max_depth = c->h_counters[LFQ_GC_MAXDEPTH];
    }
    const int64_t per_block = 3 * max_depth + 72;
    int n_big_blocks = c->n_cu;                         /* 8-wave workgroups: one per CU beside the light kernel */
    const int64_t budget = (int64_t)1 << 29;            /* 4 GiB of doubles */
    if (per_block * n_big_blocks > budget) {
        n_big_blocks = (int)std::max<int64_t>(8, budget / per_block);
    }
    LFQ_TRY(grow(&c->d_scratch, &c->scratch_doubles, per_block * n_big_blocks));
    if (!c->d_longs) {
        /* row-split bookkeeping: 64 Ki column records, 8 Mi segment cells (128 MiB); a column that does not
         * get its cells simply runs unsplit */
        c->long_cap = 1 << 16;
        c->pool_cells = kn.split_pool_cells;                    /* 0 disables row splitting */
        LFQ_TRY_HIP(hipMalloc((void **)&c->d_longs, (size_t)c->long_cap * sizeof(LfqLong)));
        LFQ_TRY_HIP(hipMalloc((void **)&c->d_pool, (size_t)std::max(c->pool_cells, 1) * sizeof(LfqSegCell)));
        LFQ_TRY_HIP(hipEventCreateWithFlags(&c->ev_segw, hipEventDisableTiming));
        LFQ_TRY_HIP(hipEventCreateWithFlags(&c->ev_prep, hipEventDisableTiming));
        LFQ_TRY_HIP(hipEventCreateWithFlags(&c->ev_mid, hipEventDisableTiming));
    }

    /* Segments: the count kernel is HBM-bound and leaves the VALUs mostly idle, the DP kernels are
     * latency/issue-bound and touch little memory.  Cutting the batch into segments lets the DP of
     * segment s run (on other streams) under the count kernel of segment s+1.  The running Bonferroni
     * prefix is carried from segment to segment on the device (LFQ_GC_TESTED). */
    int n_seg = 1;   /* measured on C3: with the current kernels overlapping count and DP loses (both want wave slots
                      * and VALU issue); kept switchable for experiments via LFQ_SEGMENTS */
    n_seg = kn.segments;
    c->cur_segments = n_seg;
    LFQ_TRY_HIP(hipEventRecord(c->ev_join[2], st));
    LFQ_TRY_HIP(hipStreamWaitEvent(dps, c->ev_join[2], 0));   /* dps starts after the memset */

    for (int s = 0; s < n_seg; s++) {
        const int64_t c0 = ncols * s / n_seg, c1 = ncols * (s + 1) / n_seg;
        LfqWork W;
        W.tested_prefix = c->d_prefix;
        W.entries = c->d_entries + c0;
        W.counters = c->d_counters + s * LFQ_NCOUNTERS;
        W.gcounters = gcounters;
        W.block_sums = (int32_t *)(c->d_tiles + 2 * (c0 / 4096 + 8 * s));
        W.long_cap = c->long_cap / n_seg;
        W.pool_cells = c->pool_cells / n_seg;
        W.longs = c->d_longs + (int64_t)s * W.long_cap;
        W.pool = c->d_pool + (int64_t)s * W.pool_cells;
        W.unsplit = c->d_unsplit + c0;

        LFQ_TRY_HIP(hipEventRecord(c->ev_cnt[s][0], st));
        LFQ_TRY(lfq_launch_count(T, c0, c1, P, c->d_luts, d_counts, c->d_flags, max_depth, st));
        LFQ_TRY_HIP(hipEventRecord(c->ev_cnt[s][1], st));

        LFQ_TRY_HIP(hipStreamWaitEvent(dps, c->ev_cnt[s][1], 0));
        LFQ_TRY(lfq_launch_scan(T, c0, c1, c->d_flags, d_counts, W, dps));
        if (P.approx_n > 0) {
            /* -t: the Poisson gate over the listed columns, then the list without the ones it gave up */
            LFQ_TRY(grow(&c->d_approx_mu, &c->approx_mu_bytes, (c1 - c0) * 8));
            LFQ_TRY(lfq_launch_approx_gate(T, P, c->d_luts, d_counts, W, c1 - c0, (double *)c->d_approx_mu, c->d_flags, dps));
            LFQ_TRY(lfq_launch_scan(T, c0, c1, c->d_flags, d_counts, W, dps, true));
        }
        LFQ_TRY_HIP(hipEventRecord(c->ev_scan[s], dps));
        if (n_seg == 1 && !indel_mode && c->leader && !kn.no_sb_precompute) {
            /* DP4 tuples of the mid / big class alleles with >= 16 alt bases -> host; Fisher tests start now */
            if (P.lazy_strand) {        /* the count kernel left the strands out: count them for the heavy columns here */
                LFQ_TRY(lfq_launch_strand_heavy(T, W, d_counts, c->d_tuples_mapped, c->d_nheavy_mapped, c->heavy_cap, 16, dps));
            } else {
                LFQ_TRY(lfq_launch_gather_heavy(W, d_counts, c->d_tuples_mapped, c->d_nheavy_mapped, c->heavy_cap, 16, dps));
            }
            LFQ_TRY_HIP(hipEventRecord(c->ev_heavy, dps));
            {
                std::lock_guard<std::mutex> lk(*c->lm);
                c->leader_go++;
                c->sb_pending++;
            }
            c->lcv->notify_all();
        }
        const int64_t seg_cols = c1 - c0;
        /* the light kernel is throughput work and persistent: it must leave wave slots for the short
         * latency-bound kernels of the long columns, or they only start when it ends */
        const int light_waves_per_cu = kn.light_kernel == 0 ? kn.screen_waves_per_cu : 10;
        const int n_light_waves = (int)std::min<int64_t>((int64_t)c->n_cu * light_waves_per_cu, std::max<int64_t>(seg_cols / 8, 4));
        const int n_mid_waves = (int)std::min<int64_t>((int64_t)c->n_cu * 4, std::max<int64_t>(seg_cols, 4));
        const bool run_big = !kn.skip_big, run_mid = !kn.skip_mid;     /* profiling aid: run the DP classes in isolation */
        const bool dbg_sync = kn.debug_sync != 0;                      /* debugging aid: serialize and name the stages */
#define LFQ_DBG_STAGE(name)                                                        \
    do {                                                                           \
        if (dbg_sync) {                                                            \
            fprintf(stderr, "[lfq] %s launched\n", name);                          \
            hipError_t e_ = hipDeviceSynchronize();                                \
            fprintf(stderr, "[lfq] %s done (%d)\n", name, (int)e_);                \
        }                                                                          \
    } while (0)
        /* Stream plan (4 streams = the 4 hardware queues; more streams would share queues and serialise):
         *   dps     scan -> screen (light columns) -> retry
         *   side[0] [scan] -> big prep -> row segments of the big class -> fold tree + emission -> unsplit big columns
         *   side[1] [scan] -> mid kernel (first stretch of rows) -> [prep] -> row segments of the mid class -> fold tree
         * (Round 1 ran the prep kernel in front of the light kernel on dps: beside the lane-group kernel's 10 waves per
         * CU it starved.  The screen kernel runs 8 lighter waves per CU and is no longer the longest chain.) */
        LFQ_TRY_HIP(hipStreamWaitEvent(side1, c->ev_scan[s], 0));
        LFQ_TRY_HIP(hipEventRecord(c->ev_side[1][s][0], side1));
        LFQ_TRY_HIP(hipStreamWaitEvent(side0, c->ev_scan[s], 0));
        LFQ_TRY_HIP(hipEventRecord(c->ev_side[0][s][0], side0));
        if (run_big) {
            LFQ_TRY(lfq_launch_dp_big_prep(T, P, c->d_luts, d_counts, W, d_pvals, pvals_capacity, c->n_cu * 2, side0));
            LFQ_DBG_STAGE("prep");
        }
        LFQ_TRY_HIP(hipEventRecord(c->ev_prep, side0));
        if (run_mid) {
            LFQ_TRY(lfq_launch_dp_mid(T, P, c->d_luts, d_counts, W, d_pvals, pvals_capacity, n_mid_waves, side1));
            LFQ_DBG_STAGE("mid");
        }
        if (run_big) {
            LFQ_TRY(lfq_launch_dp_seg(1, T, P, c->d_luts, W, c->n_cu * 8, side0));
            LFQ_DBG_STAGE("seg big");
            LFQ_TRY(lfq_launch_dp_combine(1, P, d_counts, W, d_pvals, pvals_capacity, c->n_cu, side0));
            LFQ_DBG_STAGE("combine big");
            LFQ_TRY(lfq_launch_dp_big(T, P, c->d_luts, d_counts, W, d_pvals, pvals_capacity, c->d_scratch, per_block,
                                      n_big_blocks, side0));
            LFQ_DBG_STAGE("big");
        }
        LFQ_TRY_HIP(hipStreamWaitEvent(side1, c->ev_prep, 0));     /* K = 250..252 of the big class lands in class 1 */
        if (run_mid || run_big) {
            LFQ_TRY(lfq_launch_dp_seg(0, T, P, c->d_luts, W, c->n_cu * 8, side1));
            LFQ_DBG_STAGE("seg mid");
            LFQ_TRY(lfq_launch_dp_combine(0, P, d_counts, W, d_pvals, pvals_capacity, c->n_cu, side1));
            LFQ_DBG_STAGE("combine mid");
        }
        LFQ_TRY_HIP(hipEventRecord(c->ev_light[s][0], dps));
        if (!kn.skip_light) {
            if (kn.light_kernel == 2) {                 /* A/B switch: the one-column-per-wavefront kernel */
                LFQ_TRY(lfq_launch_dp_light(T, P, c->d_luts, d_counts, W, d_pvals, pvals_capacity, n_light_waves, dps));
            } else {
                LFQ_TRY(lfq_launch_dp_quad(T, P, c->d_luts, d_counts, W, c->d_retry + c0, d_pvals, pvals_capacity,
                                           n_light_waves, indel_mode ? c->kreg_hint_indel : c->kreg_hint, dps));
            }
        }
        LFQ_TRY_HIP(hipEventRecord(c->ev_light[s][1], dps));
        for (int i = 0; i < 2; i++) {
            LFQ_TRY_HIP(hipEventRecord(c->ev_side[i][s][1], side_i[i]));
        }
    }
    /* Join.  A caller that passed its own stream gets everything joined back into it.  On the library's own streams
     * (layer 2) the batch ends on the dps stream instead and `st` carries nothing but the memsets and count kernels:
     * the count kernel of the NEXT batch (another context on the same device streams) then starts as soon as this
     * batch's count kernel is done and streams through HBM while this batch's DP kernels -- latency / issue-bound,
     * < 1 GB of traffic, on the high-priority streams -- run beside it. */
    hipStream_t jn = (st != c->stream || single_stream) ? st : dps;
    LFQ_TRY_HIP(hipEventRecord(c->ev[1], st));                     /* all count kernels done */
    LFQ_TRY_HIP(hipEventRecord(c->ev_join[0], side0));
    LFQ_TRY_HIP(hipEventRecord(c->ev_join[1], side1));
    LFQ_TRY_HIP(hipEventRecord(c->ev_join[2], dps));
    for (int i = 0; i < 3; i++) {
        LFQ_TRY_HIP(hipStreamWaitEvent(jn, c->ev_join[i], 0));
    }
    if (P.lazy_strand && d_pvals && pvals_capacity > 0) {
        /* DP4 of the columns that made it into the sparse output (lofreq_call.c:853-857) */
        LFQ_TRY(lfq_launch_strand_pvals(T, d_pvals, gcounters + LFQ_GC_PVALS, pvals_capacity, c->n_cu, jn));
    }
    /* the batch's counters and the two ends of its CSR offsets travel to pinned memory as part of the batch:
     * lfq_batch_finish then only waits for ev[3] (a synchronous hipMemcpy there takes the null stream, and the null
     * stream's turn can sit behind unrelated work queued on the device) */
    LFQ_TRY_HIP(hipMemcpyAsync(c->h_counters, c->d_counters, (LFQ_MAX_SEGMENTS + 1) * LFQ_NCOUNTERS * sizeof(int32_t),
                               hipMemcpyDeviceToHost, jn));
    {
        uint64_t *h_ends = reinterpret_cast<uint64_t *>(c->h_counters + (LFQ_MAX_SEGMENTS + 1) * LFQ_NCOUNTERS);
        if (c->cur_col_off) {
            LFQ_TRY_HIP(hipMemcpyAsync(h_ends, c->cur_col_off, 8, hipMemcpyDeviceToHost, jn));
            LFQ_TRY_HIP(hipMemcpyAsync(h_ends + 1, c->cur_col_off + c->cur_ncols, 8, hipMemcpyDeviceToHost, jn));
        }
    }
    LFQ_TRY_HIP(hipEventRecord(c->ev[3], jn));
    c->batch_recorded = 1;
    return LFQ_OK;
}

int lfq_snv_batch_device(lfq_ctx *c, const lfq_conf *conf, const lfq_tracks *tr, lfq_col_counts *d_counts,
                         lfq_col_pvals *d_pvals, int64_t pvals_capacity, void *stream_or_null)
{
    if (!c) {
        return LFQ_ERR_INVALID;
    }
    if (!c->lazy_forced) {
        c->lazy_now = !c->dense_strand;
    }
    return batch_device_impl(c, conf, tr, d_counts, d_pvals, pvals_capacity, stream_or_null, false);
}

int lfq_set_indel_arrays_on_host(lfq_ctx *c, int on)
{
    if (!c) {
        return LFQ_ERR_INVALID;
    }
    c->indel_host_arrays = on ? 1 : 0;
    return LFQ_OK;
}

int lfq_set_pileup_nt_packed(lfq_ctx *c, int on)
{
    if (!c) {
        return LFQ_ERR_INVALID;
    }
    c->plp_nt_bytes = on ? 0 : 1;
    return LFQ_OK;
}

int lfq_set_baq_hmm_params(lfq_ctx *c, float gap_open, float gap_ext)
{
    if (!c || !(gap_open > 0.f) || !(gap_open < 0.5f) || !(gap_ext > 0.f) || !(gap_ext < 1.f)) {
        return LFQ_ERR_INVALID;             /* (NaN included; 1 - 2 d and 1 - e are transition probabilities) */
    }
    c->baq_par_d = gap_open;
    c->baq_par_e = gap_ext;
    return LFQ_OK;
}

int lfq_pack_nt_track(const uint8_t *nt_bytes, int64_t n_obs, uint8_t *packed_out)
{
    if (n_obs < 0 || (n_obs > 0 && (!nt_bytes || !packed_out))) {
        return LFQ_ERR_INVALID;
    }
    const int64_t full = n_obs / 8;
    for (int64_t g = 0; g < full; g++) {
        for (int k = 0; k < 4; k++) {
            packed_out[4 * g + k] = (uint8_t)((nt_bytes[8 * g + k] & 15) | ((nt_bytes[8 * g + 4 + k] & 15) << 4));
        }
    }
    if (n_obs & 7) {
        uint8_t last[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        memcpy(last, nt_bytes + 8 * full, (size_t)(n_obs & 7));
        for (int k = 0; k < 4; k++) {
            packed_out[4 * full + k] = (uint8_t)((last[k] & 15) | ((last[4 + k] & 15) << 4));
        }
    }
    return LFQ_OK;
}

int lfq_set_dense_strand_counts(lfq_ctx *c, int on)
{
    if (!c) {
        return LFQ_ERR_INVALID;
    }
    c->dense_strand = on ? 1 : 0;
    return LFQ_OK;
}

int lfq_indel_batch_device(lfq_ctx *c, const lfq_conf *conf, const lfq_tracks *tr, lfq_col_counts *d_counts,
                           lfq_col_pvals *d_pvals, int64_t pvals_capacity, void *stream_or_null)
{
    return batch_device_impl(c, conf, tr, d_counts, d_pvals, pvals_capacity, stream_or_null, true);
}

int lfq_batch_finish(lfq_ctx *c, lfq_batch_stats *stats)
{
    if (!c) {
        return LFQ_ERR_INVALID;
    }
    /* wait for THIS batch (its last event), not for the stream: the streams are shared with the other contexts of
     * the device, and a batch of one of them may already be queued behind this one */
    LFQ_TRY_HIP(hipEventSynchronize(c->ev[3]));
    /* counters + first and last CSR offset (the batch's observation count) were copied by the batch itself */
    const uint64_t *h_ends = reinterpret_cast<const uint64_t *>(c->h_counters + (LFQ_MAX_SEGMENTS + 1) * LFQ_NCOUNTERS);
    const int64_t batch_obs = (c->cur_col_off && c->cur_ncols > 0) ? (int64_t)(h_ends[1] - h_ends[0]) : 0;
    c->cur_count_read = batch_obs * c->cur_obs_bytes_x2 / 2 + c->cur_ncols * c->cur_col_bytes;
    c->cur_count_written = c->cur_ncols * (int64_t)(sizeof(lfq_col_counts) + 1);
    const int32_t *g = c->h_counters + LFQ_MAX_SEGMENTS * LFQ_NCOUNTERS;
    memset(&c->times, 0, sizeof(c->times));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev[0], c->ev[3]) == hipSuccess) c->times.ms_total = ms;
    float dp_end = 0.f;
    for (int s = 0; s < c->cur_segments; s++) {
        if (hipEventElapsedTime(&ms, c->ev_cnt[s][0], c->ev_cnt[s][1]) == hipSuccess) c->times.ms_count += ms;
        if (hipEventElapsedTime(&ms, c->ev_cnt[s][1], c->ev_scan[s]) == hipSuccess) c->times.ms_scan += ms;
        if (hipEventElapsedTime(&ms, c->ev_light[s][0], c->ev_light[s][1]) == hipSuccess) c->times.ms_dp_light += ms;
        if (hipEventElapsedTime(&ms, c->ev_side[0][s][0], c->ev_side[0][s][1]) == hipSuccess) c->times.ms_dp_big += ms;
        if (hipEventElapsedTime(&ms, c->ev_side[1][s][0], c->ev_side[1][s][1]) == hipSuccess) c->times.ms_dp_mid += ms;
    }
    /* DP time that is NOT hidden under a count kernel: last count kernel's end -> everything done */
    if (c->cur_segments > 0 && hipEventElapsedTime(&dp_end, c->ev[1], c->ev[3]) == hipSuccess) {
        c->times.ms_dp = dp_end;
    }
    c->times.n_segments = c->cur_segments;
    memset(&c->work, 0, sizeof(c->work));
    memcpy(&c->work.cells, g + LFQ_GC_CELLS, 8);
    memcpy(&c->work.rows, g + LFQ_GC_ROWS, 8);
    c->work.n_light_retry = g[LFQ_GC_SCREEN_RETRY];
    c->work.n_approx_pruned = g[LFQ_GC_APPROX_PRUNED];
    for (int s = 0; s < c->cur_segments; s++) {
        const int32_t *sc = c->h_counters + s * LFQ_NCOUNTERS;
        c->work.n_light += sc[LFQ_CNT_LIGHT];
        c->work.n_mid += sc[LFQ_CNT_MID];
        c->work.n_big += sc[LFQ_CNT_BIG];
    }
    {
        /* the screen-kernel variant for this context's NEXT batch: the smallest one that leaves at most 1 in 2000 light
         * columns of THIS batch to the retry kernel (lfq_launch_dp_quad) */
        const int32_t *sc = c->h_counters;
        const int64_t n_light = sc[LFQ_CNT_LIGHT];
        int hint = 64;
        for (int f = LFQ_NKHIST - 1; f >= 0; f--) {
            if ((int64_t)sc[LFQ_CNT_KHIST + f] >= n_light - n_light / 2000) {
                hint = lfq_khist_thr(f) + 1;
            }
        }
        if (n_light > 0) {
            (c->cur_indel_mode ? c->kreg_hint_indel : c->kreg_hint) = hint;
        }
    }
    c->work.bytes_read_count = c->cur_count_read;
    c->work.bytes_written_count = c->cur_count_written;
    if (stats) {
        stats->n_tested = g[LFQ_GC_TESTED];
        stats->n_pvals = std::min<int64_t>(g[LFQ_GC_PVALS], c->cur_pvals_cap);
        stats->n_obs = batch_obs;
    }
    if (g[LFQ_GC_OVERFLOW]) {
        return LFQ_ERR_CAPACITY;
    }
    return LFQ_OK;
}

/* profiling aid (not part of the public header): raw device counters of the last batch */
int lfq_debug_counters(lfq_ctx *c, int32_t *out16)
{
    if (!c || !out16) {
        return LFQ_ERR_INVALID;
    }
    memcpy(out16, c->h_counters, 64 * sizeof(int32_t));   /* segment 0; the caller's buffer holds 64 values */
    return LFQ_OK;
}

int lfq_last_dp_work(lfq_ctx *c, lfq_dp_work *w)
{
    if (!c || !w) {
        return LFQ_ERR_INVALID;
    }
    *w = c->work;
    return LFQ_OK;
}

int lfq_last_kernel_times(lfq_ctx *c, lfq_kernel_times *t)
{
    if (!c || !t) {
        return LFQ_ERR_INVALID;
    }
    *t = c->times;
    return LFQ_OK;
}

/* host tracks -> one padded device allocation (16-byte contract); device tracks pass through */
static int stage_tracks(lfq_ctx *c, const lfq_tracks *tr, int tracks_on_device, lfq_tracks *dev_out)
{
    const int64_t ncols = tr->ncols;
    lfq_tracks dev = *tr;
    if (ncols > 0 && (!tr->nt || !tr->bq || !tr->mq || !tr->col_off || !tr->ref_base)) {
        return LFQ_ERR_INVALID;         /* (baq and sq may be NULL: track off) */
    }
    if (!tracks_on_device) {
        /* host buffers: stage them (padded to the 16-byte contract) in one device allocation */
        const uint64_t n_obs = tr->col_off[ncols];
        const int64_t trk = (int64_t)((n_obs + 15) / 16 * 16) + 16;
        int64_t need = 5 * trk + (ncols + 1) * 8 + (ncols + 16) + 2 * (ncols * 4 + 16) + 64;
        /* the copies go to the context's own upload stream: the device's main stream is shared by its contexts, and a
         * second context's upload queued there would wait behind this one's count kernel instead of running beside it.
         * The batch's first kernel waits for the copies (event), the copies for the previous batch of this context. */
        if (!c->up_stream && hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking) != hipSuccess) {
            c->up_stream = nullptr;
            return LFQ_ERR_HIP;
        }
        if (!c->ev_up && hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming) != hipSuccess) {
            return LFQ_ERR_HIP;
        }
        hipStream_t ups = c->up_stream;
        LFQ_TRY(order_after_batch(c, ups));            /* the previous batch may still read the staging area */
        LFQ_TRY(grow(&c->d_stage, &c->stage_bytes, need));
        uint8_t *p = c->d_stage;
        auto put = [&](const void *src, int64_t bytes, int64_t reserve) -> uint8_t * {
            uint8_t *dst = p;
            p += (reserve + 15) / 16 * 16;
            if (!src) {
                return nullptr;
            }
            if (bytes > 0 && hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, ups) != hipSuccess) {
                return nullptr;
            }
            return dst;
        };
        /* a host producer that packs its nt track (lfq_pack_nt_track, or nibble by nibble as the columns arrive) sends
         * half the bytes of that track over PCIe and gets the 1.5-bytes-per-observation count kernel */
        dev.nt = put(tr->nt, (tr->flags & LFQ_TRACKS_NT_PACKED) ? (int64_t)((n_obs + 7) / 8 * 4) : (int64_t)n_obs, trk);
        dev.bq = put(tr->bq, (int64_t)n_obs, trk);
        dev.baq = put(tr->baq, (int64_t)n_obs, trk);
        dev.mq = put(tr->mq, (int64_t)n_obs, trk);
        dev.sq = put(tr->sq, (int64_t)n_obs, trk);
        dev.col_off = (const uint64_t *)put(tr->col_off, (ncols + 1) * 8, (ncols + 1) * 8);
        dev.ref_base = put(tr->ref_base, ncols, ncols + 16);
        dev.coverage_plp = (const int32_t *)put(tr->coverage_plp, ncols * 4, ncols * 4 + 16);
        dev.num_bases = (const int32_t *)put(tr->num_bases, ncols * 4, ncols * 4 + 16);
        if (!dev.nt || !dev.bq || !dev.mq || !dev.col_off || !dev.ref_base) {
            return LFQ_ERR_HIP;
        }
        if (dev.max_col_obs <= 0) {
            uint64_t md = 0;
            for (int64_t i = 0; i < ncols; i++) {
                md = std::max(md, tr->col_off[i + 1] - tr->col_off[i]);
            }
            dev.max_col_obs = (int64_t)md;
        }
        LFQ_TRY_HIP(hipEventRecord(c->ev_up, ups));
        LFQ_TRY_HIP(hipStreamWaitEvent(c->stream, c->ev_up, 0));
    }

    *dev_out = dev;
    return LFQ_OK;
}

/* layer 2, asynchronous half: stage the tracks if they are host buffers and launch the kernels of the batch */
int lfq_call_snvs_submit(lfq_ctx *c, const lfq_conf *conf, const lfq_tracks *tr, int tracks_on_device)
{
    if (!c || !conf || !tr || tr->ncols < 0) {
        return LFQ_ERR_INVALID;
    }
    if (c->sub_ncols >= 0) {
        return LFQ_ERR_INVALID;         /* a submitted batch has not been collected: one batch in flight per context */
    }
    c->sub_t0 = lfq_now_ms();
    if (tr->ncols == 0) {
        c->sub_ncols = 0;
        return LFQ_OK;
    }
    LFQ_TRY_HIP(hipSetDevice(c->device));
    const int64_t ncols = tr->ncols;
    lfq_tracks dev;
    LFQ_TRY(stage_tracks(c, tr, tracks_on_device, &dev));
    LFQ_TRY(grow(&c->d_counts, &c->counts_cap, ncols));
    LFQ_TRY(grow(&c->d_pvals, &c->pvals_cap, ncols));
    LFQ_TRY(lfq_snv_batch_device(c, conf, &dev, c->d_counts, c->d_pvals, c->pvals_cap, c->stream));
    c->sub_ncols = ncols;
    c->sub_ref_host = tracks_on_device ? nullptr : tr->ref_base;
    c->sub_t1 = lfq_now_ms();
    return LFQ_OK;
}

/* blocks until the kernels of the batch submitted on this context are done (nothing else): a caller that keeps two
 * contexts busy waits here, submits the next batch on the other context, and only then collects this one */
int lfq_call_snvs_wait(lfq_ctx *c)
{
    if (!c || c->sub_ncols < 0) {
        return LFQ_ERR_INVALID;
    }
    if (c->sub_ncols == 0) {
        return LFQ_OK;
    }
    LFQ_TRY_HIP(hipSetDevice(c->device));
    LFQ_TRY_HIP(hipEventSynchronize(c->ev[3]));
    return LFQ_OK;
}

/* layer 2, second half: wait for the batch submitted on this context, fetch the sparse records, exact emit test,
 * strand bias, records; conf's Bonferroni bookkeeping as the per-column loop does it */
int lfq_call_snvs_collect(lfq_ctx *c, lfq_conf *conf, lfq_snv_record *records, int64_t records_capacity,
                          int64_t *n_records, lfq_col_counts *h_counts_or_null, lfq_batch_stats *stats_out)
{
    if (!c || !conf || !n_records || c->sub_ncols < 0) {
        return LFQ_ERR_INVALID;
    }
    *n_records = 0;
    const int64_t ncols = c->sub_ncols;
    c->sub_ncols = -1;
    if (ncols == 0) {
        if (stats_out) memset(stats_out, 0, sizeof(*stats_out));
        return LFQ_OK;
    }
    LFQ_TRY_HIP(hipSetDevice(c->device));
    double tp[4];
    tp[0] = lfq_now_ms();
    lfq_batch_stats st;
    LFQ_TRY(lfq_batch_finish(c, &st));
    tp[1] = lfq_now_ms();
    LfqPin<lfq_col_pvals> h_pv(c, (size_t)st.n_pvals);
    LFQ_PIN_OK(h_pv);
    if (st.n_pvals > 0) {
        LFQ_TRY_HIP(hipMemcpy(h_pv.data(), c->d_pvals, (size_t)st.n_pvals * sizeof(lfq_col_pvals),
                              hipMemcpyDeviceToHost));
    }
    if (h_counts_or_null) {
        LFQ_TRY_HIP(hipMemcpy(h_counts_or_null, c->d_counts, (size_t)ncols * sizeof(lfq_col_counts),
                              hipMemcpyDeviceToHost));
    }
    tp[2] = lfq_now_ms();
    if (c->leader) {        /* this batch's strand-bias tables: usually long done (they ran under the DP kernels) */
        std::unique_lock<std::mutex> lk(*c->lm);
        c->lcv->wait(lk, [&] { return c->sb_pending == 0 || c->leader_stop; });
    }
    /* the reference base of a surviving column travels in its record (lfq_col_pvals.ref_base) */
    int rc = lfq_finalize_pvals(conf, h_pv.data(), st.n_pvals, nullptr, c->sub_ref_host, records, records_capacity,
                                n_records);
    tp[3] = lfq_now_ms();
    if (lfq_timing_on) {
        fprintf(stderr, "[lfq timing] launch %.3f  wait %.3f  d2h %.3f  finalize %.3f ms (kernels %.3f ms, %ld records)\n",
                c->sub_t1 - c->sub_t0, tp[1] - tp[0], tp[2] - tp[1], tp[3] - tp[2], (double)c->times.ms_total, (long)st.n_pvals);
    }
    /* Bonferroni bookkeeping of the per-column loop (lofreq_call.c:794-801) */
    if (st.n_tested > 0) {
        if (conf->bonf_dynamic) {
            conf->bonf_subst = ((conf->bonf_subst == 1) ? 0 : conf->bonf_subst) + 3 * st.n_tested;
        }
        conf->num_snv_tests += 3 * st.n_tested;
    }
    if (stats_out) {
        *stats_out = st;
    }
    return rc;
}

/* second half for a sharded run: the sparse device records as they are (shard-local running Bonferroni factors, no
 * emit test yet) -- the caller exchanges its test counts, rebases the factors (lfq_shard_rebase_bonferroni) and only
 * then runs lfq_finalize_pvals.  conf is not touched. */
int lfq_call_snvs_collect_pvals(lfq_ctx *c, lfq_col_pvals *pvals, int64_t pvals_capacity, int64_t *n_pvals,
                                lfq_batch_stats *stats_out)
{
    if (!c || !n_pvals || c->sub_ncols < 0 || pvals_capacity < 0 || (pvals_capacity > 0 && !pvals)) {
        return LFQ_ERR_INVALID;
    }
    *n_pvals = 0;
    const int64_t ncols = c->sub_ncols;
    c->sub_ncols = -1;
    if (ncols == 0) {
        if (stats_out) memset(stats_out, 0, sizeof(*stats_out));
        return LFQ_OK;
    }
    LFQ_TRY_HIP(hipSetDevice(c->device));
    lfq_batch_stats st;
    LFQ_TRY(lfq_batch_finish(c, &st));
    if (c->leader) {        /* the strand-bias precompute of this batch is finished before its context is reused */
        std::unique_lock<std::mutex> lk(*c->lm);
        c->lcv->wait(lk, [&] { return c->sb_pending == 0 || c->leader_stop; });
    }
    if (stats_out) {
        *stats_out = st;
    }
    *n_pvals = st.n_pvals;
    if (st.n_pvals > pvals_capacity) {
        return LFQ_ERR_CAPACITY;
    }
    if (st.n_pvals > 0) {
        LfqPin<lfq_col_pvals> h_pv(c, (size_t)st.n_pvals);
        LFQ_PIN_OK(h_pv);
        LFQ_TRY_HIP(hipMemcpy(h_pv.data(), c->d_pvals, (size_t)st.n_pvals * sizeof(lfq_col_pvals), hipMemcpyDeviceToHost));
        memcpy(pvals, h_pv.data(), (size_t)st.n_pvals * sizeof(lfq_col_pvals));
    }
    return LFQ_OK;
}

int lfq_call_snvs_batch(lfq_ctx *c, lfq_conf *conf, const lfq_tracks *tr, int tracks_on_device,
                        lfq_snv_record *records, int64_t records_capacity, int64_t *n_records,
                        lfq_col_counts *h_counts_or_null, lfq_batch_stats *stats_out)
{
    if (!c || !conf || !tr || !n_records || tr->ncols < 0) {
        return LFQ_ERR_INVALID;
    }
    *n_records = 0;
    /* nobody looks at the dense entries unless h_counts is asked for: strand counts only where a record comes out */
    c->lazy_forced = 1;
    c->lazy_now = h_counts_or_null == nullptr;
    const int rc = lfq_call_snvs_submit(c, conf, tr, tracks_on_device);
    c->lazy_forced = 0;
    LFQ_TRY(rc);
    return lfq_call_snvs_collect(c, conf, records, records_capacity, n_records, h_counts_or_null, stats_out);
}

int lfq_call_indel_tests_batch(lfq_ctx *c, lfq_conf *conf, const lfq_tracks *tr, int tracks_on_device,
                               lfq_indel_call *calls, int64_t calls_capacity, int64_t *n_calls,
                               lfq_batch_stats *stats_out)
{
    if (!c || !conf || !tr || !n_calls || tr->ncols < 0) {
        return LFQ_ERR_INVALID;
    }
    *n_calls = 0;
    if (tr->ncols == 0) {
        if (stats_out) memset(stats_out, 0, sizeof(*stats_out));
        return LFQ_OK;
    }
    LFQ_TRY_HIP(hipSetDevice(c->device));
    const int64_t ncols = tr->ncols;
    lfq_tracks dev;
    LFQ_TRY(stage_tracks(c, tr, tracks_on_device, &dev));
    LFQ_TRY(grow(&c->d_counts, &c->counts_cap, ncols));
    LFQ_TRY(grow(&c->d_pvals, &c->pvals_cap, ncols));
    LFQ_TRY(lfq_indel_batch_device(c, conf, &dev, c->d_counts, c->d_pvals, c->pvals_cap, c->stream));
    lfq_batch_stats st;
    LFQ_TRY(lfq_batch_finish(c, &st));
    LfqPin<lfq_col_pvals> h_pv(c, (size_t)st.n_pvals);
    LFQ_PIN_OK(h_pv);
    if (st.n_pvals > 0) {
        LFQ_TRY_HIP(hipMemcpy(h_pv.data(), c->d_pvals, (size_t)st.n_pvals * sizeof(lfq_col_pvals),
                              hipMemcpyDeviceToHost));
    }
    std::sort(h_pv.data(), h_pv.data() + h_pv.size(), [](const lfq_col_pvals &a, const lfq_col_pvals &b) { return a.col < b.col; });
    int64_t n_out = 0;
    int rc = LFQ_OK;
    for (size_t pi = 0; pi < h_pv.size(); pi++) {
        const lfq_col_pvals &r = h_pv[pi];
        const long double pv = lfq_pvalue_from_log(r.logp[0], r.status[0]);
        if (pv * (long long)r.bonf < conf->sig) {                 /* lofreq_call.c:326 / :384 */
            if (n_out >= calls_capacity) {
                rc = LFQ_ERR_CAPACITY;
                break;
            }
            lfq_indel_call &o = calls[n_out++];
            o.test = r.col;
            o.bonf = r.bonf;
            o.pvalue = pv;
            o.qual = (int)(-10.0 * log10l(pv));                   /* PROB_TO_PHREDQUAL, utils.h:45 */
            o.count = r.counts.alt_counts[0];
        }
    }
    *n_calls = n_out;
    /* every pseudo-column is one test (lofreq_call.c:693-696, 715-718) */
    if (conf->bonf_dynamic) {
        conf->bonf_indel += st.n_tested;
    }
    conf->num_indel_tests += st.n_tested;
    if (stats_out) {
        *stats_out = st;
    }
    return rc;
}

namespace {

/* pseudo-columns of a run of indel tests, host side */
struct IndelPack {
    std::vector<uint8_t> nt, bq, baq, mq, sq, ref;
    std::vector<uint64_t> off{0};
    struct Meta {
        int64_t col;
        int32_t side, event;
    };
    std::vector<Meta> meta;
    int64_t max_obs = 0;
    void clear()
    {
        nt.clear(); bq.clear(); baq.clear(); mq.clear(); sq.clear(); ref.clear();
        off.assign(1, 0);
        meta.clear();
        max_obs = 0;
    }
};

inline uint8_t q8(int q)            /* phred int -> track byte; -1 (n/a) -> 255 */
{
    return q < 0 ? (uint8_t)LFQ_Q_MISSING : (uint8_t)std::min(q, 254);
}

inline int nt4_of(char ch)          /* bam_nt4_table for the letters the poly-AT rule looks at */
{
    switch (ch) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return 4;
    }
}

/* plp_to_ins_errprobs / plp_to_del_errprobs (snpcaller.c:502-623) as track bytes for one tested event */
void pack_indel_test(IndelPack &pk, const lfq_indel_columns *b, const lfq_conf *conf, int sd, int64_t c, int64_t ev)
{
    const lfq_indel_side &S = b->side[sd];
    const bool use_mq = (conf->flag & LFQ_USE_MQ) != 0;
    const bool use_sq = (conf->flag & LFQ_USE_SQ) != 0 && S.rd_sq;
    const bool use_aq = (conf->flag & LFQ_USE_IDAQ) != 0 && S.rd_aq;
    /* sized once, filled through raw pointers: this loop moves every read of every tested column */
    const int64_t n_ne = S.ne_off[c + 1] - S.ne_off[c];
    const int64_t n_rd = S.rd_off[S.ev_off[c + 1]] - S.rd_off[S.ev_off[c]];
    const size_t base = pk.nt.size(), total = base + (size_t)(n_ne + n_rd);
    pk.nt.resize(total);
    pk.bq.resize(total);
    pk.baq.resize(total);
    pk.mq.resize(total);
    pk.sq.resize(total);
    uint8_t *p_nt = pk.nt.data() + base, *p_bq = pk.bq.data() + base, *p_baq = pk.baq.data() + base,
            *p_mq = pk.mq.data() + base, *p_sq = pk.sq.data() + base;
    {
        const int16_t *q = S.ne_q + S.ne_off[c], *m = (use_mq && S.ne_mq) ? S.ne_mq + S.ne_off[c] : nullptr;
        memset(p_nt, 0, (size_t)n_ne);
        memset(p_baq, LFQ_Q_MISSING, (size_t)n_ne);
        memset(p_sq, LFQ_Q_MISSING, (size_t)n_ne);
        for (int64_t i = 0; i < n_ne; i++) {
            p_bq[i] = (uint8_t)std::min(std::max((int)q[i], 0), 254);
        }
        if (m) {
            for (int64_t i = 0; i < n_ne; i++) {
                p_mq[i] = q8(m[i]);
            }
        } else {
            memset(p_mq, LFQ_Q_MISSING, (size_t)n_ne);
        }
    }
    int64_t w = n_ne;
    for (int64_t e = S.ev_off[c]; e < S.ev_off[c + 1]; e++) {
        const bool me = e == ev;                     /* strcmp(it->key, key) == 0 (snpcaller.c:540) */
        for (int64_t i = S.rd_off[e]; i < S.rd_off[e + 1]; i++, w++) {
            p_nt[w] = me ? 1 : 0;
            p_bq[w] = (uint8_t)std::min(std::max((int)S.rd_q[i], 0), 254);
            p_baq[w] = me && use_aq ? q8(S.rd_aq[i]) : (uint8_t)LFQ_Q_MISSING;
            p_mq[w] = use_mq && S.rd_mq ? q8(S.rd_mq[i]) : (uint8_t)LFQ_Q_MISSING;
            p_sq[w] = use_sq ? q8(S.rd_sq[i]) : (uint8_t)LFQ_Q_MISSING;
        }
    }
    const uint64_t end = pk.nt.size();
    pk.max_obs = std::max<int64_t>(pk.max_obs, (int64_t)(end - pk.off.back()));
    pk.off.push_back(end);
    pk.ref.push_back('A');
    pk.meta.push_back({c, sd, (int32_t)ev});
}

}  // namespace

int lfq_call_indels_batch(lfq_ctx *c, lfq_conf *conf, const lfq_indel_columns *b, lfq_indel_record *recs,
                          int64_t cap, int64_t *n_records, int64_t *n_tests_out)
{
    if (!c || !conf || !b || !n_records || b->ncols < 0) {
        return LFQ_ERR_INVALID;
    }
    *n_records = 0;
    int64_t n_tests = 0, n_out = 0;
    IndelPack pk;
    std::vector<lfq_indel_call> calls;
    const uint64_t flush_obs = 256u << 20;            /* pseudo-column bytes per track per device batch */
    /* columns that came out of lfq_readset_pileup_indels on this context still have their quality arrays in HBM:
     * the pseudo-columns are then built by lfq_indel_pack_kernel instead of on the host */
    const bool dev_pack = c->plp_indel && b == &c->plp_indel->cols && c->d_plp_ne && !lfq_knobs().indel_host_pack;
    for (int sd = 0; sd < 2 && !dev_pack; sd++) {
        if (b->ncols > 0 && b->side[sd].ne_off[b->ncols] > 0 && !b->side[sd].ne_q) {
            return LFQ_ERR_INVALID;         /* device-only columns that are no longer the context's current ones */
        }
    }
    std::vector<LfqIndelTestDesc> descs;
    uint64_t dev_obs = 0;
    int16_t *d_rd = nullptr;                          /* event-read arrays of both sides, uploaded once */
    int64_t rd_n[2] = {0, 0};
    if (dev_pack) {
        LFQ_TRY_HIP(hipSetDevice(c->device));
        for (int sd = 0; sd < 2; sd++) {
            rd_n[sd] = b->side[sd].rd_off[b->side[sd].ev_off[b->ncols]];
        }
        const int64_t tot = 4 * (rd_n[0] + rd_n[1]);
        if (tot > 0) {
            LFQ_TRY(grow(&c->d_tmp[3], &c->tmp_bytes[3], tot * 2));
            d_rd = (int16_t *)c->d_tmp[3];
            int64_t o = 0;
            for (int sd = 0; sd < 2; sd++) {
                const int16_t *src[4] = {b->side[sd].rd_q, b->side[sd].rd_aq, b->side[sd].rd_mq, b->side[sd].rd_sq};
                for (int k = 0; k < 4; k++, o += rd_n[sd]) {
                    if (rd_n[sd] > 0) {
                        LFQ_TRY_HIP(hipMemcpyAsync(d_rd + o, src[k], (size_t)rd_n[sd] * 2, hipMemcpyHostToDevice, c->stream));
                    }
                }
            }
        }
    }

    double t_flush = 0.0, t_tests = 0.0;
    const double t_begin = lfq_now_ms();
    auto flush = [&]() -> int {
        if (pk.meta.empty()) {
            return LFQ_OK;
        }
        const double tf0 = lfq_now_ms();
        lfq_tracks tr;
        memset(&tr, 0, sizeof(tr));
        tr.ncols = (int64_t)pk.meta.size();
        tr.max_col_obs = pk.max_obs;
        uint8_t *d_trk = nullptr;
        if (dev_pack) {
            const int64_t nt_ = (int64_t)descs.size(), trk = (int64_t)((dev_obs + 15) / 16 * 16) + 32;
            auto al = [](int64_t x) { return (x + 255) / 256 * 256; };
            const int64_t o_desc = 0, o_off = o_desc + al(nt_ * (int64_t)sizeof(LfqIndelTestDesc)), o_ref = o_off + al((nt_ + 1) * 8),
                          o_trk = o_ref + al(nt_ + 16), total = o_trk + 5 * trk;
            LFQ_TRY(grow(&c->d_tmp[4], &c->tmp_bytes[4], total));
            d_trk = c->d_tmp[4];
            /* descriptors, offsets and reference bytes in one pinned block (laid out as on the device), one copy */
            LfqPin<uint8_t> hp(c, (size_t)o_trk);
            LFQ_PIN_OK(hp);
            memcpy(hp.data() + o_desc, descs.data(), (size_t)nt_ * sizeof(LfqIndelTestDesc));
            memcpy(hp.data() + o_off, pk.off.data(), (size_t)(nt_ + 1) * 8);
            memcpy(hp.data() + o_ref, pk.ref.data(), (size_t)nt_);
            LFQ_TRY_HIP(hipMemcpyAsync(d_trk, hp.data(), (size_t)o_trk, hipMemcpyHostToDevice, c->stream));
            /* the pinned block goes back to the pool at the end of this scope: the copy must have read it by then */
            LFQ_TRY_HIP(hipStreamSynchronize(c->stream));
            LFQ_TRY_HIP(hipMemsetAsync(d_trk + o_trk, 0, (size_t)(5 * trk), c->stream));      /* the 16-byte tails are read */
            LfqIndelPackArgs A;
            memset(&A, 0, sizeof(A));
            A.tests = (const LfqIndelTestDesc *)(d_trk + o_desc);
            A.n_tests = nt_;
            A.ne_q[0] = c->d_plp_ne;
            A.ne_mq[0] = c->d_plp_ne + c->plp_ne_total[0];
            A.ne_q[1] = c->d_plp_ne + 2 * c->plp_ne_total[0];
            A.ne_mq[1] = c->d_plp_ne + 2 * c->plp_ne_total[0] + c->plp_ne_total[1];
            int64_t o = 0;
            for (int sd = 0; sd < 2; sd++) {
                A.rd_q[sd] = d_rd + o; o += rd_n[sd];
                A.rd_aq[sd] = d_rd + o; o += rd_n[sd];
                A.rd_mq[sd] = d_rd + o; o += rd_n[sd];
                A.rd_sq[sd] = d_rd + o; o += rd_n[sd];
            }
            A.use_mq = (conf->flag & LFQ_USE_MQ) ? 1 : 0;
            A.use_sq = (conf->flag & LFQ_USE_SQ) ? 1 : 0;
            A.use_aq = (conf->flag & LFQ_USE_IDAQ) ? 1 : 0;
            A.nt = d_trk + o_trk;
            A.bq = A.nt + trk;
            A.baq = A.bq + trk;
            A.mq = A.baq + trk;
            A.sq = A.mq + trk;
            LFQ_TRY(lfq_launch_indel_pack(A, c->stream));
            tr.nt = A.nt; tr.bq = A.bq; tr.baq = A.baq; tr.mq = A.mq; tr.sq = A.sq;
            tr.col_off = (const uint64_t *)(d_trk + o_off);
            tr.ref_base = d_trk + o_ref;
        } else {
            const size_t pad = 32;
            for (auto *v : {&pk.nt, &pk.bq, &pk.baq, &pk.mq, &pk.sq}) {
                v->resize(v->size() + pad, 0);            /* 16-byte tail contract of the track format */
            }
            tr.nt = pk.nt.data();
            tr.bq = pk.bq.data();
            tr.baq = pk.baq.data();
            tr.mq = pk.mq.data();
            tr.sq = pk.sq.data();
            tr.col_off = pk.off.data();
            tr.ref_base = pk.ref.data();
        }
        calls.resize(pk.meta.size());
        int64_t nc = 0;
        lfq_batch_stats st;
        const double tf1 = lfq_now_ms();
        LFQ_TRY(lfq_call_indel_tests_batch(c, conf, &tr, dev_pack ? 1 : 0, calls.data(), (int64_t)calls.size(), &nc, &st));
        t_tests += lfq_now_ms() - tf1;
        t_flush += tf1 - tf0;
        if (st.n_tested != (int64_t)pk.meta.size()) {
            return LFQ_ERR_INVALID;                   /* every packed event must have been a test */
        }
        n_tests += st.n_tested;
        for (int64_t i = 0; i < nc; i++) {
            const IndelPack::Meta &m = pk.meta[(size_t)calls[i].test];
            const lfq_indel_side &S = b->side[m.side];
            if (n_out >= cap) {
                return LFQ_ERR_CAPACITY;
            }
            lfq_indel_record &r = recs[n_out++];
            memset(&r, 0, sizeof(r));
            r.col = m.col;
            r.side = m.side;
            r.event = m.event;
            r.qual = calls[i].qual;
            r.count = calls[i].count;
            r.bonf = calls[i].bonf;
            r.pvalue = calls[i].pvalue;
            r.af = r.count / ((float)b->coverage_plp[m.col] - b->num_tails[m.col]);   /* lofreq_call.c:334 */
            r.dp = b->coverage_plp[m.col] - b->num_tails[m.col];                       /* lofreq_call.c:132 */
            r.ref_fw = S.non_fw[m.col];
            r.ref_rv = S.non_rv[m.col];
            r.alt_fw = S.ev_fw[m.event];
            r.alt_rv = S.ev_rv[m.event];
            r.sb = lfq_sb_phred(r.ref_fw, r.ref_rv, r.alt_fw, r.alt_rv);
            r.hrun = b->hrun ? b->hrun[m.col] : 0;
        }
        pk.clear();
        descs.clear();
        dev_obs = 0;
        return LFQ_OK;
    };

    /* the gates of call_vars' indel part over the columns [c0, c1): emit(col, side, event) for every event that is tested */
    auto scan = [&](int64_t c0, int64_t c1, auto &&emit) -> int {
    for (int64_t col = c0; col < c1; col++) {
        if (b->ref_base[col] == 'N') {
            continue;                                                        /* lofreq_call.c:892 */
        }
        if (b->num_non_indels[col] + b->num_ins[col] + b->num_dels[col] < conf->min_cov) {
            continue;                                                        /* :626 */
        }
        /* low-AF 1-bp A/T insertion AND deletion of the same base at one column: skipped (:649-681) */
        bool ign[5] = {false, false, false, false, false};
        const int64_t ne_ins = b->side[0].ne_off[col + 1] - b->side[0].ne_off[col];
        const int64_t ne_del = b->side[1].ne_off[col + 1] - b->side[1].ne_off[col];
        if (b->num_ins[col] && ne_ins && b->num_dels[col] && ne_del) {
            int cnt[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
            for (int sd = 0; sd < 2; sd++) {
                const lfq_indel_side &S = b->side[sd];
                for (int64_t e = S.ev_off[col]; e < S.ev_off[col + 1]; e++) {
                    const char *key = S.key_chars + S.key_off[e];
                    if (S.key_off[e + 1] - S.key_off[e] == 1 && (key[0] == 'A' || key[0] == 'T')) {
                        cnt[sd][nt4_of(key[0])] = (int)(S.rd_off[e + 1] - S.rd_off[e]);
                    }
                }
            }
            const float denom = (float)(b->coverage_plp[col] - b->num_tails[col]);
            for (int i = 0; i < 5; i++) {
                if (cnt[0][i] && cnt[1][i] && cnt[0][i] / denom < 0.05f && cnt[1][i] / denom < 0.05f) {
                    ign[i] = true;
                }
            }
        }
        for (int sd = 0; sd < 2; sd++) {
            const lfq_indel_side &S = b->side[sd];
            if (!(sd == 0 ? b->num_ins[col] : b->num_dels[col])) {
                continue;                                                    /* :684 / :706 */
            }
            for (int64_t e = S.ev_off[col]; e < S.ev_off[col + 1]; e++) {
                const char *key = S.key_chars + S.key_off[e];
                if (S.key_off[e + 1] - S.key_off[e] == 1 && ign[nt4_of(key[0])]) {
                    continue;                                                /* :687-689 / :709-711 */
                }
                LFQ_TRY(emit(col, sd, e));
            }
        }
    }
    return LFQ_OK;
    };
    /* one tested event of the device-packed path: where its pseudo-column comes from and goes to */
    auto describe = [&](int64_t col, int sd, int64_t e, uint64_t out_off) {
        const lfq_indel_side &S = b->side[sd];
        LfqIndelTestDesc D;
        memset(&D, 0, sizeof(D));
        D.out_off = (int64_t)out_off;
        D.ne_off = S.ne_off[col];
        D.ne_len = (int32_t)(S.ne_off[col + 1] - S.ne_off[col]);
        D.rd_begin = S.rd_off[S.ev_off[col]];
        D.rd_len = (int32_t)(S.rd_off[S.ev_off[col + 1]] - D.rd_begin);
        D.me_begin = (int32_t)(S.rd_off[e] - D.rd_begin);
        D.me_len = (int32_t)(S.rd_off[e + 1] - S.rd_off[e]);
        D.side = sd;
        return D;
    };
    auto append = [&](const LfqIndelTestDesc &D, int64_t col, int64_t e) {
        descs.push_back(D);
        dev_obs = (uint64_t)D.out_off + (uint64_t)(D.ne_len + D.rd_len);
        pk.max_obs = std::max<int64_t>(pk.max_obs, D.ne_len + D.rd_len);
        pk.off.push_back(dev_obs);
        pk.ref.push_back('A');
        pk.meta.push_back({col, D.side, (int32_t)e});
    };
    struct PartTests {
        std::vector<LfqIndelTestDesc> descs;        /* out_off relative to the part's first observation */
        std::vector<IndelPack::Meta> meta;
        uint64_t obs = 0;
    };
    PartTests part_tests[8];
    int scan_parts = 1;
    bool scanned = false;
    if (dev_pack) {
        /* the scan over the columns (1 M of them for 1 Mb, nearly all without an event) split over a few threads; the
         * tests of the parts are appended in column order afterwards, so descriptors, offsets and flushes are those of
         * the one-thread loop unless a part alone exceeds a device batch (then that loop runs) */
        lfq_for_reads(b->ncols, [&](int64_t c0, int64_t c1, int part) {
            PartTests &P = part_tests[part];
            (void)scan(c0, c1, [&](int64_t col, int sd, int64_t e) -> int {
                const LfqIndelTestDesc D = describe(col, sd, e, P.obs);
                P.descs.push_back(D);
                P.meta.push_back({col, sd, (int32_t)e});
                P.obs += (uint64_t)(D.ne_len + D.rd_len);
                return LFQ_OK;
            });
        }, &scan_parts);
        scanned = true;
        for (int q = 0; q < scan_parts; q++) {
            scanned = scanned && part_tests[q].obs < flush_obs;
        }
    }
    if (scanned) {
        size_t total_tests = 0;
        for (int q = 0; q < scan_parts; q++) {
            total_tests += part_tests[q].descs.size();
        }
        descs.reserve(total_tests);
        pk.off.reserve(total_tests + 1);
        pk.meta.reserve(total_tests);
        for (int q = 0; q < scan_parts; q++) {
            const PartTests &P = part_tests[q];
            if (dev_obs + P.obs >= flush_obs) {
                LFQ_TRY(flush());
            }
            const uint64_t first = dev_obs;
            const size_t at = descs.size(), m = P.descs.size();
            descs.insert(descs.end(), P.descs.begin(), P.descs.end());
            pk.meta.insert(pk.meta.end(), P.meta.begin(), P.meta.end());
            pk.ref.insert(pk.ref.end(), m, (uint8_t)'A');
            pk.off.resize(pk.off.size() + m);
            uint64_t *off_out = pk.off.data() + pk.off.size() - m;
            for (size_t i = 0; i < m; i++) {
                LfqIndelTestDesc &D = descs[at + i];
                D.out_off += (int64_t)first;
                const int64_t len = (int64_t)D.ne_len + D.rd_len;
                pk.max_obs = std::max(pk.max_obs, len);
                off_out[i] = (uint64_t)(D.out_off + len);
            }
            dev_obs = first + P.obs;
        }
    } else {
        LFQ_TRY(scan(0, b->ncols, [&](int64_t col, int sd, int64_t e) -> int {
            if (dev_pack) {
                append(describe(col, sd, e, dev_obs), col, e);
            } else {
                pack_indel_test(pk, b, conf, sd, col, e);
            }
            if ((dev_pack ? dev_obs : (uint64_t)pk.nt.size()) >= flush_obs) {
                return flush();
            }
            return LFQ_OK;
        }));
    }
    LFQ_TRY(flush());
    if (lfq_timing_on) {
        const double all = lfq_now_ms() - t_begin;
        fprintf(stderr, "[lfq timing] indel calls: test descriptors %.1f  upload + pack %.1f  tests batch %.1f ms (%ld tests, %ld records)\n",
                all - t_flush - t_tests, t_flush, t_tests, (long)n_tests, (long)n_out);
    }
    *n_records = n_out;
    if (n_tests_out) {
        *n_tests_out = n_tests;
    }
    return LFQ_OK;
}

/* ---- resident read set ------------------------------------------------------------------------------------
 * The reads of one contig region, uploaded once; BAQ / IDAQ, source quality and both pileups work on the device
 * copy and leave their per-base results (lb, ai, ad, sq) there for the next stage.  The host arrays handed to
 * lfq_readset_create stay the caller's and must outlive the read set: the sparse host-side steps (geometry from the
 * CIGARs, the indel event tables) read them in place. */
enum { LFQ_RSC_BLOB = 0, LFQ_RSC_TAGS = 1, LFQ_RSC_PMAX = 2, LFQ_RSC_TAGFL = 3, LFQ_RSC_PINFL = 4 };

static void rs_cache_free(int kind, void *p)
{
    if (p) {
        (void)(kind == LFQ_RSC_PINFL ? hipHostFree(p) : hipFree(p));
    }
}

/* `bytes` of device memory (pinned host memory for LFQ_RSC_PINFL): the cached block of this kind if it is large enough */
static void *rs_cache_take(lfq_ctx *c, int kind, size_t bytes, size_t *cap_out)
{
    auto &e = c->rs_cache[kind];
    if (e.p && e.cap >= bytes) {
        void *p = e.p;
        *cap_out = e.cap;
        e.p = nullptr;
        e.cap = 0;
        return p;
    }
    rs_cache_free(kind, e.p);
    e.p = nullptr;
    e.cap = 0;
    void *p = nullptr;
    const size_t want = std::max<size_t>(bytes, 256);
    const hipError_t rc = (kind == LFQ_RSC_PINFL) ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
    if (rc != hipSuccess) {
        return nullptr;
    }
    *cap_out = want;
    return p;
}

static void rs_cache_give(lfq_ctx *c, int kind, void *p, size_t cap)
{
    if (!p) {
        return;
    }
    auto &e = c->rs_cache[kind];
    if (!e.p || cap > e.cap) {
        rs_cache_free(kind, e.p);
        e.p = p;
        e.cap = cap;
    } else {
        rs_cache_free(kind, p);
    }
}

struct lfq_readset {
    lfq_ctx *c;
    size_t cap[5];                      /* capacities of blob, tag_blob, d_pmax, d_tagfl, h_fl_pin (rs_cache_*) */
    int64_t n, n_bases, n_cig, ref_len;
    const int32_t *pos;
    const int64_t *cigar_off, *seq_off;
    const uint32_t *cigar;
    const uint8_t *seq, *qual, *mapq, *reverse;
    const char *ref;
    const uint8_t *h_bi, *h_bd, *h_ai, *h_ad, *h_flags;     /* tag bytes on the host, as given (may be null) */
    const int32_t *h_sq;
    uint8_t *blob, *tag_blob;           /* inputs; lb / ai / ad computed by lfq_readset_baq */
    uint8_t *d_pos, *d_coff, *d_soff, *d_cig, *d_seq, *d_qual, *d_ref, *d_mapq, *d_rev, *d_bi, *d_bd, *d_lb, *d_ai,
            *d_ad, *d_fl, *d_sqb;
    bool has_lb, has_idaq, has_sqb, has_bi, has_bd;
    std::vector<uint8_t> fl;            /* per read: bit 0..3 = has BI / BD / ai / ad (host flags or from the device BAQ) */
    std::vector<int32_t> sq32;          /* source quality per read once computed */
    /* lfq_readset_baq returns when its kernels are queued: what follows on the device is ordered by the stream, what the
     * host needs (which reads got an ai / ad tag) arrives in pinned memory and is picked up by readset_baq_wait */
    uint8_t *d_tagfl, *h_fl_pin;        /* [n] bit 0: ai, bit 1: ad written by the kernels; [n] merged flags on their way back */
    hipEvent_t ev_baq;
    bool baq_pending, baq_idaq;
    int32_t *d_pmax;                    /* position-sorted reads: running maximum of the end coordinates (lazily) */
    int pmax_state;                     /* 0 unknown, 1 sorted (d_pmax valid), 2 unsorted */
    /* lfq_readset_create returns while the reads are still crossing PCIe (a helper thread feeds the copies of the caller's
     * pageable arrays to the upload stream): host-only work of the next step -- the BAQ geometry -- runs meanwhile, and
     * every step calls readset_upload_wait before its first device operation on the read set */
    std::thread *up_thread;
    std::atomic<int> up_stage;          /* 1: everything but BI / BD has landed (what lfq_readset_baq reads), 2: all of it */
    std::atomic<int> up_chunks;         /* bases + qualities of the reads [0, n * up_chunks / up_nchunks) have landed */
    int up_nchunks;
    std::atomic<int> up_rc;        /* written by the helper thread at the end of each stage */
    LfqPin<uint8_t> *up_fl;             /* the flag bytes on their way out (pinned; handed back once the copies are done) */
};

/* the reads, qualities, CIGARs and the contig are on the device (BI / BD may still be on their way: the BAQ kernels do
 * not read them, and 600 of the 1300 MB of a 2 M-read region then cross PCIe under those kernels) */
static int readset_upload_wait_inputs(lfq_readset *rs)
{
    if (rs && rs->up_thread) {
        while (rs->up_stage.load(std::memory_order_acquire) < 1) {
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        return rs->up_rc;
    }
    return rs ? rs->up_rc.load() : (int)LFQ_OK;
}

/* ... and for a launch over the reads up to (and including) r_last: the small arrays and the chunks of bases and
 * qualities that hold them */
static int readset_upload_wait_reads(lfq_readset *rs, int64_t r_last)
{
    if (rs && rs->up_thread) {
        int need = 1;
        while (need < rs->up_nchunks && rs->n * need / rs->up_nchunks <= r_last) {
            need++;
        }
        while (rs->up_chunks.load(std::memory_order_acquire) < need) {
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    return rs ? rs->up_rc.load() : (int)LFQ_OK;
}

static int readset_upload_wait(lfq_readset *rs)
{
    if (rs && rs->up_thread) {
        rs->up_thread->join();
        delete rs->up_thread;
        rs->up_thread = nullptr;
    }
    if (rs && rs->up_fl) {
        delete rs->up_fl;
        rs->up_fl = nullptr;
    }
    return rs ? rs->up_rc.load() : (int)LFQ_OK;
}

/* the host side of an lfq_readset_baq that is still running: merged tag flags (bit 0 BI, 1 BD, 2 ai, 3 ad) into rs->fl */
static int readset_baq_wait(lfq_readset *rs)
{
    if (!rs || !rs->baq_pending) {
        return LFQ_OK;
    }
    rs->baq_pending = false;
    if (hipEventSynchronize(rs->ev_baq) != hipSuccess) {
        return LFQ_ERR_HIP;
    }
    if (rs->baq_idaq && rs->h_fl_pin) {
        memcpy(rs->fl.data(), rs->h_fl_pin, (size_t)rs->n);
    }
    return LFQ_OK;
}

void lfq_readset_destroy(lfq_readset *rs)
{
    if (rs) {
        (void)readset_upload_wait(rs);
        (void)readset_baq_wait(rs);
        if (rs->ev_baq) (void)hipEventDestroy(rs->ev_baq);
        (void)hipStreamSynchronize(rs->c->stream);      /* nothing queued may still use what goes back to the cache */
        rs_cache_give(rs->c, LFQ_RSC_PINFL, rs->h_fl_pin, rs->cap[LFQ_RSC_PINFL]);
        rs_cache_give(rs->c, LFQ_RSC_TAGFL, rs->d_tagfl, rs->cap[LFQ_RSC_TAGFL]);
        rs_cache_give(rs->c, LFQ_RSC_BLOB, rs->blob, rs->cap[LFQ_RSC_BLOB]);
        rs_cache_give(rs->c, LFQ_RSC_TAGS, rs->tag_blob, rs->cap[LFQ_RSC_TAGS]);
        rs_cache_give(rs->c, LFQ_RSC_PMAX, rs->d_pmax, rs->cap[LFQ_RSC_PMAX]);
        delete rs;
    }
}

int lfq_readset_create(lfq_ctx *c, const lfq_pileup_reads *rd, const lfq_pileup_indel_tags *tg, lfq_readset **out)
{
    if (!c || !rd || !out || rd->n_reads < 0
        || (rd->n_reads > 0 && (!rd->pos || !rd->cigar_off || !rd->cigar || !rd->seq_off || !rd->seq || !rd->ref))) {
        return LFQ_ERR_INVALID;
    }
    *out = nullptr;
    LFQ_TRY_HIP(hipSetDevice(c->device));
    lfq_readset *rs = new lfq_readset();
    rs->c = c;
    rs->n = rd->n_reads;
    rs->n_bases = rs->n > 0 ? rd->seq_off[rs->n] : 0;
    rs->n_cig = rs->n > 0 ? rd->cigar_off[rs->n] : 0;
    rs->ref_len = rd->ref_len;
    rs->pos = rd->pos; rs->cigar_off = rd->cigar_off; rs->seq_off = rd->seq_off; rs->cigar = rd->cigar;
    rs->seq = rd->seq; rs->qual = rd->qual; rs->mapq = rd->mapq; rs->reverse = rd->reverse; rs->ref = rd->ref;
    rs->h_bi = tg ? tg->bi : nullptr; rs->h_bd = tg ? tg->bd : nullptr;
    rs->h_ai = tg ? tg->ai : nullptr; rs->h_ad = tg ? tg->ad : nullptr;
    rs->h_flags = tg ? tg->tag_flags : nullptr;
    rs->h_sq = tg ? tg->sq : nullptr;
    rs->blob = nullptr;
    rs->tag_blob = nullptr;
    rs->d_pmax = nullptr;
    rs->d_tagfl = nullptr;
    rs->h_fl_pin = nullptr;
    memset(rs->cap, 0, sizeof(rs->cap));
    rs->pmax_state = 0;
    rs->up_thread = nullptr;
    rs->up_stage.store(0);
    rs->up_chunks.store(0);
    rs->up_nchunks = 1;
    rs->up_rc = LFQ_OK;
    rs->up_fl = nullptr;
    rs->has_lb = rs->has_idaq = rs->has_sqb = rs->has_bi = rs->has_bd = false;
    const int64_t n = rs->n, nb = rs->n_bases;
    {
        const uint32_t have = (rs->h_bi ? 1u : 0u) | (rs->h_bd ? 2u : 0u) | (rs->h_ai ? 4u : 0u) | (rs->h_ad ? 8u : 0u);
        rs->fl.assign((size_t)std::max<int64_t>(n, 1), (uint8_t)have);      /* no per-read flags: every read has every tag given */
        for (int64_t r = 0; rs->h_flags && r < n; r++) {
            rs->fl[(size_t)r] = (uint8_t)(rs->h_flags[r] & have);
        }
    }
    if (n == 0) {
        *out = rs;
        return LFQ_OK;
    }
    auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
    int64_t off = 0;
    auto take = [&](int64_t bytes) { const int64_t o = off; off += al(bytes); return o; };
    /* per-base arrays only where there is something to put: device allocations of this size are not free.  The
     * outputs of lfq_readset_baq (lb, ai, ad) get their own allocation when that step runs. */
    const int64_t o_pos = take(n * 4), o_coff = take((n + 1) * 8), o_soff = take((n + 1) * 8), o_cig = take(rs->n_cig * 4),
                  o_seq = take(nb + 16), o_qual = take(rd->qual ? nb + 16 : 0), o_ref = take(rs->ref_len + 1), o_mapq = take(n),
                  o_rev = take(n), o_bi = take(rs->h_bi ? nb + 16 : 0), o_bd = take(rs->h_bd ? nb + 16 : 0),
                  o_lb = take(rd->baq ? nb + 16 : 0), o_fl = take(n), o_sqb = take(n);
    rs->blob = (uint8_t *)rs_cache_take(c, LFQ_RSC_BLOB, (size_t)off, &rs->cap[LFQ_RSC_BLOB]);
    if (!rs->blob) {
        delete rs;
        return LFQ_ERR_NOMEM;
    }
    uint8_t *d = rs->blob;
    rs->d_pos = d + o_pos; rs->d_coff = d + o_coff; rs->d_soff = d + o_soff; rs->d_cig = d + o_cig; rs->d_seq = d + o_seq;
    rs->d_qual = d + o_qual; rs->d_ref = d + o_ref; rs->d_mapq = d + o_mapq; rs->d_rev = d + o_rev; rs->d_bi = d + o_bi;
    rs->d_bd = d + o_bd; rs->d_lb = rd->baq ? d + o_lb : nullptr; rs->d_ai = nullptr; rs->d_ad = nullptr; rs->d_fl = d + o_fl;
    rs->d_sqb = d + o_sqb;
    if (!c->up_stream && hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking) != hipSuccess) {
        c->up_stream = nullptr;
        lfq_readset_destroy(rs);
        return LFQ_ERR_HIP;
    }
    rs->up_fl = new LfqPin<uint8_t>(c, (size_t)n);       /* not from rs->fl: see LfqPin */
    if (!rs->up_fl->ok()) {
        lfq_readset_destroy(rs);
        return LFQ_ERR_NOMEM;
    }
    memcpy(rs->up_fl->data(), rs->fl.data(), (size_t)n);
    /* The order of the copies follows what lfq_readset_baq, usually the first step, needs: the small per-read arrays, then
     * bases and qualities in chunks of reads -- its launch over reads [a, b) starts when the chunks up to read b have
     * landed (readset_upload_wait_reads) --, BI / BD, which it does not read, last. */
    struct Copy { uint8_t *dst; const void *src; int64_t bytes; int chunks_after, stage_after; };
    std::vector<Copy> todo;
    int64_t up_bytes = 0;
    auto add = [&](uint8_t *dst, const void *src, int64_t bytes) {
        if (src && bytes > 0) {
            todo.push_back({dst, src, bytes, 0, 0});
            up_bytes += bytes;
        }
    };
    add(rs->d_pos, rd->pos, n * 4);
    add(rs->d_coff, rd->cigar_off, (n + 1) * 8);
    add(rs->d_soff, rd->seq_off, (n + 1) * 8);
    add(rs->d_cig, rd->cigar, rs->n_cig * 4);
    add(rs->d_ref, rd->ref, rs->ref_len);
    add(rs->d_mapq, rd->mapq, n);
    add(rs->d_rev, rd->reverse, n);
    add(rs->d_lb, rd->baq, nb);
    add(rs->d_sqb, rd->sq, n);
    add(rs->d_fl, rs->up_fl->data(), n);
    const int n_chunks = nb >= ((int64_t)64 << 20) ? 4 : 1;
    rs->up_nchunks = n_chunks;
    for (int j = 0; j < n_chunks; j++) {
        const int64_t b0 = rd->seq_off[n * j / n_chunks], b1 = rd->seq_off[n * (j + 1) / n_chunks];
        add(rs->d_seq + b0, rd->seq + b0, b1 - b0);
        add(rs->d_qual + b0, rd->qual ? rd->qual + b0 : nullptr, b1 - b0);
        if (!todo.empty()) {
            todo.back().chunks_after = j + 1;
        }
    }
    if (!todo.empty()) {
        todo.back().stage_after = 1;
    }
    add(rs->d_bi, rs->h_bi, nb);
    add(rs->d_bd, rs->h_bd, nb);
    if (!todo.empty()) {
        todo.back().stage_after = 2;
    }
    rs->has_bi = rs->h_bi != nullptr;
    rs->has_bd = rs->h_bd != nullptr;
    rs->has_lb = rd->baq != nullptr;
    rs->has_sqb = rd->sq != nullptr;
    const int device = c->device;
    hipStream_t ups = c->up_stream;
    auto run = [rs, todo, n_chunks, device, ups]() {
        int rc = hipSetDevice(device) == hipSuccess ? LFQ_OK : LFQ_ERR_HIP;
        for (const Copy &x : todo) {
            if (rc == LFQ_OK && hipMemcpyAsync(x.dst, x.src, (size_t)x.bytes, hipMemcpyHostToDevice, ups) != hipSuccess) {
                rc = LFQ_ERR_HIP;
            }
            if (x.chunks_after || x.stage_after) {      /* progress is published when the copies have landed */
                if (hipStreamSynchronize(ups) != hipSuccess && rc == LFQ_OK) {
                    rc = LFQ_ERR_HIP;
                }
                rs->up_rc = rc;
                if (x.chunks_after) {
                    rs->up_chunks.store(x.chunks_after, std::memory_order_release);
                }
                if (x.stage_after) {
                    rs->up_stage.store(x.stage_after, std::memory_order_release);
                }
            }
        }
        rs->up_rc = rc;
        rs->up_chunks.store(n_chunks, std::memory_order_release);
        rs->up_stage.store(2, std::memory_order_release);
    };
    const int up_mode = lfq_knobs().sync_upload;        /* 0: helper thread from 8 MB on, 1: never, 2: always */
    if (up_mode == 2 || (up_bytes >= ((int64_t)8 << 20) && up_mode == 0)) {
        rs->up_thread = new std::thread(run);           /* readset_upload_wait joins it */
    } else {
        run();
        const int rc = readset_upload_wait(rs);
        if (rc != LFQ_OK) {
            lfq_readset_destroy(rs);
            return rc;
        }
    }
    *out = rs;
    return LFQ_OK;
}

/* position-sorted reads (what mpileup requires) take the column-major pileup kernels, which find the reads that can
 * overlap a position by binary search: they need the running maximum of the end coordinates.  -> device array or null */
static const int32_t *readset_pmax(lfq_ctx *c, lfq_readset *rs, hipStream_t st)
{
    if (rs->pmax_state == 0) {
        rs->pmax_state = 2;
        const int64_t n = rs->n;
        LfqPin<int32_t> pmax(c, (size_t)std::max<int64_t>(n, 1));
        if (!lfq_knobs().pileup_atomic && n > 0 && pmax.ok()) {
            /* two passes split over a few threads: end coordinate and running maximum inside a part (and whether the
             * part is sorted), then the maximum of the parts before it */
            int32_t part_max[8];
            bool part_sorted[8];
            int parts = 1;
            lfq_for_reads(n, [&](int64_t r0, int64_t r1, int part) {
                int32_t run = INT32_MIN;
                bool sorted = true;
                for (int64_t r = r0; r < r1; r++) {
                    const uint32_t *cg = rs->cigar + rs->cigar_off[r];
                    const int nc = (int)(rs->cigar_off[r + 1] - rs->cigar_off[r]);
                    int64_t e = rs->pos[r];
                    for (int k = 0; k < nc; k++) {
                        const int op = cg[k] & 0xf;
                        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) {
                            e += cg[k] >> 4;
                        }
                    }
                    run = std::max<int32_t>(run, (int32_t)std::min<int64_t>(e, INT32_MAX));
                    pmax[(size_t)r] = run;
                    sorted = sorted && (r == 0 || rs->pos[r] >= rs->pos[r - 1]);     /* r0 - 1 belongs to the part before */
                }
                part_max[part] = run;
                part_sorted[part] = sorted;
            }, &parts);
            bool sorted = true;
            for (int q = 0; q < parts; q++) {
                sorted = sorted && part_sorted[q];
            }
            if (sorted && parts > 1) {
                int32_t before[8];
                before[0] = INT32_MIN;
                for (int q = 1; q < parts; q++) {
                    before[q] = std::max(before[q - 1], part_max[q - 1]);
                }
                lfq_for_reads(n, [&](int64_t r0, int64_t r1, int part) {
                    const int32_t m = before[part];
                    for (int64_t r = r0; r < r1 && pmax[(size_t)r] < m; r++) {     /* the running maximum only grows */
                        pmax[(size_t)r] = m;
                    }
                });
            }
            if (sorted) {
                rs->d_pmax = (int32_t *)rs_cache_take(c, LFQ_RSC_PMAX, (size_t)n * 4, &rs->cap[LFQ_RSC_PMAX]);
                if (rs->d_pmax
                    && hipMemcpyAsync(rs->d_pmax, pmax.data(), (size_t)n * 4, hipMemcpyHostToDevice, st) == hipSuccess
                    && hipStreamSynchronize(st) == hipSuccess) {
                    rs->pmax_state = 1;
                } else if (rs->d_pmax) {
                    rs_cache_give(c, LFQ_RSC_PMAX, rs->d_pmax, rs->cap[LFQ_RSC_PMAX]);
                    rs->d_pmax = nullptr;
                }
            }
        }
    }
    return rs->pmax_state == 1 ? rs->d_pmax : nullptr;
}

int lfq_readset_fetch_tags(lfq_ctx *c, lfq_readset *rs, uint8_t *lb_out, uint8_t *ai_out, uint8_t *ad_out, uint8_t *tag_flags)
{
    if (!c || !rs || rs->c != c) {
        return LFQ_ERR_INVALID;
    }
    if (rs->n == 0) {
        return LFQ_OK;
    }
    if ((lb_out && !rs->has_lb) || ((ai_out || ad_out || tag_flags) && !rs->has_idaq)) {
        return LFQ_ERR_INVALID;
    }
    LFQ_TRY_HIP(hipSetDevice(c->device));
    LFQ_TRY(readset_upload_wait(rs));
    LFQ_TRY(readset_baq_wait(rs));
    if (lb_out) LFQ_TRY_HIP(hipMemcpyAsync(lb_out, rs->d_lb, (size_t)rs->n_bases, hipMemcpyDeviceToHost, c->stream));
    if (ai_out) LFQ_TRY_HIP(hipMemcpyAsync(ai_out, rs->d_ai, (size_t)rs->n_bases, hipMemcpyDeviceToHost, c->stream));
    if (ad_out) LFQ_TRY_HIP(hipMemcpyAsync(ad_out, rs->d_ad, (size_t)rs->n_bases, hipMemcpyDeviceToHost, c->stream));
    LFQ_TRY_HIP(hipStreamSynchronize(c->stream));
    if (tag_flags) {
        for (int64_t r = 0; r < rs->n; r++) {
            tag_flags[r] = (uint8_t)((rs->fl[(size_t)r] >> 2) & 3u);       /* bit 0: ai, bit 1: ad (lfq_baq_idaq_batch) */
        }
    }
    return LFQ_OK;
}

static int readset_from_baq_reads(lfq_ctx *c, const lfq_baq_reads *rd, lfq_readset **rs)
{
    lfq_pileup_reads pr;
    memset(&pr, 0, sizeof(pr));
    pr.n_reads = rd->n_reads;
    pr.pos = rd->pos; pr.cigar_off = rd->cigar_off; pr.cigar = rd->cigar; pr.seq_off = rd->seq_off;
    pr.seq = rd->seq; pr.qual = rd->qual; pr.ref = rd->ref; pr.ref_len = rd->ref_len;
    return lfq_readset_create(c, &pr, nullptr, rs);
}

int lfq_baq_batch(lfq_ctx *c, const lfq_baq_reads *rd, int baq_extended, uint8_t *lb_out)
{
    return lfq_baq_idaq_batch(c, rd, baq_extended, lb_out, nullptr, nullptr, nullptr);
}

int lfq_baq_idaq_batch(lfq_ctx *c, const lfq_baq_reads *rd, int baq_extended, uint8_t *lb_out, uint8_t *ai_out,
                       uint8_t *ad_out, uint8_t *tag_flags)
{
    const bool want_idaq = ai_out && ad_out && tag_flags;
    if (!c || !rd || rd->n_reads < 0 || (rd->n_reads > 0 && (!rd->pos || !rd->cigar_off || !rd->cigar || !rd->seq_off
                                                              || !rd->seq || !rd->qual || !rd->ref || !lb_out))) {
        return LFQ_ERR_INVALID;
    }
    if (rd->n_reads == 0) {
        return LFQ_OK;
    }
    lfq_readset *rs = nullptr;
    LFQ_TRY(readset_from_baq_reads(c, rd, &rs));
    int rc = lfq_readset_baq(c, rs, baq_extended, want_idaq ? 1 : 0);
    if (rc == LFQ_OK) {
        rc = lfq_readset_fetch_tags(c, rs, lb_out, want_idaq ? ai_out : nullptr, want_idaq ? ad_out : nullptr,
                                    want_idaq ? tag_flags : nullptr);
    }
    lfq_readset_destroy(rs);
    return rc;
}

/* bam_prob_realn_core_ext for every read of the set: lb (and ai / ad) stay on the device */
int lfq_readset_baq(lfq_ctx *c, lfq_readset *rs, int baq_extended, int want_idaq_i)
{
    if (!c || !rs || rs->c != c || (rs->n > 0 && !rs->qual)) {
        return LFQ_ERR_INVALID;
    }
    const bool want_idaq = want_idaq_i != 0;
    const lfq_readset *rd = rs;             /* the host views carry the names the geometry code below uses */
    const int64_t n = rs->n;
    if (n == 0) {
        return LFQ_OK;
    }
    LFQ_TRY_HIP(hipSetDevice(c->device));
    LFQ_TRY(readset_baq_wait(rs));
    LFQ_TRY_HIP(hipStreamSynchronize(c->stream));   /* the pinned geometry / order buffers below may still be on their way out */
    double tmb[5] = {lfq_now_ms(), 0, 0, 0, 0};
    /* geometry of every read: alignment window and band width (bam_md_ext.c:312-380, :396-399) */
    /* (pinned, grow-only host buffers: 28 bytes per read are written once by the threads below and go out by DMA; a
     * std::vector would zero 56 MB for 2 M reads first and be copied through a staging buffer afterwards) */
    const int64_t h_bytes = (n * (int64_t)sizeof(LfqBaqRead) + 255) / 256 * 256, ord_bytes = (n * 4 + 255) / 256 * 256;
    if (h_bytes + ord_bytes > c->pin_bytes) {
        if (c->h_pin) (void)hipHostFree(c->h_pin);
        c->h_pin = nullptr;
        c->pin_bytes = 0;
        LFQ_TRY_HIP(hipHostMalloc((void **)&c->h_pin, (size_t)(h_bytes + ord_bytes), hipHostMallocDefault));
        c->pin_bytes = h_bytes + ord_bytes;
    }
    LfqBaqRead *h = (LfqBaqRead *)c->h_pin;
    int32_t *order = (int32_t *)(c->h_pin + h_bytes);
    /* (from the context's pinned pool: fresh memory would be page-faulted in by the threads below while the upload
     * thread is pinning the caller's arrays -- the two fight over the address-space lock) */
    LfqPin<int32_t> width(c, (size_t)n);
    LFQ_PIN_OK(width);
    const bool use_lds = lfq_knobs().baq_lds != 0;
    int max_lq = 0, max_w = 0;
    int part_lq[8] = {0, 0, 0, 0, 0, 0, 0, 0}, part_w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t part_narrow[9] = {0}, part_band8[9] = {0}, part_plain[9] = {0};
    LfqPin<uint8_t> has_id(c, (size_t)n);                            /* the read has an I or D operation (what idaq looks at) */
    LFQ_PIN_OK(has_id);
    const bool reg_kernel = lfq_knobs().baq_kernel == 0;    /* the register kernel also has a band-8 instantiation */
    int part_lrn[8] = {0}, part_lqn[8] = {0};
    int parts = 1;
    lfq_for_reads(n, [&](int64_t r_begin, int64_t r_end, int part) {
    int max_lq = 0, max_w = 0, lrn = 0, lqn = 0;    /* of this part */
    int64_t n_nar = 0, n_b8 = 0, n_pl = 0;
    for (int64_t r = r_begin; r < r_end; r++) {
        LfqBaqRead &o = h[(size_t)r];
        const int l_qseq = (int)(rd->seq_off[r + 1] - rd->seq_off[r]);
        const uint32_t *cg = rd->cigar + rd->cigar_off[r];
        const int n_cigar = (int)(rd->cigar_off[r + 1] - rd->cigar_off[r]);
        int x = rd->pos[r], y = 0, yb = -1, ye = -1, xb = -1, xe = -1;
        bool indel_op = false;
        for (int k = 0; k < n_cigar; ++k) {
            const int op = cg[k] & 0xf, l = cg[k] >> 4;
            indel_op = indel_op || op == 1 || op == 2;
            if (op == 0 || op == 7 || op == 8) {
                if (yb < 0) yb = y;
                if (xb < 0) xb = x;
                ye = y + l; xe = x + l;
                x += l; y += l;
            } else if (op == 4 || op == 1) {
                y += l;
            } else if (op == 2 || op == 3) {
                x += l;
            }
        }
        has_id[(size_t)r] = indel_op ? 1 : 0;
        int bw = 7;
        if (abs((xe - xb) - (ye - yb)) > bw) bw = abs((xe - xb) - (ye - yb)) + 3;
        xb -= yb + bw / 2; if (xb < 0) xb = 0;
        xe += l_qseq - ye + bw / 2;
        if (xe - xb - l_qseq > bw) {
            xb += (xe - xb - l_qseq - bw) / 2, xe -= (xe - xb - l_qseq - bw) / 2;
        }
        if (xe > rd->ref_len) xe = (int)rd->ref_len;      /* the reference stops at the string's NUL */
        o.pos = rd->pos[r];
        o.l_qseq = l_qseq;
        o.xb = xb;
        o.l_ref = xe - xb;
        o.bw = bw;
        o.n_cigar = n_cigar;
        o.cigar_off = rd->cigar_off[r];
        int wr = 0;
        if (l_qseq > 0 && o.l_ref > 0) {
            int b2 = std::max(o.l_ref, l_qseq);
            if (b2 > bw) b2 = bw;
            if (b2 < abs(o.l_ref - l_qseq)) b2 = abs(o.l_ref - l_qseq);
            max_lq = std::max(max_lq, l_qseq);
            max_w = std::max(max_w, (b2 * 2 + 1) * 3 + 6);
            wr = (b2 * 2 + 1) * 3 + 6;
        }
        width[(size_t)r] = wr;
        /* narrow-band reads (rows of at most LFQ_BAQ_LDS_CELLS cells, a short reference window) run in the register kernel */
        if (use_lds && wr <= LFQ_BAQ_LDS_CELLS && o.l_ref <= LFQ_BAQ_LDS_MAX_LREF) {
            n_nar++;
            n_pl += (want_idaq && indel_op) ? 0 : 1;
            lrn = std::max(lrn, o.l_ref);
            lqn = std::max(lqn, l_qseq);
        } else if (use_lds && reg_kernel && wr == LFQ_BAQ_BAND8_CELLS && o.l_ref <= LFQ_BAQ_LDS_MAX_LREF) {
            n_b8++;                                     /* band 8: a deletion of odd length (bam_md_ext.c:353-356) */
            lqn = std::max(lqn, l_qseq);
        }
    }
    part_lq[part] = max_lq;
    part_w[part] = max_w;
    part_narrow[part + 1] = n_nar;
    part_plain[part + 1] = n_pl;
    part_band8[part + 1] = n_b8;
    part_lrn[part] = lrn;
    part_lqn[part] = lqn;
    }, &parts);
    int max_lref_narrow = 0, max_lq_narrow = 0;
    for (int p = 0; p < parts; p++) {
        max_lq = std::max(max_lq, part_lq[p]);
        max_w = std::max(max_w, part_w[p]);
        max_lref_narrow = std::max(max_lref_narrow, part_lrn[p]);
        max_lq_narrow = std::max(max_lq_narrow, part_lqn[p]);
        part_narrow[p + 1] += part_narrow[p];
        part_plain[p + 1] += part_plain[p];
        part_band8[p + 1] += part_band8[p];
    }
    tmb[1] = lfq_now_ms();
    /* launch order: the narrow-band reads first, in input order (neighbouring reads share their reference window in the
     * caches), then the band-8 reads, the others behind them.  Every part of the read range knows where its reads go. */
    /* (with idaq the narrow-band reads without an I / D operation come first: for them the idaq instantiation does
     * nothing the plain one does not do -- ai / ad stay '~', no tag flag -- but runs 17 % longer) */
    const int64_t n_narrow = part_narrow[parts], n_band8 = part_band8[parts], n_plain = part_plain[parts];
    lfq_for_reads(n, [&](int64_t r_begin, int64_t r_end, int part) {
        int64_t pi = part_plain[part], ni = n_plain + (part_narrow[part] - part_plain[part]), bi = n_narrow + part_band8[part];
        int64_t wi = n - 1 - (r_begin - part_narrow[part] - part_band8[part]);    /* wide reads before this part */
        for (int64_t r = r_begin; r < r_end; r++) {
            const bool short_ref = h[(size_t)r].l_ref <= LFQ_BAQ_LDS_MAX_LREF;
            if (use_lds && width[(size_t)r] <= LFQ_BAQ_LDS_CELLS && short_ref) {
                if (want_idaq && has_id[(size_t)r]) {
                    order[(size_t)ni++] = (int32_t)r;
                } else {
                    order[(size_t)pi++] = (int32_t)r;
                }
            } else if (use_lds && reg_kernel && width[(size_t)r] == LFQ_BAQ_BAND8_CELLS && short_ref) {
                order[(size_t)bi++] = (int32_t)r;
            } else {
                order[(size_t)wi--] = (int32_t)r;
            }
        }
    });
    const int64_t n_bases = rs->n_bases;
    if (!rs->tag_blob) {                        /* lb (+ ai, ad): resident from here on */
        const int64_t each = (n_bases + 16 + 255) / 256 * 256;
        rs->tag_blob = (uint8_t *)rs_cache_take(c, LFQ_RSC_TAGS, (size_t)(each * (want_idaq ? 3 : 1)), &rs->cap[LFQ_RSC_TAGS]);
        if (!rs->tag_blob) {
            return LFQ_ERR_NOMEM;
        }
        rs->d_lb = rs->tag_blob;
        rs->d_ai = want_idaq ? rs->tag_blob + each : nullptr;
        rs->d_ad = want_idaq ? rs->tag_blob + 2 * each : nullptr;
        LFQ_TRY_HIP(hipEventCreateWithFlags(&rs->ev_baq, hipEventDisableTiming));
        if (want_idaq) {
            rs->d_tagfl = (uint8_t *)rs_cache_take(c, LFQ_RSC_TAGFL, (size_t)n, &rs->cap[LFQ_RSC_TAGFL]);
            rs->h_fl_pin = (uint8_t *)rs_cache_take(c, LFQ_RSC_PINFL, (size_t)n, &rs->cap[LFQ_RSC_PINFL]);
            if (!rs->d_tagfl || !rs->h_fl_pin) {
                return LFQ_ERR_NOMEM;
            }
        }
    } else if (want_idaq && !rs->d_ai) {
        return LFQ_ERR_INVALID;                 /* a second BAQ pass that suddenly wants ai / ad: make a new read set */
    }
    /* the quality table outlives the call: the copy below is asynchronous and this function returns when it is queued */
    static const float *const h_q2p = [] {
        static float t[256];
        for (int i = 0; i < 256; i++) {
            t[i] = (float)pow(10, -i / 10.);                 /* kprobaln_ext.c:121-123 */
        }
        return (const float *)t;
    }();
    /* per-call device data: geometry, launch order, the quality table (the reads themselves are resident) */
    uint8_t *d_blob = nullptr;
    auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
    const int64_t o_reads = 0, o_q2p = o_reads + al(n * (int64_t)sizeof(LfqBaqRead)), o_ord = o_q2p + al(1024),
                  total = o_ord + al(n * 4);
    LFQ_TRY(grow(&c->d_tmp[0], &c->tmp_bytes[0], total));
    d_blob = c->d_tmp[0];
    int rc = LFQ_OK;
    auto up = [&](int64_t off, const void *src, int64_t bytes) {
        if (rc == LFQ_OK && bytes > 0 && hipMemcpyAsync(d_blob + off, src, (size_t)bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
            rc = LFQ_ERR_HIP;
        }
    };
    up(o_reads, h, n * (int64_t)sizeof(LfqBaqRead));
    up(o_q2p, h_q2p, 1024);
    up(o_ord, order, n * 4);
    if (rc == LFQ_OK && (hipMemsetAsync(rs->d_lb, 0, (size_t)std::max<int64_t>(n_bases, 1), c->stream) != hipSuccess
                         || (want_idaq && (hipMemsetAsync(rs->d_ai, '~', (size_t)n_bases, c->stream) != hipSuccess
                                           || hipMemsetAsync(rs->d_ad, '~', (size_t)n_bases, c->stream) != hipSuccess
                                           || hipMemsetAsync(rs->d_tagfl, 0, (size_t)n, c->stream) != hipSuccess)))) {
        rc = LFQ_ERR_HIP;
    }
    tmb[2] = lfq_now_ms();
    double *d_scr = nullptr;
    int32_t *d_expect = nullptr;
    uint8_t *d_tmp8 = nullptr;
    if (rc == LFQ_OK && max_lq > 0) {
        LfqBaqArgs A;
        memset(&A, 0, sizeof(A));
        A.reads = (const LfqBaqRead *)(d_blob + o_reads);
        A.seq_off = (const int64_t *)rs->d_soff;
        A.cigar = (const uint32_t *)rs->d_cig;
        A.seq = rs->d_seq;
        A.qual = rs->d_qual;
        A.ref = rs->d_ref;
        A.lb_out = rs->d_lb;
        A.qual2prob = (const float *)(d_blob + o_q2p);
        A.n_reads = n;
        A.rows = max_lq + 1;
        A.W = max_w;
        A.baq_extended = baq_extended ? 1 : 0;
        A.par_d = c->baq_par_d;
        A.par_e = c->baq_par_e;
        /* waves per launch from a 4 GiB scratch budget */
        const int64_t per_wave = ((int64_t)A.rows * A.W + 2 * (int64_t)A.W + 2 * ((int64_t)A.rows + 2)) * 64 * 8;
        /* the kernel is a chain of dependent HBM accesses per lane: it needs several wavefronts per SIMD in
         * flight, i.e. scratch for them -- up to half of the free HBM, at most 32 GiB */
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        int64_t budget_b = std::min<int64_t>((int64_t)32 << 30, (int64_t)(free_b / 2));
        if (lfq_knobs().baq_scratch_mb >= 0) {
            budget_b = (int64_t)lfq_knobs().baq_scratch_mb << 20;
        }
        /* + 2: the three classes of reads round up to whole wavefronts separately */
        int64_t waves = std::max<int64_t>(1, std::min<int64_t>((n + 63) / 64 + 2, budget_b / per_wave));
        auto keep = [&](auto **slot, int64_t *have, int64_t need) {
            if (need > *have) {
                if (*slot) (void)hipFree(*slot);
                *slot = nullptr;
                *have = 0;
                if (hipMalloc((void **)slot, (size_t)need) != hipSuccess) {
                    rc = LFQ_ERR_NOMEM;
                    return;
                }
                *have = need;
            }
        };
        keep(&c->d_baq_scr, &c->baq_scr_bytes, waves * per_wave);
        keep(&c->d_baq_expect, &c->baq_expect_bytes, waves * A.rows * 64 * 4);
        keep(&c->d_baq_tmp8, &c->baq_tmp8_bytes, waves * 2 * A.rows * 64);
        if (want_idaq) {
            keep(&c->d_baq_itab, &c->baq_itab_bytes, waves * LFQ_BAQ_MAX_INDELS * 4 * 64 * 4);
            keep(&c->d_baq_terms, &c->baq_terms_bytes, waves * (int64_t)LFQ_BAQ_MAX_TERMS * 64 * 8);
            A.itab = c->d_baq_itab;
            A.terms = c->d_baq_terms;
            A.ai_out = rs->d_ai;
            A.ad_out = rs->d_ad;
            A.tag_flags = rs->d_tagfl;
        }
        d_scr = c->d_baq_scr;
        d_expect = c->d_baq_expect;
        d_tmp8 = c->d_baq_tmp8;
        A.scratch = d_scr;
        A.expect = d_expect;
        A.tmp8 = d_tmp8;
        A.order = (const int32_t *)(d_blob + o_ord);
        A.max_lref = max_lref_narrow;
        A.lds_rows = max_lq_narrow + 1;
        /* The reads with a wider band are few: band 8 (a deletion of odd length; the register kernel's second
         * instantiation) and everything beyond (the all-HBM kernel, a handful of latency-bound wavefronts).  They run
         * beside the narrow-band launches on the side streams, in scratch slots of their own behind the narrow ones'
         * (wavefront w of a launch owns slot w).  The narrow-band kernel runs one wavefront per SIMD, so a launch is cut
         * to a whole number of rounds over the SIMDs: a launch of 7.3 rounds takes as long as one of 8. */
        const int64_t n_wide = n - n_narrow - n_band8;
        const int64_t waves_wide = (n_wide + 63) / 64, waves_b8 = (n_band8 + 63) / 64;
        const bool beside = n_narrow > 0 && waves_wide + waves_b8 > 0 && waves_wide + waves_b8 < waves / 4
                            && c->side[0] != nullptr && c->side[1] != nullptr && !lfq_knobs().single_stream;
        int64_t waves_n = beside ? waves - waves_wide - waves_b8 : waves;      /* slots of a narrow launch */
        const int64_t round = (int64_t)c->n_cu * 4;
        if ((n_narrow + 63) / 64 > waves_n && waves_n > round) {       /* more than one launch: whole rounds each */
            waves_n = waves_n / round * round;
        }
        auto at_slot = [&](int64_t slot) {          /* the arguments with the scratch of wavefront slot `slot` first */
            LfqBaqArgs X = A;
            X.scratch = A.scratch + (size_t)slot * (size_t)(per_wave / 8);
            X.expect = A.expect + (size_t)slot * A.rows * 64;
            X.tmp8 = A.tmp8 + (size_t)slot * 2 * A.rows * 64;
            if (want_idaq) {
                X.itab = A.itab + (size_t)slot * LFQ_BAQ_MAX_INDELS * 4 * 64;
                X.terms = A.terms + (size_t)slot * LFQ_BAQ_MAX_TERMS * 64;
            }
            return X;
        };
        /* The reads may still be crossing PCIe (lfq_readset_create): a launch waits for the chunks of bases and qualities
         * that hold its reads -- the plain narrow-band launches walk the reads in input order, so the first one starts
         * after a quarter of them --, everything else for all of them. */
        if (beside && hipEventRecord(c->ev_join[0], c->stream) != hipSuccess) {
            rc = LFQ_ERR_HIP;       /* the side streams start after the uploads / memsets queued on c->stream so far */
        }
        {
            LfqBaqArgs Ap = A;                      /* the plain instantiation: no indel table */
            Ap.itab = nullptr;
            Ap.terms = nullptr;
            Ap.ai_out = Ap.ad_out = nullptr;
            Ap.tag_flags = nullptr;
            for (int64_t first = 0; rc == LFQ_OK && first < n_plain; first += waves_n * 64) {
                const int64_t cnt = std::min<int64_t>(waves_n * 64, n_plain - first);
                rc = readset_upload_wait_reads(rs, order[(size_t)(first + cnt - 1)]);
                Ap.first_read = (int32_t)first;
                if (rc == LFQ_OK) {
                    rc = lfq_launch_baq(Ap, cnt, 1, c->stream);
                }
            }
        }
        if (rc == LFQ_OK) {
            rc = readset_upload_wait_inputs(rs);
        }
        if (beside) {
            /* wide-band and band-8 reads on the side streams, beside the narrow-band launches; c->stream ends after them */
            if (rc == LFQ_OK && n_wide > 0) {
                LfqBaqArgs Aw = at_slot(waves_n + waves_b8);
                Aw.first_read = (int32_t)(n_narrow + n_band8);
                if (hipStreamWaitEvent(c->side[0], c->ev_join[0], 0) != hipSuccess) {
                    rc = LFQ_ERR_HIP;
                }
                if (rc == LFQ_OK) {
                    rc = lfq_launch_baq(Aw, n_wide, 0, c->side[0]);
                }
            }
            if (rc == LFQ_OK && n_band8 > 0) {
                LfqBaqArgs Ab = at_slot(waves_n);
                Ab.first_read = (int32_t)n_narrow;
                if (hipStreamWaitEvent(c->side[1], c->ev_join[0], 0) != hipSuccess) {
                    rc = LFQ_ERR_HIP;
                }
                if (rc == LFQ_OK) {
                    rc = lfq_launch_baq(Ab, n_band8, 2, c->side[1]);
                }
            }
            if (rc == LFQ_OK && (hipEventRecord(c->ev_join[1], c->side[0]) != hipSuccess
                                 || hipEventRecord(c->ev_join[2], c->side[1]) != hipSuccess)) {
                rc = LFQ_ERR_HIP;
            }
        }
        for (int64_t first = n_plain; rc == LFQ_OK && first < n_narrow; first += waves_n * 64) {
            A.first_read = (int32_t)first;
            rc = lfq_launch_baq(A, std::min<int64_t>(waves_n * 64, n_narrow - first), 1, c->stream);
        }
        if (beside) {
            if (rc == LFQ_OK && (hipStreamWaitEvent(c->stream, c->ev_join[1], 0) != hipSuccess
                                 || hipStreamWaitEvent(c->stream, c->ev_join[2], 0) != hipSuccess)) {
                rc = LFQ_ERR_HIP;
            }
        } else {
            for (int64_t first = n_narrow; rc == LFQ_OK && first < n_narrow + n_band8; first += waves * 64) {
                A.first_read = (int32_t)first;
                rc = lfq_launch_baq(A, std::min<int64_t>(waves * 64, n_narrow + n_band8 - first), 2, c->stream);
            }
            for (int64_t first = n_narrow + n_band8; rc == LFQ_OK && first < n; first += waves * 64) {
                A.first_read = (int32_t)first;
                rc = lfq_launch_baq(A, std::min<int64_t>(waves * 64, n - first), 0, c->stream);
            }
        }
    }
    tmb[3] = lfq_now_ms();
    /* which reads got an ai / ad tag (bam_md_ext.c:238-243) joins the resident flags as bits 2, 3 on the device; the
     * merged byte travels to pinned memory for the host's event tables.  Nothing here waits for the kernels. */
    if (rc == LFQ_OK && want_idaq) {
        rc = lfq_launch_flag_merge(rs->d_fl, rs->d_tagfl, n, c->stream);
        if (rc == LFQ_OK && hipMemcpyAsync(rs->h_fl_pin, rs->d_fl, (size_t)n, hipMemcpyDeviceToHost, c->stream) != hipSuccess) {
            rc = LFQ_ERR_HIP;
        }
    }
    if (rc == LFQ_OK && hipEventRecord(rs->ev_baq, c->stream) != hipSuccess) {
        rc = LFQ_ERR_HIP;
    }
    if (rc != LFQ_OK) {
        (void)hipStreamSynchronize(c->stream);
        return rc;
    }
    rs->baq_pending = true;
    rs->baq_idaq = want_idaq;
    tmb[4] = lfq_now_ms();
    if (lfq_timing_on) {
        (void)hipStreamSynchronize(c->stream);
        fprintf(stderr, "[lfq timing] baq: geometry %.1f  order + allocations + uploads %.1f  scratch + launches %.1f  kernels (sync, timing only) %.1f ms\n",
                tmb[1] - tmb[0], tmb[2] - tmb[1], tmb[3] - tmb[2], lfq_now_ms() - tmb[3]);
    }
    rs->has_lb = true;
    if (want_idaq) {
        rs->has_idaq = true;
        rs->h_ai = rs->h_ad = nullptr;          /* superseded by the device result */
    }
    return rc;
}

int lfq_pileup_snv_tracks(lfq_ctx *c, const lfq_pileup_reads *rd, int64_t region_begin, int64_t region_end,
                          int min_plp_bq, lfq_tracks *out, int64_t *col_pos_out)
{
    if (!c || !rd || !out || region_end < region_begin || rd->n_reads < 0
        || (rd->n_reads > 0 && (!rd->pos || !rd->cigar_off || !rd->cigar || !rd->seq_off || !rd->seq || !rd->qual
                                || !rd->mapq || !rd->reverse || !rd->ref))) {
        return LFQ_ERR_INVALID;
    }
    lfq_readset *rs = nullptr;
    LFQ_TRY(lfq_readset_create(c, rd, nullptr, &rs));
    const int rc = lfq_readset_pileup_snv(c, rs, region_begin, region_end, min_plp_bq, out, col_pos_out);
    lfq_readset_destroy(rs);            /* the tracks live in the context, not in the read set; waits for the scatter pass */
    return rc;
}

int lfq_readset_pileup_snv(lfq_ctx *c, lfq_readset *rs, int64_t region_begin, int64_t region_end, int min_plp_bq,
                           lfq_tracks *out, int64_t *col_pos_out)
{
    if (!c || !rs || rs->c != c || !out || region_end < region_begin
        || (rs->n > 0 && (!rs->qual || !rs->mapq || !rs->reverse))) {
        return LFQ_ERR_INVALID;
    }
    const lfq_readset *rd = rs;
    memset(out, 0, sizeof(*out));
    const int64_t n = rs->n, width = region_end - region_begin;
    if (n == 0 || width == 0) {
        return LFQ_OK;
    }
    LFQ_TRY_HIP(hipSetDevice(c->device));
    LFQ_TRY(readset_upload_wait(rs));
    auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
    /* per-position counters (kept until the next call) */
    const int64_t o_cov = 0, o_nb = o_cov + al(width * 4), o_cur = o_nb + al(width * 4), o_cidx = o_cur + al(width * 4),
                  total = o_cidx + al(width * 4);
    /* Everything goes to the main stream, behind the BAQ kernels if they are still running (the scatter pass reads their
     * lb bytes; pass 0 beside them was measured: 2 ms alone, 14 ms squeezed between wavefronts that hold 416 of a SIMD's
     * 512 registers, with the host waiting for its result).  Nothing waits for the scatter pass: the tracks are complete
     * in stream order (see the header) -- the host goes on with the indel tests while it runs. */
    hipStream_t ps = c->stream;
    LFQ_TRY(order_after_batch(c, c->stream));          /* the tracks of the previous call may still be a running batch's input */
    LFQ_TRY(grow(&c->d_plp_in, &c->plp_in_bytes, total));
    uint8_t *d = c->d_plp_in;
    LFQ_TRY_HIP(hipMemsetAsync(d + o_cov, 0, (size_t)(o_cidx - o_cov), ps));
    LfqPileupArgs A;
    memset(&A, 0, sizeof(A));
    A.n_reads = n;
    A.pos = (const int32_t *)rs->d_pos;
    A.cigar_off = (const int64_t *)rs->d_coff;
    A.seq_off = (const int64_t *)rs->d_soff;
    A.cigar = (const uint32_t *)rs->d_cig;
    A.seq = rs->d_seq;
    A.qual = rs->d_qual;
    A.baq = rs->has_lb ? rs->d_lb : nullptr;
    A.mapq = rs->d_mapq;
    A.reverse = rs->d_rev;
    A.sq = rs->has_sqb ? rs->d_sqb : nullptr;
    A.begin = region_begin;
    A.width = width;
    A.min_plp_bq = min_plp_bq;
    A.cov = (int32_t *)(d + o_cov);
    A.nb = (int32_t *)(d + o_nb);
    A.cursor = (int32_t *)(d + o_cur);
    /* position-sorted reads (the normal case): the column-major kernels; otherwise one thread per read + atomics */
    A.pmax_end = readset_pmax(c, rs, ps);
    const bool sorted = A.pmax_end != nullptr;
    LFQ_TRY(sorted ? lfq_launch_pileup_columns(A, 0, ps) : lfq_launch_pileup_count(A, ps));
    /* prefix sums on the host: 8 bytes per reference position of the region, once per region */
    LfqPin<int32_t> cov(c, (size_t)width), nb(c, (size_t)width), cidx(c, (size_t)width, -1);
    LFQ_PIN_OK(cov);
    LFQ_PIN_OK(nb);
    LFQ_PIN_OK(cidx);
    LFQ_TRY_HIP(hipMemcpyAsync(cov.data(), A.cov, (size_t)width * 4, hipMemcpyDeviceToHost, ps));
    LFQ_TRY_HIP(hipMemcpyAsync(nb.data(), A.nb, (size_t)width * 4, hipMemcpyDeviceToHost, ps));
    LFQ_TRY_HIP(hipStreamSynchronize(ps));
    /* two passes over the positions, both split over a few threads: covered positions and bases per part, then every
     * part fills its slice */
    int64_t part_cols[9] = {0}, part_obs[9] = {0}, part_max[8] = {0};
    int parts = 1;
    lfq_for_reads(width, [&](int64_t p0, int64_t p1, int part) {
        int64_t nc = 0, no = 0, mx = 0;
        for (int64_t p = p0; p < p1; p++) {
            if (cov[(size_t)p] > 0) {
                nc++;
                no += nb[(size_t)p];
                mx = std::max<int64_t>(mx, nb[(size_t)p]);
            }
        }
        part_cols[part + 1] = nc;
        part_obs[part + 1] = no;
        part_max[part] = mx;
    }, &parts);
    int64_t max_obs = 0;
    for (int q = 0; q < parts; q++) {
        part_cols[q + 1] += part_cols[q];
        part_obs[q + 1] += part_obs[q];
        max_obs = std::max(max_obs, part_max[q]);
    }
    LfqPin<uint64_t> off(c, (size_t)part_cols[parts] + 1);
    LfqPin<int32_t> h_cov(c, (size_t)part_cols[parts]), h_nb(c, (size_t)part_cols[parts]);
    LfqPin<uint8_t> h_ref(c, (size_t)part_cols[parts]);
    LFQ_PIN_OK(off);
    LFQ_PIN_OK(h_cov);
    LFQ_PIN_OK(h_nb);
    LFQ_PIN_OK(h_ref);
    off[0] = 0;
    lfq_for_reads(width, [&](int64_t p0, int64_t p1, int part) {
        size_t ci = (size_t)part_cols[part];
        uint64_t run = (uint64_t)part_obs[part];
        for (int64_t p = p0; p < p1; p++) {
            if (cov[(size_t)p] <= 0) {
                continue;
            }
            cidx[(size_t)p] = (int32_t)ci;
            if (col_pos_out) {
                col_pos_out[ci] = region_begin + p;
            }
            h_cov[ci] = cov[(size_t)p];
            h_nb[ci] = nb[(size_t)p];
            const int64_t gp = region_begin + p;
            char rb = (gp < rd->ref_len) ? rd->ref[gp] : 'N';           /* plp.c:818-823 */
            if (!(rb == 'A' || rb == 'C' || rb == 'T' || rb == 'G' || rb == 'N')) {
                rb = 'N';
            }
            h_ref[ci] = (uint8_t)rb;
            run += (uint64_t)nb[(size_t)p];
            off[ci + 1] = run;
            ci++;
        }
    });
    const int64_t ncols = (int64_t)h_cov.size();
    const int64_t n_obs = (int64_t)off.back(), trk = al(n_obs + 32);
    const bool nt_packed = !c->plp_nt_bytes;
    const int64_t t_off = 0, t_ref = t_off + al((ncols + 1) * 8), t_cov = t_ref + al(ncols + 16), t_nb = t_cov + al(ncols * 4 + 16),
                  t_nt = t_nb + al(ncols * 4 + 16), t_bq = t_nt + trk, t_baq = t_bq + trk, t_mq = t_baq + trk,
                  t_sq = t_mq + trk, t_ntp = t_sq + (rs->has_sqb ? trk : 0), t_total = t_ntp + (nt_packed ? al(trk / 2 + 16) : 0);
    LFQ_TRY(grow(&c->d_plp_out, &c->plp_out_bytes, t_total));
    uint8_t *t = c->d_plp_out;
    LFQ_TRY_HIP(hipMemsetAsync(t + t_nt, 0, (size_t)(t_total - t_nt), ps));            /* the 16-byte tails are read */
    LFQ_TRY_HIP(hipMemcpyAsync(t + t_off, off.data(), (size_t)(ncols + 1) * 8, hipMemcpyHostToDevice, ps));
    if (ncols > 0) {
        LFQ_TRY_HIP(hipMemcpyAsync(t + t_ref, h_ref.data(), (size_t)ncols, hipMemcpyHostToDevice, ps));
        LFQ_TRY_HIP(hipMemcpyAsync(t + t_cov, h_cov.data(), (size_t)ncols * 4, hipMemcpyHostToDevice, ps));
        LFQ_TRY_HIP(hipMemcpyAsync(t + t_nb, h_nb.data(), (size_t)ncols * 4, hipMemcpyHostToDevice, ps));
    }
    LFQ_TRY_HIP(hipMemcpyAsync(d + o_cidx, cidx.data(), (size_t)width * 4, hipMemcpyHostToDevice, ps));
    /* the pinned blocks these copies read go back to the pool when this function returns: they are waited for here (a
     * few megabytes on a stream that carries nothing else); the scatter pass below touches no host memory */
    LFQ_TRY_HIP(hipStreamSynchronize(ps));
    A.col_index = (const int32_t *)(d + o_cidx);
    A.col_off = (const uint64_t *)(t + t_off);
    A.t_nt = t + t_nt;
    A.t_bq = t + t_bq;
    A.t_baq = t + t_baq;
    A.t_mq = t + t_mq;
    A.t_sq = rs->has_sqb ? t + t_sq : nullptr;
    LFQ_TRY(sorted ? lfq_launch_pileup_columns(A, 1, c->stream) : lfq_launch_pileup_scatter(A, c->stream));
    if (nt_packed) {
        /* the layout the count kernel reads 1.5 instead of 2 bytes per observation of (LFQ_TRACKS_NT_PACKED): the scatter
         * pass writes bytes (two lanes, often of two wavefronts, would share a byte), one streaming pass packs them */
        LFQ_TRY(lfq_launch_pack_nt(t + t_nt, t + t_ntp, n_obs, c->stream));
    }
    /* (no wait: what consumes the tracks -- lfq_call_snvs_batch, lfq_pileup_skip_snv_columns, the uniq calls -- is
     * queued on the same stream; lfq_readset_destroy and lfq_synchronize wait for it) */
    out->nt = nt_packed ? t + t_ntp : t + t_nt;
    out->flags = nt_packed ? LFQ_TRACKS_NT_PACKED : 0;
    out->bq = t + t_bq;
    out->baq = t + t_baq;
    out->mq = t + t_mq;
    out->sq = rs->has_sqb ? t + t_sq : nullptr;
    out->col_off = (const uint64_t *)(t + t_off);
    out->ref_base = t + t_ref;
    out->coverage_plp = (const int32_t *)(t + t_cov);
    out->num_bases = (const int32_t *)(t + t_nb);
    c->d_plp_nb = (int32_t *)(t + t_nb);
    c->plp_ncols = ncols;
    out->ncols = ncols;
    out->max_col_obs = max_obs;
    return LFQ_OK;
}

/* compile_plp_col's indel fields for the reads of a region (plp.c:1019-1192): the sparse part (which read carries
 * which insertion / deletion where: straight from the CIGARs) is assembled here, the dense part (counts over all
 * pileup entries and the quality arrays of the reads WITHOUT an event at the event columns) by lfq_plp_indel_kernel */
int lfq_pileup_indel_columns(lfq_ctx *c, const lfq_pileup_reads *rd, const lfq_pileup_indel_tags *tg,
                             int64_t region_begin, int64_t region_end, int min_plp_idq,
                             const lfq_indel_columns **cols_out, int64_t *col_pos_out)
{
    if (!c || !rd || !cols_out || region_end < region_begin || rd->n_reads < 0
        || (rd->n_reads > 0 && (!rd->pos || !rd->cigar_off || !rd->cigar || !rd->seq_off || !rd->seq || !rd->mapq
                                || !rd->reverse || !rd->ref))) {
        return LFQ_ERR_INVALID;
    }
    lfq_readset *rs = nullptr;
    LFQ_TRY(lfq_readset_create(c, rd, tg, &rs));
    const int rc = lfq_readset_pileup_indels(c, rs, region_begin, region_end, min_plp_idq, cols_out, col_pos_out);
    lfq_readset_destroy(rs);            /* the columns live in the context */
    return rc;
}

int lfq_readset_pileup_indels(lfq_ctx *c, lfq_readset *rs, int64_t region_begin, int64_t region_end, int min_plp_idq,
                              const lfq_indel_columns **cols_out, int64_t *col_pos_out)
{
    if (!c || !rs || rs->c != c || !cols_out || region_end < region_begin || (rs->n > 0 && (!rs->mapq || !rs->reverse))) {
        return LFQ_ERR_INVALID;
    }
    const lfq_readset *rd = rs;
    if (c->plp_indel) {
        c->plp_indel->reset();              /* keeps the capacity: see LfqPin for what freeing a DMA target costs */
    } else {
        c->plp_indel = new LfqIndelColsOwned();
    }
    c->plp_ne_total[0] = c->plp_ne_total[1] = 0;
    LfqIndelColsOwned &O = *c->plp_indel;
    memset(&O.cols, 0, sizeof(O.cols));
    *cols_out = &O.cols;
    const int64_t n = rs->n, width = region_end - region_begin;
    /* tag bytes on the host where the caller gave them; ai / ad computed by lfq_readset_baq are fetched per event */
    const uint8_t *t_bi = rs->h_bi, *t_bd = rs->h_bd, *t_ai = rs->h_ai, *t_ad = rs->h_ad, *t_fl = rs->fl.data();
    const int32_t *t_sq = rs->h_sq ? rs->h_sq : (rs->sq32.empty() ? nullptr : rs->sq32.data());

    double tm[8] = {lfq_now_ms(), 0, 0, 0, 0, 0, 0, 0};
    /* 1. events from the CIGARs, in read (= pileup) order */
    struct Ev { int64_t pos; int64_t read; int32_t qpos, indel; };
    std::vector<Ev> evs;
    std::vector<Ev> evs_part[8];                    /* per thread, concatenated in read order below */
    /* (host arrays only: runs while the counter kernel of step 2 does) */
    auto scan_events = [&]() {
    lfq_for_reads(n, [&](int64_t r_begin, int64_t r_end, int part) {
    std::vector<Ev> &evs = evs_part[part];
    for (int64_t r = r_begin; r < r_end; r++) {
        const uint32_t *cg = rd->cigar + rd->cigar_off[r];
        const int n_cigar = (int)(rd->cigar_off[r + 1] - rd->cigar_off[r]);
        const int64_t s0 = rd->seq_off[r];
        const int l_qseq = (int)(rd->seq_off[r + 1] - s0);
        const uint32_t fl = t_fl[r];
        int64_t x = rd->pos[r];
        int y = 0;
        for (int k = 0; k < n_cigar; ++k) {
            const int op = cg[k] & 0xf, l = cg[k] >> 4;
            if (op == 0 || op == 7 || op == 8 || op == 2 || op == 3) {
                const bool is_del = op == 2 || op == 3;
                int indel = 0;                                      /* htslib resolve_cigar2: peek at the next operation */
                if (l > 0 && k + 1 < n_cigar) {
                    const int op2 = cg[k + 1] & 0xf, l2 = cg[k + 1] >> 4;
                    if (op2 == 2) {
                        indel = -l2;
                    } else if (op2 == 1) {
                        indel = l2;
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
                        indel = l3 > 0 ? l3 : 0;
                    }
                }
                const int64_t p = x + l - 1;
                if (indel != 0 && p >= region_begin && p < region_end) {
                    int qpos = is_del ? y : y + l - 1;
                    qpos = qpos < l_qseq ? qpos : l_qseq - 1;
                    const int iq = (t_bi && (fl & 1u) && qpos >= 0) ? (int)t_bi[s0 + qpos] - 33 : 0;
                    const int dq = (t_bd && (fl & 2u) && qpos >= 0) ? (int)t_bd[s0 + qpos] - 33 : 0;
                    if (!(iq < min_plp_idq || dq < min_plp_idq)) {      /* plp.c:1062 */
                        evs.push_back({p, r, qpos, indel});
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
    });
    for (int p = 0; p < 8; p++) {
        evs.insert(evs.end(), evs_part[p].begin(), evs_part[p].end());
    }
    std::stable_sort(evs.begin(), evs.end(), [](const Ev &a, const Ev &b) { return a.pos < b.pos; });
    };
    const uint8_t *g_ai = nullptr, *g_ad = nullptr;     /* per event, when the qualities come from the device */
    std::vector<int32_t> qsum[2];           /* per column: quality sum of the reads without an event, from the kernel */
    /* 5. consensus indel (plp.c:1236-1270): the largest sum of qualities of one event against the sum over the
     * reads without an event of that side */
    auto consensus = [&]() {
    O.cons_indel.assign(O.cov.size(), 0);
    for (int64_t col = 0; col < (int64_t)O.cov.size(); col++) {
        for (int sd = 0; sd < 2; sd++) {
            const LfqIndelColsOwned::Side &S = O.side[sd];
            if (S.ev_off[(size_t)col] == S.ev_off[(size_t)col + 1]) {
                continue;                               /* no event of this side: nothing can exceed the non-event sum */
            }
            int64_t best = 0, non = 0;
            for (int64_t e = S.ev_off[(size_t)col]; e < S.ev_off[(size_t)col + 1]; e++) {
                int64_t sum = 0;
                for (int64_t i = S.rd_off[(size_t)e]; i < S.rd_off[(size_t)e + 1]; i++) {
                    sum += S.rd_q[(size_t)i];
                }
                best = std::max(best, sum);
            }
            if (!qsum[sd].empty()) {
                non = qsum[sd][(size_t)col];
            } else {
                for (int64_t i = S.ne_off[(size_t)col]; i < S.ne_off[(size_t)col + 1]; i++) {
                    non += S.ne_q[(size_t)i];
                }
            }
            if (best > non) {
                O.cons_indel[(size_t)col] = 1;
            }
        }
    }
    };

    if (n == 0 || width == 0) {
        for (int sd = 0; sd < 2; sd++) {
            O.side[sd].ne_off.assign(1, 0);
            O.side[sd].ev_off.assign(1, 0);
            O.side[sd].key_off.assign(1, 0);
            O.side[sd].rd_off.assign(1, 0);
        }
        consensus();
    } else {
        /* 2. dense counters on the device */
        LFQ_TRY_HIP(hipSetDevice(c->device));
        LFQ_TRY(readset_upload_wait(rs));
        /* Steps 2 and 3 read nothing lfq_readset_baq writes (of the flag bytes only the BI / BD bits, which its merge
         * kernel leaves as they are): while its kernels are still running they go to another stream and run beside them --
         * the BAQ kernels hold one wavefront per SIMD and 416 of its 512 registers, these kernels need 40. */
        hipStream_t ps = (rs->baq_pending && c->dps && !lfq_knobs().single_stream) ? c->dps : c->stream;
        auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
        const int64_t o_cnt = 0, o_cur = o_cnt + 9 * al(width * 4), o_off = o_cur + 2 * al(width * 4),
                      total = o_off + 2 * al(width * 8);
        LFQ_TRY(grow(&c->d_tmp[1], &c->tmp_bytes[1], total));
        uint8_t *d = c->d_tmp[1];
        int16_t *d_ne = nullptr;
        int rc = LFQ_OK;
        auto up = [&](int64_t off, const void *src, int64_t bytes) {
            if (rc == LFQ_OK && src && bytes > 0
                && hipMemcpyAsync(d + off, src, (size_t)bytes, hipMemcpyHostToDevice, ps) != hipSuccess) {
                rc = LFQ_ERR_HIP;
            }
        };
        if (rc == LFQ_OK && hipMemsetAsync(d + o_cnt, 0, (size_t)(o_off - o_cnt), ps) != hipSuccess) {
            rc = LFQ_ERR_HIP;
        }
        LfqPlpIndelArgs A;
        memset(&A, 0, sizeof(A));
        A.n_reads = n;
        A.pos = (const int32_t *)rs->d_pos;
        A.cigar_off = (const int64_t *)rs->d_coff;
        A.seq_off = (const int64_t *)rs->d_soff;
        A.cigar = (const uint32_t *)rs->d_cig;
        A.bi = rs->has_bi ? rs->d_bi : nullptr;
        A.bd = rs->has_bd ? rs->d_bd : nullptr;
        A.tag_flags = rs->d_fl;
        A.mapq = rs->d_mapq;
        A.reverse = rs->d_rev;
        A.begin = region_begin;
        A.width = width;
        A.min_plp_idq = min_plp_idq;
        int32_t **cnt[9] = {&A.cov, &A.tails, &A.non_indels, &A.n_ins, &A.n_dels, &A.non_ins_fw, &A.non_del_fw,
                            &A.ne_qsum[0], &A.ne_qsum[1]};
        for (int i = 0; i < 9; i++) {
            *cnt[i] = (int32_t *)(d + o_cnt + i * al(width * 4));
        }
        /* the nine per-position counters come back into pinned memory (grow-only): DMA instead of a staged copy */
        int32_t *h[9] = {nullptr};
        {
            const int64_t need = 9 * al(width * 4);
            if (need > c->pin2_bytes) {
                if (c->h_pin2) (void)hipHostFree(c->h_pin2);
                c->h_pin2 = nullptr;
                c->pin2_bytes = 0;
                if (hipHostMalloc((void **)&c->h_pin2, (size_t)need, hipHostMallocDefault) != hipSuccess) {
                    return LFQ_ERR_NOMEM;
                }
                c->pin2_bytes = need;
            }
            for (int i = 0; i < 9; i++) {
                h[i] = (int32_t *)(c->h_pin2 + i * al(width * 4));
            }
        }
        if (rc == LFQ_OK) {
            A.pmax_end = readset_pmax(c, rs, ps);
            rc = A.pmax_end ? lfq_launch_plp_indel_columns(A, 0, ps) : lfq_launch_plp_indel(A, 0, ps);
        }
        const bool have_qsum = A.pmax_end != nullptr;       /* the column-major kernel sums the qualities itself */
        for (int i = 0; i < (have_qsum ? 9 : 7) && rc == LFQ_OK; i++) {
            if (hipMemcpyAsync(h[i], *cnt[i], (size_t)width * 4, hipMemcpyDeviceToHost, ps) != hipSuccess) {
                rc = LFQ_ERR_HIP;
            }
        }
        scan_events();
        tm[1] = lfq_now_ms();
        if (rc == LFQ_OK && hipStreamSynchronize(ps) != hipSuccess) {
            rc = LFQ_ERR_HIP;
        }
        tm[2] = lfq_now_ms();
        /* 3. columns = covered positions; quality arrays of the reads without an event at the event positions */
        LfqPin<int64_t> pos_off_ins(c, (size_t)width), pos_off_del(c, (size_t)width);      /* DMA sources: pinned */
        if (!pos_off_ins.ok() || !pos_off_del.ok()) {
            rc = LFQ_ERR_NOMEM;
        }
        int64_t *const pos_off[2] = {pos_off_ins.data(), pos_off_del.data()};
        std::vector<int32_t> col_of;                /* column index of an event position */
        int64_t ne_total[2] = {0, 0};
        const bool host_arrays = c->indel_host_arrays || !have_qsum;   /* device-only needs the sums from the kernel */
        if (rc == LFQ_OK) {
            std::fill(pos_off[0], pos_off[0] + width, (int64_t)-1);
            std::fill(pos_off[1], pos_off[1] + width, (int64_t)-1);
            col_of.assign((size_t)width, -1);
            std::vector<uint8_t> has_ev((size_t)width, 0);
            for (const Ev &e : evs) {
                has_ev[(size_t)(e.pos - region_begin)] = 1;
            }
            /* two passes over the positions, both split over a few threads: count the covered positions and the
             * non-event reads at event positions per part, then every part fills its slice of the column arrays */
            int64_t part_cov[9] = {0}, part_ne[2][9] = {{0}, {0}};
            int parts = 1;
            lfq_for_reads(width, [&](int64_t p0, int64_t p1, int part) {
                int64_t nc = 0, ne0 = 0, ne1 = 0;
                for (int64_t p = p0; p < p1; p++) {
                    if (h[0][(size_t)p] <= 0) {
                        continue;
                    }
                    nc++;
                    if (has_ev[(size_t)p]) {
                        ne0 += h[2][(size_t)p] + h[4][(size_t)p];
                        ne1 += h[2][(size_t)p] + h[3][(size_t)p];
                    }
                }
                part_cov[part + 1] = nc;
                part_ne[0][part + 1] = ne0;
                part_ne[1][part + 1] = ne1;
            }, &parts);
            for (int q = 0; q < parts; q++) {
                part_cov[q + 1] += part_cov[q];
                part_ne[0][q + 1] += part_ne[0][q];
                part_ne[1][q + 1] += part_ne[1][q];
            }
            const size_t n_cov = (size_t)part_cov[parts];
            ne_total[0] = part_ne[0][parts];
            ne_total[1] = part_ne[1][parts];
            for (auto *v : {&O.cov, &O.tails, &O.non_indels, &O.n_ins, &O.n_dels, &O.hrun}) {
                v->resize(n_cov);
            }
            O.ref_base.resize(n_cov);
            if (have_qsum) {
                qsum[0].resize(n_cov);
                qsum[1].resize(n_cov);
            }
            for (int sd = 0; sd < 2; sd++) {
                O.side[sd].non_fw.resize(n_cov);
                O.side[sd].non_rv.resize(n_cov);
                O.side[sd].ne_off.resize(n_cov + 1);
                O.side[sd].ev_off.reserve(n_cov + 1);
                O.side[sd].ne_off[0] = 0;
            }
            lfq_for_reads(width, [&](int64_t p0, int64_t p1, int part) {
                size_t ci = (size_t)part_cov[part];
                int64_t run[2] = {part_ne[0][part], part_ne[1][part]};
                for (int64_t p = p0; p < p1; p++) {
                    if (h[0][(size_t)p] <= 0) {
                        continue;
                    }
                    const int64_t gp = region_begin + p;
                    if (col_pos_out) {
                        col_pos_out[ci] = gp;
                    }
                    char rb = (gp < rd->ref_len) ? rd->ref[gp] : 'N';       /* plp.c:818-823 */
                    if (!(rb == 'A' || rb == 'C' || rb == 'T' || rb == 'G' || rb == 'N')) {
                        rb = 'N';
                    }
                    O.ref_base[ci] = (uint8_t)rb;
                    O.cov[ci] = h[0][(size_t)p];
                    O.tails[ci] = h[1][(size_t)p];
                    O.non_indels[ci] = h[2][(size_t)p];
                    O.n_ins[ci] = h[3][(size_t)p];
                    O.n_dels[ci] = h[4][(size_t)p];
                    int hr = 1;                                             /* get_hrun, plp.c:744-787 */
                    if (gp + 1 < rd->ref_len) {
                        const int ch = toupper((unsigned char)rd->ref[gp + 1]);
                        for (int64_t i = gp + 2; i < rd->ref_len && toupper((unsigned char)rd->ref[i]) == ch; i++) {
                            hr++;
                        }
                        for (int64_t i = gp; i >= 0 && toupper((unsigned char)rd->ref[i]) == ch; i--) {
                            hr++;
                        }
                    }
                    O.hrun[ci] = hr;
                    if (have_qsum) {
                        qsum[0][ci] = h[7][(size_t)p];
                        qsum[1][ci] = h[8][(size_t)p];
                    }
                    const int32_t ne_cnt[2] = {h[2][(size_t)p] + h[4][(size_t)p], h[2][(size_t)p] + h[3][(size_t)p]};
                    const int32_t fw[2] = {h[5][(size_t)p], h[6][(size_t)p]};
                    for (int sd = 0; sd < 2; sd++) {
                        O.side[sd].non_fw[ci] = fw[sd];
                        O.side[sd].non_rv[ci] = ne_cnt[sd] - fw[sd];
                        if (has_ev[(size_t)p]) {
                            pos_off[sd][(size_t)p] = run[sd];
                            run[sd] += ne_cnt[sd];
                            col_of[(size_t)p] = (int32_t)ci;
                        }
                        O.side[sd].ne_off[ci + 1] = run[sd];
                    }
                    ci++;
                }
            });
            const int64_t ne_all = ne_total[0] + ne_total[1];
            if (ne_all > 0 && grow(&c->d_plp_ne, &c->plp_ne_cap, ne_all * 2) != LFQ_OK) {
                rc = LFQ_ERR_NOMEM;
            }
            d_ne = c->d_plp_ne;
            if (rc == LFQ_OK && ne_all > 0) {
                for (int sd = 0; sd < 2; sd++) {
                    up(o_off + sd * al(width * 8), pos_off[sd], width * 8);
                    A.ne_off[sd] = (const int64_t *)(d + o_off + sd * al(width * 8));
                    A.cursor[sd] = (int32_t *)(d + o_cur + sd * al(width * 4));
                }
                A.ne_q[0] = d_ne;
                A.ne_mq[0] = d_ne + ne_total[0];
                A.ne_q[1] = d_ne + 2 * ne_total[0];
                A.ne_mq[1] = d_ne + 2 * ne_total[0] + ne_total[1];
                if (rc == LFQ_OK) {
                    rc = A.pmax_end ? lfq_launch_plp_indel_columns(A, 1, ps) : lfq_launch_plp_indel(A, 1, ps);
                }
                for (int sd = 0; sd < 2 && rc == LFQ_OK && host_arrays; sd++) {
                    O.side[sd].ne_q.resize((size_t)ne_total[sd]);
                    O.side[sd].ne_mq.resize((size_t)ne_total[sd]);
                    if (ne_total[sd] > 0
                        && (hipMemcpyAsync(O.side[sd].ne_q.data(), A.ne_q[sd], (size_t)ne_total[sd] * 2, hipMemcpyDeviceToHost, ps) != hipSuccess
                            || hipMemcpyAsync(O.side[sd].ne_mq.data(), A.ne_mq[sd], (size_t)ne_total[sd] * 2, hipMemcpyDeviceToHost, ps) != hipSuccess)) {
                        rc = LFQ_ERR_HIP;
                    }
                }
                /* (waited for behind the event tables below) */
            }
        }
        tm[3] = lfq_now_ms();
        /* ai / ad of the event reads when lfq_readset_baq left them on the device: the gather is queued behind the BAQ
         * kernels on their stream and lands in pinned memory; nothing waits for it until the event tables and the consensus
         * flags -- which need neither ai / ad nor the tag bits of the reads -- are built (that host work used to start
         * when the last BAQ kernel had ended: 7 ms of an idle GPU per region). */
        LfqPin<int64_t> idx(c, evs.size());
        LfqPin<uint8_t> g_pin(c, 2 * evs.size());
        const bool gather_aq = rc == LFQ_OK && rs->has_idaq && !evs.empty();
        if (gather_aq) {
            LFQ_PIN_OK(idx);
            LFQ_PIN_OK(g_pin);
            for (size_t i = 0; i < evs.size(); i++) {
                idx[i] = rd->seq_off[evs[i].read] + evs[i].qpos;
            }
            const int64_t ne = (int64_t)evs.size();
            if (grow(&c->d_tmp[2], &c->tmp_bytes[2], ne * 10) != LFQ_OK) {
                rc = LFQ_ERR_NOMEM;
            } else {
                uint8_t *dg = c->d_tmp[2];
                if (hipMemcpyAsync(dg, idx.data(), (size_t)ne * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess
                    || lfq_launch_gather2(rs->d_ai, rs->d_ad, (const int64_t *)dg, ne, dg + ne * 8, dg + ne * 9, c->stream) != LFQ_OK
                    || hipMemcpyAsync(g_pin.data(), dg + ne * 8, (size_t)ne * 2, hipMemcpyDeviceToHost, c->stream) != hipSuccess) {
                    rc = LFQ_ERR_HIP;
                }
                g_ai = g_pin.data();
                g_ad = g_pin.data() + ne;
            }
        }
        if (rc != LFQ_OK) {
            (void)hipStreamSynchronize(ps);
            (void)hipStreamSynchronize(c->stream);
            return rc;
        }
        /* the quality arrays stay resident (c->d_plp_ne): lfq_call_indels_batch builds its pseudo-columns from them on the device */
        c->plp_ne_total[0] = ne_total[0];
        c->plp_ne_total[1] = ne_total[1];
        tm[4] = lfq_now_ms();
        /* 4. event tables: per column and side, events in order of first appearance (uthash iterates in insertion
         * order), their reads in pileup order (add_ins_sequence / add_del_sequence, utils.c) */
        /* Columns with events are few and independent of one another: the event list is cut at position boundaries into a
         * few parts, every part builds the tables of its columns on its own thread, and the parts are appended in order
         * (offsets shifted by what came before; the ev_off entries of the event-less columns in between are range fills). */
        for (int sd = 0; sd < 2; sd++) {
            LfqIndelColsOwned::Side &S = O.side[sd];
            S.ev_off.push_back(0);
            S.key_off.push_back(0);
            S.rd_off.push_back(0);
        }
        const int64_t ncols = (int64_t)O.cov.size();
        struct PartTables {
            LfqIndelColsOwned::Side side[2];        /* key_off / rd_off: local running totals, no leading 0 */
            std::vector<int64_t> cols;              /* columns with events, ascending */
            std::vector<int64_t> ev_after[2];       /* local event count of each side after each of them */
            std::vector<int64_t> rd_ev[2];          /* event index of each entry of side[sd].rd_q (for rd_aq, filled last) */
        };
        const int n_parts = (int)std::max<size_t>(1, std::min<size_t>(8, evs.size() / (size_t)std::max<int64_t>(lfq_knobs().host_par_min / 48, 1)));
        std::vector<PartTables> pt((size_t)n_parts);
        std::vector<size_t> cut((size_t)n_parts + 1, evs.size());
        cut[0] = 0;
        for (int t = 1; t < n_parts; t++) {
            size_t k = evs.size() * (size_t)t / (size_t)n_parts;
            while (k < evs.size() && k > 0 && evs[k].pos == evs[k - 1].pos) {
                k++;
            }
            cut[(size_t)t] = std::max(k, cut[(size_t)t - 1]);
        }
        auto build = [&](int t) {
            PartTables &P = pt[(size_t)t];
            std::vector<std::string> keys;
            std::vector<std::vector<size_t>> members;
            std::string key;
            size_t ei = cut[(size_t)t];
            const size_t e_end = cut[(size_t)t + 1];
            while (ei < e_end) {
                const int64_t ppos = evs[ei].pos - region_begin;
                size_t e1 = ei;
                while (e1 < e_end && evs[e1].pos - region_begin == ppos) {
                    e1++;
                }
                if (h[0][(size_t)ppos] <= 0) {      /* (cannot happen: a read with an event covers its position) */
                    ei = e1;
                    continue;
                }
                P.cols.push_back(col_of[(size_t)ppos]);
                for (int sd = 0; sd < 2; sd++) {
                    LfqIndelColsOwned::Side &S = P.side[sd];
                    keys.clear();                       /* (reused across columns: no allocation in the common case) */
                    for (auto &m : members) {
                        m.clear();
                    }
                    size_t n_keys = 0;
                    for (size_t i = ei; i < e1; i++) {
                        const Ev &e = evs[i];
                        if ((e.indel > 0) != (sd == 0)) {
                            continue;
                        }
                        key.clear();
                        if (sd == 0) {                                  /* inserted bases, plp.c:1082-1086 */
                            const int64_t s0 = rd->seq_off[e.read], lq = rd->seq_off[e.read + 1] - s0;
                            for (int j = 1; j <= e.indel; j++) {
                                const int64_t q = e.qpos + j;
                                const uint8_t code = q < lq ? rd->seq[s0 + q] : 4;
                                key.push_back("ACGTN"[code > 4 ? 4 : code]);
                            }
                        } else {                                        /* deleted reference bases, :1127-1131 */
                            for (int j = 1; j <= -e.indel; j++) {
                                const int64_t g = e.pos + j;
                                key.push_back(g < rd->ref_len ? (char)toupper((unsigned char)rd->ref[g]) : 'N');
                            }
                        }
                        size_t ki = 0;
                        while (ki < n_keys && keys[ki] != key) {
                            ki++;
                        }
                        if (ki == n_keys) {
                            keys.push_back(key);
                            if (members.size() <= n_keys) {
                                members.emplace_back();
                            }
                            n_keys++;
                        }
                        members[ki].push_back(i);
                    }
                    for (size_t ki = 0; ki < n_keys; ki++) {
                        int fw = 0, rv = 0;
                        for (size_t i : members[ki]) {
                            const Ev &e = evs[i];
                            const int64_t s0 = rd->seq_off[e.read];
                            const uint32_t fl = t_fl[e.read];
                            const uint8_t *qa = sd == 0 ? t_bi : t_bd;
                            const bool has_q = qa && (fl & (sd == 0 ? 1u : 2u));
                            S.rd_q.push_back((int16_t)(has_q ? (int)qa[s0 + e.qpos] - 33 : 0));
                            S.rd_aq.push_back((int16_t)-1);                  /* filled when the BAQ kernels are through */
                            P.rd_ev[sd].push_back((int64_t)i);
                            S.rd_mq.push_back((int16_t)rd->mapq[e.read]);
                            const int32_t sq = t_sq ? t_sq[e.read] : -1;
                            S.rd_sq.push_back((int16_t)(sq > 32767 ? 32767 : sq));
                            if (rd->reverse[e.read]) {
                                rv++;
                            } else {
                                fw++;
                            }
                        }
                        S.ev_fw.push_back(fw);
                        S.ev_rv.push_back(rv);
                        S.key_chars.insert(S.key_chars.end(), keys[ki].begin(), keys[ki].end());
                        S.key_off.push_back((int64_t)S.key_chars.size());
                        S.rd_off.push_back((int64_t)S.rd_q.size());
                    }
                    P.ev_after[sd].push_back((int64_t)S.ev_fw.size());
                }
                ei = e1;
            }
        };
        {
            const std::function<void(int)> task = [&](int t) { build(t); };
            LfqLoopPool &pool = LfqLoopPool::instance();
            if (n_parts > 1 && pool.try_run(n_parts, task)) {
                build(0);
                pool.finish();
            } else {
                std::vector<std::thread> th;
                for (int t = 1; t < n_parts; t++) {
                    th.emplace_back(build, t);
                }
                build(0);
                for (auto &x : th) {
                    x.join();
                }
            }
        }
        int64_t col_done = 0;                       /* columns [0, col_done) have their ev_off entries */
        std::vector<int64_t> rd_ev[2];
        for (int t = 0; t < n_parts; t++) {
            PartTables &P = pt[(size_t)t];
            int64_t ev_base[2], rd_base[2], key_base[2];
            for (int sd = 0; sd < 2; sd++) {
                LfqIndelColsOwned::Side &S = O.side[sd];
                const LfqIndelColsOwned::Side &L = P.side[sd];
                ev_base[sd] = (int64_t)S.ev_fw.size();
                rd_base[sd] = (int64_t)S.rd_q.size();
                key_base[sd] = (int64_t)S.key_chars.size();
                S.ev_fw.insert(S.ev_fw.end(), L.ev_fw.begin(), L.ev_fw.end());
                S.ev_rv.insert(S.ev_rv.end(), L.ev_rv.begin(), L.ev_rv.end());
                S.key_chars.insert(S.key_chars.end(), L.key_chars.begin(), L.key_chars.end());
                S.rd_q.insert(S.rd_q.end(), L.rd_q.begin(), L.rd_q.end());
                S.rd_aq.insert(S.rd_aq.end(), L.rd_aq.begin(), L.rd_aq.end());
                S.rd_mq.insert(S.rd_mq.end(), L.rd_mq.begin(), L.rd_mq.end());
                S.rd_sq.insert(S.rd_sq.end(), L.rd_sq.begin(), L.rd_sq.end());
                rd_ev[sd].insert(rd_ev[sd].end(), P.rd_ev[sd].begin(), P.rd_ev[sd].end());
                for (int64_t v : L.key_off) {
                    S.key_off.push_back(v + key_base[sd]);
                }
                for (int64_t v : L.rd_off) {
                    S.rd_off.push_back(v + rd_base[sd]);
                }
            }
            for (size_t i = 0; i < P.cols.size(); i++) {
                const int64_t col = P.cols[i];
                for (int sd = 0; sd < 2; sd++) {
                    LfqIndelColsOwned::Side &S = O.side[sd];
                    /* the event-less columns before this one repeat the running event count */
                    S.ev_off.insert(S.ev_off.end(), (size_t)(col - col_done), S.ev_off.back());
                    S.ev_off.push_back(ev_base[sd] + P.ev_after[sd][i]);
                }
                col_done = col + 1;
            }
        }
        for (int sd = 0; sd < 2; sd++) {            /* the event-less columns behind the last event */
            O.side[sd].ev_off.insert(O.side[sd].ev_off.end(), (size_t)(ncols - col_done), O.side[sd].ev_off.back());
        }
        if (have_qsum) {
            consensus();                            /* (sums from the counter kernel: needs nothing the scatter pass writes) */
        }
        /* 4b. the scatter pass, the BAQ kernels and the gather behind them: the alignment qualities of the event reads
         * (plp.c:1069-1073, 1113-1117); from here on the ai / ad bits of rs->fl (t_fl) are valid */
        if (hipStreamSynchronize(ps) != hipSuccess || (gather_aq && hipStreamSynchronize(c->stream) != hipSuccess)
            || readset_baq_wait(rs) != LFQ_OK) {
            return LFQ_ERR_HIP;
        }
        for (int sd = 0; sd < 2; sd++) {
            LfqIndelColsOwned::Side &S = O.side[sd];
            const uint8_t *aa = sd == 0 ? t_ai : t_ad, *ga = sd == 0 ? g_ai : g_ad;
            for (size_t j = 0; j < rd_ev[sd].size(); j++) {
                const Ev &e = evs[(size_t)rd_ev[sd][j]];
                if (!(t_fl[e.read] & (sd == 0 ? 4u : 8u))) {
                    continue;                           /* no ai / ad tag on this read: -1 */
                }
                if (ga) {
                    S.rd_aq[j] = (int16_t)((int)ga[(size_t)rd_ev[sd][j]] - 33);
                } else if (aa) {
                    S.rd_aq[j] = (int16_t)((int)aa[rd->seq_off[e.read] + e.qpos] - 33);
                }
            }
        }
        if (!have_qsum) {
            consensus();
        }
    }
    tm[5] = lfq_now_ms();
    tm[6] = lfq_now_ms();
    if (lfq_timing_on) {
        fprintf(stderr, "[lfq timing] indel pileup: counters queued + events %.1f  wait %.1f  columns + scatter queued %.1f  gather queued %.1f  tables + consensus + wait + ai / ad %.1f  - %.1f ms\n",
                tm[1] - tm[0], tm[2] - tm[1], tm[3] - tm[2], tm[4] - tm[3], tm[5] - tm[4], tm[6] - tm[5]);
    }
    /* 6. publish */
    lfq_indel_columns &C = O.cols;
    C.cons_indel = O.cons_indel.data();
    C.ncols = (int64_t)O.cov.size();
    C.ref_base = O.ref_base.data();
    C.coverage_plp = O.cov.data();
    C.num_tails = O.tails.data();
    C.num_non_indels = O.non_indels.data();
    C.num_ins = O.n_ins.data();
    C.num_dels = O.n_dels.data();
    C.hrun = O.hrun.data();
    for (int sd = 0; sd < 2; sd++) {
        LfqIndelColsOwned::Side &S = O.side[sd];
        S.key_chars.push_back('\0');
        lfq_indel_side &T = C.side[sd];
        T.non_fw = S.non_fw.data();
        T.non_rv = S.non_rv.data();
        T.ne_off = S.ne_off.data();
        T.ne_q = S.ne_q.empty() && S.ne_off.back() > 0 ? nullptr : S.ne_q.data();       /* device-only: lfq_set_indel_arrays_on_host */
        T.ne_mq = S.ne_mq.empty() && S.ne_off.back() > 0 ? nullptr : S.ne_mq.data();
        T.ev_off = S.ev_off.data();
        T.key_off = S.key_off.data();
        T.key_chars = S.key_chars.data();
        T.ev_fw = S.ev_fw.data();
        T.ev_rv = S.ev_rv.data();
        T.rd_off = S.rd_off.data();
        T.rd_q = S.rd_q.data();
        T.rd_aq = S.rd_aq.data();
        T.rd_mq = S.rd_mq.data();
        T.rd_sq = S.rd_sq.data();
    }
    return LFQ_OK;
}

/* lofreq_uniq.c:262-268: an AF parsed from the VCF that is out of bounds is reset, not rejected */
static inline float lfq_uniq_reset_af(float af)
{
    if (af < 0.0 || af > 1.0) {
        return af < 0.0 ? 0.01f : 1.0f;
    }
    return af;
}

/* uniq_snv with --use-det-lim (lofreq_uniq.c:274-333) for a batch of columns: the default varcall_conf
 * (init_varcall_conf), alt_counts = {(int)(af * n_err_probs), 0, 0}, snpcaller(bonf 1, alpha 0.01f) */
int lfq_uniq_detlim_batch(lfq_ctx *c, const lfq_tracks *tr, int tracks_on_device, const float *af,
                          uint8_t *detectable, long double *pvalue_or_null)
{
    if (!c || !tr || !af || !detectable || tr->ncols < 0) {
        return LFQ_ERR_INVALID;
    }
    const int64_t ncols = tr->ncols;
    if (ncols == 0) {
        return LFQ_OK;
    }
    /* an AF outside [0, 1] is logged and RESET by the reference (af < 0 -> 0.01, af > 1 -> 1.0, lofreq_uniq.c:262-268:
     * LOG_FATAL there does not exit) and the variant is processed with the new value; a NaN passes both comparisons
     * there and is refused here */
    LFQ_TRY_HIP(hipSetDevice(c->device));
    LfqPin<float> h_af(c, (size_t)ncols);
    LFQ_PIN_OK(h_af);
    for (int64_t i = 0; i < ncols; i++) {
        if (af[i] != af[i]) {
            return LFQ_ERR_INVALID;
        }
        h_af[(size_t)i] = lfq_uniq_reset_af(af[i]);
    }
    lfq_conf conf;
    lfq_conf_init(&conf);
    conf.bonf_dynamic = 0;
    conf.bonf_subst = 1;                                /* int bonf = 1 (:286) */
    conf.sig = 0.01f;                                   /* float alpha = 0.01 (:287) */
    lfq_tracks dev;
    LFQ_TRY(stage_tracks(c, tr, tracks_on_device, &dev));
    LFQ_TRY(grow(&c->d_counts, &c->counts_cap, ncols));
    LFQ_TRY(grow(&c->d_pvals, &c->pvals_cap, ncols));
    LFQ_TRY(grow(&c->d_detlim, &c->detlim_cap, ncols));
    LFQ_TRY_HIP(hipMemcpyAsync(c->d_detlim, h_af.data(), (size_t)ncols * 4, hipMemcpyHostToDevice, c->stream));
    c->detlim_af = c->d_detlim;
    int rc = lfq_snv_batch_device(c, &conf, &dev, c->d_counts, c->d_pvals, c->pvals_cap, c->stream);
    c->detlim_af = nullptr;
    LFQ_TRY(rc);
    lfq_batch_stats st;
    LFQ_TRY(lfq_batch_finish(c, &st));
    LfqPin<lfq_col_pvals> h_pv(c, (size_t)st.n_pvals);
    LFQ_PIN_OK(h_pv);
    if (st.n_pvals > 0) {
        LFQ_TRY_HIP(hipMemcpy(h_pv.data(), c->d_pvals, (size_t)st.n_pvals * sizeof(lfq_col_pvals), hipMemcpyDeviceToHost));
    }
    memset(detectable, 0, (size_t)ncols);
    if (pvalue_or_null) {
        for (int64_t i = 0; i < ncols; i++) {
            pvalue_or_null[i] = LDBL_MAX;               /* snpcaller's "not computed" (snpcaller.c:1100-1103) */
        }
    }
    const int bonf = 1;
    const float alpha = 0.01f;
    for (size_t pi = 0; pi < h_pv.size(); pi++) {
        const lfq_col_pvals &p = h_pv[pi];
        if (p.col < 0 || p.col >= ncols || p.status[0] == LFQ_PV_NONE) {
            continue;
        }
        const long double pv = lfq_pvalue_from_log(p.logp[0], p.status[0]);
        if (pvalue_or_null) {
            pvalue_or_null[p.col] = pv;
        }
        detectable[p.col] = (pv * (float)bonf < alpha) ? 1 : 0;         /* :314 */
    }
    return LFQ_OK;
}

/* uniq_snv's default branch for a batch of columns (lofreq_uniq.c:254-256, 335-393): the device counts the bases of
 * every nucleotide (base_count, plp.c:128-132); coverage, the binomial test and the phred value are per-variant scalar
 * work on the host */
int lfq_uniq_binom_batch(lfq_ctx *c, const lfq_tracks *tr, int tracks_on_device, const float *af, const char *alt_base,
                         int32_t *uq_out, double *pvalue_or_null)
{
    if (!c || !tr || !af || !alt_base || !uq_out || tr->ncols < 0) {
        return LFQ_ERR_INVALID;
    }
    const int64_t ncols = tr->ncols;
    if (ncols == 0) {
        return LFQ_OK;
    }
    LFQ_TRY_HIP(hipSetDevice(c->device));
    lfq_tracks dev;
    LFQ_TRY(stage_tracks(c, tr, tracks_on_device, &dev));
    LfqTracksDev T;
    memset(&T, 0, sizeof(T));
    T.nt = dev.nt;
    T.bq = dev.bq;
    T.col_off = dev.col_off;
    T.ncols = ncols;
    T.nt_packed = (dev.flags & LFQ_TRACKS_NT_PACKED) ? 1 : 0;
    LFQ_TRY(grow(&c->d_counts, &c->counts_cap, ncols));             /* reused as int32[4] per column */
    int32_t *d_nt = reinterpret_cast<int32_t *>(c->d_counts);
    LFQ_TRY(lfq_launch_ntcount(T, d_nt, c->stream));
    LfqPin<int32_t> h_nt(c, (size_t)ncols * 4), h_cov(c, dev.coverage_plp ? (size_t)ncols : 0);
    LfqPin<uint64_t> h_off(c, (size_t)ncols + 1);
    LFQ_PIN_OK(h_nt);
    LFQ_PIN_OK(h_cov);
    LFQ_PIN_OK(h_off);
    LFQ_TRY_HIP(hipMemcpyAsync(h_nt.data(), d_nt, (size_t)ncols * 16, hipMemcpyDeviceToHost, c->stream));
    LFQ_TRY_HIP(hipMemcpyAsync(h_off.data(), dev.col_off, ((size_t)ncols + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    if (dev.coverage_plp) {
        LFQ_TRY_HIP(hipMemcpyAsync(h_cov.data(), dev.coverage_plp, (size_t)ncols * 4, hipMemcpyDeviceToHost, c->stream));
    }
    LFQ_TRY_HIP(hipStreamSynchronize(c->stream));
    for (int64_t i = 0; i < ncols; i++) {
        uq_out[i] = -1;
        if (pvalue_or_null) {
            pvalue_or_null[i] = -1.0;
        }
        const int coverage = dev.coverage_plp ? h_cov[(size_t)i] : (int)(h_off[(size_t)i + 1] - h_off[(size_t)i]);
        if (coverage < 1) {
            continue;                                           /* :254-256 */
        }
        const char ab = alt_base[i];
        const int code = (ab == 'A' || ab == 'a') ? 0 : (ab == 'C' || ab == 'c') ? 1 : (ab == 'G' || ab == 'g') ? 2
                         : (ab == 'T' || ab == 't') ? 3 : 4;
        /* bam_nt4_table sends every other letter to N: those are the observations of the column that are in none of
         * the four nucleotide counts */
        const int n_col = (int)(h_off[(size_t)i + 1] - h_off[(size_t)i]);
        const int32_t *cn = &h_nt[(size_t)i * 4];
        const int alt_count = code < 4 ? cn[code] : n_col - cn[0] - cn[1] - cn[2] - cn[3];
        int st = 0;
        const double pv = lfq_binom_cdf(coverage, alt_count, (double)lfq_uniq_reset_af(af[i]), &st);   /* :262-268, :381; one-sided */
        if (st != 0) {
            continue;                                           /* "binom() failed": no UQ tag */
        }
        uq_out[i] = phred_safe(pv);                             /* :386 */
        if (pvalue_or_null) {
            pvalue_or_null[i] = pv;
        }
    }
    return LFQ_OK;
}

int lfq_pileup_skip_snv_columns(lfq_ctx *c, const uint8_t *skip, int64_t ncols)
{
    if (!c || !skip || ncols < 0 || !c->d_plp_out || ncols != c->plp_ncols) {
        return LFQ_ERR_INVALID;
    }
    if (ncols == 0) {
        return LFQ_OK;
    }
    LFQ_TRY_HIP(hipSetDevice(c->device));
    /* the skip bytes go through a pinned block and are applied on the device, in stream order before the batch */
    LfqPin<uint8_t> h(c, (size_t)ncols);
    LFQ_PIN_OK(h);
    memcpy(h.data(), skip, (size_t)ncols);
    LFQ_TRY(order_after_batch(c, c->stream));
    LFQ_TRY(grow(&c->d_tmp[2], &c->tmp_bytes[2], ncols));
    LFQ_TRY_HIP(hipMemcpyAsync(c->d_tmp[2], h.data(), (size_t)ncols, hipMemcpyHostToDevice, c->stream));
    LFQ_TRY(lfq_launch_skip_columns(c->d_plp_nb, c->d_tmp[2], ncols, c->stream));
    LFQ_TRY_HIP(hipStreamSynchronize(c->stream));
    return LFQ_OK;
}

/* source_qual for a batch of reads (plp.c:427-593): counting + DP on the device, phred conversion here */
int lfq_source_qual_batch(lfq_ctx *c, const lfq_baq_reads *rd, int def_nm_q, int min_bq, const uint8_t *ign,
                          int32_t *sq_out, uint8_t *sq_byte)
{
    if (!c || !rd || !sq_out || rd->n_reads < 0 || def_nm_q > 255
        || (rd->n_reads > 0 && (!rd->pos || !rd->cigar_off || !rd->cigar || !rd->seq_off || !rd->seq || !rd->qual
                                || !rd->ref))) {
        return LFQ_ERR_INVALID;
    }
    if (rd->n_reads == 0) {
        return LFQ_OK;
    }
    lfq_readset *rs = nullptr;
    LFQ_TRY(readset_from_baq_reads(c, rd, &rs));
    const int rc = lfq_readset_source_qual(c, rs, def_nm_q, min_bq, ign, sq_out);
    if (rc == LFQ_OK && sq_byte) {
        for (int64_t r = 0; r < rd->n_reads; r++) {
            const int q = sq_out[r];
            sq_byte[r] = (uint8_t)(q < 0 ? 0 : (q > 254 ? 254 : q));
        }
    }
    lfq_readset_destroy(rs);
    return rc;
}

int lfq_readset_source_qual(lfq_ctx *c, lfq_readset *rs, int def_nm_q, int min_bq, const uint8_t *ign, int32_t *sq_out)
{
    if (!c || !rs || rs->c != c || def_nm_q > 255 || (rs->n > 0 && !rs->qual)) {
        return LFQ_ERR_INVALID;
    }
    const lfq_readset *rd = rs;
    const int64_t n = rs->n;
    if (n == 0) {
        return LFQ_OK;
    }
    LFQ_TRY_HIP(hipSetDevice(c->device));
    LFQ_TRY(readset_upload_wait(rs));
    int64_t max_ops = 0;                                /* bound of K: one operation per base or CIGAR element */
    for (int64_t r = 0; r < n; r++) {
        max_ops = std::max<int64_t>(max_ops, (rd->seq_off[r + 1] - rd->seq_off[r]) + (rd->cigar_off[r + 1] - rd->cigar_off[r]));
    }
    const int n_blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n + 3) / 4, (int64_t)c->n_cu * 2));
    const int64_t scratch_cells = max_ops + 1 > LFQ_SRCQ_LDS_CELLS ? max_ops + 1 : 0;
    auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
    const int64_t o_ign = 0, o_prob = o_ign + (ign ? al(rd->ref_len) : 0), o_st = o_prob + al(n * 8), o_scr = o_st + al(n),
                  total = o_scr + (int64_t)n_blocks * 4 * 2 * scratch_cells * 8;
    uint8_t *d = nullptr;
    if (hipMalloc((void **)&d, (size_t)total) != hipSuccess) {
        return LFQ_ERR_NOMEM;
    }
    int rc = LFQ_OK;
    LfqPin<double> prob(c, (size_t)n);
    LfqPin<uint8_t> st(c, (size_t)n);
    if (!prob.ok() || !st.ok()) {
        rc = LFQ_ERR_NOMEM;
    }
    if (ign && hipMemcpyAsync(d + o_ign, ign, (size_t)rd->ref_len, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
        rc = LFQ_ERR_HIP;
    }
    if (rc == LFQ_OK) {
        LfqSrcqArgs A;
        memset(&A, 0, sizeof(A));
        A.n_reads = n;
        A.pos = (const int32_t *)rs->d_pos;
        A.cigar_off = (const int64_t *)rs->d_coff;
        A.seq_off = (const int64_t *)rs->d_soff;
        A.cigar = (const uint32_t *)rs->d_cig;
        A.seq = rs->d_seq;
        A.qual = rs->d_qual;
        A.ref = (const char *)rs->d_ref;
        A.ref_len = rd->ref_len;
        A.ign = ign ? d + o_ign : nullptr;
        A.nonmatch_qual = def_nm_q;
        A.min_bq = min_bq;
        A.scratch = scratch_cells ? (double *)(d + o_scr) : nullptr;
        A.scratch_cells = scratch_cells;
        A.prob = (double *)(d + o_prob);
        A.status = d + o_st;
        rc = lfq_launch_srcq(A, c->d_luts, n_blocks, c->stream);
    }
    if (rc == LFQ_OK
        && (hipMemcpyAsync(prob.data(), d + o_prob, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess
            || hipMemcpyAsync(st.data(), d + o_st, (size_t)n, hipMemcpyDeviceToHost, c->stream) != hipSuccess)) {
        rc = LFQ_ERR_HIP;
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == LFQ_OK) {
        rc = LFQ_ERR_HIP;
    }
    (void)hipFree(d);
    if (rc != LFQ_OK) {
        return rc;
    }
    const int perfect = (int)(-10.0L * log10l(LDBL_MIN));       /* PROB_TO_PHREDQUAL(LDBL_MIN) = 49314, plp.c:521 */
    rs->sq32.resize((size_t)n);
    LfqPin<uint8_t> sqb(c, (size_t)n);
    LFQ_PIN_OK(sqb);
    for (int64_t r = 0; r < n; r++) {
        int q;
        if (st[(size_t)r] == LFQ_SRCQ_NA) {
            q = -1;
        } else if (st[(size_t)r] == LFQ_SRCQ_PERFECT) {
            q = perfect;
        } else {
            const double x = 1.0 - prob[(size_t)r];             /* PROB_TO_PHREDQUAL(1.0 - src_prob), plp.c:567 */
            /* log10l(0) = -inf and a negative argument gives NaN: the x86-64 long double -> int conversion
             * of either is INT_MIN ("integer indefinite"); mplp_func then stores 0 */
            q = (x > 0.0) ? (int)(-10.0L * log10l((long double)x)) : INT32_MIN;
        }
        if (sq_out) {
            sq_out[r] = q;
        }
        rs->sq32[(size_t)r] = q < 0 ? 0 : q;                    /* what the sq tag holds (plp.c:731-734) */
        sqb[(size_t)r] = (uint8_t)(q < 0 ? 0 : (q > 254 ? 254 : q));
    }
    /* the byte of the packed sq track, resident for lfq_readset_pileup_snv */
    LFQ_TRY_HIP(hipMemcpy(rs->d_sqb, sqb.data(), (size_t)n, hipMemcpyHostToDevice));
    rs->has_sqb = true;
    return LFQ_OK;
}

int lfq_synth_fill_device(lfq_ctx *c, uint64_t seed, uint32_t depth, uint32_t plant_period, int64_t col_begin,
                          int64_t ncols, uint8_t *d_nt, uint8_t *d_bq, uint8_t *d_baq, uint8_t *d_mq,
                          uint64_t *d_col_off, uint8_t *d_ref_base, void *stream_or_null)
{
    return lfq_synth_fill_device_layout(c, seed, depth, plant_period, col_begin, ncols, d_nt, d_bq, d_baq, d_mq, d_col_off,
                                        d_ref_base, 0, stream_or_null);
}

int lfq_synth_fill_device_layout(lfq_ctx *c, uint64_t seed, uint32_t depth, uint32_t plant_period, int64_t col_begin,
                                 int64_t ncols, uint8_t *d_nt, uint8_t *d_bq, uint8_t *d_baq, uint8_t *d_mq,
                                 uint64_t *d_col_off, uint8_t *d_ref_base, int nt_packed, void *stream_or_null)
{
    if (!c || !d_nt || !d_bq || !d_baq || !d_mq || !d_col_off || !d_ref_base || ncols < 0 || depth == 0) {
        return LFQ_ERR_INVALID;
    }
    LFQ_TRY_HIP(hipSetDevice(c->device));
    lfq_synth_spec s;
    memset(&s, 0, sizeof(s));
    s.seed = seed;
    s.depth = depth;
    s.plant_period = plant_period;
    for (int q = 0; q < 64; q++) {
        const long double p = powl(10.0L, -(long double)q / 10.0L);
        const long double t = floorl(p * 18446744073709551616.0L);
        s.err_thresh[q] = (t >= 18446744073709551615.0L) ? UINT64_MAX : (uint64_t)t;
    }
    hipStream_t st = stream_or_null ? (hipStream_t)stream_or_null : c->stream;
    LFQ_TRY(order_after_batch(c, st));                 /* the target may be the tracks of a batch that is still running */
    return lfq_launch_synth(&s, col_begin, ncols, d_nt, d_bq, d_baq, d_mq, d_col_off, d_ref_base, nt_packed ? 1 : 0, st);
}

}  // extern "C"

/* ---- the exchange of a sharded run from C (include/lofreq_amd.h, "N processes") ------------------------------------ */
#include <dlfcn.h>
namespace {
typedef int (*lfq_nccl_allgather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
lfq_nccl_allgather_fn lfq_rccl_allgather()
{
    static lfq_nccl_allgather_fn fn = [] {
        /* the copy the process already has (PyTorch brings its own) before the system one */
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (int pass = 0; pass < 2; pass++) {
            for (const char *nm : names) {
                void *h = dlopen(nm, RTLD_NOW | (pass == 0 ? RTLD_NOLOAD : 0));
                if (h) {
                    void *f = dlsym(h, "ncclAllGather");
                    if (f) {
                        return (lfq_nccl_allgather_fn)f;
                    }
                }
            }
        }
        return (lfq_nccl_allgather_fn) nullptr;
    }();
    return fn;
}

lfq_host_allgather_fn g_host_allgather = nullptr;
void *g_host_allgather_user = nullptr;

/* all-gather of `bytes` bytes per rank through device staging buffers of the context */
int shard_allgather_bytes(lfq_ctx *c, void *comm, int world, int rank, const void *mine, size_t bytes, void *all)
{
    if (world > 1 && g_host_allgather) {
        /* a launcher-supplied host transport (MPI, files, a test double) instead of RCCL */
        return g_host_allgather(g_host_allgather_user, world, rank, mine, all, bytes) == 0 ? LFQ_OK : LFQ_ERR_HIP;
    }
    if (world == 1 || !comm) {
        if (world != 1) {
            return LFQ_ERR_INVALID;
        }
        memcpy(all, mine, bytes);
        return LFQ_OK;
    }
    lfq_nccl_allgather_fn ag = lfq_rccl_allgather();
    if (!ag || !c) {
        return LFQ_ERR_UNSUPPORTED;
    }
    (void)rank;
    LFQ_TRY_HIP(hipSetDevice(c->device));
    const int64_t padded = (int64_t)((bytes + 255) / 256 * 256);
    LFQ_TRY(grow(&c->d_tmp[2], &c->tmp_bytes[2], padded * (world + 1)));
    uint8_t *d_send = c->d_tmp[2], *d_recv = c->d_tmp[2] + padded;
    LFQ_TRY_HIP(hipMemcpyAsync(d_send, mine, bytes, hipMemcpyHostToDevice, c->stream));
    if (ag(d_send, d_recv, (size_t)padded, /* ncclUint8 */ 1, comm, c->stream) != 0) {
        return LFQ_ERR_HIP;
    }
    LfqPin<uint8_t> h(c, (size_t)padded * world);
    LFQ_PIN_OK(h);
    LFQ_TRY_HIP(hipMemcpyAsync(h.data(), d_recv, (size_t)padded * world, hipMemcpyDeviceToHost, c->stream));
    LFQ_TRY_HIP(hipStreamSynchronize(c->stream));
    for (int r = 0; r < world; r++) {
        memcpy((uint8_t *)all + (size_t)r * bytes, h.data() + (size_t)r * padded, bytes);
    }
    return LFQ_OK;
}
}  // namespace

int lfq_shard_allgather(lfq_ctx *c, void *comm, int world, int rank, const void *mine, int64_t bytes, void *all)
{
    if (world < 1 || rank < 0 || rank >= world || bytes < 0 || (bytes > 0 && (!mine || !all))) {
        return LFQ_ERR_INVALID;
    }
    if (bytes == 0) {
        return LFQ_OK;
    }
    return shard_allgather_bytes(c, comm, world, rank, mine, (size_t)bytes, all);
}

int lfq_shard_set_host_allgather(lfq_host_allgather_fn fn, void *user)
{
    g_host_allgather = fn;
    g_host_allgather_user = user;
    return LFQ_OK;
}

int lfq_shard_exchange_counts(lfq_ctx *c, void *comm, int world, int rank, const int64_t *local, int n, int64_t *all_out,
                              int64_t *prefix_out)
{
    if (world < 1 || rank < 0 || rank >= world || n < 0 || (n > 0 && (!local || !all_out))) {
        return LFQ_ERR_INVALID;
    }
    if (n == 0) {
        return LFQ_OK;
    }
    LFQ_TRY(shard_allgather_bytes(c, comm, world, rank, local, (size_t)n * 8, all_out));
    if (prefix_out) {
        for (int i = 0; i < n; i++) {
            int64_t p = 0;
            for (int r = 0; r < rank; r++) {
                p += all_out[(size_t)r * n + i];
            }
            prefix_out[i] = p;
        }
    }
    return LFQ_OK;
}

int lfq_shard_rebase_bonferroni(lfq_col_pvals *pvals, int64_t n, int64_t prefix_tested)
{
    if (n < 0 || (n > 0 && !pvals) || prefix_tested < 0) {
        return LFQ_ERR_INVALID;
    }
    for (int64_t i = 0; i < n; i++) {
        pvals[i].bonf += 3 * prefix_tested;         /* every tested column of an earlier shard: 3 tests (lofreq_call.c:794-801) */
    }
    return LFQ_OK;
}

int lfq_shard_advance_conf(lfq_conf *conf, int64_t total_tested)
{
    if (!conf || total_tested < 0) {
        return LFQ_ERR_INVALID;
    }
    if (total_tested > 0) {
        if (conf->bonf_dynamic) {
            conf->bonf_subst = (conf->bonf_subst == 1 ? 0 : conf->bonf_subst) + 3 * total_tested;
        }
        conf->num_snv_tests += 3 * total_tested;
    }
    return LFQ_OK;
}

int lfq_shard_gather_records(lfq_ctx *c, void *comm, int world, int rank, const lfq_snv_record *recs, int64_t n,
                             int64_t col_offset, lfq_snv_record *out, int64_t capacity, int64_t *n_out)
{
    if (world < 1 || rank < 0 || rank >= world || n < 0 || (n > 0 && !recs) || !n_out || capacity < 0 || (capacity > 0 && !out)) {
        return LFQ_ERR_INVALID;
    }
    std::vector<int64_t> counts((size_t)world);
    LFQ_TRY(shard_allgather_bytes(c, comm, world, rank, &n, 8, counts.data()));
    int64_t total = 0, most = 0;
    for (int r = 0; r < world; r++) {
        total += counts[(size_t)r];
        most = std::max(most, counts[(size_t)r]);
    }
    *n_out = total;
    if (most == 0) {                /* the same on every rank: nobody enters the second collective */
        return LFQ_OK;
    }
    /* `capacity` is a local value (a caller may want the records on rank 0 only): the decision to enter the second
     * all-gather must not depend on it, or the ranks with enough room wait for ever for the ones without */
    std::vector<lfq_snv_record> mine((size_t)most), all((size_t)most * world);
    memset((void *)mine.data(), 0, (size_t)most * sizeof(lfq_snv_record));
    for (int64_t i = 0; i < n; i++) {
        mine[(size_t)i] = recs[i];
        mine[(size_t)i].col += col_offset;
    }
    LFQ_TRY(shard_allgather_bytes(c, comm, world, rank, mine.data(), (size_t)most * sizeof(lfq_snv_record), all.data()));
    int64_t o = 0;
    for (int r = 0; r < world; r++) {
        for (int64_t i = 0; i < counts[(size_t)r] && o < capacity; i++) {
            out[o++] = all[(size_t)r * most + (size_t)i];
        }
    }
    return total > capacity ? LFQ_ERR_CAPACITY : LFQ_OK;      /* *n_out says how many there are */
}
