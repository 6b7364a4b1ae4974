/*
 * lfq_ctx.h -- what the host-side files of the library share: the context (lfq_ctx), the helper-thread pool of the
 * host loops, grow-only device buffers, the pinned pool, and the few internal functions that cross file boundaries.
 * Internal: nothing outside lofreq_amd/csrc includes it.
 *
 *   lfq_api.hip        context, streams, the batch driver (layer 1), the call_snvs loop (layer 2), uniq, generator
 *   lfq_indel_api.hip  the indel tests (lfq_call_indel_tests_batch, lfq_call_indels_batch)
 *   lfq_readset.hip    the resident read set: upload, BAQ / IDAQ, source quality, both pileups
 *   lfq_shard.hip      device choice of a worker, the exchange of a sharded run
 */
#ifndef LFQ_CTX_H
#define LFQ_CTX_H

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

/* most threads a host loop is cut for (the arrays of per-part results are this long); how many it really uses:
 * LFQ_HOST_LOOP_THREADS, the process's CPU budget, the loop's length */
#define LFQ_HOST_PARTS 16

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
        const unsigned want = (unsigned)std::min<long>(std::max<long>(lfq_knobs().host_loop_threads, 1), LFQ_HOST_PARTS);
        const int n = spin_us_ < 0 ? 0 : (int)std::min(want - 1u, hw - 1u);
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
        const unsigned want = (unsigned)std::min<long>(std::max<long>(lfq_knobs().host_loop_threads, 1), LFQ_HOST_PARTS);
        parts = (int)std::min<int64_t>(std::min<unsigned>(hw ? hw : 1u, want), n / std::max<int64_t>(par_min / 2, 1));
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

/* std::vector whose resize() leaves new elements of a trivial type uninitialised: the per-region column arrays (tens of MB)
 * are resized by one thread and then written in full by the pool's threads -- value-initialising them first was a serial
 * memset of every array, milliseconds per region. */
template <class T>
struct LfqNoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { typedef LfqNoInitAlloc<U> other; };
    LfqNoInitAlloc() = default;
    template <class U> LfqNoInitAlloc(const LfqNoInitAlloc<U> &) {}
    template <class U> void construct(U *p) { ::new ((void *)p) U; }
    template <class U, class A0, class... Args> void construct(U *p, A0 &&a0, Args &&...args)
    {
        ::new ((void *)p) U(std::forward<A0>(a0), std::forward<Args>(args)...);
    }
};
template <class T> using LfqVec = std::vector<T, LfqNoInitAlloc<T>>;

struct LfqIndelColsOwned {
    lfq_indel_columns cols;
    LfqVec<uint8_t> ref_base, cons_indel;
    LfqVec<int32_t> cov, tails, non_indels, n_ins, n_dels, hrun;
    struct Side {
        LfqVec<int32_t> non_fw, non_rv, ev_fw, ev_rv;
        LfqVec<int64_t> ne_off, ev_off, key_off, rd_off;
        LfqVec<int16_t> ne_q, ne_mq, rd_q, rd_aq, rd_mq, rd_sq;
        LfqVec<char> key_chars;
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
    int cur_sparse_counts;           /* this batch's count kernel stored the dense entries of the tested columns only */
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
    int priv_stream_on;
    hipStream_t priv_stream;         /* lfq_set_private_stream: this context's own launch stream (null = the device's shared one) */
    hipStream_t up_stream;           /* lfq_readset_create's uploads and the staging copies of host tracks (created on first use) */
    hipEvent_t ev_up;                /* end of the staging copies of a batch of host tracks */
    hipEvent_t ev_apply;             /* lfq_readset_pileup_snv: the column positions are final (created on first use) */
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
    int dense_counts;                /* lfq_set_dense_counts: 0 = a caller's dense array may keep stale entries for untested columns */
    int dense_strand;                /* lfq_set_dense_strand_counts: layer 1 / async layer 2 fill the strand fields of every dense entry */
    int lazy_forced;                 /* set by lfq_call_snvs_batch around its submit */
    hipEvent_t ev_baq_t[2];          /* around the BAQ kernels of the last lfq_readset_baq / lfq_baq_batch call (timing; created on first use) */
    int32_t baq_launches;            /* kernel launches between them on the context's stream */
    int64_t baq_reads, baq_bases;    /* reads / bases of that call */
    int batch_gate;                  /* lfq_set_batch_gate: what this context's next count kernel waits for (LFQ_GATE_*) */
    int lazy_now;                    /* this batch: strand counts only for the columns of the sparse output */
    int64_t sub_ncols;               /* batch submitted with lfq_call_snvs_submit and not collected yet: its columns, else -1 */
    int batch_recorded;              /* ev[3] has been recorded: a batch of this context may still be running */
    float baq_par_d, baq_par_e;      /* lfq_set_baq_hmm_params; kpa_ext_par_lofreq_illumina (kprobaln_ext.c:50) by default */
    int plp_nt_bytes;                /* lfq_set_pileup_nt_packed(ctx, 0): the device pileup hands out one nt byte per observation */
    int plp_unsorted_ok;             /* lfq_set_pileup_unsorted(ctx, 1): reads that are not position-sorted go to the read-major kernels
                                      * instead of being refused (a column's observations then arrive in no fixed order) */
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
    uint8_t *d_baq_nflag;            /* one byte per wavefront of the plain narrow-band launches: the wavefront meets an N */
    int64_t baq_nflag_bytes;
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

/* Host temporaries a DMA reads or writes come from a grow-only pool of pinned blocks owned by the context (lfq_api.hip) */
void *lfq_pin_acquire(lfq_ctx *c, size_t bytes, int *slot);

template <typename T>
struct LfqPin {
    lfq_ctx *c;
    T *p = nullptr;
    size_t n = 0;
    int slot = -1;
    LfqPin(lfq_ctx *ctx, size_t count) : c(ctx), n(count)
    {
        p = (T *)lfq_pin_acquire(c, std::max<size_t>(count, 1) * sizeof(T), &slot);
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

/* ---- internal functions that cross file boundaries (lfq_api.hip) ---- */
int lfq_make_params(const lfq_conf *conf, const lfq_tracks *tr, LfqParams *P, bool indel_mode);
extern "C" {
/* everything the library queues that rewrites what a running batch still reads waits for the batch's last event first */
int lfq_order_after_batch(lfq_ctx *c, hipStream_t st);
int lfq_batch_device_impl(lfq_ctx *c, const lfq_conf *conf, const lfq_tracks *tr, lfq_col_counts *d_counts,
                          lfq_col_pvals *d_pvals, int64_t pvals_capacity, void *stream_or_null, bool indel_mode);
int lfq_stage_tracks(lfq_ctx *c, const lfq_tracks *tr, int tracks_on_device, lfq_tracks *dev_out);
}

#endif
