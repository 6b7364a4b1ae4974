/*
 * lfq_host.cpp -- host-side pieces of the SNV calling path that stay on the CPU, as in the
 * reference: 80-bit p-value conversion with the errno/fenv clamp, the running-Bonferroni emit
 * test, strand bias (Fisher's exact test), VCF record text, multiple-testing corrections and the
 * final `lofreq filter` step.  Per *reported variant* work only -- negligible cost.
 *
 * No HIP in this file.  Reference citations are relative to src/lofreq/.
 *
 * Provenance of two blocks that follow their originals statement by statement, because their results are integers in the
 * VCF (SB, FILTER) that must come out bit-identical, so the order of the floating-point operations IS the specification
 * (tests/test_ref_parts.py compares them bitwise with the reference's own objects):
 *   - Fisher's exact test (log_choose .. lfq_fisher_exact, lfq_sb_phred): after LoFreq's fet.c:13-99, which its header marks
 *     "Taken from samtools 0.1.18 (r982:295)" -- Heng Li's kfunc.c kt_fisher_exact (MIT licence), implemented with ideas from
 *     http://www.langsrud.com/fisher.htm;
 *   - the multiple-testing corrections (lfq_bonf_corr, lfq_holm_bonf_corr, lfq_fdr): after LoFreq's multtest.c:66-189
 *     (LoFreq, MIT licence; its FDR follows the Benjamini-Hochberg step-up procedure).
 */
#include <errno.h>
#include <fenv.h>
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <sched.h>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "lofreq_amd.h"
#include "lfq_internal.h"

/* the environment knobs, parsed once (lfq_internal.h).
 *
 * A release build reads TEN variables, none of which changes a result (DESIGN.md "Environment knobs"): LFQ_TIMING,
 * LFQ_SINGLE_STREAM, LFQ_DEBUG_SYNC, LFQ_PRIVATE_STREAM, LFQ_SYNC_UPLOAD, LFQ_BAQ_SCRATCH_MB, LFQ_HOST_THREADS,
 * LFQ_HOST_LOOP_THREADS, LFQ_HOST_SPIN_US and torchrun's LOCAL_WORLD_SIZE.  Everything else -- the launch-shape and
 * fallback selectors the knob tests drive (tests/test_gpu_knobs.py) and LFQ_DEBUG_SKIP, which drops DP classes and with them
 * calls -- exists only in the tuning build (-DLFQ_TUNE: lofreq_amd/liblofreq_amd_tune.so, `make tune`; same kernels, same
 * objects, this one file compiled twice). */
#ifdef LFQ_TUNE
#define LFQ_TUNE_I(name, dflt) geti(name, dflt)
#define LFQ_TUNE_HAS(name) has(name)
#else
#define LFQ_TUNE_I(name, dflt) ((long)(dflt))
#define LFQ_TUNE_HAS(name) false
#endif
const LfqKnobs &lfq_knobs(void)
{
    static const LfqKnobs k = [] {
        LfqKnobs x;
        memset(&x, 0, sizeof(x));
        auto geti = [](const char *name, long dflt) -> long {
            const char *e = getenv(name);
            return (e && *e) ? atol(e) : dflt;
        };
        auto has = [](const char *name) { return getenv(name) != nullptr; };
        /* ---- release: ten variables ---- */
        x.timing = has("LFQ_TIMING");
        x.single_stream = has("LFQ_SINGLE_STREAM");
        x.debug_sync = has("LFQ_DEBUG_SYNC");
        x.private_stream = (int)geti("LFQ_PRIVATE_STREAM", 0);
        x.sync_upload = (int)geti("LFQ_SYNC_UPLOAD", 0);
        x.baq_scratch_mb = geti("LFQ_BAQ_SCRATCH_MB", -1);
        x.host_threads = (int)geti("LFQ_HOST_THREADS", -1);
        x.host_loop_threads = geti("LFQ_HOST_LOOP_THREADS", 8);
        x.host_spin_us = geti("LFQ_HOST_SPIN_US", 2000);
        x.local_world_size = (int)std::max(1L, geti("LOCAL_WORLD_SIZE", 1));
        /* ---- tuning build only (defaults otherwise) ---- */
        x.no_sb_precompute = LFQ_TUNE_HAS("LFQ_NO_SB_PRECOMPUTE");
#ifdef LFQ_TUNE
        if (const char *sk = getenv("LFQ_DEBUG_SKIP")) {
            x.skip_light = strstr(sk, "light") != nullptr;
            x.skip_mid = strstr(sk, "mid") != nullptr;
            x.skip_big = strstr(sk, "big") != nullptr;
        }
        if (const char *lk = getenv("LFQ_LIGHT_KERNEL")) {
            x.light_kernel = !strcmp(lk, "wave") ? 2 : 0;
        }
#endif
        x.screen_waves_per_cu = (int)LFQ_TUNE_I("LFQ_SCREEN_WAVES_PER_CU", -1);      /* < 1: by the context's gate and the batch's depth */
        x.screen_rounds = (int)std::max(1L, LFQ_TUNE_I("LFQ_SCREEN_ROUNDS", 24));
        x.phase1_chunks = (int)std::max(1L, LFQ_TUNE_I("LFQ_PHASE1_CHUNKS", LFQ_PHASE1_CHUNKS));
        {
            const long both = LFQ_TUNE_I("LFQ_SEG_MAX", -1);
            const long big = LFQ_TUNE_I("LFQ_SEG_MAX_BIG", both), mid = LFQ_TUNE_I("LFQ_SEG_MAX_MID", both);
            x.seg_max = big < 0 ? -1 : (int)std::min((long)LFQ_SEG_MAX, std::max(2L, big));
            x.seg_max_mid = mid < 0 ? -1 : (int)std::min((long)LFQ_SEG_MAX, std::max(2L, mid));
        }
        x.seg_budget_mid = (int)std::max(1L, LFQ_TUNE_I("LFQ_SEG_BUDGET_MID", 4096));
        x.seg_budget_big = (int)std::max(1L, LFQ_TUNE_I("LFQ_SEG_BUDGET_BIG", 4096));
        x.split_pool_cells = (int)std::max(0L, LFQ_TUNE_I("LFQ_SPLIT_POOL_CELLS", 8L << 20));
        x.count_multi_below = LFQ_TUNE_I("LFQ_COUNT_MULTI_BELOW", 4096);
        {
            const long w = LFQ_TUNE_I("LFQ_COUNT_WAVES_PER_WG", 16);
            x.count_waves_per_wg = (w == 4 || w == 8 || w == 12) ? (int)w : 16;
        }
        {
            const long u = LFQ_TUNE_I("LFQ_COUNT_AHEAD_DEEP", 2);
            x.count_ahead_deep = (u == 3 || u == 4) ? (int)u : 2;
        }
        x.count_cols_per_wave = (int)std::min(std::max(LFQ_TUNE_I("LFQ_COUNT_COLS_PER_WAVE", 1), 1L), 16L);
        x.big_on_side = LFQ_TUNE_HAS("LFQ_BIG_ON_SIDE");
        x.sb_par_min_cost = LFQ_TUNE_I("LFQ_SB_PAR_MIN_COST", 4000);
        x.pileup_tiles = (int)LFQ_TUNE_I("LFQ_PILEUP_TILES", 1);
        x.baq_one_variant = LFQ_TUNE_HAS("LFQ_BAQ_ONE_VARIANT") ? 1 : 0;
        x.count_lpg4_below = LFQ_TUNE_I("LFQ_COUNT_LPG4_BELOW", 320);
        x.count_lpg8_below = LFQ_TUNE_I("LFQ_COUNT_LPG8_BELOW", 900);
        x.host_par_min = std::max(1L, LFQ_TUNE_I("LFQ_HOST_PAR_MIN", 200000));
        x.indel_host_pack = LFQ_TUNE_HAS("LFQ_INDEL_HOST_PACK");
        x.pileup_atomic = LFQ_TUNE_HAS("LFQ_PILEUP_ATOMIC");
        x.baq_lds = LFQ_TUNE_I("LFQ_BAQ_LDS", 1) != 0;
        x.baq_idaq_beside = (int)LFQ_TUNE_I("LFQ_BAQ_IDAQ_BESIDE", 0);
        x.tail_light = (int)std::min(2L, std::max(0L, LFQ_TUNE_I("LFQ_TAIL_LIGHT", 1)));
        x.count_shallow_wgs_none = (int)std::min(4L, std::max(0L, LFQ_TUNE_I("LFQ_COUNT_SHALLOW_WGS_NONE", 2)));
        x.count_lean_lds_pad = (int)std::min(160000L, std::max(0L, LFQ_TUNE_I("LFQ_COUNT_LEAN_LDS_PAD", 0)));
        x.count_shallow_lds_pad = (int)std::min(120000L, std::max(0L, LFQ_TUNE_I("LFQ_COUNT_SHALLOW_LDS_PAD", 0)));
        x.join_on_side = (int)LFQ_TUNE_I("LFQ_JOIN_ON_SIDE", 1);
        x.heavy_after_screen = (int)LFQ_TUNE_I("LFQ_HEAVY_AFTER_SCREEN", 1);
        (void)geti;
        (void)has;
        return x;
    }();
    return k;
}

namespace {

/* utils.h:42 */
inline double phred_to_prob(int q)
{
    return (q == INT_MAX) ? DBL_MIN : pow(10.0, -1.0 * q / 10.0);
}

/* utils.h:45: 80-bit log10, C truncation */
inline int prob_to_phred(long double p) { return (int)(-10.0 * log10l(p)); }

/* utils.h:46 */
inline int prob_to_phred_safe(double p) { return (p <= 0.0) ? INT_MAX : (int)(-10.0 * log10l(p)); }

/* utils.c:66-76 */
inline int eps_cmp(double a, double b)
{
    if (fabs(a - b) < DBL_EPSILON) {
        return 0;
    }
    return a < b ? -1 : (a > b ? 1 : 0);
}

struct IndexedP {
    double p;
    int64_t i;
};

/* the reference sorts with libc qsort + dbl_cmp (multtest.c:51-57); glibc's qsort is a stable
 * merge sort for these sizes, so a stable sort with the same comparator reproduces its order */
void sort_indexed(std::vector<IndexedP> &v)
{
    std::stable_sort(v.begin(), v.end(), [](const IndexedP &a, const IndexedP &b) { return eps_cmp(a.p, b.p) < 0; });
}

/* ---- Fisher's exact test: fet.c:13-99 = samtools 0.1.18 (r982:295) kfunc.c, Heng Li, MIT licence; see the file header ---- */

double log_choose(int n, int k)                             /* fet.c:13-17 */
{
    if (k == 0 || n == k) {
        return 0;
    }
    int sg;   /* lgamma_r: same value as lgamma (fet.c:16), without the shared signgam write */
    return lgamma_r(n + 1, &sg) - lgamma_r(k + 1, &sg) - lgamma_r(n - k + 1, &sg);
}

double hypergeom(int n11, int n1_, int n_1, int n)          /* fet.c:26-29 */
{
    return exp(log_choose(n1_, n11) + log_choose(n - n1_, n_1 - n11) - log_choose(n, n_1));
}

struct HyperAcc {
    int n11, n1_, n_1, n;
    double p;
};

/* fet.c:37-61: table probability, updated incrementally except at every 11th n11 */
double hypergeom_next(int n11, int n1_, int n_1, int n, HyperAcc &s)
{
    if (n1_ || n_1 || n) {
        s.n11 = n11;
        s.n1_ = n1_;
        s.n_1 = n_1;
        s.n = n;
    } else {
        if (n11 % 11 && n11 + s.n - s.n1_ - s.n_1) {
            if (n11 == s.n11 + 1) {
                s.p *= (double)(s.n1_ - s.n11) / n11 * (s.n_1 - s.n11) / (n11 + s.n - s.n1_ - s.n_1);
                s.n11 = n11;
                return s.p;
            }
            if (n11 == s.n11 - 1) {
                s.p *= (double)s.n11 / (s.n1_ - n11) * (s.n11 + s.n - s.n1_ - s.n_1) / (s.n_1 - n11);
                s.n11 = n11;
                return s.p;
            }
        }
        s.n11 = n11;
    }
    s.p = hypergeom(s.n11, s.n1_, s.n_1, s.n);
    return s.p;
}

}  // namespace

unsigned lfq_cpu_budget(void)
{
    static const unsigned budget = [] {
        unsigned n = std::thread::hardware_concurrency();
        n = n ? n : 1u;
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0) {
            const int a = CPU_COUNT(&set);
            if (a > 0) {
                n = std::min<unsigned>(n, (unsigned)a);
            }
        }
        /* cgroup v2: "<quota> <period>" or "max <period>"; v1: cpu.cfs_quota_us / cpu.cfs_period_us */
        long long quota = -1, period = -1;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32];
            if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) {
                quota = atoll(q);
            }
            fclose(f);
        } else {
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (fscanf(g, "%lld", &quota) != 1) quota = -1;
                fclose(g);
            }
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(g, "%lld", &period) != 1) period = -1;
                fclose(g);
            }
        }
        if (quota > 0 && period > 0) {
            n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
        }
        return std::max(1u, n);
    }();
    return budget;
}

namespace {

/* A small persistent pool for the host finishing step (strand-bias Fisher tests): spawning threads per
 * batch costs more than the work.  Threads are created on first use and parked on a condition variable. */
class LfqPool {
public:
    static LfqPool &instance()
    {
        static LfqPool p;
        return p;
    }
    int size() const { return (int)threads_.size(); }
    /* run `f` on `helpers` pool threads and on the caller; returns when all of them are done */
    void run(const std::function<void()> &f, int helpers)
    {
        helpers = std::max(0, std::min(helpers, size()));
        if (helpers == 0) {
            f();
            return;
        }
        /* one batch at a time -- and a caller that finds the pool taken (several host threads with a context each: the
         * genome runs) does its batch alone at once instead of queueing for helpers */
        std::unique_lock<std::mutex> call_lock(call_m_, std::try_to_lock);
        if (!call_lock.owns_lock()) {
            f();
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &f;
            want_ = helpers;
            pending_ = helpers;
            generation_++;
        }
        cv_start_.notify_all();
        f();
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [&] { return pending_ == 0; });
        job_ = nullptr;
    }

private:
    LfqPool()
    {
        unsigned hw = lfq_cpu_budget();
        const LfqKnobs &kn = lfq_knobs();
        hw = std::max(1u, hw / (unsigned)kn.local_world_size);   /* one process per GPU (torchrun): share the cores */
        int n = (int)std::min<unsigned>(hw > 1 ? hw - 1 : 0, 63u);
        if (kn.host_threads >= 0) {
            n = std::max(0, std::min(kn.host_threads - 1, 255));
        }
        for (int i = 0; i < n; i++) {
            threads_.emplace_back([this, i] { loop(i); });
        }
    }
    ~LfqPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            generation_++;
        }
        cv_start_.notify_all();
        for (auto &t : threads_) {
            t.join();
        }
    }
    void loop(int idx)
    {
        int seen = 0;
        for (;;) {
            const std::function<void()> *job = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_start_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) {
                    return;
                }
                if (idx >= want_) {
                    continue;           /* not needed for this batch */
                }
                job = job_;
            }
            (*job)();
            {
                std::lock_guard<std::mutex> lk(m_);
                pending_--;
            }
            cv_done_.notify_one();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_, call_m_;
    std::condition_variable cv_start_, cv_done_;
    const std::function<void()> *job_ = nullptr;
    int generation_ = 0, want_ = 0, pending_ = 0;
    bool stop_ = false;
};

}  // namespace

namespace {

struct SbKey {
    int32_t a, b, c, d;
    bool operator==(const SbKey &o) const { return a == o.a && b == o.b && c == o.c && d == o.d; }
};
struct SbKeyHash {
    size_t operator()(const SbKey &k) const
    {
        uint64_t h = (uint64_t)(uint32_t)k.a * 0x9E3779B97F4A7C15ull;
        h ^= ((uint64_t)(uint32_t)k.b + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
        h ^= ((uint64_t)(uint32_t)k.c + 0x165667B1ull) * 0x9E3779B97F4A7C15ull + (h << 6);
        h ^= ((uint64_t)(uint32_t)k.d + 0x27D4EB2Full) * 0xC2B2AE3D27D4EB4Full + (h >> 3);
        return (size_t)h;
    }
};

class LfqSbCache {
public:
    static LfqSbCache &instance()
    {
        static LfqSbCache c;
        return c;
    }
    /* results of one context's precompute; entries of other batches / contexts stay (content-addressed: a key
     * always maps to the same value) until the table gets large */
    void publish(std::vector<std::pair<SbKey, int>> &items)
    {
        std::lock_guard<std::mutex> lk(m_);
        if (map_.size() + items.size() > ((size_t)1 << 18)) {
            map_.clear();                       /* the cache never grows without bound */
        }
        map_.reserve(map_.size() + items.size());
        for (auto &it : items) {
            map_.emplace(it.first, it.second);
        }
    }
    bool lookup(const SbKey &k, int *sb)
    {
        std::lock_guard<std::mutex> lk(m_);
        return lookup_locked(k, sb);
    }
    /* a batch of lookups under one lock (the finish step does ~10^3 of them) */
    template <typename F>
    void with_lock(F f)
    {
        std::lock_guard<std::mutex> lk(m_);
        f();
    }
    bool lookup_locked(const SbKey &k, int *sb)
    {
        auto it = map_.find(k);
        if (it == map_.end()) {
            return false;
        }
        *sb = it->second;
        return true;
    }

private:
    std::mutex m_;
    std::unordered_map<SbKey, int, SbKeyHash> map_;
};

}  // namespace

void lfq_sb_precompute(const int32_t *tuples, int64_t n)
{
    std::vector<std::pair<SbKey, int>> items;
    items.reserve((size_t)std::max<int64_t>(n, 0));
    for (int64_t i = 0; i < n; i++) {
        const int32_t *t = tuples + 4 * i;
        if (t[0] | t[1] | t[2] | t[3]) {
            items.push_back({SbKey{t[0], t[1], t[2], t[3]}, 0});
        }
    }
    /* most expensive first (cost ~ range of the hypergeometric sum), one item at a time */
    std::sort(items.begin(), items.end(), [](const std::pair<SbKey, int> &x, const std::pair<SbKey, int> &y) {
        return (int64_t)x.first.c + x.first.d > (int64_t)y.first.c + y.first.d;
    });
    std::atomic<int64_t> next(0);
    const int64_t m = (int64_t)items.size();
    auto work = [&]() {
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= m) {
                break;
            }
            const SbKey &k = items[(size_t)i].first;
            items[(size_t)i].second = lfq_sb_phred(k.a, k.b, k.c, k.d);
        }
    };
    LfqPool::instance().run(work, (int)std::min<int64_t>(LfqPool::instance().size(), m / 2));
    LfqSbCache::instance().publish(items);
}

extern "C" {

int lfq_abi_version(void) { return LFQ_ABI_VERSION; }

const char *lfq_strerror(int status)
{
    switch (status) {
    case LFQ_OK: return "ok";
    case LFQ_ERR_INVALID: return "invalid argument";
    case LFQ_ERR_NO_DEVICE: return "no usable HIP device";
    case LFQ_ERR_NOMEM: return "out of memory";
    case LFQ_ERR_CAPACITY: return "output capacity too small";
    case LFQ_ERR_UNSUPPORTED: return "unsupported option (the reference rejects it too)";
    case LFQ_ERR_HIP: return "HIP runtime error";
    default: return "unknown error";
    }
}

/* snpcaller.c:627-651 */
void lfq_conf_init(lfq_conf *c)
{
    memset(c, 0, sizeof(*c));
    c->min_bq = 6;
    c->min_alt_bq = 6;
    c->def_alt_bq = 0;
    c->min_jq = 0;
    c->min_alt_jq = 0;
    c->def_alt_jq = 0;
    c->bonf_dynamic = 1;
    c->min_cov = 1;
    c->bonf_subst = 1;
    c->sig = 0.01;
    c->flag = LFQ_USE_MQ | LFQ_USE_BAQ;
    c->num_snv_tests = 0;
    c->bonf_indel = 1;
    c->num_indel_tests = 0;
    c->flag |= LFQ_USE_IDAQ;
    c->approx_threshold_n = -1;      /* snpcaller.c:650 */
}

/* expl() with the reference's clamp (snpcaller.c:1047-1059 / 1169-1188).  For
 * LFQ_PV_LOG_FECLAMP the device saw the reference's tail-sum exp() chain underflow, which
 * leaves FE_UNDERFLOW raised in the reference no matter what expl() itself does. */
long double lfq_pvalue_from_log(double logp, int status)
{
    if (status == LFQ_PV_NONE) {
        return LDBL_MAX;
    }
    if (status == LFQ_PV_UNDERFLOW) {
        return LDBL_MIN;        /* expl() of anything below -11399 underflows: clamp of snpcaller.c:1054-1055 */
    }
    errno = 0;
    feclearexcept(FE_ALL_EXCEPT);
    long double pv = expl(logp);
    const int errsv = errno;
    const bool bad = errsv || fetestexcept(FE_INVALID | FE_DIVBYZERO | FE_OVERFLOW | FE_UNDERFLOW)
                     || status == LFQ_PV_LOG_FECLAMP;
    if (bad) {
        pv = (pv < DBL_EPSILON) ? LDBL_MIN : LDBL_MAX;
    }
    return pv;
}

/* fet.c:62-99 */
double lfq_fisher_exact(int n11, int n12, int n21, int n22, double *left_out, double *right_out,
                        double *two_out)
{
    HyperAcc acc;
    const int row1 = n11 + n12, col1 = n11 + n21, tot = n11 + n12 + n21 + n22;
    const int hi = (col1 < row1) ? col1 : row1;
    int lo = row1 + col1 - tot;
    if (lo < 0) {
        lo = 0;
    }
    *two_out = *left_out = *right_out = 1.;
    if (lo == hi) {
        return 1.;
    }
    const double q = hypergeom_next(n11, row1, col1, tot, acc);
    int i, j;
    double left, right;
    double p = hypergeom_next(lo, 0, 0, 0, acc);
    for (left = 0., i = lo + 1; p < 0.99999999 * q; ++i) {
        left += p;
        p = hypergeom_next(i, 0, 0, 0, acc);
    }
    --i;
    if (p < 1.00000001 * q) {
        left += p;
    } else {
        --i;
    }
    p = hypergeom_next(hi, 0, 0, 0, acc);
    for (right = 0., j = hi - 1; p < 0.99999999 * q; --j) {
        right += p;
        p = hypergeom_next(j, 0, 0, 0, acc);
    }
    ++j;
    if (p < 1.00000001 * q) {
        right += p;
    } else {
        ++j;
    }
    *two_out = left + right;
    if (*two_out > 1.) {
        *two_out = 1.;
    }
    if (abs(i - n11) < abs(j - n11)) {
        right = 1. - left + q;
    } else {
        left = 1.0 - right + q;
    }
    *left_out = left;
    *right_out = right;
    return q;
}

/* lofreq_call.c:117-129 */
int lfq_sb_phred(int ref_fw, int ref_rv, int alt_fw, int alt_rv)
{
    if ((ref_fw + ref_rv) == 0 && (alt_fw == 0 || alt_rv == 0)) {
        return INT_MAX;
    }
    double l, r, two;
    (void)lfq_fisher_exact(ref_fw, ref_rv, alt_fw, alt_rv, &l, &r, &two);
    return prob_to_phred_safe(two);
}

/* lofreq_call.c:1523-1527: float / integer division, then log10l */
int lfq_snvqual_thresh(float sig, int64_t bonf_subst)
{
    int t = INT_MAX;
    if (bonf_subst) {
        t = prob_to_phred(sig / (long long)bonf_subst);
        if (t < 0) {
            t = 0;
        }
    }
    return t;
}

/* The tail of call_snvs (lofreq_call.c:817-871) for the columns the device did not prune. */
int lfq_finalize_pvals(const lfq_conf *conf, const lfq_col_pvals *pvals, int64_t n_pvals,
                       const int32_t *coverage_plp_or_null, const uint8_t *ref_base,
                       lfq_snv_record *records, int64_t records_capacity, int64_t *n_records)
{
    if (!conf || (!pvals && n_pvals > 0) || !n_records || n_pvals < 0) {
        return LFQ_ERR_INVALID;
    }
    const bool timing = lfq_knobs().timing != 0;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tf[6] = {now(), 0, 0, 0, 0, 0};
    std::vector<int64_t> order((size_t)n_pvals);
    for (int64_t i = 0; i < n_pvals; i++) {
        order[(size_t)i] = i;
    }
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return pvals[a].col < pvals[b].col; });

    /* expl() / log10l() on 80-bit values are the cost of this loop: p-values and QUALs first, in parallel */
    struct PvQ {
        long double pv[3];
        int qual[3];
    };
    std::vector<PvQ> pvq((size_t)n_pvals);
    {
        std::atomic<int64_t> nextp(0);
        auto workp = [&]() {
            for (;;) {
                const int64_t i0 = nextp.fetch_add(64);
                if (i0 >= n_pvals) {
                    break;
                }
                for (int64_t i = i0; i < std::min(n_pvals, i0 + 64); i++) {
                    const lfq_col_pvals &r = pvals[i];
                    for (int a = 0; a < 3; a++) {
                        pvq[(size_t)i].pv[a] = lfq_pvalue_from_log(r.logp[a], r.status[a]);
                        pvq[(size_t)i].qual[a] = (r.status[a] != LFQ_PV_NONE) ? prob_to_phred(pvq[(size_t)i].pv[a]) : 0;
                    }
                }
            }
        };
        LfqPool::instance().run(workp, n_pvals >= 256 ? std::min(LfqPool::instance().size(), 8) : 0);
    }

    tf[1] = now();
    static const char acgt[4] = {'A', 'C', 'G', 'T'};
    int64_t n_out = 0;
    for (int64_t oi = 0; oi < n_pvals; oi++) {
        const lfq_col_pvals &r = pvals[order[(size_t)oi]];
        const lfq_col_counts &cn = r.counts;
        const double bonf = (double)r.bonf;
        const long double *pv = pvq[(size_t)order[(size_t)oi]].pv;
        const int *quals = pvq[(size_t)order[(size_t)oi]].qual;
        int kmax = 0;
        for (int a = 0; a < 3; a++) {
            kmax = std::max(kmax, cn.alt_counts[a]);
        }
        /* snpcaller() returns all-LDBL_MAX if the most frequent allele is not significant
         * (snpcaller.c:1155) */
        bool main_ok = false;
        for (int a = 0; a < 3; a++) {
            if (cn.alt_counts[a] == kmax && kmax > 0 && r.status[a] != LFQ_PV_NONE) {
                main_ok = !(pv[a] * bonf > (double)conf->sig);
                break;
            }
        }
        if (!main_ok) {
            continue;
        }
        char refc = ref_base ? (char)ref_base[r.col] : (char)r.ref_base;      /* the record carries it */
        int ref_code = -1;
        for (int x = 0; x < 4; x++) {
            if (acgt[x] == refc) {
                ref_code = x;
            }
        }
        if (ref_code < 0) {
            continue;
        }
        const int cov = coverage_plp_or_null ? coverage_plp_or_null[r.col] : cn.coverage;
        int ai = 0;
        for (int x = 0; x < 4; x++) {
            if (x == ref_code) {
                continue;
            }
            const int a = ai++;
            if (pv[a] * bonf < conf->sig) {                 /* lofreq_call.c:832 */
                if (n_out >= records_capacity) {
                    *n_records = n_out;
                    return LFQ_ERR_CAPACITY;
                }
                lfq_snv_record &o = records[n_out++];
                memset(&o, 0, sizeof(o));
                o.col = r.col;
                o.pvalue = pv[a];
                o.qual = quals[a];                          /* lofreq_call.c:863 */
                o.dp = cov;
                o.alt_raw_count = cn.alt_raw_counts[a];     /* lofreq_call.c:835 */
                o.ref_fw = cn.ref_fw;                       /* lofreq_call.c:853-857 */
                o.ref_rv = cn.ref_rv;
                o.alt_fw = cn.alt_fw[a];
                o.alt_rv = cn.alt_raw_counts[a] - cn.alt_fw[a];
                o.hqa = cn.alt_counts[a];                   /* lofreq_call.c:860 */
                o.sb = 0;                                   /* filled below */
                o.ref = refc;
                o.alt = acgt[x];
            }
        }
    }
    /* Strand bias (report_var, lofreq_call.c:117-129): Fisher's exact test walks the hypergeometric
     * support with lgamma -- tens of microseconds per record at depth 1e4, the dominant host cost once
     * the columns themselves run on the GPU.  Records are independent: spread them over host threads. */
    {
        const int64_t n = n_out;
        std::atomic<int64_t> next(0);
        /* expensive tables were precomputed while the DP kernels ran (lfq_sb_precompute); the rest here */
        tf[2] = now();
        /* (a context waits for ITS OWN precompute before it gets here -- lfq_call_snvs_collect; other contexts'
         * batches in flight are none of this call's business, and a miss is simply computed below) */
        LfqSbCache &cache = LfqSbCache::instance();
        tf[3] = now();
        std::vector<int64_t> miss;
        cache.with_lock([&] {
            for (int64_t i = 0; i < n; i++) {
                lfq_snv_record &o = records[i];
                if (!cache.lookup_locked(SbKey{o.ref_fw, o.ref_rv, o.alt_fw, o.alt_rv}, &o.sb)) {
                    miss.push_back(i);
                }
            }
        });
        const int64_t nm = (int64_t)miss.size();
        auto work = [&]() {
            for (;;) {
                const int64_t j = next.fetch_add(1);    /* one record at a time: their costs differ by 100x */
                if (j >= nm) {
                    break;
                }
                lfq_snv_record &o = records[miss[(size_t)j]];
                o.sb = lfq_sb_phred(o.ref_fw, o.ref_rv, o.alt_fw, o.alt_rv);
            }
        };
        int64_t cost = 0;
        for (int64_t j = 0; j < nm; j++) {
            cost += records[miss[(size_t)j]].alt_fw + records[miss[(size_t)j]].alt_rv;
        }
        /* waking the pool costs more than a few cheap tables */
        const int helpers = cost < lfq_knobs().sb_par_min_cost ? 0 : (int)std::min<int64_t>(LfqPool::instance().size(), nm / 4);
        LfqPool::instance().run(work, helpers);
        tf[4] = now();
        if (timing) {
            fprintf(stderr, "[lfq timing] finalize: p-values+QUAL %.3f  records %.3f  wait for SB precompute %.3f  SB lookups %.3f ms (%ld misses of %ld)\n",
                    tf[1] - tf[0], tf[2] - tf[1], tf[3] - tf[2], tf[4] - tf[3], (long)nm, (long)n);
        }
    }
    *n_records = n_out;
    return LFQ_OK;
}

/* vcf_write_var + vcf_var_sprintf_info (vcf.c:469-497, 608-629) */
int lfq_format_snv_record(char *buf, int buflen, const char *chrom, int64_t pos0, const lfq_snv_record *rec,
                          const char *filter_or_null)
{
    const float af = rec->alt_raw_count / (float)rec->dp;   /* lofreq_call.c:835 */
    return snprintf(buf, (size_t)buflen, "%s\t%ld\t.\t%c\t%c\t%d\t%s\tDP=%d;AF=%f;SB=%d;DP4=%d,%d,%d,%d;HQA=%d\n",
                    chrom, (long)(pos0 + 1), rec->ref, rec->alt, rec->qual,
                    filter_or_null ? filter_or_null : ".", rec->dp, af, rec->sb, rec->ref_fw, rec->ref_rv,
                    rec->alt_fw, rec->alt_rv, rec->hqa);
}

int lfq_filter_indel_records(const lfq_indel_record *recs, int64_t n, int indelqual_thresh, int apply_defaults,
                             int32_t *keep)
{
    if (n < 0 || (n > 0 && (!recs || !keep))) {
        return LFQ_ERR_INVALID;
    }
    for (int64_t i = 0; i < n; i++) {
        bool ok = true;
        if (indelqual_thresh > 0 && recs[i].qual > -1 && recs[i].qual < indelqual_thresh) {
            ok = false;                                   /* apply_indelqual_threshold */
        }
        if (apply_defaults && recs[i].dp < 10) {
            ok = false;                                   /* DEFAULT_MIN_COV, lofreq_filter.c:210-236 */
        }
        keep[i] = ok ? 1 : 0;
    }
    return LFQ_OK;
}

int lfq_format_indel_record(char *buf, int buflen, const char *chrom, int64_t pos0, const char *ref,
                            const char *alt, int qual, int dp, float af, int sb, int ref_fw, int ref_rv,
                            int alt_fw, int alt_rv, int hrun, const char *filter_or_null)
{
    return snprintf(buf, (size_t)buflen, "%s\t%ld\t.\t%s\t%s\t%d\t%s\tDP=%d;AF=%f;SB=%d;DP4=%d,%d,%d,%d;INDEL;HRUN=%d\n",
                    chrom, (long)(pos0 + 1), ref, alt, qual, filter_or_null ? filter_or_null : ".", dp, af, sb,
                    ref_fw, ref_rv, alt_fw, alt_rv, hrun);
}

/* many records at once; returns the number of bytes the full text needs (written if it fits) */
int64_t lfq_format_vcf(char *buf, int64_t buflen, const char *chrom, const int64_t *pos0_or_null,
                       const lfq_snv_record *recs, int64_t n, const uint8_t *keep_or_null,
                       const char *filter_or_null)
{
    if (n >= 256) {
        /* text in parallel into fixed slots, then one compaction pass (snprintf with %f is the cost) */
        constexpr int SLOT = 320;
        std::vector<char> slots((size_t)n * SLOT);
        std::vector<int32_t> lens((size_t)n, 0);
        std::atomic<int64_t> next(0);
        std::atomic<int> bad(0);
        auto work = [&]() {
            for (;;) {
                const int64_t i0 = next.fetch_add(32);
                if (i0 >= n) {
                    break;
                }
                for (int64_t i = i0; i < std::min(n, i0 + 32); i++) {
                    if (keep_or_null && !keep_or_null[i]) {
                        continue;
                    }
                    const int64_t pos0 = pos0_or_null ? pos0_or_null[i] : recs[i].col;
                    const int len = lfq_format_snv_record(&slots[(size_t)i * SLOT], SLOT, chrom, pos0, &recs[i], filter_or_null);
                    if (len < 0 || len >= SLOT) {
                        bad.store(1);
                    }
                    lens[(size_t)i] = len;
                }
            }
        };
        LfqPool::instance().run(work, std::min(LfqPool::instance().size(), 8));
        if (!bad.load()) {
            int64_t used = 0;
            for (int64_t i = 0; i < n; i++) {
                const int len = lens[(size_t)i];
                if (buf && len > 0 && used + len <= buflen) {
                    memcpy(buf + used, &slots[(size_t)i * SLOT], (size_t)len);
                }
                used += len;
            }
            return used;
        }
        /* a line did not fit its slot (very long contig name): fall through to the serial path */
    }
    int64_t used = 0;
    char line[512];
    for (int64_t i = 0; i < n; i++) {
        if (keep_or_null && !keep_or_null[i]) {
            continue;
        }
        const int64_t pos0 = pos0_or_null ? pos0_or_null[i] : recs[i].col;
        const int len = lfq_format_snv_record(line, (int)sizeof(line), chrom, pos0, &recs[i], filter_or_null);
        if (len < 0) {
            return LFQ_ERR_INVALID;
        }
        if (buf && used + len <= buflen) {
            memcpy(buf + used, line, (size_t)len);
        }
        used += len;
    }
    return used;
}

/* multtest.c:66-81 */
void lfq_bonf_corr(double *pvals, int64_t n, int64_t num_tests)
{
    const int64_t fac = (num_tests < 1) ? n : num_tests;
    for (int64_t i = 0; i < n; i++) {
        pvals[i] *= fac;
    }
}

/* multtest.c:91-136 */
void lfq_holm_bonf_corr(double *pvals, int64_t n, double alpha, int64_t num_tests)
{
    std::vector<IndexedP> ix((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        ix[(size_t)i] = {pvals[i], i};
    }
    sort_indexed(ix);
    int64_t lp = (num_tests < 1) ? n : num_tests;
    double seen = n ? ix[0].p : 0.0;
    for (int64_t i = 0; i < n; i++) {
        if (eps_cmp(ix[(size_t)i].p, seen) != 0) {
            lp = (num_tests < 1) ? n - i : num_tests - i;
            seen = ix[(size_t)i].p;
        }
        const double tp = ix[(size_t)i].p * 1. / lp;
        if (eps_cmp(tp, alpha) < 0) {
            pvals[ix[(size_t)i].i] = ix[(size_t)i].p * lp;
        }
    }
}

/* multtest.c:148-189 (Benjamini-Hochberg with the reference's float division) */
int64_t lfq_fdr(const double *pvals, int64_t n, double alpha, int64_t num_tests, int64_t *rejected_idx)
{
    std::vector<IndexedP> ix((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        ix[(size_t)i] = {pvals[i], i};
    }
    sort_indexed(ix);
    const int64_t m = (num_tests < 1) ? n : num_tests;
    int64_t nrej = 0;
    for (int64_t i = n; i > 0; i--) {
        if (ix[(size_t)(i - 1)].p < (alpha * i / (float)m)) {
            nrej = i;
            break;
        }
    }
    if (rejected_idx) {
        for (int64_t i = 0; i < nrej; i++) {
            rejected_idx[i] = ix[(size_t)i].i;
        }
    }
    return nrej;
}

/* ---- `lofreq uniq`, default mode: the binomial test (binom.c:52-69 -> cdflib90's cdfbin, which = 1) ---------------
 * P(X <= k) for X ~ Binomial(n, pr).  The reference goes through the regularised incomplete beta function (cumbin,
 * dcdflib.c:4966-5031 -> bratio, TOMS 708); here the binomial probabilities themselves are summed in 80-bit
 * arithmetic, from the largest term of the wanted tail outwards with the exact term ratios, until the terms no longer
 * register: below the mode the lower tail directly, above it 1 - upper tail.  Agrees with the reference's compiled
 * cdflib to ~1e-13 relative (tests/test_uniq.py); `status` gets cdfbin's code (0, -5 n <= 0, -4 k outside [0, n],
 * -6 pr outside [0, 1]). */
double lfq_binom_cdf(int n, int k, double pr, int *status_or_null)
{
    int st = 0;
    double out = 0.0;
    if (n <= 0) {
        st = -5;
    } else if (k < 0 || k > n) {
        st = -4;
    } else if (!(pr >= 0.0 && pr <= 1.0)) {
        st = -6;
    } else if (k >= n || pr <= 0.0) {
        out = 1.0;                                  /* cumbin: s >= xn; cumbet: x <= 0 */
    } else if (pr >= 1.0) {
        out = 0.0;                                  /* all mass at n > k */
    } else {
        const long double N = n, P = pr, Q = 1.0L - P, odds = P / Q;
        auto log_term = [&](int i) {
            const long double I = i;
            return lgammal(N + 1.0L) - lgammal(I + 1.0L) - lgammal(N - I + 1.0L) + I * logl(P) + (N - I) * log1pl(-P);
        };
        const int mode = (int)floorl((N + 1.0L) * P);
        if (k < mode) {
            /* lower tail, terms fall away from i = k downwards: t(i-1) / t(i) = i / (n - i + 1) / odds */
            long double t = expl(log_term(k)), sum = t;
            for (int i = k; i > 0 && t > sum * 1e-25L; i--) {
                t *= (long double)i / (N - (long double)i + 1.0L) / odds;
                sum += t;
            }
            out = (double)sum;
        } else {
            /* upper tail from i = k + 1 upwards: t(i+1) / t(i) = (n - i) / (i + 1) * odds */
            long double t = expl(log_term(k + 1)), sum = t;
            for (int i = k + 1; i < n && t > sum * 1e-25L; i++) {
                t *= (N - (long double)i) / ((long double)i + 1.0L) * odds;
                sum += t;
            }
            const long double r = 1.0L - sum;
            out = (double)(r < 0.0L ? 0.0L : r);
        }
    }
    if (status_or_null) {
        *status_or_null = st;
    }
    return st ? 0.0 : out;
}

/* apply_uniq_filter_mtc (lofreq_uniq.c:140-206): PHREDQUAL_TO_PROB of the integer UQ values (no UQ tag = 0), bonf /
 * holm / fdr over ntests (0 = the number of variants), filtered when the corrected value exceeds alpha.
 * mtc_type as in multtest.h: 1 bonf, 2 holm, 3 fdr.  pass[i] = 1: the variant keeps PASS. */
int lfq_uniq_mtc(const int32_t *uq, int64_t n, int mtc_type, double alpha, int64_t ntests, uint8_t *pass)
{
    if ((!uq || !pass) && n > 0) {
        return LFQ_ERR_INVALID;
    }
    if (mtc_type < 1 || mtc_type > 3 || n < 0) {
        return LFQ_ERR_INVALID;
    }
    if (!ntests) {
        ntests = n;
    }
    std::vector<double> pr((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        pr[(size_t)i] = phred_to_prob(uq[i] < 0 ? 0 : uq[i]);
    }
    if (mtc_type == 1) {
        lfq_bonf_corr(pr.data(), n, ntests);
    } else if (mtc_type == 2) {
        lfq_holm_bonf_corr(pr.data(), n, alpha, ntests);
    } else {
        std::vector<int64_t> idx((size_t)std::max<int64_t>(n, 1));
        const int64_t nrej = lfq_fdr(pr.data(), n, alpha, ntests, idx.data());
        for (int64_t i = 0; i < nrej; i++) {
            pr[(size_t)idx[(size_t)i]] = -1;
        }
    }
    for (int64_t i = 0; i < n; i++) {
        pass[i] = pr[(size_t)i] > alpha ? 0 : 1;
    }
    return LFQ_OK;
}

/* `lofreq filter` as `lofreq call` runs it on its own output (lofreq_call.c:1506-1538):
 * SNV QUAL threshold (lofreq_filter.c:313-323) and, unless --no-defaults, DP >= 10
 * (lofreq_filter.c:1095-1097, 270-305) and the strand-bias FDR filter, alpha 0.001, with the
 * "alt mostly on one strand" compound rule (lofreq_filter.c:57, 210-236, 582-677, 1089-1094). */
int lfq_filter_records(const lfq_snv_record *records, int64_t n, int snvqual_thresh, int apply_defaults,
                       uint8_t *keep)
{
    if ((!records && n > 0) || !keep || n < 0) {
        return LFQ_ERR_INVALID;
    }
    for (int64_t i = 0; i < n; i++) {
        keep[i] = 1;
    }
    if (apply_defaults && n > 0) {
        const double alpha = 0.001;
        std::vector<double> sbp((size_t)n);
        std::vector<int64_t> rej((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            sbp[(size_t)i] = phred_to_prob(records[i].sb);            /* lofreq_filter.c:611 */
        }
        const int64_t nrej = lfq_fdr(sbp.data(), n, alpha, n, rej.data());
        for (int64_t i = 0; i < nrej; i++) {
            const lfq_snv_record &r = records[rej[(size_t)i]];
            const float ratio = std::max(r.alt_fw, r.alt_rv) / (float)(r.alt_fw + r.alt_rv);   /* :227 */
            if (ratio > 0.85) {
                keep[rej[(size_t)i]] = 0;
            }
        }
        for (int64_t i = 0; i < n; i++) {
            if (records[i].dp < 10) {
                keep[i] = 0;
            }
        }
    }
    if (snvqual_thresh) {
        for (int64_t i = 0; i < n; i++) {
            if (records[i].qual > -1 && records[i].qual < snvqual_thresh) {   /* lofreq_filter.c:319 */
                keep[i] = 0;
            }
        }
    }
    return LFQ_OK;
}

#define LFQ_HOST_TRY(expr) do { const int rc_ = (expr); if (rc_ != LFQ_OK) return rc_; } while (0)
/* ---- `lofreq filter`, every mode (lofreq_filter.c) ------------------------------------------------------------------ */
void lfq_filter_conf_init(lfq_filter_conf *c)
{
    if (!c) {
        return;
    }
    memset(c, 0, sizeof(*c));
    c->dp_min = c->dp_max = -1;                      /* lofreq_filter.c:1093-1097 */
    c->af_min = c->af_max = -1;
    c->sb_alpha = c->snvqual_alpha = c->indelqual_alpha = 0.01;     /* DEFAULT_SIG, defaults.h:68 */
}

void lfq_filter_conf_defaults(lfq_filter_conf *c)
{
    if (!c) {
        return;
    }
    if (c->sb_mtc_type == LFQ_MTC_NONE && !c->sb_thresh) {           /* lofreq_filter.c:1184-1197 */
        c->sb_mtc_type = LFQ_MTC_FDR;
        c->sb_alpha = 0.001;
    }
    if (c->dp_min < 0) {
        c->dp_min = 10;
    }
}

namespace {

/* the three apply_*_filter_mtc functions (lofreq_filter.c:376-677) are one: correct the probabilities of the selected
 * variants, report which come out below alpha */
int filter_mtc(std::vector<double> &p, int mtc_type, double alpha, int64_t *ntests, std::vector<uint8_t> &signif)
{
    const int64_t n = (int64_t)p.size();
    signif.assign((size_t)n, 0);
    if (n <= 0) {
        return LFQ_OK;
    }
    if (!*ntests) {
        *ntests = n;                                 /* :407-409; a smaller predefined count only earns a warning there */
    }
    if (mtc_type == LFQ_MTC_BONF) {
        lfq_bonf_corr(p.data(), n, *ntests);
    } else if (mtc_type == LFQ_MTC_HOLMBONF) {
        lfq_holm_bonf_corr(p.data(), n, alpha, *ntests);
    } else if (mtc_type == LFQ_MTC_FDR) {
        std::vector<int64_t> rej((size_t)n);
        const int64_t nrej = lfq_fdr(p.data(), n, alpha, *ntests, rej.data());
        std::fill(p.begin(), p.end(), DBL_MAX);      /* :437-445 */
        for (int64_t i = 0; i < nrej; i++) {
            p[(size_t)rej[(size_t)i]] = -1;
        }
    } else {
        return LFQ_ERR_INVALID;
    }
    for (int64_t i = 0; i < n; i++) {
        signif[(size_t)i] = p[(size_t)i] < alpha ? 1 : 0;
    }
    return LFQ_OK;
}

bool alt_mostly_on_one_strand(const lfq_filter_var &v)         /* :214-236 */
{
    const float ratio = std::max(v.alt_fw, v.alt_rv) / (float)(v.alt_fw + v.alt_rv);
    return ratio > 0.85;
}

const char *mtc_name(int t)                          /* mtc_str, multtest.c:205-214 */
{
    return t == LFQ_MTC_BONF ? "bonf" : (t == LFQ_MTC_HOLMBONF ? "holmbonf" : (t == LFQ_MTC_FDR ? "fdr" : "none"));
}

}  // namespace

int lfq_filter_vars(lfq_filter_conf *c, const lfq_filter_var *vars, int64_t n, uint32_t *fail)
{
    if (!c || n < 0 || (n > 0 && (!vars || !fail))) {
        return LFQ_ERR_INVALID;
    }
    /* main_filter's checks (:1177-1227) */
    if ((c->only_indels && c->only_snvs) || (c->dp_max > 0 && c->dp_max < c->dp_min)
        || (c->af_max > 0 && c->af_max < c->af_min) || c->af_max > 1.0
        || (c->sb_thresh && c->sb_mtc_type != LFQ_MTC_NONE) || (c->snvqual_thresh && c->snvqual_mtc_type != LFQ_MTC_NONE)
        || (c->indelqual_thresh && c->indelqual_mtc_type != LFQ_MTC_NONE)) {
        return LFQ_ERR_INVALID;
    }
    /* first pass (:1246-1281): the corrections see EVERY variant of the file, also the ones --only-snvs / --only-indels
     * drop afterwards; order sb, indel quality, SNV quality */
    std::vector<uint8_t> sb_sig, iq_sig, sq_sig;
    std::vector<int64_t> sb_ix, iq_ix, sq_ix;
    /* A quirk of the reference that decides FILTER columns: apply_af_filter tests `errno == ERANGE` after its strtof
     * without clearing errno first (:252-261), and switches AF filtering off for the rest of the run if it is set.  The
     * first pass leaves it set whenever one of its pow(10, -q / 10) underflows (QUAL or SB beyond ~3070: glibc reports
     * the subnormal / zero result) -- with a correction requested and such a variant in the file, -a / -A do nothing.
     * Same libm here, so the same calls are watched. */
    errno = 0;
    auto qual_of = [](const lfq_filter_var &v) { return v.qual == -1 ? INT_MAX : v.qual; };      /* :824-831 */
    if (c->sb_mtc_type != LFQ_MTC_NONE) {
        std::vector<double> p;
        for (int64_t i = 0; i < n; i++) {
            if (!c->sb_incl_indels && vars[i].is_indel) {
                continue;
            }
            p.push_back(phred_to_prob(vars[i].sb));
            sb_ix.push_back(i);
        }
        LFQ_HOST_TRY(filter_mtc(p, c->sb_mtc_type, c->sb_alpha, &c->sb_ntests, sb_sig));
    }
    if (c->indelqual_mtc_type != LFQ_MTC_NONE) {
        std::vector<double> p;
        for (int64_t i = 0; i < n; i++) {
            if (vars[i].is_indel) {
                p.push_back(phred_to_prob(qual_of(vars[i])));
                iq_ix.push_back(i);
            }
        }
        LFQ_HOST_TRY(filter_mtc(p, c->indelqual_mtc_type, c->indelqual_alpha, &c->indelqual_ntests, iq_sig));
    }
    if (c->snvqual_mtc_type != LFQ_MTC_NONE) {
        std::vector<double> p;
        for (int64_t i = 0; i < n; i++) {
            if (!vars[i].is_indel) {
                p.push_back(phred_to_prob(qual_of(vars[i])));
                sq_ix.push_back(i);
            }
        }
        LFQ_HOST_TRY(filter_mtc(p, c->snvqual_mtc_type, c->snvqual_alpha, &c->snvqual_ntests, sq_sig));
    }
    const bool af_off = errno == ERANGE;
    /* second pass (:1323-1386) */
    for (int64_t i = 0; i < n; i++) {
        const lfq_filter_var &v = vars[i];
        uint32_t f = 0;
        if ((c->only_snvs && v.is_indel) || (c->only_indels && !v.is_indel)) {
            fail[i] = LFQ_FILT_DROPPED;
            continue;
        }
        if (!af_off && c->af_min > 0.0 && v.af < c->af_min) f |= LFQ_FILT_AF_MIN;          /* apply_af_filter */
        if (!af_off && c->af_max > 0.0 && v.af > c->af_max) f |= LFQ_FILT_AF_MAX;
        if (c->dp_min > 0 && v.dp < c->dp_min) f |= LFQ_FILT_DP_MIN;            /* apply_dp_filter */
        if (c->dp_max > 0 && v.dp > c->dp_max) f |= LFQ_FILT_DP_MAX;
        if (!v.is_indel && c->snvqual_thresh && v.qual > -1 && v.qual < c->snvqual_thresh) f |= LFQ_FILT_SNVQUAL;
        if (v.is_indel && c->indelqual_thresh && v.qual > -1 && v.qual < c->indelqual_thresh) f |= LFQ_FILT_INDELQUAL;
        if (c->sb_thresh && (!v.is_indel || c->sb_incl_indels) && v.sb > c->sb_thresh
            && (c->sb_no_compound || alt_mostly_on_one_strand(v))) {
            f |= LFQ_FILT_SB;                                                   /* apply_sb_threshold */
        }
        fail[i] = f;
    }
    /* quality corrections: a variant that is NOT significant is filtered (:1337-1339, 1347-1349) */
    for (size_t k = 0; k < sq_ix.size(); k++) {
        if (!sq_sig[k] && fail[sq_ix[k]] != LFQ_FILT_DROPPED) fail[sq_ix[k]] |= LFQ_FILT_SNVQUAL;
    }
    for (size_t k = 0; k < iq_ix.size(); k++) {
        if (!iq_sig[k] && fail[iq_ix[k]] != LFQ_FILT_DROPPED) fail[iq_ix[k]] |= LFQ_FILT_INDELQUAL;
    }
    /* strand bias: the other way round -- significant bias, and the compound rule, filters (:660-666, 1361-1367) */
    for (size_t k = 0; k < sb_ix.size(); k++) {
        const int64_t i = sb_ix[k];
        if (sb_sig[k] && fail[i] != LFQ_FILT_DROPPED && (c->sb_no_compound || alt_mostly_on_one_strand(vars[i]))) {
            fail[i] |= LFQ_FILT_SB;
        }
    }
    return LFQ_OK;
}

int lfq_filter_id(const lfq_filter_conf *c, uint32_t bit, char *buf, int buflen)
{
    if (!c || !buf || buflen < 1) {
        return 0;
    }
    int k = 0;
    buf[0] = 0;
    switch (bit) {                                    /* cfg_filter_to_vcf_header, :682-786 */
    case LFQ_FILT_AF_MIN: if (c->af_min > 0) k = snprintf(buf, (size_t)buflen, "min_af_%f", c->af_min); break;
    case LFQ_FILT_AF_MAX: if (c->af_max > 0) k = snprintf(buf, (size_t)buflen, "max_af_%f", c->af_max); break;
    case LFQ_FILT_DP_MIN: if (c->dp_min > 0) k = snprintf(buf, (size_t)buflen, "min_dp_%d", c->dp_min); break;
    case LFQ_FILT_DP_MAX: if (c->dp_max > 0) k = snprintf(buf, (size_t)buflen, "max_dp_%d", c->dp_max); break;
    case LFQ_FILT_SB:
        if (c->sb_thresh > 0) k = snprintf(buf, (size_t)buflen, "max_sb_%d", c->sb_thresh);
        else if (c->sb_mtc_type != LFQ_MTC_NONE) k = snprintf(buf, (size_t)buflen, "sb_%s", mtc_name(c->sb_mtc_type));
        break;
    case LFQ_FILT_SNVQUAL:
        if (c->snvqual_thresh > 0) k = snprintf(buf, (size_t)buflen, "min_snvqual_%d", c->snvqual_thresh);
        else if (c->snvqual_mtc_type != LFQ_MTC_NONE) k = snprintf(buf, (size_t)buflen, "snvqual_%s", mtc_name(c->snvqual_mtc_type));
        break;
    case LFQ_FILT_INDELQUAL:
        if (c->indelqual_thresh > 0) k = snprintf(buf, (size_t)buflen, "min_indelqual_%d", c->indelqual_thresh);
        else if (c->indelqual_mtc_type != LFQ_MTC_NONE) k = snprintf(buf, (size_t)buflen, "indelqual_%s", mtc_name(c->indelqual_mtc_type));
        break;
    default: break;
    }
    return k < 0 ? 0 : (k >= buflen ? buflen - 1 : k);
}

int lfq_filter_string(const lfq_filter_conf *c, uint32_t f, char *buf, int buflen)
{
    if (!c || !buf || buflen < 5) {
        return 0;
    }
    /* the order main_filter applies them in: af, dp, quality, sb (:1323-1370) */
    static const uint32_t order[] = {LFQ_FILT_AF_MIN, LFQ_FILT_AF_MAX, LFQ_FILT_DP_MIN, LFQ_FILT_DP_MAX, LFQ_FILT_SNVQUAL,
                                     LFQ_FILT_INDELQUAL, LFQ_FILT_SB};
    int used = 0;
    buf[0] = 0;
    for (uint32_t b : order) {
        if (!(f & b)) {
            continue;
        }
        char id[96];
        const int k = lfq_filter_id(c, b, id, (int)sizeof(id));
        if (k <= 0 || used + k + 2 > buflen) {
            continue;
        }
        if (used) {
            buf[used++] = ';';
        }
        memcpy(buf + used, id, (size_t)k + 1);
        used += k;
    }
    if (!used) {
        used = snprintf(buf, (size_t)buflen, "PASS");
    }
    return used;
}

int lfq_filter_header_lines(const lfq_filter_conf *c, char *buf, int buflen)
{
    if (!c || !buf || buflen < 1) {
        return 0;
    }
    std::string s;
    char id[96], line[512];
    auto add = [&](const char *fmt, auto... a) {
        snprintf(line, sizeof(line), fmt, a...);
        s += line;
    };
    if (lfq_filter_id(c, LFQ_FILT_AF_MIN, id, sizeof(id))) add("##FILTER=<ID=%s,Description=\"Minimum allele frequency %f\">\n", id, c->af_min);
    if (lfq_filter_id(c, LFQ_FILT_AF_MAX, id, sizeof(id))) add("##FILTER=<ID=%s,Description=\"Maximum allele frequency %f\">\n", id, c->af_max);
    if (lfq_filter_id(c, LFQ_FILT_DP_MIN, id, sizeof(id))) add("##FILTER=<ID=%s,Description=\"Minimum Coverage %d\">\n", id, c->dp_min);
    if (lfq_filter_id(c, LFQ_FILT_DP_MAX, id, sizeof(id))) add("##FILTER=<ID=%s,Description=\"Maximum Coverage %d\">\n", id, c->dp_max);
    if (lfq_filter_id(c, LFQ_FILT_SB, id, sizeof(id))) {
        if (c->sb_thresh > 0) add("##FILTER=<ID=%s,Description=\"Maximum Strand-Bias Phred %d\">\n", id, c->sb_thresh);
        else add("##FILTER=<ID=%s,Description=\"Strand-Bias Multiple Testing Correction: %s corr. pvalue > %f\">\n", id, mtc_name(c->sb_mtc_type), c->sb_alpha);
    }
    if (lfq_filter_id(c, LFQ_FILT_SNVQUAL, id, sizeof(id))) {
        if (c->snvqual_thresh > 0) add("##FILTER=<ID=%s,Description=\"Minimum SNV Quality (Phred) %d\">\n", id, c->snvqual_thresh);
        else add("##FILTER=<ID=%s,Description=\"SNV Quality Multiple Testing Correction: %s corr. pvalue < %f\">\n", id, mtc_name(c->snvqual_mtc_type), c->snvqual_alpha);
    }
    if (lfq_filter_id(c, LFQ_FILT_INDELQUAL, id, sizeof(id))) {
        if (c->indelqual_thresh > 0) add("##FILTER=<ID=%s,Description=\"Minimum Indel Quality (Phred) %d\">\n", id, c->indelqual_thresh);
        else add("##FILTER=<ID=%s,Description=\"Indel Quality Multiple Testing Correction: %s corr. pvalue < %f\">\n", id, mtc_name(c->indelqual_mtc_type), c->indelqual_alpha);
    }
    const int k = (int)std::min<size_t>(s.size(), (size_t)buflen - 1);
    memcpy(buf, s.data(), (size_t)k);
    buf[k] = 0;
    return (int)s.size();
}

static float af_through_text(float af)               /* the filter reads AF back from the "%f" text (apply_af_filter, :252) */
{
    char t[64];
    snprintf(t, sizeof(t), "%f", af);
    return strtof(t, nullptr);
}

void lfq_filter_var_from_snv(const lfq_snv_record *r, lfq_filter_var *o)
{
    memset(o, 0, sizeof(*o));
    o->is_indel = 0;
    o->qual = r->qual;
    o->dp = r->dp;
    o->sb = r->sb;
    o->alt_fw = r->alt_fw;
    o->alt_rv = r->alt_rv;
    o->af = af_through_text(r->alt_raw_count / (float)r->dp);     /* report_var's AF, lofreq_call.c:835 */
}

void lfq_filter_var_from_indel(const lfq_indel_record *r, lfq_filter_var *o)
{
    memset(o, 0, sizeof(*o));
    o->is_indel = 1;
    o->qual = r->qual;
    o->dp = r->dp;
    o->sb = r->sb;
    o->alt_fw = r->alt_fw;
    o->alt_rv = r->alt_rv;
    o->af = af_through_text(r->af);
}

}  // extern "C"
