/*
 * lfq_shard.hip -- N processes, one per GPU: which device a worker takes, and the exchange of a sharded run
 * (test counts all-gathered, Bonferroni factors rebased, records gathered) over RCCL by dlopen or a caller's all-gather.
 */
#include "lfq_ctx.h"

#include <sys/mman.h>
#include <time.h>

extern "C" {

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
        /* a launcher-supplied host transport (MPI, files, gloo, a test double) instead of RCCL */
        return g_host_allgather(g_host_allgather_user, world, rank, mine, all, bytes) == 0 ? LFQ_OK : LFQ_ERR_HIP;
    }
    if (!comm) {
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

/* ---- the record gather in two halves: started (nothing waits for the device), collected a step later ---------------------
 * A handle owns its staging buffers (device + pinned host), a high-priority stream of its own and an event; handles go back to
 * a process-wide free list, so that a pipelined caller allocates nothing per step (hipMalloc / hipHostMalloc synchronise). */
struct lfq_shard_gather {
    int device = -1;
    int world = 0;
    size_t piece = 0, padded = 0;
    size_t cap_dev = 0, cap_host = 0;           /* bytes: d_buf holds (world + 1) * padded, h_buf the same */
    uint8_t *d_buf = nullptr, *h_buf = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;
    bool on_device = false, want_all = false;
    std::vector<uint8_t> host_all;              /* host transport / one process: the gathered pieces */
};

namespace {
std::mutex g_gather_m;
std::vector<lfq_shard_gather *> g_gather_free;

lfq_shard_gather *gather_acquire(int device)
{
    std::lock_guard<std::mutex> lk(g_gather_m);
    for (size_t i = 0; i < g_gather_free.size(); i++) {
        if (g_gather_free[i]->device == device) {
            lfq_shard_gather *g = g_gather_free[i];
            g_gather_free.erase(g_gather_free.begin() + (long)i);
            return g;
        }
    }
    lfq_shard_gather *g = new (std::nothrow) lfq_shard_gather();
    if (g) {
        g->device = device;
    }
    return g;
}

void gather_release(lfq_shard_gather *g)
{
    std::lock_guard<std::mutex> lk(g_gather_m);
    g_gather_free.push_back(g);
}
}  // namespace

int lfq_shard_gather_start(lfq_ctx *c, void *comm, int world, int rank, const void *piece, int64_t piece_bytes, int want_all,
                           lfq_shard_gather **out)
{
    if (!out || world < 1 || rank < 0 || rank >= world || piece_bytes <= 0 || !piece) {
        return LFQ_ERR_INVALID;
    }
    *out = nullptr;
    const bool rccl = comm != nullptr;
    if (rccl && (!c || !lfq_rccl_allgather())) {
        return LFQ_ERR_UNSUPPORTED;
    }
    if (!rccl && world > 1 && !g_host_allgather) {
        return LFQ_ERR_INVALID;
    }
    lfq_shard_gather *g = gather_acquire(rccl ? c->device : -1);
    if (!g) {
        return LFQ_ERR_NOMEM;
    }
    g->world = world;
    g->piece = (size_t)piece_bytes;
    g->want_all = want_all != 0;
    g->on_device = rccl;
    if (!rccl) {
        /* a host transport (or one process): the collective is the caller's blocking all-gather, done here */
        g->host_all.resize((size_t)world * g->piece);
        int rc = LFQ_OK;
        if (world == 1) {
            memcpy(g->host_all.data(), piece, g->piece);
        } else {
            rc = g_host_allgather(g_host_allgather_user, world, rank, piece, g->host_all.data(), g->piece) == 0 ? LFQ_OK : LFQ_ERR_HIP;
        }
        if (rc != LFQ_OK) {
            gather_release(g);
            return rc;
        }
        *out = g;
        return LFQ_OK;
    }
    auto fail = [&](int rc) {
        gather_release(g);
        return rc;
    };
    if (hipSetDevice(c->device) != hipSuccess) {
        return fail(LFQ_ERR_HIP);
    }
    g->padded = (g->piece + 255) / 256 * 256;
    const size_t need = g->padded * (size_t)(world + 1);
    if (!g->stream) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);           /* hi = the numerically lowest = the highest priority */
        if (hipStreamCreateWithPriority(&g->stream, hipStreamNonBlocking, hi) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev, hipEventDisableTiming) != hipSuccess) {
            return fail(LFQ_ERR_HIP);
        }
    }
    if (g->cap_dev < need) {
        if (g->d_buf) {
            (void)hipFree(g->d_buf);
            g->d_buf = nullptr;
        }
        if (hipMalloc((void **)&g->d_buf, need) != hipSuccess) {
            g->cap_dev = 0;
            return fail(LFQ_ERR_NOMEM);
        }
        g->cap_dev = need;
    }
    if (g->cap_host < need) {
        if (g->h_buf) {
            (void)hipHostFree(g->h_buf);
            g->h_buf = nullptr;
        }
        if (hipHostMalloc((void **)&g->h_buf, need, hipHostMallocDefault) != hipSuccess) {
            g->cap_host = 0;
            return fail(LFQ_ERR_NOMEM);
        }
        g->cap_host = need;
    }
    uint8_t *h_send = g->h_buf, *h_recv = g->h_buf + g->padded;
    uint8_t *d_send = g->d_buf, *d_recv = g->d_buf + g->padded;
    memcpy(h_send, piece, g->piece);
    if (hipMemcpyAsync(d_send, h_send, g->padded, hipMemcpyHostToDevice, g->stream) != hipSuccess) {
        return fail(LFQ_ERR_HIP);
    }
    if (lfq_rccl_allgather()(d_send, d_recv, g->padded, /* ncclUint8 */ 1, comm, g->stream) != 0) {
        return fail(LFQ_ERR_HIP);
    }
    if (g->want_all &&
        hipMemcpyAsync(h_recv, d_recv, g->padded * (size_t)world, hipMemcpyDeviceToHost, g->stream) != hipSuccess) {
        return fail(LFQ_ERR_HIP);
    }
    if (hipEventRecord(g->ev, g->stream) != hipSuccess) {
        return fail(LFQ_ERR_HIP);
    }
    *out = g;
    return LFQ_OK;
}

int lfq_shard_gather_wait(lfq_shard_gather *g, void *all, int64_t all_bytes)
{
    if (!g) {
        return LFQ_ERR_INVALID;
    }
    int rc = LFQ_OK;
    const size_t total = (size_t)g->world * g->piece;
    if (all && (all_bytes < 0 || (size_t)all_bytes < total)) {
        rc = LFQ_ERR_CAPACITY;                  /* (the collective is still waited for: the buffers go back to the pool) */
    }
    if (g->on_device) {
        if (hipEventSynchronize(g->ev) != hipSuccess) {
            rc = LFQ_ERR_HIP;
        } else if (all && rc == LFQ_OK) {
            if (!g->want_all) {
                rc = LFQ_ERR_INVALID;
            } else {
                const uint8_t *h_recv = g->h_buf + g->padded;
                for (int r = 0; r < g->world; r++) {
                    memcpy((uint8_t *)all + (size_t)r * g->piece, h_recv + (size_t)r * g->padded, g->piece);
                }
            }
        }
    } else if (all && rc == LFQ_OK) {
        memcpy(all, g->host_all.data(), total);
    }
    gather_release(g);
    return rc;
}

int lfq_shard_set_host_allgather(lfq_host_allgather_fn fn, void *user)
{
    g_host_allgather = fn;
    g_host_allgather_user = user;
    return LFQ_OK;
}

/* ---- a host transport of the library's own for the ranks of ONE node: all-gather through POSIX shared memory ---------------
 * The per-step test counts are a few host integers per rank; over a loopback TCP ring (gloo) eight ranks take ~2 ms for
 * them, a collective staged through the GPU waits for wave slots behind the count kernels.  Here a rank writes its piece into
 * its slot of a shared segment, publishes the collective's number with release order and reads the others' slots once their
 * numbers have arrived: microseconds.  Two buffers per slot, taken in turn: a rank can only be one collective ahead of the
 * slowest one (it needs everybody's number k + 1 to finish collective k + 1), so buffer k & 1 is never rewritten (collective
 * k + 2) before everybody has read collective k.  Pieces larger than a buffer go in several rounds. */
namespace {
constexpr size_t LFQ_SHM_BUF = 32768;
struct LfqShmSlot {
    std::atomic<uint64_t> seq;
    uint8_t pad[56];
    uint8_t buf[2][LFQ_SHM_BUF];
};
struct LfqShm {
    LfqShmSlot *slots = nullptr;
    size_t bytes = 0;
    int world = 0, rank = 0;
    uint64_t k = 0;                 /* collectives (rounds) done */
    double timeout_s = 600.;
    char name[256] = {0};
};
LfqShm g_shm;

int shm_allgather(void *user, int world, int rank, const void *send, void *recv, size_t bytes)
{
    LfqShm *m = (LfqShm *)user;
    if (!m->slots || world != m->world || rank != m->rank) {
        return -1;
    }
    for (size_t off = 0; off < bytes; off += LFQ_SHM_BUF) {
        const size_t n = std::min(LFQ_SHM_BUF, bytes - off);
        const uint64_t k = m->k++;
        LfqShmSlot &mine = m->slots[rank];
        memcpy(mine.buf[k & 1], (const uint8_t *)send + off, n);
        mine.seq.store(k + 1, std::memory_order_release);
        const double t0 = lfq_now_ms();
        for (int r = 0; r < world; r++) {
            LfqShmSlot &s = m->slots[r];
            long spins = 0;
            while (s.seq.load(std::memory_order_acquire) < k + 1) {
                LFQ_CPU_PAUSE();
                if ((++spins & 1023) == 0) {
                    const double waited = lfq_now_ms() - t0;
                    if (waited > m->timeout_s * 1e3) {
                        return -1;
                    }
                    if (waited > 0.2) {             /* a rank that is late by more than a moment: stop burning the core */
                        struct timespec ts = {0, 50000};
                        nanosleep(&ts, nullptr);
                    }
                }
            }
            memcpy((uint8_t *)recv + (size_t)r * bytes + off, s.buf[k & 1], n);
        }
    }
    return 0;
}
}  // namespace

int lfq_shard_shm_open(const char *name, int world, int rank)
{
    if (!name || !*name || strlen(name) >= sizeof(g_shm.name) || world < 1 || rank < 0 || rank >= world) {
        return LFQ_ERR_INVALID;
    }
    (void)lfq_shard_shm_close();
    const size_t bytes = sizeof(LfqShmSlot) * (size_t)world;
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) {
        return LFQ_ERR_UNSUPPORTED;
    }
    if (ftruncate(fd, (off_t)bytes) != 0) {            /* (a fresh segment reads as zeros: every number starts at 0) */
        close(fd);
        return LFQ_ERR_NOMEM;
    }
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        return LFQ_ERR_NOMEM;
    }
    g_shm.slots = (LfqShmSlot *)p;
    g_shm.bytes = bytes;
    g_shm.world = world;
    g_shm.rank = rank;
    g_shm.k = 0;
    snprintf(g_shm.name, sizeof(g_shm.name), "%s", name);
    g_host_allgather = shm_allgather;
    g_host_allgather_user = &g_shm;
    return LFQ_OK;
}

int lfq_shard_shm_unlink(void)
{
    if (!g_shm.slots || !g_shm.name[0]) {
        return LFQ_ERR_INVALID;
    }
    shm_unlink(g_shm.name);                             /* the mappings stay until every rank has closed */
    return LFQ_OK;
}

int lfq_shard_shm_close(void)
{
    if (!g_shm.slots) {
        return LFQ_OK;
    }
    if (g_host_allgather == shm_allgather) {
        g_host_allgather = nullptr;
        g_host_allgather_user = nullptr;
    }
    munmap(g_shm.slots, g_shm.bytes);
    g_shm.slots = nullptr;
    g_shm.name[0] = 0;
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
