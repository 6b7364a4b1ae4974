/*
 * lfq_shard.hip -- N processes, one per GPU: which device a worker takes, and the exchange of a sharded run
 * (test counts all-gathered, Bonferroni factors rebased, records gathered) over RCCL by dlopen or a caller's all-gather.
 */
#include "lfq_ctx.h"

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
