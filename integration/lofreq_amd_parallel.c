/*
 * lofreq_amd_parallel.c -- see lofreq_amd_parallel.h.  Plain C against include/lofreq_amd.h; RCCL is looked up at
 * run time (dlopen), like the library does, so a build without RCCL still links and can use the files transport.
 *
 * What it replaces in the reference: the log-sum + `bcftools concat` + `lofreq filter` epilogue of the parallel
 * wrapper (lofreq2_call_pparallel.py:131-185, 685-707).  What it does not: cutting the genome into bins and starting
 * the workers (lofreq2_call_pparallel.py:590-667 stays as it is; it only has to export the four LFQ_PAR_* variables).
 */
#define _GNU_SOURCE
#include "lofreq_amd_parallel.h"

#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef struct { char internal[128]; } lfq_nccl_id;            /* ncclUniqueId (nccl.h: NCCL_UNIQUE_ID_BYTES 128) */
typedef int (*nccl_get_id_fn)(lfq_nccl_id *);
typedef int (*nccl_init_rank_fn)(void **, int, lfq_nccl_id, int);
typedef int (*nccl_destroy_fn)(void *);

struct lfq_par {
    int world, rank;
    int files;                      /* 1: the files transport is asked for */
    int shm;                        /* 1: the library's shared-memory transport (one node; lfq_shard_shm_open) */
    int installed;                  /* 1: ... and installed (lfq_shard_set_host_allgather) */
    lfq_ctx *ctx;
    void *comm;                     /* ncclComm_t */
    void *rccl;                     /* dlopen handle */
    nccl_destroy_fn comm_destroy;
    char rdv[900];
    long seq;                       /* collectives done so far (files transport) */
    double timeout_s;
    uint64_t job;                   /* this run's nonce: every rendezvous file starts with it (see job_nonce) */
    int made_job_file;              /* 1: this run's rank 0 wrote <rdv>.job (the handshake of job_nonce) and removes it at exit */
};

/* Every file that passes through the rendezvous path starts with this header.  A file another run left behind under
 * the same LFQ_PAR_RENDEZVOUS (a crashed run's <rdv>.id / <rdv>.ag*, the closing barrier's files of a finished one)
 * carries another nonce and is waited past like a file that is not there yet. */
typedef struct { uint64_t magic, job; } rdv_hdr;
#define RDV_MAGIC 0x3152415051464cULL       /* "LFQPAR1" */

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void nap(void)
{
    struct timespec ts = {0, 2000000};      /* 2 ms */
    nanosleep(&ts, NULL);
}

/* write `n` bytes to `path` so that a reader never sees a partial file: temp name + rename */
static int put_file(uint64_t job, const char *path, const void *buf, size_t n)
{
    char tmp[1024];
    rdv_hdr h = {RDV_MAGIC, job};
    FILE *f;
    snprintf(tmp, sizeof(tmp), "%s.tmp%ld", path, (long)getpid());
    f = fopen(tmp, "wb");
    if (!f) {
        return -1;
    }
    if (fwrite(&h, sizeof(h), 1, f) != 1 || (n > 0 && fwrite(buf, 1, n, f) != n)) {
        fclose(f);
        unlink(tmp);
        return -1;
    }
    if (fclose(f) != 0 || rename(tmp, path) != 0) {
        unlink(tmp);
        return -1;
    }
    return 0;
}

/* wait until `path` exists with this run's header and exactly `n` payload bytes, then read it */
static int get_file(uint64_t job, const char *path, void *buf, size_t n, double timeout_s)
{
    const double t0 = now_s();
    for (;;) {
        struct stat st;
        if (stat(path, &st) == 0 && (size_t)st.st_size == n + sizeof(rdv_hdr)) {
            FILE *f = fopen(path, "rb");
            if (f) {
                rdv_hdr h = {0, 0};
                const size_t hg = fread(&h, sizeof(h), 1, f);
                const size_t got = (hg == 1 && n) ? fread(buf, 1, n, f) : 0;
                fclose(f);
                if (hg == 1 && h.magic == RDV_MAGIC && h.job == job && got == n) {
                    return 0;
                }
            }
        }
        if (now_s() - t0 > timeout_s) {
            return -1;
        }
        nap();
    }
}

/* lfq_host_allgather_fn: collective number `seq` of rank r lives in <rdv>.ag<seq>.<r> */
static int files_allgather(void *user, int world, int rank, const void *send, void *recv, size_t bytes)
{
    lfq_par *p = (lfq_par *)user;
    char path[1024];
    int r;
    const long k = p->seq++;
    if (world != p->world || rank != p->rank) {
        return -1;
    }
    snprintf(path, sizeof(path), "%s.ag%ld.%d", p->rdv, k, rank);
    if (put_file(p->job, path, send, bytes) != 0) {
        return -1;
    }
    for (r = 0; r < world; r++) {
        if (r == rank) {
            memcpy((char *)recv + (size_t)r * bytes, send, bytes);
            continue;
        }
        snprintf(path, sizeof(path), "%s.ag%ld.%d", p->rdv, k, r);
        if (get_file(p->job, path, (char *)recv + (size_t)r * bytes, bytes, p->timeout_s) != 0) {
            return -1;
        }
    }
    if (k >= 2) {       /* everybody who reads collective k has read k - 2 of everybody: that one can go */
        snprintf(path, sizeof(path), "%s.ag%ld.%d", p->rdv, k - 2, rank);
        unlink(path);
    }
    return 0;
}

static int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    char *end = NULL;
    long v;
    if (!e || !*e) {
        return dflt;
    }
    v = strtol(e, &end, 10);
    return (end == e || *end) ? dflt : (int)v;
}

/* 64 bits nobody else has: /dev/urandom, or pid + clocks where that cannot be read */
static uint64_t fresh_token(void)
{
    uint64_t t = 0;
    struct timespec ts;
    FILE *f = fopen("/dev/urandom", "rb");
    if (f) {
        if (fread(&t, sizeof(t), 1, f) != 1) {
            t = 0;
        }
        fclose(f);
    }
    clock_gettime(CLOCK_REALTIME, &ts);
    t ^= ((uint64_t)getpid() << 40) ^ ((uint64_t)ts.tv_sec << 20) ^ (uint64_t)ts.tv_nsec;
    return t ? t : 1;
}

/* The run's nonce.  LFQ_PAR_JOB, when the launcher exports one with the other LFQ_PAR_* variables (any string that is
 * new per run: its pid and start time will do), is taken as it is.  Without it the ranks agree on a nonce by a handshake
 * that no file of ANOTHER run under the same LFQ_PAR_RENDEZVOUS can take part in -- a crashed earlier attempt, a torchrun
 * restart with the same run id and port, a second `lofreq call` inside the same srun step (what the launcher exports is
 * the same for all of those, so it is not used):
 *   rank r > 0   draws a token T_r, writes <rdv>.hello.<r> = T_r, waits for a <rdv>.job whose r-th token IS T_r (a stale
 *                .job cannot hold it), takes its nonce and answers <rdv>.ack.<r> = T_r under that nonce;
 *   rank 0       draws the nonce, removes what it finds of <rdv>.id / <rdv>.job, publishes <rdv>.job = {nonce, the tokens
 *                of the hello files it currently sees} again whenever a hello file changes (a stale hello is overwritten
 *                by its rank's fresh one), and is done when every rank's ack carries the nonce and that rank's token.
 * Every later file (<rdv>.id, <rdv>.ag<k>.<r>) starts with the nonce; files with another one are waited past. */
static int job_nonce(lfq_par *p)
{
    const char *e = getenv("LFQ_PAR_JOB");
    char path[1024];
    const double t0 = now_s();
    const size_t job_bytes = sizeof(uint64_t) * (size_t)(1 + p->world);
    uint64_t *msg;
    int r, rc = -1;
    if (e && *e) {
        uint64_t h = 1469598103934665603ULL;                /* FNV-1a of the string */
        for (; *e; e++) {
            h = (h ^ (uint64_t)(unsigned char)*e) * 1099511628211ULL;
        }
        p->job = h ? h : 1;
        return 0;
    }
    msg = (uint64_t *)calloc((size_t)(1 + p->world) * 2, sizeof(uint64_t));      /* {nonce, tokens[world]} now and last published */
    if (!msg) {
        return -1;
    }
    if (p->rank != 0) {
        const uint64_t mine = fresh_token();
        snprintf(path, sizeof(path), "%s.hello.%d", p->rdv, p->rank);
        if (put_file(0, path, &mine, sizeof(mine)) == 0) {
            for (;;) {
                snprintf(path, sizeof(path), "%s.job", p->rdv);
                if (get_file(0, path, msg, job_bytes, 0.0) == 0 && msg[1 + p->rank] == mine && msg[0] != 0) {
                    p->job = msg[0];
                    snprintf(path, sizeof(path), "%s.ack.%d", p->rdv, p->rank);
                    rc = put_file(p->job, path, &mine, sizeof(mine));
                    break;
                }
                if (now_s() - t0 > p->timeout_s) {
                    break;
                }
                nap();
            }
        }
    } else {
        uint64_t *pub = msg + 1 + p->world;
        int published = 0;
        msg[0] = fresh_token();
        snprintf(path, sizeof(path), "%s.id", p->rdv);
        unlink(path);
        snprintf(path, sizeof(path), "%s.job", p->rdv);
        unlink(path);
        for (;;) {
            int have = 1, acked = 1;
            for (r = 1; r < p->world; r++) {
                uint64_t t = 0;
                snprintf(path, sizeof(path), "%s.hello.%d", p->rdv, r);
                if (get_file(0, path, &t, sizeof(t), 0.0) == 0 && t != 0) {
                    msg[1 + r] = t;
                } else if (msg[1 + r] == 0) {
                    have = 0;
                }
            }
            if (have && (!published || memcmp(msg, pub, job_bytes) != 0)) {
                snprintf(path, sizeof(path), "%s.job", p->rdv);
                if (put_file(0, path, msg, job_bytes) != 0) {
                    break;
                }
                memcpy(pub, msg, job_bytes);
                published = 1;
            }
            for (r = 1; r < p->world && published; r++) {
                uint64_t t = 0;
                snprintf(path, sizeof(path), "%s.ack.%d", p->rdv, r);
                if (!(get_file(msg[0], path, &t, sizeof(t), 0.0) == 0 && t == msg[1 + r])) {
                    acked = 0;
                }
            }
            if (published && acked) {
                p->job = msg[0];
                p->made_job_file = 1;
                for (r = 1; r < p->world; r++) {           /* the handshake's files have done their work */
                    snprintf(path, sizeof(path), "%s.hello.%d", p->rdv, r);
                    unlink(path);
                    snprintf(path, sizeof(path), "%s.ack.%d", p->rdv, r);
                    unlink(path);
                }
                rc = 0;
                break;
            }
            if (now_s() - t0 > p->timeout_s) {
                break;
            }
            nap();
        }
    }
    free(msg);
    return rc;
}

/* what a run that died left of THIS rank under the rendezvous path (a finished one leaves its closing barrier's file) */
static void unlink_stale(const lfq_par *p)
{
    char path[1024];
    long k;
    for (k = 0; k < 64; k++) {
        snprintf(path, sizeof(path), "%s.ag%ld.%d", p->rdv, k, p->rank);
        unlink(path);
    }
}

int lfq_par_world(const lfq_par *p) { return p ? p->world : 1; }
int lfq_par_rank(const lfq_par *p) { return p ? p->rank : 0; }
lfq_ctx *lfq_par_ctx(lfq_par *p) { return p ? p->ctx : NULL; }

static void *rccl_open(void)
{
    const char *names[] = {"librccl.so", "librccl.so.1", NULL};
    int pass, i;
    for (pass = 0; pass < 2; pass++) {              /* the copy the process already has before a fresh one */
        for (i = 0; names[i]; i++) {
            void *h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) {
                return h;
            }
        }
    }
    return NULL;
}

int lfq_par_init(lfq_par **out, int need_gpu)
{
    lfq_par *p;
    const char *rdv = getenv("LFQ_PAR_RENDEZVOUS"), *tr = getenv("LFQ_PAR_TRANSPORT");
    const int world = env_int("LFQ_PAR_WORLD", 1), rank = env_int("LFQ_PAR_RANK", 0);
    int rc, dev = -1;

    if (!out) {
        return LFQ_ERR_INVALID;
    }
    *out = NULL;
    if (world <= 1) {
        return LFQ_OK;
    }
    if (rank < 0 || rank >= world || !rdv || !*rdv || strlen(rdv) >= sizeof(p->rdv)) {
        return LFQ_ERR_INVALID;
    }
    p = (lfq_par *)calloc(1, sizeof(*p));
    if (!p) {
        return LFQ_ERR_NOMEM;
    }
    p->world = world;
    p->rank = rank;
    p->timeout_s = (double)env_int("LFQ_PAR_TIMEOUT_S", 600);
    strcpy(p->rdv, rdv);
    p->files = (tr && strcmp(tr, "files") == 0) ? 1 : 0;
    p->shm = (tr && strcmp(tr, "shm") == 0) ? 1 : 0;

    if (lfq_abi_version() != LFQ_ABI_VERSION) {        /* lfq_conf / record layouts belong to the version */
        free(p);
        return LFQ_ERR_UNSUPPORTED;
    }
    if (need_gpu) {
        if (!p->files && !p->shm && !getenv("LFQ_DEVICE")) {
            const int n = lfq_device_count();
            dev = n > 0 ? rank % n : LFQ_ERR_NO_DEVICE;     /* RCCL: one GPU per rank */
        } else {
            dev = lfq_pick_device(0, NULL);
        }
        if (dev < 0) {
            free(p);
            return dev;
        }
        rc = lfq_create(&p->ctx, dev);
        if (rc != LFQ_OK) {
            free(p);
            return rc;
        }
    }
    if (p->files) {
        unlink_stale(p);
    }
    if (job_nonce(p) != 0) {
        lfq_par_destroy(p);
        return LFQ_ERR_INVALID;                     /* the rendezvous path is not writable, or rank 0 never came */
    }
    if (p->files) {
        lfq_shard_set_host_allgather(files_allgather, p);
        p->installed = 1;
    } else if (p->shm) {
        /* the workers of ONE node: every collective through the library's shared segment, named after the run's nonce (new per
         * run by construction); a first all-gather tells rank 0 that everybody has it mapped, then the name goes */
        char name[64];
        int64_t one = 1, *all = (int64_t *)malloc(sizeof(int64_t) * (size_t)world);
        snprintf(name, sizeof(name), "/lofreq_amd.%016llx", (unsigned long long)p->job);
        rc = all ? lfq_shard_shm_open(name, world, rank) : LFQ_ERR_NOMEM;
        if (rc == LFQ_OK) {
            p->shm = 2;                             /* open: lfq_par_destroy closes it */
            rc = lfq_shard_allgather(NULL, NULL, world, rank, &one, 8, all);
        }
        free(all);
        if (rc == LFQ_OK && rank == 0) {
            (void)lfq_shard_shm_unlink();
        }
        if (rc != LFQ_OK) {
            lfq_par_destroy(p);
            return rc;
        }
    } else {
        /* ncclUniqueId of rank 0 through <rdv>.id, then ncclCommInitRank on every rank (blocks until all are in) */
        lfq_nccl_id id;
        char path[1024];
        nccl_get_id_fn get_id;
        nccl_init_rank_fn init_rank;
        p->rccl = rccl_open();
        get_id = p->rccl ? (nccl_get_id_fn)dlsym(p->rccl, "ncclGetUniqueId") : NULL;
        init_rank = p->rccl ? (nccl_init_rank_fn)dlsym(p->rccl, "ncclCommInitRank") : NULL;
        p->comm_destroy = p->rccl ? (nccl_destroy_fn)dlsym(p->rccl, "ncclCommDestroy") : NULL;
        if (!get_id || !init_rank || !p->ctx) {
            lfq_par_destroy(p);
            return LFQ_ERR_UNSUPPORTED;             /* no RCCL here (or no GPU asked for): use LFQ_PAR_TRANSPORT=files */
        }
        snprintf(path, sizeof(path), "%s.id", p->rdv);
        memset(&id, 0, sizeof(id));
        if (rank == 0) {
            if (get_id(&id) != 0 || put_file(p->job, path, &id, sizeof(id)) != 0) {
                lfq_par_destroy(p);
                return LFQ_ERR_HIP;
            }
        } else if (get_file(p->job, path, &id, sizeof(id), p->timeout_s) != 0) {
            lfq_par_destroy(p);
            return LFQ_ERR_HIP;
        }
        if (init_rank(&p->comm, world, id, rank) != 0) {
            p->comm = NULL;
            lfq_par_destroy(p);
            return LFQ_ERR_HIP;
        }
    }
    *out = p;
    return LFQ_OK;
}

void lfq_par_destroy(lfq_par *p)
{
    if (!p) {
        return;
    }
    if (p->installed) {
        char path[1024];
        long k;
        int64_t one = 1, *all = (int64_t *)malloc(sizeof(int64_t) * (size_t)p->world);
        if (all) {      /* a last collective as a barrier: nobody still needs this rank's files afterwards ... */
            (void)lfq_shard_allgather(NULL, NULL, p->world, p->rank, &one, 8, all);
            free(all);
        }
        lfq_shard_set_host_allgather(NULL, NULL);
        for (k = p->seq - 3; k < p->seq - 1; k++) {     /* ... except the barrier's own, which the last one out leaves */
            if (k >= 0) {
                snprintf(path, sizeof(path), "%s.ag%ld.%d", p->rdv, k, p->rank);
                unlink(path);
            }
        }
    }
    if (p->shm == 2) {
        int64_t one = 1, *all = (int64_t *)malloc(sizeof(int64_t) * (size_t)p->world);
        if (all) {      /* nobody unmaps while somebody still reads */
            (void)lfq_shard_allgather(NULL, NULL, p->world, p->rank, &one, 8, all);
            free(all);
        }
        (void)lfq_shard_shm_close();
    }
    if (p->comm && p->comm_destroy) {
        p->comm_destroy(p->comm);
    }
    if (p->rank == 0) {
        char path[1024];
        if (!p->files && !p->shm) {
            snprintf(path, sizeof(path), "%s.id", p->rdv);
            unlink(path);
        }
        if (p->made_job_file) {
            snprintf(path, sizeof(path), "%s.job", p->rdv);
            unlink(path);
        }
    }
    if (p->ctx) {
        lfq_destroy(p->ctx);
    }
    free(p);
}

int lfq_par_merge_snvs(lfq_par *p, lfq_conf *conf, lfq_col_pvals *pvals, int64_t n_pvals, int64_t n_tested_columns,
                       int64_t n_indel_tests, lfq_snv_record **records_out, int64_t *n_records_out)
{
    int64_t local[2], prefix[2], *all, total_tested = 0, total_indel = 0, n_mine = 0, n_all = 0, cap;
    lfq_snv_record *mine = NULL, *merged = NULL;
    int rc, r;

    if (!p || !conf || !records_out || !n_records_out || n_pvals < 0 || (n_pvals > 0 && !pvals)) {
        return LFQ_ERR_INVALID;
    }
    *records_out = NULL;
    *n_records_out = 0;
    all = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)p->world);
    if (!all) {
        return LFQ_ERR_NOMEM;
    }
    /* (1) one all-gather of {tested columns, indel tests} per rank: the global totals and this rank's prefix */
    local[0] = n_tested_columns;
    local[1] = n_indel_tests;
    rc = lfq_shard_exchange_counts(p->ctx, p->comm, p->world, p->rank, local, 2, all, prefix);
    if (rc != LFQ_OK) {
        free(all);
        return rc;
    }
    for (r = 0; r < p->world; r++) {
        total_tested += all[2 * r];
        total_indel += all[2 * r + 1];
    }
    free(all);
    /* (2) the shard-local running factors become the single-process ones, then the exact emit test (lofreq_call.c:832) */
    /* only with the dynamic factor: under `-b N` every column already carries the fixed N (lofreq_call.c:794 is the
       only place the factor moves), and adding 3 * prefix would over-correct every rank but the first */
    if (conf->bonf_dynamic) {
        rc = lfq_shard_rebase_bonferroni(pvals, n_pvals, prefix[0]);
        if (rc != LFQ_OK) {
            return rc;
        }
    }
    cap = 3 * n_pvals + 1;
    mine = (lfq_snv_record *)malloc(sizeof(lfq_snv_record) * (size_t)cap);
    if (!mine) {
        return LFQ_ERR_NOMEM;
    }
    rc = lfq_finalize_pvals(conf, pvals, n_pvals, NULL, NULL, mine, cap, &n_mine);
    if (rc != LFQ_OK) {
        free(mine);
        return rc;
    }
    /* (3) the records travel to rank 0 (every rank takes part in the collectives; only rank 0 keeps the result) */
    {
        int64_t *sizes = (int64_t *)malloc(sizeof(int64_t) * (size_t)p->world);
        int64_t room;
        if (!sizes) {
            free(mine);
            return LFQ_ERR_NOMEM;
        }
        rc = lfq_shard_allgather(p->ctx, p->comm, p->world, p->rank, &n_mine, 8, sizes);
        for (r = 0; rc == LFQ_OK && r < p->world; r++) {
            n_all += sizes[r];
        }
        free(sizes);
        if (rc != LFQ_OK) {
            free(mine);
            return rc;
        }
        room = p->rank == 0 ? n_all : 0;
        if (room > 0) {
            merged = (lfq_snv_record *)malloc(sizeof(lfq_snv_record) * (size_t)room);
            if (!merged) {
                free(mine);
                return LFQ_ERR_NOMEM;       /* (the peers time out in their collective: out of memory is fatal anyway) */
            }
        }
        rc = lfq_shard_gather_records(p->ctx, p->comm, p->world, p->rank, mine, n_mine, 0, merged, room, &n_all);
        if (rc != LFQ_OK && !(rc == LFQ_ERR_CAPACITY && p->rank != 0)) {
            free(mine);
            free(merged);
            return rc;
        }
    }
    free(mine);
    /* (4) conf as the single-process loop leaves it: 3 tests per tested column (lofreq_call.c:794-801), one per indel
     * test (:693-696) */
    lfq_shard_advance_conf(conf, total_tested);
    if (total_indel > 0) {
        if (conf->bonf_dynamic) {
            conf->bonf_indel = (conf->bonf_indel == 1 ? 0 : conf->bonf_indel) + total_indel;
        }
        conf->num_indel_tests += total_indel;
    }
    *records_out = merged;
    *n_records_out = n_all;
    return LFQ_OK;
}

int lfq_par_gather_bytes(lfq_par *p, const void *mine, int64_t n, void **out, int64_t *n_out)
{
    int64_t *sizes, most = 0, total = 0, o = 0;
    char *send, *all;
    int rc, r;

    if (!p || n < 0 || (n > 0 && !mine) || !out || !n_out) {
        return LFQ_ERR_INVALID;
    }
    *out = NULL;
    *n_out = 0;
    sizes = (int64_t *)malloc(sizeof(int64_t) * (size_t)p->world);
    if (!sizes) {
        return LFQ_ERR_NOMEM;
    }
    rc = lfq_shard_allgather(p->ctx, p->comm, p->world, p->rank, &n, 8, sizes);
    if (rc != LFQ_OK) {
        free(sizes);
        return rc;
    }
    for (r = 0; r < p->world; r++) {
        total += sizes[r];
        if (sizes[r] > most) {
            most = sizes[r];
        }
    }
    *n_out = total;
    if (most == 0) {
        free(sizes);
        return LFQ_OK;
    }
    send = (char *)calloc((size_t)most, 1);
    all = (char *)malloc((size_t)most * (size_t)p->world);
    if (!send || !all) {
        free(send);
        free(all);
        free(sizes);
        return LFQ_ERR_NOMEM;
    }
    if (n > 0) {
        memcpy(send, mine, (size_t)n);
    }
    rc = lfq_shard_allgather(p->ctx, p->comm, p->world, p->rank, send, most, all);
    free(send);
    if (rc == LFQ_OK && p->rank == 0) {
        char *cat = (char *)malloc((size_t)total + 1);
        if (!cat) {
            rc = LFQ_ERR_NOMEM;
        } else {
            for (r = 0; r < p->world; r++) {
                memcpy(cat + o, all + (size_t)r * (size_t)most, (size_t)sizes[r]);
                o += sizes[r];
            }
            cat[total] = 0;
            *out = cat;
        }
    }
    free(all);
    free(sizes);
    return rc;
}
