/*
 * lfq_readset.hip -- the resident read set: the reads of a region uploaded once; BAQ / IDAQ, source
 * quality and both pileups on the device copy; the host-buffer entry points of those steps around a temporary read set.
 */
#include "lfq_ctx.h"

extern "C" {

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

#define LFQ_UP_CHUNKS 6
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
    std::atomic<int> up_chunks;         /* bases + qualities of the reads [0, up_bound[up_chunks]) have landed */
    int up_nchunks;
    int64_t up_bound[LFQ_UP_CHUNKS + 1]; /* chunk j = reads [up_bound[j], up_bound[j + 1]); the first ones are small: 1, 2, 4
                                          * rounds of the BAQ kernel over the SIMDs, so that its first launch starts early */
    std::atomic<int> up_rc;        /* written by the helper thread at the end of each stage */
    LfqPin<uint8_t> *up_fl;             /* the flag bytes on their way out (pinned; handed back once the copies are done) */
    /* Pinned caller arrays (lfq_host_alloc): the copies are DMA transfers queued by lfq_readset_create itself, and what
     * waits for them is a STREAM, not the host -- events instead of the helper thread's progress counters */
    bool up_events;
    hipEvent_t ev_chunk[LFQ_UP_CHUNKS]; /* bases + qualities of the reads up to chunk j have landed */
    hipEvent_t ev_stage[2];             /* [0]: everything but BI / BD, [1]: all of it */
};

/* the reads, qualities, CIGARs and the contig are on the device (BI / BD may still be on their way: the BAQ kernels do
 * not read them, and 600 of the 1300 MB of a 2 M-read region then cross PCIe under those kernels).  Every one of the three
 * waits below makes the given stream(s) wait when the uploads are tracked by events, and the host otherwise. */
static int readset_upload_wait_inputs(lfq_readset *rs, std::initializer_list<hipStream_t> streams)
{
    if (rs && rs->up_events) {
        for (hipStream_t st : streams) {
            if (st && hipStreamWaitEvent(st, rs->ev_stage[0], 0) != hipSuccess) {
                return LFQ_ERR_HIP;
            }
        }
        return LFQ_OK;
    }
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
static int readset_upload_wait_reads(lfq_readset *rs, int64_t r_last, hipStream_t st)
{
    if (rs && (rs->up_thread || rs->up_events)) {
        int need = 1;
        while (need < rs->up_nchunks && rs->up_bound[need] <= r_last) {
            need++;
        }
        if (rs->up_events) {
            return hipStreamWaitEvent(st, rs->ev_chunk[need - 1], 0) == hipSuccess ? (int)LFQ_OK : (int)LFQ_ERR_HIP;
        }
        while (rs->up_chunks.load(std::memory_order_acquire) < need) {
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    return rs ? rs->up_rc.load() : (int)LFQ_OK;
}

/* all of it.  st = null: the host waits (and the pinned flag bytes go back to the pool) */
static int readset_upload_wait(lfq_readset *rs, hipStream_t st = nullptr)
{
    if (rs && rs->up_events) {
        if (st) {
            return hipStreamWaitEvent(st, rs->ev_stage[1], 0) == hipSuccess ? (int)LFQ_OK : (int)LFQ_ERR_HIP;
        }
        if (hipEventSynchronize(rs->ev_stage[1]) != hipSuccess) {
            return LFQ_ERR_HIP;
        }
    }
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

/* is `p` pinned (or registered) host memory?  The runtime copies from such memory by DMA and hipMemcpyAsync returns at once */
static bool lfq_is_pinned(const void *p)
{
    if (!p) {
        return true;                        /* nothing to copy */
    }
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();            /* plain malloc'ed memory: an error, by design of that call */
        return false;
    }
    return at.type == hipMemoryTypeHost;
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
        for (hipEvent_t e : {rs->ev_chunk[0], rs->ev_chunk[1], rs->ev_chunk[2], rs->ev_chunk[3], rs->ev_chunk[4], rs->ev_chunk[5],
                             rs->ev_stage[0], rs->ev_stage[1]}) {
            if (e) (void)hipEventDestroy(e);
        }
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
    /* chunks of reads: small first ones -- the reads of one round of the BAQ register kernel over the SIMDs (one wavefront
     * of 64 reads each), what lfq_readset_baq's first launch takes, then two and four -- and the rest in equal parts */
    const int64_t first_reads = (int64_t)c->n_cu * 4 * 64;
    const int n_chunks = nb >= ((int64_t)64 << 20) ? (n >= 16 * first_reads ? LFQ_UP_CHUNKS : 4) : 1;
    rs->up_nchunks = n_chunks;
    rs->up_bound[0] = 0;
    if (n_chunks == LFQ_UP_CHUNKS) {
        /* 1, 2, 4 rounds (the link delivers reads about twice as fast as the kernel takes them: while a launch runs, the
         * reads of one twice its size arrive), the rest in three equal parts */
        rs->up_bound[1] = first_reads;
        rs->up_bound[2] = 3 * first_reads;
        rs->up_bound[3] = 7 * first_reads;
        for (int j = 4; j <= n_chunks; j++) {
            rs->up_bound[j] = 7 * first_reads + (n - 7 * first_reads) * (j - 3) / (n_chunks - 3);
        }
    } else {
        for (int j = 1; j <= n_chunks; j++) {
            rs->up_bound[j] = n * j / n_chunks;
        }
    }
    for (int j = 0; j < n_chunks; j++) {
        const int64_t b0 = rd->seq_off[rs->up_bound[j]], b1 = rd->seq_off[rs->up_bound[j + 1]];
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
    if (up_mode == 0 && up_bytes >= ((int64_t)8 << 20) && lfq_is_pinned(rd->seq) && lfq_is_pinned(rd->qual)
        && lfq_is_pinned(rs->h_bi) && lfq_is_pinned(rs->h_bd)) {
        /* the big arrays are pinned: every copy is queued right here (the small pageable ones, if any, are staged by the
         * runtime inside the call), events mark the stages, and nothing on the host ever waits for the link */
        int rc = LFQ_OK;
        for (int j = 0; j < LFQ_UP_CHUNKS + 2 && rc == LFQ_OK; j++) {
            hipEvent_t *e = j < LFQ_UP_CHUNKS ? &rs->ev_chunk[j] : &rs->ev_stage[j - LFQ_UP_CHUNKS];
            if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) {
                rc = LFQ_ERR_HIP;
            }
        }
        bool stage0_done = false;
        for (const Copy &x : todo) {
            if (rc == LFQ_OK && hipMemcpyAsync(x.dst, x.src, (size_t)x.bytes, hipMemcpyHostToDevice, ups) != hipSuccess) {
                rc = LFQ_ERR_HIP;
            }
            if (rc == LFQ_OK && x.chunks_after && hipEventRecord(rs->ev_chunk[x.chunks_after - 1], ups) != hipSuccess) {
                rc = LFQ_ERR_HIP;
            }
            if (rc == LFQ_OK && x.stage_after && hipEventRecord(rs->ev_stage[x.stage_after - 1], ups) != hipSuccess) {
                rc = LFQ_ERR_HIP;
            }
            if (rc == LFQ_OK && x.stage_after == 2 && !stage0_done && hipEventRecord(rs->ev_stage[0], ups) != hipSuccess) {
                rc = LFQ_ERR_HIP;           /* (no BI / BD: the last copy ends both stages) */
            }
            stage0_done = stage0_done || x.stage_after != 0;
        }
        for (int j = n_chunks; j < LFQ_UP_CHUNKS && rc == LFQ_OK; j++) {   /* fewer chunks than events: the unused ones are the last one */
            if (hipEventRecord(rs->ev_chunk[j], ups) != hipSuccess) {
                rc = LFQ_ERR_HIP;
            }
        }
        if (rc != LFQ_OK) {
            (void)hipStreamSynchronize(ups);
            lfq_readset_destroy(rs);
            return rc;
        }
        rs->up_events = true;
        rs->up_rc = LFQ_OK;
        rs->up_chunks.store(n_chunks, std::memory_order_release);
        rs->up_stage.store(2, std::memory_order_release);
    } else if (up_mode == 2 || (up_bytes >= ((int64_t)8 << 20) && up_mode == 0)) {
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
            int32_t part_max[LFQ_HOST_PARTS];
            bool part_sorted[LFQ_HOST_PARTS];
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
                int32_t before[LFQ_HOST_PARTS];
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

/* Reads that are not position-sorted: the column-major kernels cannot take them (no window to search), the read-major ones hand
 * a column's observations out in the order their threads get there (an atomic cursor), so that the last bits of a p-value can
 * differ from run to run.  The reference cannot take such input at all (bam_mplp_auto needs a coordinate-sorted file,
 * plp.c:1406-1447): refused unless the caller asked for it (lfq_set_pileup_unsorted; the tuning build's LFQ_PILEUP_ATOMIC sends
 * even sorted reads that way, for the tests that hold the two kernel families against each other). */
static bool readset_unsorted_ok(const lfq_ctx *c, const lfq_readset *rs)
{
    return rs->n == 0 || c->plp_unsorted_ok || lfq_knobs().pileup_atomic;
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
    const int64_t h_bytes = (n * (int64_t)sizeof(LfqBaqGeom) + 255) / 256 * 256, ord_bytes = (n * 4 + 255) / 256 * 256;
    if (h_bytes + ord_bytes > c->pin_bytes) {
        if (c->h_pin) (void)hipHostFree(c->h_pin);
        c->h_pin = nullptr;
        c->pin_bytes = 0;
        LFQ_TRY_HIP(hipHostMalloc((void **)&c->h_pin, (size_t)(h_bytes + ord_bytes), hipHostMallocDefault));
        c->pin_bytes = h_bytes + ord_bytes;
    }
    LfqBaqGeom *h = (LfqBaqGeom *)c->h_pin;
    int32_t *order = (int32_t *)(c->h_pin + h_bytes);
    /* (from the context's pinned pool: fresh memory would be page-faulted in by the threads below while the upload
     * thread is pinning the caller's arrays -- the two fight over the address-space lock) */
    LfqPin<int32_t> width(c, (size_t)n);
    LFQ_PIN_OK(width);
    const bool use_lds = lfq_knobs().baq_lds != 0;
    int max_lq = 0, max_w = 0;
    int part_lq[LFQ_HOST_PARTS] = {0}, part_w[LFQ_HOST_PARTS] = {0};
    int64_t part_narrow[LFQ_HOST_PARTS + 1] = {0}, part_band8[LFQ_HOST_PARTS + 1] = {0}, part_plain[LFQ_HOST_PARTS + 1] = {0};
    LfqPin<uint8_t> has_id(c, (size_t)n);                            /* the read has an I or D operation (what idaq looks at) */
    LFQ_PIN_OK(has_id);
    int part_lrn[LFQ_HOST_PARTS] = {0}, part_lqn[LFQ_HOST_PARTS] = {0};
    int parts = 1;
    lfq_for_reads(n, [&](int64_t r_begin, int64_t r_end, int part) {
    int max_lq = 0, max_w = 0, lrn = 0, lqn = 0;    /* of this part */
    int64_t n_nar = 0, n_b8 = 0, n_pl = 0;
    for (int64_t r = r_begin; r < r_end; r++) {
        LfqBaqGeom &o = h[(size_t)r];
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
        o.xb = xb;
        o.l_ref = xe - xb;
        o.bw = bw;
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
        } else if (use_lds && wr == LFQ_BAQ_BAND8_CELLS && o.l_ref <= LFQ_BAQ_LDS_MAX_LREF) {
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
            } else if (use_lds && width[(size_t)r] == LFQ_BAQ_BAND8_CELLS && short_ref) {
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
    const int64_t o_reads = 0, o_q2p = o_reads + al(n * (int64_t)sizeof(LfqBaqGeom)), o_ord = o_q2p + al(1024),
                  total = o_ord + al(n * 4);
    LFQ_TRY(grow(&c->d_tmp[0], &c->tmp_bytes[0], total));
    d_blob = c->d_tmp[0];
    int rc = LFQ_OK;
    auto up = [&](int64_t off, const void *src, int64_t bytes) {
        if (rc == LFQ_OK && bytes > 0 && hipMemcpyAsync(d_blob + off, src, (size_t)bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
            rc = LFQ_ERR_HIP;
        }
    };
    up(o_reads, h, n * (int64_t)sizeof(LfqBaqGeom));
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
        A.geom = (const LfqBaqGeom *)(d_blob + o_reads);
        A.pos = (const int32_t *)rs->d_pos;
        A.cigar_off = (const int64_t *)rs->d_coff;
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
        /* + 3: the four groups of reads (plain narrow, narrow with indels, band 8, wide) round up to whole wavefronts separately */
        int64_t waves = std::max<int64_t>(1, std::min<int64_t>((n + 63) / 64 + 3, budget_b / per_wave));
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
        /* several contexts of one process (a host thread each) size their scratch from the same "free" figure at the same time: a
         * context that comes too late for its share takes fewer wavefronts per launch instead of failing -- down to one round of the
         * SIMDs, below which the kernel would leave part of the device idle */
        while (rc == LFQ_ERR_NOMEM && waves > (int64_t)c->n_cu * 4) {
            (void)hipGetLastError();
            rc = LFQ_OK;
            waves = std::max<int64_t>((int64_t)c->n_cu * 4, waves / 2);
            keep(&c->d_baq_scr, &c->baq_scr_bytes, waves * per_wave);
        }
        if (rc == LFQ_OK && c->baq_scr_bytes / per_wave < waves) {
            waves = c->baq_scr_bytes / per_wave;
        }
        keep(&c->d_baq_expect, &c->baq_expect_bytes, waves * A.rows * 64 * 4);
        keep(&c->d_baq_tmp8, &c->baq_tmp8_bytes, waves * 2 * A.rows * 64);
        if (!lfq_knobs().baq_one_variant) {
            keep(&c->d_baq_nflag, &c->baq_nflag_bytes, (n + 63) / 64 + 64);
        }
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
        const bool side_ok = n_narrow > 0 && c->side[0] != nullptr && c->side[1] != nullptr && !lfq_knobs().single_stream;
        const bool beside = side_ok && waves_wide + waves_b8 > 0 && waves_wide + waves_b8 < waves / 4;
        /* The narrow-band reads with an indel operation (the IDAQ instantiation: 4 % of the bench's reads) as well, when every
         * group has scratch slots of its own: behind the plain launches on the same stream they were a launch of a third of
         * a round with the machine to itself (0.8 ms per 479 K reads); beside them they fill the plain launch's last round. */
        const int64_t waves_i = (n_narrow - n_plain + 63) / 64;
        const bool beside_i = side_ok && lfq_knobs().baq_idaq_beside && n_plain > 0 && waves_i > 0 && waves_i < waves / 4
                              && (n_plain + 63) / 64 + waves_i + (beside ? waves_wide + waves_b8 : 0) <= waves;
        /* (they share the band-8 launch's stream: the first side stream tends to be multiplexed onto the hardware queue of
         * c->stream, where a launch waits for the one before it whatever stream it came from) */
        const bool use_side0 = beside && n_wide > 0, use_side1 = (beside && n_band8 > 0) || beside_i;
        int64_t waves_n = waves - (beside ? waves_wide + waves_b8 : 0) - (beside_i ? waves_i : 0);      /* slots of a narrow launch */
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
        if ((use_side0 || use_side1) && hipEventRecord(c->ev_join[0], c->stream) != hipSuccess) {
            rc = LFQ_ERR_HIP;       /* the side streams start after the uploads / memsets queued on c->stream so far */
        }
        /* timing events around the call's BAQ kernels (lfq_last_baq_times) */
        if (!c->ev_baq_t[0] && (hipEventCreate(&c->ev_baq_t[0]) != hipSuccess || hipEventCreate(&c->ev_baq_t[1]) != hipSuccess)) {
            rc = LFQ_ERR_HIP;
        }
        if (rc == LFQ_OK && hipEventRecord(c->ev_baq_t[0], c->stream) != hipSuccess) {
            rc = LFQ_ERR_HIP;
        }
        c->baq_launches = 0;
        c->baq_reads = n;
        c->baq_bases = n_bases;
        {
            LfqBaqArgs Ap = A;                      /* the plain instantiation: no indel table */
            Ap.itab = nullptr;
            Ap.terms = nullptr;
            Ap.ai_out = Ap.ad_out = nullptr;
            Ap.tag_flags = nullptr;
            /* (while the reads are still arriving the first launch takes one round only: it starts when the small first
             * chunk of lfq_readset_create has landed, and the link stays ahead of the kernel from there on) */
            const bool early = (rs->up_thread || rs->up_events) && rs->up_nchunks == LFQ_UP_CHUNKS && waves_n >= 4 * round
                               && n_plain > 16 * round * 64;
            int64_t ramp = early ? round : waves_n;             /* wavefronts of the next launch: 1, 2, 4 rounds, then all slots */
            std::vector<std::pair<LfqBaqArgs, int64_t>> with_n;
            for (int64_t first = 0, cnt = 0; rc == LFQ_OK && first < n_plain; first += cnt) {
                cnt = std::min<int64_t>(std::min(ramp, waves_n) * 64, n_plain - first);
                ramp = ramp < 4 * round ? ramp * 2 : waves_n;
                rc = readset_upload_wait_reads(rs, order[(size_t)(first + cnt - 1)], c->stream);
                Ap.first_read = (int32_t)first;
                /* (first is a multiple of 64: the launches are cut to whole wavefronts) */
                Ap.nflag = (c->d_baq_nflag && !lfq_knobs().baq_one_variant) ? c->d_baq_nflag + first / 64 : nullptr;
                if (rc == LFQ_OK) {
                    rc = lfq_launch_baq(Ap, cnt, 1, c->stream, Ap.nflag ? 1 : 0);
                    c->baq_launches++;
                    if (Ap.nflag) {
                        with_n.push_back({Ap, cnt});
                    }
                }
            }
            /* The instantiation with the N case for every launch's flagged wavefronts, behind all of them: a wavefront of
             * this kernel needs an empty SIMD even to find that it has nothing to do, and a launch that came up while the
             * indel counter pass of the chain kept every SIMD busy waited for that kernel to drain (1.2 ms per region for
             * wavefronts that leave at once).  Here the other streams are idle or done; the scratch slots of a launch's
             * wavefronts are free again (every earlier launch has finished: stream order). */
            for (const auto &w : with_n) {
                if (rc == LFQ_OK) {
                    rc = lfq_launch_baq(w.first, w.second, 1, c->stream, 2);
                }
            }
        }
        if (rc == LFQ_OK) {
            rc = readset_upload_wait_inputs(rs, {c->stream, c->side[0], c->side[1]});
        }
        if (use_side0 || use_side1) {
            /* wide-band, band-8 and (beside_i) the narrow-band reads with indels on the side streams, beside the plain narrow-band
             * launches; c->stream ends after them */
            if (rc == LFQ_OK && use_side0 && hipStreamWaitEvent(c->side[0], c->ev_join[0], 0) != hipSuccess) {
                rc = LFQ_ERR_HIP;
            }
            if (rc == LFQ_OK && beside && n_wide > 0) {
                LfqBaqArgs Aw = at_slot(waves_n + waves_b8);
                Aw.first_read = (int32_t)(n_narrow + n_band8);
                rc = lfq_launch_baq(Aw, n_wide, 0, c->side[0]);
            }
            if (rc == LFQ_OK && use_side1 && hipStreamWaitEvent(c->side[1], c->ev_join[0], 0) != hipSuccess) {
                rc = LFQ_ERR_HIP;
            }
            if (rc == LFQ_OK && beside_i) {
                LfqBaqArgs Ai = at_slot(waves_n + (beside ? waves_b8 + waves_wide : 0));
                Ai.first_read = (int32_t)n_plain;
                rc = lfq_launch_baq(Ai, n_narrow - n_plain, 1, c->side[1]);
                c->baq_launches++;
            }
            if (rc == LFQ_OK && beside && n_band8 > 0) {
                LfqBaqArgs Ab = at_slot(waves_n);
                Ab.first_read = (int32_t)n_narrow;
                rc = lfq_launch_baq(Ab, n_band8, 2, c->side[1]);
            }
            if (rc == LFQ_OK && ((use_side0 && hipEventRecord(c->ev_join[1], c->side[0]) != hipSuccess)
                                 || (use_side1 && hipEventRecord(c->ev_join[2], c->side[1]) != hipSuccess))) {
                rc = LFQ_ERR_HIP;
            }
        }
        for (int64_t first = n_plain; rc == LFQ_OK && !beside_i && first < n_narrow; first += waves_n * 64) {
            A.first_read = (int32_t)first;
            rc = lfq_launch_baq(A, std::min<int64_t>(waves_n * 64, n_narrow - first), 1, c->stream);
            c->baq_launches++;
        }
        if (rc == LFQ_OK && ((use_side0 && hipStreamWaitEvent(c->stream, c->ev_join[1], 0) != hipSuccess)
                             || (use_side1 && hipStreamWaitEvent(c->stream, c->ev_join[2], 0) != hipSuccess))) {
            rc = LFQ_ERR_HIP;
        }
        if (!beside) {
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
    if (rc == LFQ_OK && c->ev_baq_t[1] && hipEventRecord(c->ev_baq_t[1], c->stream) != hipSuccess) {
        rc = LFQ_ERR_HIP;
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

int lfq_last_baq_times(lfq_ctx *c, lfq_baq_times *t)
{
    if (!c || !t) {
        return LFQ_ERR_INVALID;
    }
    memset(t, 0, sizeof(*t));
    if (!c->ev_baq_t[0] || !c->ev_baq_t[1]) {
        return LFQ_OK;                  /* no BAQ call on this context yet */
    }
    LFQ_TRY_HIP(hipSetDevice(c->device));
    LFQ_TRY_HIP(hipEventSynchronize(c->ev_baq_t[1]));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev_baq_t[0], c->ev_baq_t[1]) == hipSuccess) {
        t->ms_kernels = ms;
    }
    t->n_launches = c->baq_launches;
    t->n_reads = c->baq_reads;
    t->n_bases = c->baq_bases;
    return LFQ_OK;
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
    memset(out, 0, sizeof(*out));
    const int64_t n = rs->n, width = region_end - region_begin;
    if (n == 0 || width == 0) {
        return LFQ_OK;
    }
    LFQ_TRY_HIP(hipSetDevice(c->device));
    LFQ_TRY(readset_upload_wait(rs, c->stream));
    auto al = [](int64_t b) { return (b + 255) / 256 * 256; };
    /* per-position counters (kept until the next call) */
    const int64_t n_tiles = (width + LFQ_PLP_COMPACT_TILE - 1) / LFQ_PLP_COMPACT_TILE;
    const int64_t o_cov = 0, o_nb = o_cov + al(width * 4), o_cur = o_nb + al(width * 4), o_cidx = o_cur + al(width * 4),
                  o_tcol = o_cidx + al(width * 4), o_tobs = o_tcol + al(n_tiles * 8), o_tmax = o_tobs + al(n_tiles * 8),
                  o_tot = o_tmax + al(n_tiles * 4), o_cpos = o_tot + 256, total = o_cpos + al(width * 8);
    /* Everything goes to the main stream, behind the BAQ kernels if they are still running (the scatter pass reads their
     * lb bytes; pass 0 beside them was measured: 2 ms alone, 14 ms squeezed between wavefronts that hold 416 of a SIMD's
     * 512 registers, with the host waiting for its result).  Nothing waits for the scatter pass: the tracks are complete
     * in stream order (see the header) -- the host goes on with the indel tests while it runs. */
    hipStream_t ps = c->stream;
    LFQ_TRY(lfq_order_after_batch(c, c->stream));          /* the tracks of the previous call may still be a running batch's input */
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
    if (!sorted && !readset_unsorted_ok(c, rs)) {
        return LFQ_ERR_INVALID;             /* as mpileup: bam_mplp_auto stops at a file that is not coordinate-sorted (plp.c:1406-1447) */
    }
    LFQ_TRY(sorted ? lfq_launch_pileup_columns(A, 0, ps) : lfq_launch_pileup_count(A, ps));
    /* columns from the counters on the device (lfq_launch_plp_compact_*): the host waits once, for three numbers */
    LfqPin<int64_t> tot(c, 4);
    LFQ_PIN_OK(tot);
    LFQ_TRY(lfq_launch_plp_compact_sums(A.cov, A.nb, width, (int64_t *)(d + o_tcol), (uint64_t *)(d + o_tobs),
                                        (int32_t *)(d + o_tmax), (int64_t *)(d + o_tot), ps));
    LFQ_TRY_HIP(hipMemcpyAsync(tot.data(), d + o_tot, 24, hipMemcpyDeviceToHost, ps));
    LFQ_TRY_HIP(hipStreamSynchronize(ps));
    const int64_t ncols = tot[0], n_obs = tot[1], max_obs = tot[2];
    const int64_t trk = al(n_obs + 32);
    const bool nt_packed = !c->plp_nt_bytes;
    const int64_t t_off = 0, t_ref = t_off + al((ncols + 1) * 8), t_cov = t_ref + al(ncols + 16), t_nb = t_cov + al(ncols * 4 + 16),
                  t_nt = t_nb + al(ncols * 4 + 16), t_bq = t_nt + trk, t_baq = t_bq + trk, t_mq = t_baq + trk,
                  t_sq = t_mq + trk, t_ntp = t_sq + (rs->has_sqb ? trk : 0), t_total = t_ntp + (nt_packed ? al(trk / 2 + 16) : 0);
    LFQ_TRY(grow(&c->d_plp_out, &c->plp_out_bytes, t_total));
    uint8_t *t = c->d_plp_out;
    LFQ_TRY_HIP(hipMemsetAsync(t + t_nt, 0, (size_t)(t_total - t_nt), ps));            /* the 16-byte tails are read */
    LFQ_TRY(lfq_launch_plp_compact_apply(A.cov, A.nb, width, region_begin, rs->d_ref, rs->ref_len, (const int64_t *)(d + o_tcol),
                                         (const uint64_t *)(d + o_tobs), (const int64_t *)(d + o_tot), (int32_t *)(d + o_cidx),
                                         (uint64_t *)(t + t_off), t + t_ref, (int32_t *)(t + t_cov), (int32_t *)(t + t_nb),
                                         (int64_t *)(d + o_cpos), ps));
    /* the columns' positions for the caller: copied on the context's own upload stream behind the apply kernel, so that the
     * wait for the copy below is not a wait for the scatter pass -- and not for another context's DP chain either (the DP
     * stream, which carried this copy before, is shared by the contexts of a device) */
    hipStream_t aux = ps;
    if (!lfq_knobs().single_stream) {
        if (!c->up_stream && hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking) != hipSuccess) {
            c->up_stream = nullptr;
            return LFQ_ERR_HIP;
        }
        aux = c->up_stream;
    }
    bool copy_pending = false;
    if (col_pos_out && ncols > 0) {
        if (!c->ev_apply) {
            LFQ_TRY_HIP(hipEventCreateWithFlags(&c->ev_apply, hipEventDisableTiming));     /* kept: destroyed with the context */
        }
        if (hipEventRecord(c->ev_apply, ps) != hipSuccess || (aux != ps && hipStreamWaitEvent(aux, c->ev_apply, 0) != hipSuccess)
            || hipMemcpyAsync(col_pos_out, d + o_cpos, (size_t)ncols * 8, hipMemcpyDeviceToHost, aux) != hipSuccess) {
            (void)hipStreamSynchronize(aux);
            return LFQ_ERR_HIP;
        }
        copy_pending = true;
    }
    A.col_index = (const int32_t *)(d + o_cidx);
    A.col_off = (const uint64_t *)(t + t_off);
    A.t_nt = t + t_nt;
    A.t_bq = t + t_bq;
    A.t_baq = t + t_baq;
    A.t_mq = t + t_mq;
    A.t_sq = rs->has_sqb ? t + t_sq : nullptr;
    int rc_sc = sorted ? lfq_launch_pileup_columns(A, 1, c->stream) : lfq_launch_pileup_scatter(A, c->stream);
    if (rc_sc == LFQ_OK && nt_packed) {
        /* the layout the count kernel reads 1.5 instead of 2 bytes per observation of (LFQ_TRACKS_NT_PACKED): the scatter
         * pass writes bytes (two lanes, often of two wavefronts, would share a byte), one streaming pass packs them */
        rc_sc = lfq_launch_pack_nt(t + t_nt, t + t_ntp, n_obs, c->stream);
    }
    if (copy_pending) {                             /* on every path: the copy into the caller's array is not left in flight */
        if (hipStreamSynchronize(aux) != hipSuccess && rc_sc == LFQ_OK) {
            rc_sc = LFQ_ERR_HIP;
        }
    }
    LFQ_TRY(rc_sc);
    /* (no wait for the tracks: what consumes them -- lfq_call_snvs_batch, lfq_pileup_skip_snv_columns, the uniq calls -- is
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
    std::vector<Ev> evs_part[LFQ_HOST_PARTS];                    /* per thread, concatenated in read order below */
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
    /* by position, reads of one position in read order: every part sorts its own events on its thread ... */
    std::stable_sort(evs.begin(), evs.end(), [](const Ev &a, const Ev &b) { return a.pos < b.pos; });
    });
    /* ... and the parts, which are consecutive ranges of position-sorted reads and overlap only at their ends, are merged
     * one after the other (std::inplace_merge keeps equal positions in part order = read order) */
    const auto by_pos = [](const Ev &a, const Ev &b) { return a.pos < b.pos; };
    for (int p = 0; p < LFQ_HOST_PARTS; p++) {
        const size_t mid = evs.size();
        evs.insert(evs.end(), evs_part[p].begin(), evs_part[p].end());
        if (mid > 0 && mid < evs.size() && by_pos(evs[mid], evs[mid - 1])) {
            /* (only the tail of what is there can lie behind the new part's first event) */
            const auto first = std::upper_bound(evs.begin(), evs.begin() + (std::ptrdiff_t)mid, evs[mid], by_pos);
            std::inplace_merge(first, evs.begin() + (std::ptrdiff_t)mid, evs.end(), by_pos);
        }
    }
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
        /* Steps 2 and 3 read nothing lfq_readset_baq writes (of the flag bytes only the BI / BD bits, which its merge
         * kernel leaves as they are): while its kernels are still running they go to another stream and run beside them --
         * the BAQ kernels hold one wavefront per SIMD and 416 of its 512 registers, these kernels need 40. */
        hipStream_t ps = (rs->baq_pending && c->dps && !lfq_knobs().single_stream) ? c->dps : c->stream;
        (void)readset_pmax(c, rs, ps);      /* (its small upload is waited for on this stream: before the stream itself waits) */
        LFQ_TRY(readset_upload_wait(rs, ps));
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
            if (!A.pmax_end && !readset_unsorted_ok(c, rs)) {
                rc = LFQ_ERR_INVALID;       /* not coordinate-sorted: refused like mpileup does (lfq_set_pileup_unsorted) */
            }
        }
        if (rc == LFQ_OK) {
            rc = A.pmax_end ? lfq_launch_plp_indel_columns(A, 0, ps) : lfq_launch_plp_indel(A, 0, ps);
        }
        const bool have_qsum = A.pmax_end != nullptr;       /* the column-major kernel sums the qualities itself */
        for (int i = 0; i < (have_qsum ? 9 : 7) && rc == LFQ_OK; i++) {
            if (hipMemcpyAsync(h[i], *cnt[i], (size_t)width * 4, hipMemcpyDeviceToHost, ps) != hipSuccess) {
                rc = LFQ_ERR_HIP;
            }
        }
        scan_events();
        /* 4a. the event tables, part by part, while the counter kernel is still on its way (it waits for BI / BD, the end of
         * the upload): they need the events and the caller's arrays, nothing from the device.  Merged below (4). */
        struct PartTables {
            LfqIndelColsOwned::Side side[2];        /* key_off / rd_off: local running totals, no leading 0 */
            std::vector<int64_t> cols;              /* positions (relative to the region) with events, ascending */
            std::vector<int64_t> ev_after[2];       /* local event count of each side after each of them */
            std::vector<int64_t> rd_ev[2];          /* event index of each entry of side[sd].rd_q (for rd_aq, filled last) */
        };
        const int n_parts = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::min<long>(std::max<long>(lfq_knobs().host_loop_threads, 1), LFQ_HOST_PARTS), evs.size() / (size_t)std::max<int64_t>(lfq_knobs().host_par_min / 48, 1)));
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
                P.cols.push_back(ppos);             /* the position; its column index is known once the counters are back */
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
                                key.push_back(lfq_seq_letter(code));
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
        /* ... and appended in order (offsets shifted by what came before); per column with events: where its events end on
         * either side, and the largest quality sum of one of them */
        struct EvCol { int64_t pos; int64_t ev_after[2]; int64_t best[2]; bool has[2]; };
        std::vector<EvCol> ecols;
        std::vector<int64_t> rd_ev[2];
        for (int sd = 0; sd < 2; sd++) {
            LfqIndelColsOwned::Side &S = O.side[sd];
            S.ev_off.push_back(0);
            S.key_off.push_back(0);
            S.rd_off.push_back(0);
        }
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
                EvCol e;
                e.pos = P.cols[i];
                for (int sd = 0; sd < 2; sd++) {
                    const LfqIndelColsOwned::Side &S = O.side[sd];
                    const int64_t e0 = ecols.empty() ? 0 : ecols.back().ev_after[sd];
                    e.ev_after[sd] = ev_base[sd] + P.ev_after[sd][i];
                    e.best[sd] = 0;
                    e.has[sd] = e.ev_after[sd] > e0;
                    for (int64_t ev = e0; ev < e.ev_after[sd]; ev++) {
                        int64_t sum = 0;
                        for (int64_t k = S.rd_off[(size_t)ev]; k < S.rd_off[(size_t)ev + 1]; k++) {
                            sum += S.rd_q[(size_t)k];
                        }
                        e.best[sd] = std::max(e.best[sd], sum);
                    }
                }
                ecols.push_back(e);
            }
        }
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
        LfqVec<int32_t> col_of;                     /* column index of an event position (-1: none) */
        int64_t ne_total[2] = {0, 0};
        const bool host_arrays = c->indel_host_arrays || !have_qsum;   /* device-only needs the sums from the kernel */
        if (rc == LFQ_OK) {
            col_of.resize((size_t)width);           /* (col_of and pos_off: every position is written by the second pass) */
            std::vector<uint8_t> has_ev((size_t)width, 0);
            for (const Ev &e : evs) {
                has_ev[(size_t)(e.pos - region_begin)] = 1;
            }
            /* two passes over the positions, both split over a few threads: count the covered positions and the
             * non-event reads at event positions per part, then every part fills its slice of the column arrays */
            int64_t part_cov[LFQ_HOST_PARTS + 1] = {0}, part_ne[2][LFQ_HOST_PARTS + 1] = {{0}, {0}};
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
                    col_of[(size_t)p] = -1;
                    pos_off[0][(size_t)p] = -1;
                    pos_off[1][(size_t)p] = -1;
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
        const int64_t ncols = (int64_t)O.cov.size();
        /* the running event counts per column (the event-less columns repeat the count before them), and the consensus
         * flags of the columns with events: the largest quality sum of one event against the sum over the reads without an
         * event of that side (plp.c:1236-1270) -- the former collected at merge time, the latter from the counter kernel */
        int64_t col_done = 0;                       /* columns [0, col_done) have their ev_off entries */
        if (have_qsum) {
            O.cons_indel.assign((size_t)ncols, 0);
        }
        for (const EvCol &e : ecols) {
            const int64_t col = col_of[(size_t)e.pos];
            if (col < 0) {                          /* (cannot happen: a read with an event covers its position) */
                (void)hipStreamSynchronize(ps);     /* the pinned blocks of this call may still be a copy's source or target */
                (void)hipStreamSynchronize(c->stream);
                return LFQ_ERR_INVALID;
            }
            for (int sd = 0; sd < 2; sd++) {
                LfqIndelColsOwned::Side &S = O.side[sd];
                S.ev_off.insert(S.ev_off.end(), (size_t)(col - col_done), S.ev_off.back());
                S.ev_off.push_back(e.ev_after[sd]);
                if (have_qsum && e.has[sd] && e.best[sd] > (int64_t)qsum[sd][(size_t)col]) {
                    O.cons_indel[(size_t)col] = 1;
                }
            }
            col_done = col + 1;
        }
        for (int sd = 0; sd < 2; sd++) {            /* the event-less columns behind the last event */
            O.side[sd].ev_off.insert(O.side[sd].ev_off.end(), (size_t)(ncols - col_done), O.side[sd].ev_off.back());
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

}  // extern "C"
