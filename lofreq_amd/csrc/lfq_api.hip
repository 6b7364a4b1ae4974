/*
 * lfq_api.hip -- C-ABI entry points that own device state: context, workspace, the batch driver
 * (layer 1) and the call_snvs batch loop (layer 2).  See include/lofreq_amd.h.
 *
 * There is deliberately no CPU fallback in this file: every compute entry point needs a HIP
 * device and fails with LFQ_ERR_NO_DEVICE / LFQ_ERR_HIP otherwise.
 */
#include "lfq_ctx.h"

namespace {

/* The four streams of a device (= the four hardware queues HIP multiplexes a process's streams onto) are shared by
 * every context on that device.  Two contexts with four streams each would have their streams doubled up on the
 * same queues in an order nobody chose -- measured: the light chain of one batch started only after the big chain of
 * the same batch had finished -- while sharing them keeps the stream plan of lfq_snv_batch_device intact and simply
 * queues a second context's batch behind the first one's, which is what a caller that pipelines batches
 * (lfq_call_snvs_wait) wants anyway.  Completion is tracked per context with events, never by draining a stream. */
struct LfqDeviceStreams {
    hipStream_t stream = nullptr, dps = nullptr, side[2] = {nullptr, nullptr};
    /* "the last batch on this device is past its row-bound kernels" (lfq_batch_device_impl: behind the segment kernels of
     * the big and mid chains and behind the light chain): what the NEXT batch's count kernel waits for when its batch was
     * submitted while that one was still running */
    hipEvent_t ev_tail[3] = {nullptr, nullptr, nullptr};
    bool tail_recorded = false;
    /* "the last batch on this device is done with its kernels" (behind the join and the strand kernel of the sparse
     * records, in front of the small copies to pinned memory): what the next batch's count kernel waits for under
     * LFQ_GATE_END -- batches one after another on the device with no host round trip between them */
    hipEvent_t ev_end = nullptr;
    bool end_recorded = false;
    int refs = 0;
};
std::mutex g_streams_m;
LfqDeviceStreams g_streams[64];

bool acquire_streams(int device, lfq_ctx *c);
void release_streams(int device);

/* Host temporaries a DMA reads or writes come from a grow-only pool of pinned blocks owned by the context, never from
 * a std::vector.  Functionally pageable memory would do -- but the runtime registers a pageable range with the driver
 * for the copy, and when a multi-megabyte vector later goes back to the OS (free -> munmap) the driver's MMU notifier
 * evicts the process's hardware queues: the device sat idle for 20-30 ms before the next launch (measured on the
 * read-set chain: the SNV call after lfq_readset_pileup_snv, 3 processes in 4 -- whenever glibc served the vectors
 * from mmap; never with MALLOC_MMAP_THRESHOLD_ raised).  Blocks are handed back on scope exit and reused. */
void *pin_acquire_impl(lfq_ctx *c, size_t bytes, int *slot)
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

}  // namespace

int lfq_make_params(const lfq_conf *conf, const lfq_tracks *tr, LfqParams *P, bool indel_mode)
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
    P->seg_max = lfq_knobs().seg_max >= 0 ? lfq_knobs().seg_max : LFQ_SEG_MAX;      /* (lfq_batch_device_impl: by the gate) */
    P->seg_max_mid = lfq_knobs().seg_max_mid >= 0 ? lfq_knobs().seg_max_mid : LFQ_SEG_MAX;
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


namespace {

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
        ok = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipStreamCreateWithPriority(&d.dps, hipStreamNonBlocking, prio_hi) == hipSuccess;
        for (int i = 0; ok && i < 2; i++) {
            ok = hipStreamCreateWithPriority(&d.side[i], hipStreamNonBlocking, prio_hi) == hipSuccess;
        }
        for (int i = 0; ok && i < 3; i++) {
            ok = hipEventCreateWithFlags(&d.ev_tail[i], hipEventDisableTiming) == hipSuccess;
        }
        ok = ok && hipEventCreateWithFlags(&d.ev_end, hipEventDisableTiming) == hipSuccess;
        d.tail_recorded = false;
        d.end_recorded = false;
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
        for (int i = 0; i < 3; i++) {
            if (d.ev_tail[i]) (void)hipEventDestroy(d.ev_tail[i]);
        }
        if (d.ev_end) (void)hipEventDestroy(d.ev_end);
        d = LfqDeviceStreams();
    }
}

/* the shared tail events of the context's device: make `st` wait for the previous batch's / record this batch's */
int tail_wait(lfq_ctx *c, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_streams_m);
    LfqDeviceStreams &d = g_streams[c->device];
    if (c->stream != d.stream || c->batch_gate == LFQ_GATE_NONE) {
        return LFQ_OK;
    }
    if (c->batch_gate == LFQ_GATE_END) {
        if (d.end_recorded && hipStreamWaitEvent(st, d.ev_end, 0) != hipSuccess) {
            return LFQ_ERR_HIP;
        }
        return LFQ_OK;
    }
    if (!d.tail_recorded) {
        return LFQ_OK;
    }
    for (int i = 0; i < 3; i++) {
        if (hipStreamWaitEvent(st, d.ev_tail[i], 0) != hipSuccess) {
            return LFQ_ERR_HIP;
        }
    }
    return LFQ_OK;
}

int tail_record(lfq_ctx *c, int i, hipStream_t on)
{
    std::lock_guard<std::mutex> lk(g_streams_m);
    LfqDeviceStreams &d = g_streams[c->device];
    if (c->stream != d.stream) {
        return LFQ_OK;
    }
    if (hipEventRecord(d.ev_tail[i], on) != hipSuccess) {
        return LFQ_ERR_HIP;
    }
    d.tail_recorded = true;
    return LFQ_OK;
}

int end_record(lfq_ctx *c, hipStream_t on)
{
    std::lock_guard<std::mutex> lk(g_streams_m);
    LfqDeviceStreams &d = g_streams[c->device];
    if (c->stream != d.stream) {
        return LFQ_OK;
    }
    if (hipEventRecord(d.ev_end, on) != hipSuccess) {
        return LFQ_ERR_HIP;
    }
    d.end_recorded = true;
    return LFQ_OK;
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
        LFQ_TRY(grow(&c->d_tiles, &cap, 2 * (want / 1024 + 8 * (LFQ_MAX_SEGMENTS + 1))));
        cap = 0;
        LFQ_TRY(grow(&c->d_unsplit, &cap, want));
        cap = 0;
        LFQ_TRY(grow(&c->d_retry, &cap, want));
        c->ws_cols = want;
    }
    return LFQ_OK;
}


}  // namespace

void *lfq_pin_acquire(lfq_ctx *c, size_t bytes, int *slot) { return pin_acquire_impl(c, bytes, slot); }

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
    c->dense_counts = 1;
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
    if (ok && lfq_knobs().private_stream) {         /* LFQ_PRIVATE_STREAM: every context as after lfq_set_private_stream(ctx, 1) */
        ok = lfq_set_private_stream(c, 1) == LFQ_OK;
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
        if (c->priv_stream) (void)hipStreamDestroy(c->priv_stream);
        if (c->ev_up) (void)hipEventDestroy(c->ev_up);
        if (c->ev_apply) (void)hipEventDestroy(c->ev_apply);
        if (c->h_pin) (void)hipHostFree(c->h_pin);
        for (int i = 0; i < LFQ_PIN_SLOTS; i++) {
            if (c->pin_pool[i].p) (void)hipHostFree(c->pin_pool[i].p);
        }
        if (c->h_pin2) (void)hipHostFree(c->h_pin2);
        if (c->d_detlim) (void)hipFree(c->d_detlim);
        if (c->ev_baq_t[0]) (void)hipEventDestroy(c->ev_baq_t[0]);
        if (c->ev_baq_t[1]) (void)hipEventDestroy(c->ev_baq_t[1]);
        if (c->d_baq_scr) (void)hipFree(c->d_baq_scr);
        if (c->d_baq_expect) (void)hipFree(c->d_baq_expect);
        if (c->d_baq_tmp8) (void)hipFree(c->d_baq_tmp8);
        if (c->d_baq_nflag) (void)hipFree(c->d_baq_nflag);
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
int lfq_order_after_batch(lfq_ctx *c, hipStream_t st)
{
    if (c->batch_recorded) {
        LFQ_TRY_HIP(hipStreamWaitEvent(st, c->ev[3], 0));
    }
    return LFQ_OK;
}

int lfq_batch_device_impl(lfq_ctx *c, const lfq_conf *conf, const lfq_tracks *tr, lfq_col_counts *d_counts,
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
    LFQ_TRY(lfq_make_params(conf, tr, &P, indel_mode));
    /* Row segments per long column.  A batch whose DP kernels have the device to themselves wants SHORT chains: up to
     * LFQ_SEG_MAX segments side by side, put together by a fold tree.  A context that queues its batches without a gate
     * runs them beside the next batch's count kernel, where the chains' length is hidden and what they cost the step is
     * their WORK and their residency (fold workgroups of 121 KB LDS, seven convolutions per column): there a big column
     * is cut in two and a mid-class column runs on in one piece after its first stretch -- C3 2.90 -> 2.79-2.81 ms per
     * step, C2 0.60 -> 0.53 (profiles/NOTES.md; alone the same choice costs C3 3.32 -> 3.73).  LFQ_SEG_MAX[_MID|_BIG] override. */
    if (c->batch_gate == LFQ_GATE_NONE && !indel_mode) {
        if (kn.seg_max < 0) {
            P.seg_max = 2;
        }
        if (kn.seg_max_mid < 0) {
            P.seg_max_mid = 2;
        }
    }
    P.detlim_af = indel_mode ? nullptr : c->detlim_af;      /* set only inside lfq_uniq_detlim_batch */
    P.lazy_strand = (c->lazy_now && !indel_mode && !P.general && !P.detlim_af) ? 1 : 0;
    /* the context's own dense array inside lfq_call_snvs_batch without h_counts: nobody sees the entries of untested
     * columns; otherwise (a caller's array, or a submit whose collect may still ask for h_counts) only if the caller said
     * so (lfq_set_dense_counts) */
    P.sparse_counts = (P.lazy_strand && ((d_counts == c->d_counts && c->lazy_forced) || !c->dense_counts)) ? 1 : 0;
    P.pad_ = 0;
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
    LFQ_TRY(lfq_order_after_batch(c, st));
    /* a batch submitted while the device's previous one (another context's) is still running starts its count kernel when
     * that one is past its row-bound kernels: the count kernel streams through HBM beside the folds, combines and retries of
     * the other batch -- a few hundred wavefronts of LDS and FP64 work -- not beside its segment and screen kernels, which
     * it would slow down by as much as it gains (profiles/NOTES.md, two batches in flight) */
    LFQ_TRY(tail_wait(c, st));
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

    /* One launch sequence per batch.  (Cutting a batch into segments so that the DP of one runs under the count kernel of
     * the next was measured in rounds 1-3 and loses -- both want wave slots and VALU issue, profiles/NOTES.md; the loop
     * below is what is left of it, with one segment.) */
    const int n_seg = 1;
    c->cur_segments = n_seg;
    LFQ_TRY_HIP(hipEventRecord(c->ev_join[2], st));
    LFQ_TRY_HIP(hipStreamWaitEvent(dps, c->ev_join[2], 0));   /* dps starts after the memset */

    bool big_joined = false;
    for (int s = 0; s < n_seg; s++) {
        const int64_t c0 = ncols * s / n_seg, c1 = ncols * (s + 1) / n_seg;
        LfqWork W;
        W.tested_prefix = c->d_prefix;
        W.entries = c->d_entries + c0;
        W.counters = c->d_counters + s * LFQ_NCOUNTERS;
        W.gcounters = gcounters;
        W.block_sums = (int32_t *)(c->d_tiles + 2 * (c0 / 1024 + 8 * s));
        W.long_cap = c->long_cap / n_seg;
        W.pool_cells = c->pool_cells / n_seg;
        W.longs = c->d_longs + (int64_t)s * W.long_cap;
        W.pool = c->d_pool + (int64_t)s * W.pool_cells;
        W.unsplit = c->d_unsplit + c0;

        LFQ_TRY_HIP(hipEventRecord(c->ev_cnt[s][0], st));
        /* (a context that queues batches without a gate on the device's shared stream: its shallow count kernel takes half of a CU) */
        const bool ungated = c->batch_gate == LFQ_GATE_NONE && st == c->stream && !c->priv_stream_on && !single_stream;
        LFQ_TRY(lfq_launch_count(T, c0, c1, P, c->d_luts, d_counts, c->d_flags, max_depth, st, ungated ? kn.count_shallow_wgs_none : 0));
        c->cur_sparse_counts = P.sparse_counts && lfq_count_is_shallow(T, P, max_depth);
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
        /* DP4 tuples of the mid / big class alleles with >= 16 alt bases -> host; Fisher tests start when they arrive.  On the
         * dps stream behind the scan: in front of the screen kernel (the host starts 0.1-0.4 ms earlier) or behind it
         * (LFQ_HEAVY_AFTER_SCREEN: one launch less between the scan and the light chain, which is what a shallow batch's
         * period is made of when batches are gated on the chains' tails) */
        const bool sb_heavy = n_seg == 1 && !indel_mode && c->leader && !kn.no_sb_precompute;
        auto launch_heavy = [&]() -> int {
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
            return LFQ_OK;
        };
        const bool heavy_late = sb_heavy && kn.heavy_after_screen && !kn.skip_light && kn.light_kernel != 2;
        if (sb_heavy && !heavy_late) {
            LFQ_TRY(launch_heavy());
        }
        if (kn.tail_light == 2) {
            LFQ_TRY(tail_record(c, 2, dps));
        }
        const int64_t seg_cols = c1 - c0;
        /* the light kernel is throughput work and persistent: it must leave wave slots for the short
         * latency-bound kernels of the long columns, or they only start when it ends */
        /* How many screen wavefronts per CU follows the gate like the row segments do (round 6): alone, four make the light
         * chain short (0.69 ms at C3); a context that queues its batches without a gate runs the chain beside the next batch's
         * count kernel, where its length is hidden (1.7 ms with one wavefront per CU against a period of 2.8) and every resident
         * screen wavefront costs the count kernel a slot -- C3 2.82-2.84 -> 2.77-2.79 ms per step with one, C2 0.523-0.528 ->
         * 0.505-0.512 with one or two; shallow batches keep two (their chain is the longer part of a shorter period). */
        const int screen_auto = (c->batch_gate == LFQ_GATE_NONE && !indel_mode) ? (tr->max_col_obs >= 4096 ? 1 : 2) : 4;
        const int light_waves_per_cu = kn.light_kernel == 0 ? (kn.screen_waves_per_cu >= 1 ? kn.screen_waves_per_cu : screen_auto) : 10;
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
        /* The unsplit big columns (lfq_dp_big_kernel: what the prep kernel queued because it is too short or too wide to cut
         * -- every big column of a 1000x batch) need nothing from the segment / fold / combine kernels of the split ones.  With
         * one segment per batch the stream the count kernel ran on is idle from here on: they run there, beside that chain
         * instead of behind it (C2: 0.3 ms that used to start when the chain had ended). */
        /* (not for a context whose batches are queued without a gate: the next batch's count kernel is queued on `st`) */
        const bool big_on_st = run_big && !single_stream && !kn.big_on_side && c->batch_gate != LFQ_GATE_NONE;
        if (big_on_st) {
            LFQ_TRY_HIP(hipStreamWaitEvent(st, c->ev_prep, 0));
            LFQ_TRY(lfq_launch_dp_big(T, P, c->d_luts, d_counts, W, d_pvals, pvals_capacity, c->d_scratch, per_block,
                                      n_big_blocks, st));
            LFQ_DBG_STAGE("big");
            LFQ_TRY_HIP(hipEventRecord(c->ev_segw, st));
            big_joined = true;
        }
        if (run_big) {
            LFQ_TRY(lfq_launch_dp_seg(1, T, P, c->d_luts, W, c->n_cu * 8, side0));
            LFQ_DBG_STAGE("seg big");
            LFQ_TRY(tail_record(c, 0, side0));
            LFQ_TRY(lfq_launch_dp_combine(1, P, d_counts, W, d_pvals, pvals_capacity, c->n_cu, side0));
            LFQ_DBG_STAGE("combine big");
            if (!big_on_st) {
                LFQ_TRY(lfq_launch_dp_big(T, P, c->d_luts, d_counts, W, d_pvals, pvals_capacity, c->d_scratch, per_block,
                                          n_big_blocks, side0));
                LFQ_DBG_STAGE("big");
            }
        }
        if (!run_big) {
            LFQ_TRY(tail_record(c, 0, side0));  /* (every batch records all three tail events, wherever its chains stand) */
        }
        LFQ_TRY_HIP(hipStreamWaitEvent(side1, c->ev_prep, 0));     /* K = 250..252 of the big class lands in class 1 */
        if (run_mid || run_big) {
            LFQ_TRY(lfq_launch_dp_seg(0, T, P, c->d_luts, W, c->n_cu * 8, side1));
            LFQ_DBG_STAGE("seg mid");
            LFQ_TRY(tail_record(c, 1, side1));
            LFQ_TRY(lfq_launch_dp_combine(0, P, d_counts, W, d_pvals, pvals_capacity, c->n_cu, side1));
            LFQ_DBG_STAGE("combine mid");
        }
        if (!(run_mid || run_big)) {
            LFQ_TRY(tail_record(c, 1, side1));
        }
        LFQ_TRY_HIP(hipEventRecord(c->ev_light[s][0], dps));
        if (!kn.skip_light) {
            if (kn.light_kernel == 2) {                 /* A/B switch: the one-column-per-wavefront kernel */
                LFQ_TRY(lfq_launch_dp_light(T, P, c->d_luts, d_counts, W, d_pvals, pvals_capacity, n_light_waves, dps));
            } else {
                const int kh = indel_mode ? c->kreg_hint_indel : c->kreg_hint;
                LFQ_TRY(lfq_launch_dp_quad(T, P, c->d_luts, d_counts, W, c->d_retry + c0, d_pvals, pvals_capacity,
                                           n_light_waves, kh, dps, 1));
                if (kn.tail_light == 1) {       /* the retry kernel beside the next batch's count kernel, like the folds */
                    LFQ_TRY(tail_record(c, 2, dps));
                }
                if (heavy_late) {
                    LFQ_TRY(launch_heavy());
                }
                LFQ_TRY(lfq_launch_dp_quad(T, P, c->d_luts, d_counts, W, c->d_retry + c0, d_pvals, pvals_capacity,
                                           n_light_waves, kh, dps, 2));
            }
        }
        LFQ_TRY_HIP(hipEventRecord(c->ev_light[s][1], dps));
        if (kn.tail_light == 0 || (kn.tail_light == 1 && (kn.skip_light || kn.light_kernel == 2))) {
            LFQ_TRY(tail_record(c, 2, dps));
        }
        for (int i = 0; i < 2; i++) {
            LFQ_TRY_HIP(hipEventRecord(c->ev_side[i][s][1], side_i[i]));
        }
    }
    /* Join.  A caller that passed its own stream gets everything joined back into it.  On the library's own streams
     * (layer 2) the batch ends on the dps stream instead and `st` carries nothing but the memsets and count kernels:
     * the count kernel of the NEXT batch (another context on the same device streams) then starts as soon as this
     * batch's count kernel is done and streams through HBM while this batch's DP kernels -- latency / issue-bound,
     * < 1 GB of traffic, on the high-priority streams -- run beside it. */
    /* (LFQ_JOIN_ON_SIDE: on the big chain's stream, which is the last to end anyway -- the dps stream is then free for the NEXT
     * batch's scan as soon as this batch's retry kernel is done, instead of when its last fold is) */
    hipStream_t jn = (st != c->stream || single_stream) ? st : (kn.join_on_side ? side0 : dps);
    LFQ_TRY_HIP(hipEventRecord(c->ev[1], st));                     /* all count kernels done */
    LFQ_TRY_HIP(hipEventRecord(c->ev_join[0], side0));
    LFQ_TRY_HIP(hipEventRecord(c->ev_join[1], side1));
    LFQ_TRY_HIP(hipEventRecord(c->ev_join[2], dps));
    for (int i = 0; i < 3; i++) {
        LFQ_TRY_HIP(hipStreamWaitEvent(jn, c->ev_join[i], 0));
    }
    if (big_joined && jn != st) {
        LFQ_TRY_HIP(hipStreamWaitEvent(jn, c->ev_segw, 0));        /* the unsplit big columns on `st` */
    }
    if (P.lazy_strand && d_pvals && pvals_capacity > 0) {
        /* DP4 of the columns that made it into the sparse output (lofreq_call.c:853-857) */
        LFQ_TRY(lfq_launch_strand_pvals(T, d_pvals, gcounters + LFQ_GC_PVALS, pvals_capacity, c->n_cu, jn));
    }
    LFQ_TRY(end_record(c, jn));     /* LFQ_GATE_END: the next batch's count kernel may start here */
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
    return lfq_batch_device_impl(c, conf, tr, d_counts, d_pvals, pvals_capacity, stream_or_null, false);
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

int lfq_set_pileup_unsorted(lfq_ctx *c, int on)
{
    if (!c) {
        return LFQ_ERR_INVALID;
    }
    c->plp_unsorted_ok = on ? 1 : 0;
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

int lfq_set_batch_gate(lfq_ctx *c, int gate)
{
    if (!c || (gate != LFQ_GATE_TAIL && gate != LFQ_GATE_END && gate != LFQ_GATE_NONE)) {
        return LFQ_ERR_INVALID;
    }
    c->batch_gate = gate;
    return LFQ_OK;
}

int lfq_set_private_stream(lfq_ctx *c, int on)
{
    if (!c || !c->own_streams) {
        return LFQ_ERR_INVALID;
    }
    LFQ_TRY_HIP(hipSetDevice(c->device));
    LFQ_TRY(lfq_synchronize(c));
    LFQ_TRY_HIP(hipStreamSynchronize(c->stream));
    if (on) {
        if (!c->priv_stream) {
            LFQ_TRY_HIP(hipStreamCreateWithFlags(&c->priv_stream, hipStreamNonBlocking));
        }
        c->stream = c->priv_stream;
        c->priv_stream_on = 1;
    } else {
        std::lock_guard<std::mutex> lk(g_streams_m);
        c->stream = g_streams[c->device].stream;
        c->priv_stream_on = 0;
    }
    return LFQ_OK;
}

int lfq_set_dense_counts(lfq_ctx *c, int on)
{
    if (!c) {
        return LFQ_ERR_INVALID;
    }
    c->dense_counts = on ? 1 : 0;
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
    return lfq_batch_device_impl(c, conf, tr, d_counts, d_pvals, pvals_capacity, stream_or_null, true);
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
    if (c->cur_sparse_counts) {                     /* the shared-wavefront kernel stored the tested columns' entries only */
        c->cur_count_written = c->cur_ncols + (int64_t)c->h_counters[LFQ_MAX_SEGMENTS * LFQ_NCOUNTERS + LFQ_GC_TESTED] * (int64_t)sizeof(lfq_col_counts);
    }
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
    /* (the end of the last count kernel, not ev[1]: the stream of the count kernels may carry the unsplit big columns behind it) */
    if (c->cur_segments > 0 && hipEventElapsedTime(&dp_end, c->ev_cnt[c->cur_segments - 1][1], c->ev[3]) == hipSuccess) {
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
int lfq_stage_tracks(lfq_ctx *c, const lfq_tracks *tr, int tracks_on_device, lfq_tracks *dev_out)
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
        LFQ_TRY(lfq_order_after_batch(c, ups));            /* the previous batch may still read the staging area */
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
    LFQ_TRY(lfq_stage_tracks(c, tr, tracks_on_device, &dev));
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
    if (ncols > 0 && h_counts_or_null && c->cur_sparse_counts) {
        /* the count kernel skipped the entries of the untested columns (lfq_set_dense_counts(ctx, 0)): there is no complete
         * dense array to hand out.  The batch stays collectable without h_counts. */
        return LFQ_ERR_INVALID;
    }
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
    LFQ_TRY(lfq_stage_tracks(c, tr, tracks_on_device, &dev));
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
    LFQ_TRY(lfq_stage_tracks(c, tr, tracks_on_device, &dev));
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
    LFQ_TRY(lfq_order_after_batch(c, c->stream));
    LFQ_TRY(grow(&c->d_tmp[2], &c->tmp_bytes[2], ncols));
    LFQ_TRY_HIP(hipMemcpyAsync(c->d_tmp[2], h.data(), (size_t)ncols, hipMemcpyHostToDevice, c->stream));
    LFQ_TRY(lfq_launch_skip_columns(c->d_plp_nb, c->d_tmp[2], ncols, c->stream));
    LFQ_TRY_HIP(hipStreamSynchronize(c->stream));
    return LFQ_OK;
}

/* source_qual for a batch of reads (plp.c:427-593): counting + DP on the device, phred conversion here */
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
    LFQ_TRY(lfq_order_after_batch(c, st));                 /* the target may be the tracks of a batch that is still running */
    return lfq_launch_synth(&s, col_begin, ncols, d_nt, d_bq, d_baq, d_mq, d_col_off, d_ref_base, nt_packed ? 1 : 0, st);
}

}  // extern "C"
