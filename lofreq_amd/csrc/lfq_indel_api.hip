/*
 * lfq_indel_api.hip -- the indel tests behind the C ABI: call_indels (lofreq_call.c:619-726) as pseudo-columns on the SNV
 * path's kernels.  See include/lofreq_amd.h and DESIGN.md 1 ("Indel path").
 */
#include "lfq_ctx.h"

extern "C" {

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
    LFQ_TRY(lfq_stage_tracks(c, tr, tracks_on_device, &dev));
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
    double t_scan = 0.;                             /* the threaded scan over the columns (timing print only) */
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
        if (!b->num_ins[col] && !b->num_dels[col]) {
            continue;       /* nothing to test on either side (:684 / :706) -- nearly every column; two arrays read, not nine */
        }
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
    struct alignas(64) PartTests {          /* (a cache line of its own: the parts are filled by different threads) */
        std::vector<LfqIndelTestDesc> descs;        /* out_off relative to the part's first observation */
        std::vector<IndelPack::Meta> meta;
        uint64_t obs = 0;
    };
    PartTests part_tests[LFQ_HOST_PARTS];
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
        t_scan = lfq_now_ms() - t_begin;
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
        fprintf(stderr, "[lfq timing] indel calls: column scan %.1f of test descriptors %.1f  upload + pack %.1f  tests batch %.1f ms (%ld tests, %ld records)\n", t_scan,
                all - t_flush - t_tests, t_flush, t_tests, (long)n_tests, (long)n_out);
    }
    *n_records = n_out;
    if (n_tests_out) {
        *n_tests_out = n_tests;
    }
    return LFQ_OK;
}

}  // extern "C"
