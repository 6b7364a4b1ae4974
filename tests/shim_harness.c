/*
 * shim_harness.c -- TEST INFRASTRUCTURE (tests/test_shim.py): a mock mpileup() and a mock liblofreq_amd around
 * integration/lofreq_amd_shim.c, compiled with the reference's own headers and utils.c / log.c.
 *
 *   mock mpileup   reads columns from a binary stream (written by the test from a golden fixture), builds a
 *                  plp_col_t on its stack exactly the way compile_plp_col fills one (plp.c:797-1288: one
 *                  int_varray per nucleotide through PLP_COL_ADD_QUAL = int_varray_add_value, indel events
 *                  through add_ins_sequence / add_del_sequence, i.e. uthash in insertion order), hands it to the
 *                  shim's plp_proc_func, and FREES everything right after the callback returns, like
 *                  plp.c:1440-1445 does -- a shim that kept pointers into the column would read freed memory;
 *   mock library   lfq_call_snvs_batch / lfq_call_indels_batch dump the packed batch they are handed to a file
 *                  and report no records; no GPU, no liblofreq_amd.so.
 *
 * Nothing here is part of the product.
 */
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lofreq_amd.h"
#include "log.h"
#include "plp.h"
#include "snpcaller.h"
#include "utils.h"
#include "vcf.h"

/* globals of lofreq_call.c:84-88 the shim updates */
long long int num_snv_tests = 0;
long long int num_indel_tests = 0;
long int indel_calls_wo_idaq = 0;

void lfq_call_vars(const plp_col_t *p, void *confp);
void lfq_call_flush(varcall_conf_t *conf);
void lfq_call_shutdown(void);

static FILE *g_out;

int vcf_printf(vcf_file_t *f, char *fmt, ...)          /* vcf.h:102; the shim prints records through it */
{
    va_list ap;
    int n;
    (void)f;
    va_start(ap, fmt);
    n = vprintf(fmt, ap);
    va_end(ap);
    return n;
}

/* ---- mock liblofreq_amd ---------------------------------------------------------------------------- */
int lfq_create(lfq_ctx **ctx, int device_ordinal)
{
    (void)device_ordinal;
    *ctx = (lfq_ctx *)malloc(8);
    return LFQ_OK;
}
void lfq_destroy(lfq_ctx *ctx) { free(ctx); }
int lfq_set_dense_strand_counts(lfq_ctx *ctx, int on) { (void)ctx; (void)on; return LFQ_OK; }
int lfq_set_dense_counts(lfq_ctx *ctx, int on) { (void)ctx; (void)on; return LFQ_OK; }
int lfq_abi_version(void) { return LFQ_ABI_VERSION; }
int lfq_pick_device(int n_devices, int *slot) { (void)n_devices; if (slot) *slot = -1; return 0; }
const char *lfq_strerror(int status) { (void)status; return "mock"; }
void lfq_conf_init(lfq_conf *c)
{
    memset(c, 0, sizeof(*c));
    c->bonf_subst = 1;
    c->bonf_indel = 1;
}
int lfq_format_snv_record(char *buf, int buflen, const char *chrom, int64_t pos0, const lfq_snv_record *rec,
                          const char *filter_or_null)
{
    (void)rec; (void)filter_or_null;
    return snprintf(buf, (size_t)buflen, "%s\t%ld\n", chrom, (long)pos0 + 1);
}
int lfq_format_indel_record(char *buf, int buflen, const char *chrom, int64_t pos0, const char *ref, const char *alt,
                            int qual, int dp, float af, int sb, int ref_fw, int ref_rv, int alt_fw, int alt_rv,
                            int hrun, const char *filter_or_null)
{
    (void)qual; (void)dp; (void)af; (void)sb; (void)ref_fw; (void)ref_rv; (void)alt_fw; (void)alt_rv; (void)hrun;
    (void)filter_or_null;
    return snprintf(buf, (size_t)buflen, "%s\t%ld\t%s\t%s\n", chrom, (long)pos0 + 1, ref, alt);
}

static void put(const void *p, size_t n) { fwrite(p, 1, n, g_out); }
static void put_i64(int64_t v) { put(&v, 8); }

void *lfq_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
void lfq_host_free(void *p) { free(p); }

/* the shim submits a batch when it is full and collects it at the next flush: the mock dumps at submit (the tracks must
 * be complete by then) and hands back no records at collect */
static int64_t g_sub_ncols = -1;
int lfq_call_snvs_submit(lfq_ctx *ctx, const lfq_conf *conf, const lfq_tracks *t, int tracks_on_device)
{
    int64_t n = 0;
    if (g_sub_ncols >= 0) return LFQ_ERR_INVALID;         /* one batch in flight per context */
    g_sub_ncols = t->ncols;
    return lfq_call_snvs_batch(ctx, (lfq_conf *)conf, t, tracks_on_device, NULL, 0, &n, NULL, NULL) == LFQ_OK ? LFQ_OK : LFQ_ERR_INVALID;
}
int lfq_call_snvs_collect(lfq_ctx *ctx, lfq_conf *conf, lfq_snv_record *records, int64_t records_capacity,
                          int64_t *n_records, lfq_col_counts *h_counts_or_null, lfq_batch_stats *stats_out)
{
    (void)ctx; (void)records; (void)records_capacity; (void)h_counts_or_null; (void)stats_out;
    if (g_sub_ncols < 0) return LFQ_ERR_INVALID;
    *n_records = 0;
    conf->num_snv_tests += 3 * g_sub_ncols;     /* so that the shim's write-back of the counters can be seen */
    g_sub_ncols = -1;
    return LFQ_OK;
}

int lfq_call_snvs_batch(lfq_ctx *ctx, lfq_conf *conf, const lfq_tracks *t, int tracks_on_device,
                        lfq_snv_record *records, int64_t records_capacity, int64_t *n_records,
                        lfq_col_counts *h_counts_or_null, lfq_batch_stats *stats)
{
    const int64_t n_obs = (int64_t)t->col_off[t->ncols];
    (void)ctx; (void)records; (void)records_capacity; (void)h_counts_or_null; (void)stats;
    put("SNVB", 4);
    put_i64(t->ncols);
    put_i64(n_obs);
    put_i64(tracks_on_device);
    put_i64(t->baq != NULL);
    put_i64(t->sq != NULL);
    put_i64(t->max_col_obs);
    put_i64(conf->bonf_subst);
    put(t->col_off, (size_t)(t->ncols + 1) * 8);
    put(t->ref_base, (size_t)t->ncols);
    put(t->coverage_plp, (size_t)t->ncols * 4);
    put(t->num_bases, (size_t)t->ncols * 4);
    put_i64(t->flags);
    put(t->nt, (t->flags & LFQ_TRACKS_NT_PACKED) ? (size_t)((n_obs + 7) / 8 * 4) : (size_t)n_obs);
    put(t->bq, (size_t)n_obs);
    put(t->mq, (size_t)n_obs);
    if (t->baq) put(t->baq, (size_t)n_obs);
    if (t->sq) put(t->sq, (size_t)n_obs);
    *n_records = 0;
    return LFQ_OK;
}

int lfq_call_indels_batch(lfq_ctx *ctx, lfq_conf *conf, const lfq_indel_columns *c, lfq_indel_record *records,
                          int64_t records_capacity, int64_t *n_records, int64_t *n_tests)
{
    int s;
    (void)ctx; (void)records; (void)records_capacity;
    put("INDB", 4);
    put_i64(c->ncols);
    put(c->ref_base, (size_t)c->ncols);
    put(c->coverage_plp, (size_t)c->ncols * 4);
    put(c->num_tails, (size_t)c->ncols * 4);
    put(c->num_non_indels, (size_t)c->ncols * 4);
    put(c->num_ins, (size_t)c->ncols * 4);
    put(c->num_dels, (size_t)c->ncols * 4);
    put(c->hrun, (size_t)c->ncols * 4);
    for (s = 0; s < 2; s++) {
        const lfq_indel_side *d = &c->side[s];
        const int64_t n_ne = d->ne_off[c->ncols], n_ev = d->ev_off[c->ncols];
        const int64_t n_rd = d->rd_off[n_ev], n_key = d->key_off[n_ev];
        put_i64(n_ne); put_i64(n_ev); put_i64(n_rd); put_i64(n_key);
        put(d->non_fw, (size_t)c->ncols * 4);
        put(d->non_rv, (size_t)c->ncols * 4);
        put(d->ne_off, (size_t)(c->ncols + 1) * 8);
        put(d->ne_q, (size_t)n_ne * 2);
        put(d->ne_mq, (size_t)n_ne * 2);
        put(d->ev_off, (size_t)(c->ncols + 1) * 8);
        put(d->key_off, (size_t)(n_ev + 1) * 8);
        put(d->key_chars, (size_t)n_key);
        put(d->ev_fw, (size_t)n_ev * 4);
        put(d->ev_rv, (size_t)n_ev * 4);
        put(d->rd_off, (size_t)(n_ev + 1) * 8);
        put(d->rd_q, (size_t)n_rd * 2);
        put(d->rd_aq, (size_t)n_rd * 2);
        put(d->rd_mq, (size_t)n_rd * 2);
        put(d->rd_sq, (size_t)n_rd * 2);
    }
    *n_records = 0;
    *n_tests = 0;
    (void)conf;
    return LFQ_OK;
}

/* ---- mock mpileup ------------------------------------------------------------------------------------ */
static int32_t rd_i32(FILE *f)
{
    int32_t v = 0;
    if (fread(&v, 4, 1, f) != 1) {
        fprintf(stderr, "harness: short input\n");
        exit(2);
    }
    return v;
}

static void col_init(plp_col_t *p)
{
    int i;
    const size_t grow = 16384;                   /* plp.c:140 */
    memset(p, 0, sizeof(*p));
    for (i = 0; i < NUM_NT4; i++) {
        int_varray_init(&p->base_quals[i], grow);
        int_varray_init(&p->baq_quals[i], grow);
        int_varray_init(&p->map_quals[i], grow);
        int_varray_init(&p->source_quals[i], grow);
    }
    int_varray_init(&p->ins_quals, grow);
    int_varray_init(&p->ins_map_quals, grow);
    int_varray_init(&p->ins_source_quals, grow);
    int_varray_init(&p->del_quals, grow);
    int_varray_init(&p->del_map_quals, grow);
    int_varray_init(&p->del_source_quals, grow);
}

static void col_free(plp_col_t *p)               /* plp_col_free, plp.c:184-208 */
{
    int i;
    for (i = 0; i < NUM_NT4; i++) {
        int_varray_free(&p->base_quals[i]);
        int_varray_free(&p->baq_quals[i]);
        int_varray_free(&p->map_quals[i]);
        int_varray_free(&p->source_quals[i]);
    }
    int_varray_free(&p->ins_quals);
    int_varray_free(&p->ins_map_quals);
    int_varray_free(&p->ins_source_quals);
    int_varray_free(&p->del_quals);
    int_varray_free(&p->del_map_quals);
    int_varray_free(&p->del_source_quals);
    destruct_ins_event_counts(&p->ins_event_counts);
    destruct_del_event_counts(&p->del_event_counts);
    free(p->target);
    memset(p, 0xA5, sizeof(*p));                 /* poison: nothing of the column may be used after the callback */
}

int main(int argc, char **argv)
{
    FILE *in;
    varcall_conf_t conf;
    int32_t ncols, c;
    if (argc != 3) {
        fprintf(stderr, "usage: %s columns.bin out.bin\n", argv[0]);
        return 2;
    }
    in = fopen(argv[1], "rb");
    g_out = fopen(argv[2], "wb");
    if (!in || !g_out) {
        return 2;
    }
    memset(&conf, 0, sizeof(conf));
    conf.bonf_dynamic = rd_i32(in);
    conf.bonf_subst = rd_i32(in);
    conf.bonf_indel = 1;
    conf.no_indels = rd_i32(in);
    conf.only_indels = rd_i32(in);
    conf.flag = rd_i32(in);
    conf.sig = 0.01f;
    ncols = rd_i32(in);
    for (c = 0; c < ncols; c++) {
        plp_col_t col;
        int nt, s;
        col_init(&col);
        col.target = strdup("chr1");
        col.pos = rd_i32(in);
        col.ref_base = (char)rd_i32(in);
        col.cons_base[0] = (char)rd_i32(in);
        col.coverage_plp = rd_i32(in);
        col.num_bases = rd_i32(in);
        col.num_tails = rd_i32(in);
        col.num_non_indels = rd_i32(in);
        col.num_ins = rd_i32(in);
        col.num_dels = rd_i32(in);
        col.hrun = rd_i32(in);
        col.has_indel_aqs = rd_i32(in);
        for (nt = 0; nt < NUM_NT4; nt++) {
            const int32_t n = rd_i32(in), fw = rd_i32(in), has_baq = rd_i32(in), has_sq = rd_i32(in);
            int32_t j;
            col.fw_counts[nt] = fw;
            col.rv_counts[nt] = n - fw;
            for (j = 0; j < n; j++) {
                const int32_t bq = rd_i32(in), baq = rd_i32(in), mq = rd_i32(in), sq = rd_i32(in);
                PLP_COL_ADD_QUAL(&col.base_quals[nt], bq);
                if (has_baq) PLP_COL_ADD_QUAL(&col.baq_quals[nt], baq);
                PLP_COL_ADD_QUAL(&col.map_quals[nt], mq);
                if (has_sq) PLP_COL_ADD_QUAL(&col.source_quals[nt], sq);
            }
        }
        for (s = 0; s < 2; s++) {
            const int32_t non_fw = rd_i32(in), non_rv = rd_i32(in), n_ne = rd_i32(in);
            int32_t j, n_ev, e;
            if (s == 0) { col.non_ins_fw_rv[0] = non_fw; col.non_ins_fw_rv[1] = non_rv; }
            else        { col.non_del_fw_rv[0] = non_fw; col.non_del_fw_rv[1] = non_rv; }
            for (j = 0; j < n_ne; j++) {
                const int32_t q = rd_i32(in), mq = rd_i32(in);
                PLP_COL_ADD_QUAL(s == 0 ? &col.ins_quals : &col.del_quals, q);
                PLP_COL_ADD_QUAL(s == 0 ? &col.ins_map_quals : &col.del_map_quals, mq);
            }
            n_ev = rd_i32(in);
            for (e = 0; e < n_ev; e++) {
                char key[MAX_INDELSIZE];
                const int32_t kl = rd_i32(in);
                int32_t fw, n;
                if (kl <= 0 || kl >= MAX_INDELSIZE || fread(key, 1, (size_t)kl, in) != (size_t)kl) {
                    return 2;
                }
                key[kl] = 0;
                fw = rd_i32(in);
                n = rd_i32(in);
                for (j = 0; j < n; j++) {        /* one read at a time, like compile_plp_col (plp.c:1094-1180) */
                    const int32_t q = rd_i32(in), aq = rd_i32(in), mq = rd_i32(in), sq = rd_i32(in);
                    if (s == 0) add_ins_sequence(&col.ins_event_counts, key, q, aq, mq, sq, j < fw ? 0 : 1);
                    else        add_del_sequence(&col.del_event_counts, key, q, aq, mq, sq, j < fw ? 0 : 1);
                }
            }
        }
        lfq_call_vars(&col, &conf);              /* plp.c:1440 */
        col_free(&col);                          /* plp.c:1445 */
    }
    lfq_call_flush(&conf);
    put("DONE", 4);
    put_i64(conf.bonf_subst);
    put_i64(num_snv_tests);
    lfq_call_shutdown();
    fclose(g_out);
    fclose(in);
    return 0;
}
