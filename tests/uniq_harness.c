/*
 * uniq_harness.c -- mock mpileup + mock liblofreq_amd for integration/lofreq_amd_uniq.c (tests/test_uniq_binding.py).
 * Compiled against the REFERENCE's own plp.h / vcf.h / utils.h and linked with the reference's utils.c + log.c; plays
 * main_uniq's loop (lofreq_uniq.c:690-730): one callback per variant with a plp_col_t built the way compile_plp_col fills
 * it (int_varray_add_value per nucleotide, add_ins_sequence / add_del_sequence), freed and poisoned right after the
 * callback (plp.c:1440-1445), then lfq_uniq_flush.  The mock library writes the batch it is handed to a file and answers
 * with the values the test supplied; the harness prints every variant's INFO afterwards.
 *
 * vcf.c itself needs htslib's bgzf (absent here), so the three vcf_var_* functions the binding calls are provided by this
 * file, written from their contracts in vcf.h:119-123 (the binding's own logic is what is under test).
 *
 * input stream (int32 little endian; strings as length + bytes):
 *   use_det_lim, uni_freq (float bits), n_vars, then per variant
 *   chrom, ref, alt, info,  pos0, has_column, column_pos0, ref_base, coverage_plp, num_tails, canned_result,
 *   5 x { n, fw, n x { bq, mq } },  n_ins_events x { key, count },  n_del_events x { key, count }
 */
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lofreq_amd.h"
#include "lofreq_amd_uniq.h"
#include "log.h"
#include "plp.h"
#include "utils.h"
#include "vcf.h"

static FILE *g_out;
static int32_t *g_canned;           /* per column of the batch, in add order */
static int g_n_canned;

/* ---- the reference's vcf_var_* contracts (vcf.h:119-123), restated for var_t as this harness builds it ---- */
int vcf_var_has_info_key(char **value, const var_t *var, const char *key)
{
    const size_t kl = strlen(key);
    const char *s = var->info;
    if (value) *value = NULL;
    while (s && *s) {
        const char *e = strchr(s, ';');
        const size_t l = e ? (size_t)(e - s) : strlen(s);
        if (l >= kl && 0 == strncmp(s, key, kl) && (l == kl || s[kl] == '=')) {
            if (value && l > kl) {
                *value = (char *)calloc(l - kl, 1);
                memcpy(*value, s + kl + 1, l - kl - 1);
            }
            return 1;
        }
        s = e ? e + 1 : NULL;
    }
    return 0;
}

int vcf_var_is_indel(const var_t *var)
{
    return strlen(var->ref) > 1 || strlen(var->alt) > 1 || vcf_var_has_info_key(NULL, var, "INDEL");
}

char *vcf_var_add_to_info(var_t *var, const char *info_str)
{
    const size_t a = strlen(var->info), b = strlen(info_str);
    if (a == 0 || (a == 1 && var->info[0] == '.')) {
        free(var->info);
        var->info = strdup(info_str);
        return var->info;
    }
    var->info = (char *)realloc(var->info, a + b + 2);
    var->info[a] = ';';
    memcpy(var->info + a + 1, info_str, b + 1);
    return var->info;
}

/* ---- mock liblofreq_amd ---------------------------------------------------------------------------- */
int lfq_create(lfq_ctx **ctx, int device_ordinal)
{
    (void)device_ordinal;
    *ctx = (lfq_ctx *)malloc(8);
    return LFQ_OK;
}
void lfq_destroy(lfq_ctx *ctx) { free(ctx); }
int lfq_abi_version(void) { return LFQ_ABI_VERSION; }
int lfq_pick_device(int n_devices, int *slot) { (void)n_devices; if (slot) *slot = -1; return 0; }
const char *lfq_strerror(int status) { (void)status; return "mock"; }

static void put(const void *p, size_t n) { fwrite(p, 1, n, g_out); }
static void put_i64(int64_t v) { put(&v, 8); }

static void dump_tracks(const char *tag, const lfq_tracks *t, int tracks_on_device, const float *af)
{
    const int64_t n_obs = (int64_t)t->col_off[t->ncols];
    put(tag, 4);
    put_i64(t->ncols);
    put_i64(n_obs);
    put_i64(tracks_on_device);
    put_i64(t->baq != NULL);
    put_i64(t->sq != NULL);
    put_i64(t->max_col_obs);
    put_i64(t->coverage_plp != NULL);
    put_i64(t->flags);
    put(t->col_off, (size_t)(t->ncols + 1) * 8);
    put(t->ref_base, (size_t)t->ncols);
    if (t->coverage_plp) put(t->coverage_plp, (size_t)t->ncols * 4);
    put(af, (size_t)t->ncols * 4);
    put(t->nt, (size_t)((n_obs + 7) / 8 * 4));
    put(t->bq, (size_t)n_obs);
    put(t->mq, (size_t)n_obs);
}

int lfq_uniq_detlim_batch(lfq_ctx *ctx, const lfq_tracks *t, int tracks_on_device, const float *af, uint8_t *detectable,
                          long double *pvalue_or_null)
{
    int64_t i;
    (void)ctx; (void)pvalue_or_null;
    dump_tracks("UDET", t, tracks_on_device, af);
    if (t->ncols != g_n_canned) return LFQ_ERR_INVALID;
    for (i = 0; i < t->ncols; i++) detectable[i] = (uint8_t)g_canned[i];
    return LFQ_OK;
}

int lfq_uniq_binom_batch(lfq_ctx *ctx, const lfq_tracks *t, int tracks_on_device, const float *af, const char *alt_base,
                         int32_t *uq_out, double *pvalue_or_null)
{
    int64_t i;
    (void)ctx; (void)pvalue_or_null;
    dump_tracks("UBIN", t, tracks_on_device, af);
    put(alt_base, (size_t)t->ncols);
    if (t->ncols != g_n_canned) return LFQ_ERR_INVALID;
    for (i = 0; i < t->ncols; i++) uq_out[i] = g_canned[i];
    return LFQ_OK;
}

double lfq_binom_cdf(int n, int k, double pr, int *status_or_null)
{
    put("BINO", 4);
    put_i64(n);
    put_i64(k);
    put(&pr, 8);
    if (status_or_null) *status_or_null = (k > n) ? 3 : 0;      /* cdfbin rejects s > xn: "binom() failed", no tag */
    return 0.25;                                                 /* (int)(-10 log10l(0.25)) = 6 */
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

static char *rd_str(FILE *f)
{
    const int32_t n = rd_i32(f);
    char *s = (char *)calloc((size_t)n + 1, 1);
    if (n > 0 && fread(s, 1, (size_t)n, f) != (size_t)n) {
        fprintf(stderr, "harness: short input\n");
        exit(2);
    }
    return s;
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
    destruct_ins_event_counts(&p->ins_event_counts);
    destruct_del_event_counts(&p->del_event_counts);
    free(p->target);
    memset(p, 0xA5, sizeof(*p));                 /* poison: nothing of the column may be used after the callback */
}

int main(int argc, char **argv)
{
    FILE *in;
    lfq_uniq_binding ub;
    var_t **vars;
    int32_t n_vars, v, bits;
    if (argc != 3) {
        fprintf(stderr, "usage: %s variants.bin out.bin\n", argv[0]);
        return 2;
    }
    in = fopen(argv[1], "rb");
    g_out = fopen(argv[2], "wb");
    if (!in || !g_out) {
        return 2;
    }
    memset(&ub, 0, sizeof(ub));
    ub.use_det_lim = rd_i32(in);
    bits = rd_i32(in);
    memcpy(&ub.uni_freq, &bits, 4);
    n_vars = rd_i32(in);
    vars = (var_t **)calloc((size_t)n_vars + 1, sizeof(*vars));
    g_canned = (int32_t *)calloc((size_t)n_vars + 1, sizeof(int32_t));
    for (v = 0; v < n_vars; v++) {               /* main_uniq's loop, lofreq_uniq.c:692-726 */
        var_t *var = (var_t *)calloc(1, sizeof(var_t));
        plp_col_t col;
        int32_t has_col, col_pos, canned, nt, e, n_ev;
        int is_indel_binom;
        var->chrom = rd_str(in);
        var->ref = rd_str(in);
        var->alt = rd_str(in);
        var->info = rd_str(in);
        var->id = strdup(".");
        var->filter = strdup(".");
        var->pos = rd_i32(in);
        has_col = rd_i32(in);
        col_pos = rd_i32(in);
        vars[v] = var;
        ub.var = var;
        col_init(&col);
        col.target = strdup(var->chrom);
        col.pos = col_pos;
        col.ref_base = (char)rd_i32(in);
        col.coverage_plp = rd_i32(in);
        col.num_tails = rd_i32(in);
        canned = rd_i32(in);
        for (nt = 0; nt < NUM_NT4; nt++) {
            const int32_t n = rd_i32(in), fw = rd_i32(in);
            int32_t k;
            col.fw_counts[nt] = fw;
            col.rv_counts[nt] = n - fw;
            for (k = 0; k < n; k++) {
                int_varray_add_value(&col.base_quals[nt], rd_i32(in));
                int_varray_add_value(&col.map_quals[nt], rd_i32(in));
            }
            col.num_bases += n;
        }
        n_ev = rd_i32(in);
        for (e = 0; e < n_ev; e++) {
            char *key = rd_str(in);
            int32_t cnt = rd_i32(in);
            while (cnt-- > 0) add_ins_sequence(&col.ins_event_counts, key, 40, -1, 60, -1, cnt & 1);
            free(key);
        }
        n_ev = rd_i32(in);
        for (e = 0; e < n_ev; e++) {
            char *key = rd_str(in);
            int32_t cnt = rd_i32(in);
            while (cnt-- > 0) add_del_sequence(&col.del_event_counts, key, 40, -1, 60, -1, cnt & 1);
            free(key);
        }
        /* which variants end up as columns of the batch (the mock answers per column): everything the callback does not
         * return early for and does not test on the host */
        is_indel_binom = !ub.use_det_lim && vcf_var_is_indel(var);
        if (has_col && col_pos == var->pos && col.coverage_plp - (vcf_var_is_indel(var) ? col.num_tails : 0) >= 1
            && !is_indel_binom) {
            g_canned[g_n_canned++] = canned;
        }
        if (has_col) {                           /* no coverage: mpileup never calls back (lofreq_uniq.c:213-215) */
            lfq_uniq_snv(&col, &ub);
        }
        col_free(&col);                          /* plp.c:1440-1445 */
    }
    ub.var = NULL;                               /* lofreq_uniq.c:727 */
    lfq_uniq_flush(&ub);
    lfq_uniq_shutdown();
    for (v = 0; v < n_vars; v++) {
        printf("%ld\t%s\n", vars[v]->pos, vars[v]->info);
        free(vars[v]->chrom); free(vars[v]->ref); free(vars[v]->alt); free(vars[v]->info); free(vars[v]->id);
        free(vars[v]->filter); free(vars[v]);
    }
    free(vars);
    free(g_canned);
    fclose(in);
    fclose(g_out);
    return 0;
}
