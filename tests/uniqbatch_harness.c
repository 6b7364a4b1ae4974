/*
 * uniqbatch_harness.c -- drives integration/lofreq_amd_uniqbatch.c (the packing core of the `lofreq uniq` binding) against
 * the REAL liblofreq_amd.so on a GPU (tests/test_gpu_uniq_binding.py).  Needs include/lofreq_amd.h only.  Reads the
 * variant stream of tests/uniq_harness.c (same format, so both tests are fed by the same Python code), plays the part of
 * integration/lofreq_amd_uniq.c that tests/test_uniq_binding.py checks against the reference's headers -- the gates of
 * uniq_snv (lofreq_uniq.c:233-277) -- with plain arrays instead of plp_col_t, and prints one line per variant:
 *   <pos0> <value>     value = 1 / 0 (det-lim: UNIQ or not), the UQ number, or "-" where uniq_snv writes no tag
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lofreq_amd_uniqbatch.h"

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

typedef struct res { int set, value; } res;

static void on_result(void *user, int value)
{
    res *r = (res *)user;
    r->set = 1;
    r->value = value;
}

int main(int argc, char **argv)
{
    FILE *in;
    lfq_uniqbatch *ub = NULL;
    int32_t use_det_lim, bits, n_vars, v;
    float uni_freq;
    res *out;
    int32_t *pos;
    int rc;
    if (argc != 2) {
        fprintf(stderr, "usage: %s variants.bin\n", argv[0]);
        return 2;
    }
    in = fopen(argv[1], "rb");
    if (!in) return 2;
    use_det_lim = rd_i32(in);
    bits = rd_i32(in);
    memcpy(&uni_freq, &bits, 4);
    n_vars = rd_i32(in);
    out = (res *)calloc((size_t)n_vars + 1, sizeof(res));
    pos = (int32_t *)calloc((size_t)n_vars + 1, sizeof(int32_t));
    rc = lfq_uniqbatch_open(&ub, use_det_lim);
    if (rc != LFQ_OK) {
        fprintf(stderr, "lfq_uniqbatch_open: %s\n", lfq_strerror(rc));
        return 1;
    }
    for (v = 0; v < n_vars; v++) {
        char *chrom = rd_str(in), *ref = rd_str(in), *alt = rd_str(in), *info = rd_str(in);
        const int32_t pos0 = rd_i32(in), has_col = rd_i32(in), col_pos = rd_i32(in), ref_base = rd_i32(in);
        const int32_t coverage_plp = rd_i32(in), num_tails = rd_i32(in);
        int32_t nt, e, n_ev, coverage;
        int *q[5][2];
        lfq_uniq_col c;
        const int is_indel = strlen(ref) > 1 || strlen(alt) > 1;
        int alt_count = 0;
        float af;
        (void)rd_i32(in);                           /* the mock library's canned answer: not used here */
        pos[v] = pos0;
        memset(&c, 0, sizeof(c));
        for (nt = 0; nt < 5; nt++) {
            const int32_t n = rd_i32(in), fw = rd_i32(in);
            int32_t k;
            q[nt][0] = (int *)malloc(sizeof(int) * (size_t)(n + 1));
            q[nt][1] = (int *)malloc(sizeof(int) * (size_t)(n + 1));
            for (k = 0; k < n; k++) {
                q[nt][0][k] = rd_i32(in);
                q[nt][1][k] = rd_i32(in);
            }
            c.nt[nt].bq = q[nt][0];
            c.nt[nt].mq = q[nt][1];
            c.nt[nt].n = (size_t)n;
            c.nt[nt].fw = fw;
        }
        for (e = 0, n_ev = rd_i32(in); e < n_ev; e++) {     /* insertions: the count of the variant's own event (:361-367) */
            char *key = rd_str(in);
            const int32_t cnt = rd_i32(in);
            if (strlen(alt) > strlen(ref) && 0 == strcmp(key, alt + 1)) alt_count = cnt;
            free(key);
        }
        for (e = 0, n_ev = rd_i32(in); e < n_ev; e++) {     /* deletions (:348-358) */
            char *key = rd_str(in);
            const int32_t cnt = rd_i32(in);
            if (strlen(ref) > strlen(alt) && 0 == strcmp(key, ref + 1)) alt_count = cnt;
            free(key);
        }
        /* the gates of uniq_snv as integration/lofreq_amd_uniq.c applies them */
        coverage = coverage_plp - (is_indel ? num_tails : 0);
        if (has_col && col_pos == pos0 && coverage >= 1) {
            if (uni_freq <= 0.0) {
                const char *a = strstr(info, "AF=");
                af = a ? strtof(a + 3, NULL) : 0.f;
                if (af < 0.0 || af > 1.0) af = af < 0.0 ? 0.01f : 1.0f;
            } else {
                af = uni_freq;
            }
            c.ref_base = (char)ref_base;
            c.coverage = coverage;
            if (!use_det_lim && is_indel) rc = lfq_uniqbatch_add_count(ub, coverage, alt_count, af, &out[v]);
            else rc = lfq_uniqbatch_add_column(ub, &c, af, alt[0], &out[v]);
            if (rc != LFQ_OK) {
                fprintf(stderr, "lfq_uniqbatch_add: %s\n", lfq_strerror(rc));
                return 1;
            }
        }
        for (nt = 0; nt < 5; nt++) {                /* the column is gone after the callback (plp.c:1440-1445) */
            memset(q[nt][0], 0xA5, sizeof(int) * c.nt[nt].n);
            memset(q[nt][1], 0xA5, sizeof(int) * c.nt[nt].n);
            free(q[nt][0]);
            free(q[nt][1]);
        }
        free(chrom); free(ref); free(alt); free(info);
    }
    rc = lfq_uniqbatch_flush(ub, on_result);
    if (rc != LFQ_OK) {
        fprintf(stderr, "lfq_uniqbatch_flush: %s\n", lfq_strerror(rc));
        return 1;
    }
    lfq_uniqbatch_close(ub);
    for (v = 0; v < n_vars; v++) {
        if (out[v].set && (use_det_lim || out[v].value >= 0)) printf("%d %d\n", pos[v], out[v].value);
        else printf("%d -\n", pos[v]);
    }
    free(out);
    free(pos);
    fclose(in);
    return 0;
}
