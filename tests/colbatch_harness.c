/*
 * colbatch_harness.c -- TEST INFRASTRUCTURE (tests/test_gpu_shim.py): pileup columns -> integration/lofreq_amd_colbatch.c
 * -> the REAL liblofreq_amd.so -> VCF lines, in one process, on a GPU.  Needs no LoFreq header: the columns come from the
 * same binary stream tests/shim_harness.c (the mock-mpileup test of the plp_col_t side, no GPU) reads, and every column is
 * built in heap arrays that are poisoned and freed right after lfq_colbatch_add returns, the way mpileup frees a plp_col_t
 * after the callback (plp.c:1440-1445).
 *
 *   colbatch_harness columns.bin [min_bq=N] [min_alt_bq=N] [sig=F] [min_cov=N] > lines
 *   stream: i32 bonf_dynamic, bonf_subst, no_indels, only_indels, flag, ncols; then per column what shim_harness.c documents.
 *   stdout: the VCF lines, then "#conf bonf_subst num_snv_tests bonf_indel num_indel_tests wo_idaq".
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lofreq_amd_colbatch.h"

static int32_t rd_i32(FILE *f)
{
    int32_t v = 0;
    if (fread(&v, 4, 1, f) != 1) {
        fprintf(stderr, "harness: short input\n");
        exit(2);
    }
    return v;
}

static void emit(void *user, const char *line)
{
    (void)user;
    fputs(line, stdout);
}

static int *ints(size_t n)
{
    int *p = (int *)malloc((n ? n : 1) * sizeof(int));
    if (!p) exit(3);
    return p;
}

static void poison_free(int *p, size_t n)
{
    if (p) memset(p, 0xA5, n * sizeof(int));
    free(p);
}

int main(int argc, char **argv)
{
    FILE *in;
    lfq_colbatch *cb = NULL;
    lfq_conf conf;
    int32_t ncols, c, no_indels, only_indels;
    int a, rc;
    if (argc < 2) {
        fprintf(stderr, "usage: %s columns.bin [key=value ...]\n", argv[0]);
        return 2;
    }
    in = fopen(argv[1], "rb");
    if (!in) return 2;
    lfq_conf_init(&conf);
    conf.bonf_dynamic = rd_i32(in);
    conf.bonf_subst = rd_i32(in);
    no_indels = rd_i32(in);
    only_indels = rd_i32(in);
    conf.flag = rd_i32(in);
    for (a = 2; a < argc; a++) {
        if (!strncmp(argv[a], "min_bq=", 7)) conf.min_bq = atoi(argv[a] + 7);
        else if (!strncmp(argv[a], "min_alt_bq=", 11)) conf.min_alt_bq = atoi(argv[a] + 11);
        else if (!strncmp(argv[a], "min_cov=", 8)) conf.min_cov = atoi(argv[a] + 8);
        else if (!strncmp(argv[a], "sig=", 4)) conf.sig = strtof(argv[a] + 4, NULL);
        else return 2;
    }
    rc = lfq_colbatch_open(&cb, emit, NULL, 0);
    if (rc != LFQ_OK) {
        fprintf(stderr, "lfq_colbatch_open: %s\n", lfq_strerror(rc));
        return 4;
    }
    ncols = rd_i32(in);
    for (c = 0; c < ncols; c++) {
        lfq_col_view v;
        lfq_col_event *ev[2] = {NULL, NULL};
        int *ne[2][2] = {{NULL, NULL}, {NULL, NULL}};
        int32_t n_ne[2] = {0, 0}, n_ev[2] = {0, 0};
        int nt, s;
        int32_t e;
        char cons;
        memset(&v, 0, sizeof(v));
        v.target = "chr1";
        v.pos = rd_i32(in);
        v.ref_base = (char)rd_i32(in);
        cons = (char)rd_i32(in);
        v.coverage_plp = rd_i32(in);
        v.num_bases = rd_i32(in);
        v.num_tails = rd_i32(in);
        v.num_non_indels = rd_i32(in);
        v.num_ins = rd_i32(in);
        v.num_dels = rd_i32(in);
        v.hrun = rd_i32(in);
        v.has_indel_aqs = rd_i32(in);
        for (nt = 0; nt < 5; nt++) {
            const int32_t n = rd_i32(in), fw = rd_i32(in), has_baq = rd_i32(in), has_sq = rd_i32(in);
            int32_t j;
            int *bq = ints((size_t)n), *baq = ints((size_t)n), *mq = ints((size_t)n), *sq = ints((size_t)n);
            for (j = 0; j < n; j++) {
                bq[j] = rd_i32(in); baq[j] = rd_i32(in); mq[j] = rd_i32(in); sq[j] = rd_i32(in);
            }
            v.nt[nt].bq = bq; v.nt[nt].baq = baq; v.nt[nt].mq = mq; v.nt[nt].sq = sq;
            v.nt[nt].n = (size_t)n; v.nt[nt].n_baq = has_baq ? (size_t)n : 0; v.nt[nt].n_sq = has_sq ? (size_t)n : 0;
            v.nt[nt].fw = fw;
        }
        for (s = 0; s < 2; s++) {
            const int32_t non_fw = rd_i32(in), non_rv = rd_i32(in);
            int32_t j;
            n_ne[s] = rd_i32(in);
            if (s == 0) { v.non_ins_fw_rv[0] = non_fw; v.non_ins_fw_rv[1] = non_rv; }
            else        { v.non_del_fw_rv[0] = non_fw; v.non_del_fw_rv[1] = non_rv; }
            ne[s][0] = ints((size_t)n_ne[s]);
            ne[s][1] = ints((size_t)n_ne[s]);
            for (j = 0; j < n_ne[s]; j++) {
                ne[s][0][j] = rd_i32(in);
                ne[s][1][j] = rd_i32(in);
            }
            n_ev[s] = rd_i32(in);
            ev[s] = (lfq_col_event *)calloc((size_t)n_ev[s] + 1, sizeof(lfq_col_event));
            if (!ev[s]) return 3;
            for (e = 0; e < n_ev[s]; e++) {
                lfq_col_event *x = &ev[s][e];
                const int32_t kl = rd_i32(in);
                int32_t fw, n;
                char *key = (char *)malloc((size_t)kl + 1);
                int *q, *aq, *mq, *sq;
                if (!key || kl <= 0 || fread(key, 1, (size_t)kl, in) != (size_t)kl) return 2;
                key[kl] = 0;
                fw = rd_i32(in);
                n = rd_i32(in);
                q = ints((size_t)n); aq = ints((size_t)n); mq = ints((size_t)n); sq = ints((size_t)n);
                for (j = 0; j < n; j++) {
                    q[j] = rd_i32(in); aq[j] = rd_i32(in); mq[j] = rd_i32(in); sq[j] = rd_i32(in);
                }
                x->key = key; x->fw = fw; x->rv = n - fw;
                x->q = q; x->aq = aq; x->mq = mq; x->sq = sq;
                x->n = (size_t)n; x->n_aq = (size_t)n; x->n_sq = (size_t)n;
            }
        }
        v.ins_quals = ne[0][0]; v.ins_map_quals = ne[0][1]; v.n_ins_quals = (size_t)n_ne[0];
        v.del_quals = ne[1][0]; v.del_map_quals = ne[1][1]; v.n_del_quals = (size_t)n_ne[1];
        v.ins_events = ev[0]; v.n_ins_events = n_ev[0];
        v.del_events = ev[1]; v.n_del_events = n_ev[1];
        /* the gates of call_vars (lofreq_call.c:892-929), as integration/lofreq_amd_shim.c applies them */
        v.take_indels = !no_indels;
        v.take_snvs = !only_indels && !(cons == '+' || cons == '-');
        if (v.ref_base != 'N') {
            rc = lfq_colbatch_add(cb, &conf, &v);
            if (rc != LFQ_OK) {
                fprintf(stderr, "lfq_colbatch_add: %s\n", lfq_strerror(rc));
                return 5;
            }
        }
        for (nt = 0; nt < 5; nt++) {            /* nothing of the column may be used after the call */
            poison_free((int *)v.nt[nt].bq, v.nt[nt].n); poison_free((int *)v.nt[nt].baq, v.nt[nt].n);
            poison_free((int *)v.nt[nt].mq, v.nt[nt].n); poison_free((int *)v.nt[nt].sq, v.nt[nt].n);
        }
        for (s = 0; s < 2; s++) {
            poison_free(ne[s][0], (size_t)n_ne[s]);
            poison_free(ne[s][1], (size_t)n_ne[s]);
            for (e = 0; e < n_ev[s]; e++) {
                lfq_col_event *x = &ev[s][e];
                memset((char *)x->key, '#', strlen(x->key));
                free((char *)x->key);
                poison_free((int *)x->q, x->n); poison_free((int *)x->aq, x->n);
                poison_free((int *)x->mq, x->n); poison_free((int *)x->sq, x->n);
            }
            free(ev[s]);
        }
    }
    rc = lfq_colbatch_flush(cb, &conf);
    if (rc != LFQ_OK) {
        fprintf(stderr, "lfq_colbatch_flush: %s\n", lfq_strerror(rc));
        return 6;
    }
    printf("#conf %lld %lld %lld %lld %ld\n", (long long)conf.bonf_subst, (long long)conf.num_snv_tests,
           (long long)conf.bonf_indel, (long long)conf.num_indel_tests, lfq_colbatch_indel_calls_wo_idaq(cb));
    lfq_colbatch_close(cb);
    fclose(in);
    return 0;
}
