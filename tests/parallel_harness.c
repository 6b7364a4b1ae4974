/*
 * parallel_harness.c -- one worker of tests/test_parallel_c.py: reads its shard of sparse device records
 * (lfq_col_pvals, shard-local Bonferroni factors) and its test counts from a file, runs the merge of
 * integration/lofreq_amd_parallel.c (lfq_par_init -> lfq_par_merge_snvs -> lfq_par_gather_bytes) and, on rank 0,
 * writes the merged records + the final conf.  The environment (LFQ_PAR_*) comes from the test.
 *
 *   parallel_harness <in: int64 n_tested, int64 n_indel_tests, int64 n_pvals, n_pvals x lfq_col_pvals> <out> [gpu]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lofreq_amd_parallel.h"

int main(int argc, char **argv)
{
    lfq_par *p = NULL;
    lfq_conf conf;
    lfq_col_pvals *pv = NULL;
    lfq_snv_record *rec = NULL;
    int64_t hdr[3], n_rec = 0, n_txt = 0;
    void *txt = NULL;
    char name[64];
    FILE *f;
    int rc;

    if (argc < 3) {
        return 2;
    }
    f = fopen(argv[1], "rb");
    if (!f || fread(hdr, 8, 3, f) != 3) {
        return 3;
    }
    pv = (lfq_col_pvals *)malloc(sizeof(lfq_col_pvals) * (size_t)(hdr[2] + 1));
    if (hdr[2] > 0 && fread(pv, sizeof(lfq_col_pvals), (size_t)hdr[2], f) != (size_t)hdr[2]) {
        return 3;
    }
    fclose(f);
    rc = lfq_par_init(&p, argc > 3 ? 1 : 0);
    if (rc != LFQ_OK || !p) {
        fprintf(stderr, "lfq_par_init: %d\n", rc);
        return 4;
    }
    lfq_conf_init(&conf);
    if (getenv("LFQ_TEST_FIXED_BONF")) {        /* `lofreq call -b N`: a fixed factor, nothing to rebase */
        conf.bonf_dynamic = 0;
        conf.bonf_subst = atoll(getenv("LFQ_TEST_FIXED_BONF"));
    }
    rc = lfq_par_merge_snvs(p, &conf, pv, hdr[2], hdr[0], hdr[1], &rec, &n_rec);
    if (rc != LFQ_OK) {
        fprintf(stderr, "lfq_par_merge_snvs: %d\n", rc);
        return 5;
    }
    snprintf(name, sizeof(name), "%d\tchr%d\n", lfq_par_rank(p), lfq_par_rank(p) + 1);   /* a name table line per rank */
    rc = lfq_par_gather_bytes(p, name, (int64_t)strlen(name), &txt, &n_txt);
    if (rc != LFQ_OK) {
        fprintf(stderr, "lfq_par_gather_bytes: %d\n", rc);
        return 6;
    }
    if (lfq_par_rank(p) == 0) {
        f = fopen(argv[2], "wb");
        if (!f) {
            return 7;
        }
        fwrite(&conf, sizeof(conf), 1, f);
        fwrite(&n_rec, 8, 1, f);
        fwrite(rec, sizeof(lfq_snv_record), (size_t)n_rec, f);
        fwrite(&n_txt, 8, 1, f);
        fwrite(txt, 1, (size_t)n_txt, f);
        fclose(f);
    } else if (rec != NULL) {
        return 8;                       /* only rank 0 keeps the records */
    }
    free(rec);
    free(txt);
    free(pv);
    lfq_par_destroy(p);
    return 0;
}
