/*
 * lofreq_amd_parallel.h -- N `lofreq call` region workers, one per GPU of a node, that merge in memory.
 *
 * What the reference does with files: `lofreq call-parallel` forks one `lofreq call -r <bin>` per worker, then
 * parses the workers' logs for their test counts, sums them, concatenates the per-bin VCFs and runs `lofreq filter`
 * with the summed Bonferroni factor (src/scripts/lofreq2_call_pparallel.py:131-185, 590-707).  Here the workers are
 * ranks of one RCCL communicator: the test counts are one all-gather, the reported variants one gather to rank 0,
 * and every worker applies the EXACT running Bonferroni factor of the single-process loop (lofreq_call.c:794-801),
 * so that the merged output is the single-process output whatever the cut.
 *
 * Launch contract (environment of every worker; the launcher can be the unchanged call-parallel pool, a shell loop, ...):
 *     LFQ_PAR_WORLD       number of workers (ranks)
 *     LFQ_PAR_RANK        0 .. world-1, in genome order of the workers' regions
 *     LFQ_PAR_RENDEZVOUS  a path all workers can write next to (the ncclUniqueId of rank 0 travels through
 *                         <path>.id; nothing else does with the RCCL transport)
 *     LFQ_PAR_TRANSPORT   "rccl" (default) | "shm": the library's shared-memory all-gather (lfq_shard_shm_open; the workers of ONE
 *                         node, microseconds per collective, no RCCL needed) | "files": all-gathers through files next to the
 *                         rendezvous path -- for hosts without RCCL whose workers do not share a /dev/shm, and for the CPU tests
 *                         (installed as lfq_shard_set_host_allgather)
 *     LFQ_PAR_TIMEOUT_S   how long a worker waits for its peers (default 600)
 * Device of a worker: lfq_pick_device() (LFQ_DEVICE > LOCAL_RANK > a free slot of the node), except that with the rccl
 * transport rank r takes GPU r mod device-count unless LFQ_DEVICE says otherwise: RCCL wants one GPU per rank.
 */
#ifndef LOFREQ_AMD_PARALLEL_H
#define LOFREQ_AMD_PARALLEL_H

#include "lofreq_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lfq_par lfq_par;

/* 0, or a negative lfq_status; *out = NULL when LFQ_PAR_WORLD is unset or 1 (a plain single-process run).
 * need_gpu = 0 skips lfq_create (the exchange alone: tests, mergers that only hold records). */
int lfq_par_init(lfq_par **out, int need_gpu);
void lfq_par_destroy(lfq_par *p);
int lfq_par_world(const lfq_par *p);
int lfq_par_rank(const lfq_par *p);
lfq_ctx *lfq_par_ctx(lfq_par *p);              /* the worker's context on its GPU (NULL with need_gpu = 0) */

/* The merge of the SNV path.  Every worker hands in the sparse records of ITS columns as lfq_call_snvs_collect_pvals
 * returned them (shard-local running factors; `col` rewritten by the caller to a key that is unique and ascending
 * over the whole job, e.g. tid << 32 | pos), its number of tested columns and of indel tests, and the conf all workers
 * STARTED from.  On return, on every rank: conf is what the single-process loop over all regions leaves
 * (bonf_subst, num_snv_tests, bonf_indel, num_indel_tests: lofreq_call.c:794-801, 693-696); on rank 0:
 * *records_out (malloc'ed, caller frees) holds every worker's reported variants in rank order, `col` = the key.
 * Other ranks get *n_records_out = the total and *records_out = NULL. */
int lfq_par_merge_snvs(lfq_par *p, lfq_conf *conf, lfq_col_pvals *pvals, int64_t n_pvals, int64_t n_tested_columns,
                       int64_t n_indel_tests, lfq_snv_record **records_out, int64_t *n_records_out);

/* variable-size byte gather to rank 0 in rank order (chromosome name tables, formatted indel lines):
 * *out (malloc'ed, rank 0 only) = the concatenation, *n_out = its size */
int lfq_par_gather_bytes(lfq_par *p, const void *mine, int64_t n, void **out, int64_t *n_out);

#ifdef __cplusplus
}
#endif
#endif
