"""Region sharding of the column loop across GPUs (one process per GPU, torch.distributed).

Mirrors the reference's ``lofreq call-parallel`` model (src/scripts/lofreq2_call_pparallel.py:590-707):
contiguous genomic ranges, one per worker, no data-path exchange; what *is* exchanged is
  (1) each shard's number of tested columns -- the reference sums the per-shard
      "Number of substitution tests performed" log lines (:131-161, :685-690); here one all-gather
      of an int64 per rank, from which every rank also derives the exclusive prefix that turns its
      local running Bonferroni factor into the single-process one, and
  (2) the reported variants, gathered to rank 0 in shard order (the reference runs
      ``bcftools concat``, :164-185).
Over RCCL/xGMI on GPUs (backend "nccl"), gloo in the CPU tests.  Payloads are tens of bytes to a
few KB: latency-bound, so one collective of each kind and nothing else.
"""
import numpy as np

from . import _lib
from .caller import finalize_pvals


def shard_ranges(n_items, world_size):
    """Contiguous, near-equal ranges (lofreq2_call_pparallel.py bins, BAM header order :627-633)."""
    base, rem = divmod(int(n_items), int(world_size))
    out, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def exchange_test_counts(n_tested_local, dist=None, device=None):
    """All-gather the per-shard tested-column counts -> (counts per rank, exclusive prefix of this rank)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [int(n_tested_local)], 0
    import torch
    ws, rank = dist.get_world_size(), dist.get_rank()
    mine = torch.tensor([int(n_tested_local)], dtype=torch.int64, device=device or "cpu")
    allc = torch.zeros(ws, dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(allc, mine)
    counts = [int(x) for x in allc.cpu().tolist()]
    return counts, int(sum(counts[:rank]))


def rebase_bonferroni(pvals, prefix_tested):
    """Turn shard-local running Bonferroni factors into the single-process ones: every tested column
    of an earlier shard contributes 3 tests (lofreq_call.c:794-801)."""
    pvals = pvals.copy()
    pvals["bonf"] += 3 * int(prefix_tested)
    return pvals


def gather_records(records, col_offset, dist=None, device=None):
    """Gather reported variants to rank 0 in shard order; `col` becomes a global column index."""
    rec = records.copy()
    rec["col"] += int(col_offset)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return rec
    import torch
    ws, rank = dist.get_world_size(), dist.get_rank()
    dev = device or "cpu"
    n_mine = torch.tensor([len(rec)], dtype=torch.int64, device=dev)
    n_all = torch.zeros(ws, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(n_all, n_mine)
    n_all = [int(x) for x in n_all.cpu().tolist()]
    width = _lib.SNV_RECORD_DTYPE.itemsize
    cap = max(max(n_all), 1)
    buf = np.zeros(cap * width, np.uint8)
    buf[: len(rec) * width] = rec.view(np.uint8).reshape(-1)
    mine = torch.from_numpy(buf).to(dev)
    out = torch.zeros(ws * cap * width, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, mine)
    if rank != 0:
        return None
    out = out.cpu().numpy()
    parts = [out[r * cap * width: r * cap * width + n_all[r] * width].view(_lib.SNV_RECORD_DTYPE)
             for r in range(ws)]
    return np.concatenate(parts) if parts else rec[:0]


def finish_shard(conf, pvals, n_tested_local, ref_base, col_offset, dist=None, device=None):
    """Host + exchange half of one sharded step: exact running Bonferroni, emit test, gather.

    `pvals` are this shard's sparse device records (local Bonferroni factors, computed with the
    batch-start factor `conf.bonf_subst`, identical on every rank); returns (records on rank 0 or
    None, total tested columns).  Updates conf like the reference's single-process loop would."""
    counts, prefix = exchange_test_counts(n_tested_local, dist, device)
    if conf.bonf_dynamic:
        pvals = rebase_bonferroni(pvals, prefix)
    recs = finalize_pvals(conf, pvals, ref_base)
    allrecs = gather_records(recs, col_offset, dist, device)
    total = sum(counts)
    if total > 0:
        if conf.bonf_dynamic:
            conf.c.bonf_subst = (0 if conf.c.bonf_subst == 1 else conf.c.bonf_subst) + 3 * total
        conf.c.num_snv_tests += 3 * total
    return allrecs, total
