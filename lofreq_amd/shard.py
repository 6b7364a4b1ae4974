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


def exchange_counts(local_counts, dist=None, device=None):
    """One all-gather of a small int64 vector per rank ({tested SNV columns, indel tests}).
    -> (array [world, len(local_counts)], exclusive prefix of this rank as an array)."""
    v = np.asarray(local_counts, np.int64).reshape(-1)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return v.reshape(1, -1), np.zeros_like(v)
    import torch
    ws, rank = dist.get_world_size(), dist.get_rank()
    mine = torch.from_numpy(v.copy()).to(device or "cpu")
    allc = torch.zeros(ws * len(v), dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(allc, mine)
    allc = allc.cpu().numpy().reshape(ws, len(v))
    return allc, allc[:rank].sum(axis=0)


def exchange_test_counts(n_tested_local, dist=None, device=None):
    """All-gather the per-shard tested-column counts -> (counts per rank, exclusive prefix of this rank)."""
    allc, prefix = exchange_counts([n_tested_local], dist, device)
    return [int(x) for x in allc[:, 0]], int(prefix[0])


def rebase_bonferroni(pvals, prefix_tested):
    """Turn shard-local running Bonferroni factors into the single-process ones: every tested column
    of an earlier shard contributes 3 tests (lofreq_call.c:794-801)."""
    pvals = pvals.copy()
    pvals["bonf"] += 3 * int(prefix_tested)
    return pvals


def gather_records(records, col_offset, dist=None, device=None):
    """Gather reported variants to rank 0 in shard order; `col` becomes a global column index.
    Works for SNV and indel records (any structured dtype with a `col` field)."""
    rec = records.copy()
    rdtype = rec.dtype
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
    width = rdtype.itemsize
    cap = max(max(n_all), 1)
    buf = np.zeros(cap * width, np.uint8)
    buf[: len(rec) * width] = rec.view(np.uint8).reshape(-1)
    mine = torch.from_numpy(buf).to(dev)
    out = torch.zeros(ws * cap * width, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, mine)
    if rank != 0:
        return None
    out = out.cpu().numpy()
    parts = [out[r * cap * width: r * cap * width + n_all[r] * width].view(rdtype)
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


def finish_indel_shard(conf, bonf_indel_start, records, n_tests_local, col_offset, dist=None, device=None):
    """Sharded `call_indels`: `records` / `n_tests_local` are what lofreq_amd.call_indels returned for this
    shard when every rank started from the same `bonf_indel_start`.  A shard's local running factor is
    never larger than the single-process one, so its records are a superset of the true ones; after the
    test-count all-gather each record is re-tested with the exact factor (lofreq_call.c:326, :384:
    pvalue * bonf_indel < sig) and the survivors are gathered in shard order.  Returns (records on rank 0
    or None, total tests); conf ends up as after the single-process loop (:693-696)."""
    allc, prefix = exchange_counts([n_tests_local], dist, device)
    total = int(allc[:, 0].sum())
    rec = records.copy()
    if conf.bonf_dynamic:
        rec["bonf"] += int(prefix[0])
        keep = rec["pvalue"] * rec["bonf"].astype(np.longdouble) < np.float32(conf.sig)
        rec = rec[keep]
        conf.c.bonf_indel = int(bonf_indel_start) + total
    conf.c.num_indel_tests += total - int(n_tests_local)
    return gather_records(rec, col_offset, dist, device), total
