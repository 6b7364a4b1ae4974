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


# ---------------------------------------------------------------------------------------------------------
# region planning: call-parallel's bins (lofreq2_call_pparallel.py:590-613, 307-312), balanced by cost
# ---------------------------------------------------------------------------------------------------------
BIN_PER_THREAD = 2          # lofreq2_call_pparallel.py: "keep more bins than threads to make up for differences"
MIN_BIN_LEN = 100           # :605-607 "Regions getting too small to be efficiently processed"


def make_cost_fn(depth_profile, k_profile=None, dp_weight=2.0, bin_size=1):
    """Cost of a range = sum of depth (the streaming count phase: bytes read) + dp_weight * sum of depth * K
    (the recurrence: rows * cells) over its positions, from per-position (or per-`bin_size` window) profiles
    {chrom: array}.  `k_profile` is an estimate of the largest alt count per position (0 where nothing is
    expected); sequencing errors alone give K ~ depth / 7000, which the depth term already covers.
    -> cost_fn(chrom, begin, end), exact on window boundaries, linear inside a window."""
    import numpy as np
    pref = {}
    for chrom, d in depth_profile.items():
        d = np.asarray(d, np.float64)
        c = d.copy()
        if k_profile is not None and chrom in k_profile:
            c = c + dp_weight * d * np.asarray(k_profile[chrom], np.float64)
        pref[chrom] = np.concatenate([[0.0], np.cumsum(c)])

    def at(chrom, x):
        p = pref[chrom]
        w = x / float(bin_size)
        i = min(int(w), len(p) - 2) if len(p) > 1 else 0
        if len(p) < 2:
            return 0.0
        return p[i] + (p[i + 1] - p[i]) * min(w - i, 1.0) if w < len(p) - 1 else p[-1]

    def cost_fn(chrom, begin, end):
        return at(chrom, end) - at(chrom, begin)

    return cost_fn


def plan_regions(regions, cost_fn, world_size, bins_per_worker=BIN_PER_THREAD, balance=1.1, max_bins_per_worker=64):
    """Cut `regions` [(chrom, begin, end), ...] (BED targets or whole contigs, genome order) into bins and deal them
    to `world_size` workers.

    The reference (lofreq2_call_pparallel.py:590-613) splits the LONGEST bin in half until the biggest is shorter
    than total / (BIN_PER_THREAD * threads), then lets a process pool take bins longest first (:307-312).  Here the
    same greedy loop runs on COST instead of length (cost_fn: sum of depth + a DP term, make_cost_fn) -- equal
    lengths are the special case of uniform cost -- and, because one process per GPU owns its bins for the whole
    run instead of pulling from a pool, the dealing is the pool's schedule computed up front: bins in descending
    cost, each to the least loaded worker (LPT).  Splitting continues past the reference's 2 bins per worker until
    the heaviest worker is within `balance` of the mean (or `max_bins_per_worker` is reached).

    -> (bins, owner): bins in genome order [(chrom, begin, end)], owner[i] = worker of bin i."""
    bins = [(c, int(b), int(e)) for c, b, e in regions if int(e) > int(b)]
    if not bins:
        return [], []
    order = {}
    for c, _, _ in bins:
        order.setdefault(c, len(order))
    costs = [float(cost_fn(*b)) for b in bins]
    total = sum(costs)

    def deal(costs):
        load = [0.0] * world_size
        owner = [0] * len(costs)
        for i in sorted(range(len(costs)), key=lambda i: -costs[i]):
            w = min(range(world_size), key=lambda r: load[r])
            owner[i] = w
            load[w] += costs[i]
        return owner, load

    target = bins_per_worker
    while True:
        # the reference's loop: split the most expensive bin until it is below total / (target * workers)
        while True:
            i = max(range(len(bins)), key=lambda i: costs[i])
            c, b, e = bins[i]
            if costs[i] < total / (target * world_size) or e - b < 2 * MIN_BIN_LEN:
                break
            mid = (b + e) // 2
            bins[i:i + 1] = [(c, b, mid), (c, mid, e)]
            costs[i:i + 1] = [float(cost_fn(c, b, mid)), float(cost_fn(c, mid, e))]
        owner, load = deal(costs)
        mean = total / world_size if total > 0 else 0.0
        if mean <= 0 or max(load) <= balance * mean or target >= max_bins_per_worker:
            break
        if all(e - b < 2 * MIN_BIN_LEN for _, b, e in bins):
            break
        target *= 2
    idx = sorted(range(len(bins)), key=lambda i: (order[bins[i][0]], bins[i][1]))
    return [bins[i] for i in idx], [owner[i] for i in idx]


# tests: run the collectives even in a world of one rank (the RCCL calls of the N > 1 path on a one-GPU box)
_FORCE_COLLECTIVES = __import__("os").environ.get("LFQ_SHARD_FORCE_COLLECTIVES") == "1"


def exchange_counts(local_counts, dist=None, device=None):
    """One all-gather of a small int64 vector per rank ({tested SNV columns, indel tests}).
    -> (array [world, len(local_counts)], exclusive prefix of this rank as an array)."""
    v = np.asarray(local_counts, np.int64).reshape(-1)
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE_COLLECTIVES):
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
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE_COLLECTIVES):
        return rec
    import torch
    ws, rank = dist.get_world_size(), dist.get_rank()
    dev = device or "cpu"
    # the record counts (every rank needs the largest one: a gather moves equal-sized pieces) ...
    n_mine = torch.tensor([len(rec)], dtype=torch.int64, device=dev)
    n_all = torch.zeros(ws, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(n_all, n_mine)
    n_all = [int(x) for x in n_all.cpu().tolist()]
    width = rdtype.itemsize
    cap = max(max(n_all), 1)
    buf = np.zeros(cap * width, np.uint8)
    buf[: len(rec) * width] = rec.view(np.uint8).reshape(-1)
    mine = torch.from_numpy(buf).to(dev)
    # ... then ONE gather of the fixed-size records to rank 0 (north_star: "a single RCCL gather for the final VCF merge")
    pieces = [torch.zeros(cap * width, dtype=torch.uint8, device=dev) for _ in range(ws)] if rank == 0 else None
    dist.gather(mine, pieces, dst=0)
    if rank != 0:
        return None
    parts = [pieces[r].cpu().numpy()[: n_all[r] * width].view(rdtype) for r in range(ws)]
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


def finish_indel_bins(conf, bonf_indel_start, my_bins, n_bins_total, dist=None, device=None):
    """finish_indel_shard for call-parallel style bins: this rank ran `call_indels` on `my_bins` = [(bin_index,
    col_offset, indel records, n_tests), ...], every bin from the same `bonf_indel_start` (a bin's local running factor
    is then never larger than the single-process one: its records are a superset).  One all-gather of the per-bin test
    counts gives every bin its exact prefix (the tests of the bins before it in genome order, whoever ran them); every
    record is re-tested with the exact factor (lofreq_call.c:326, :384), the survivors are gathered and put into genome
    order on rank 0.  conf ends up as after the single-process loop (:693-696).  -> (records on rank 0 or None, total tests)"""
    counts = np.zeros(int(n_bins_total), np.int64)
    for b, _, _, n_tests in my_bins:
        counts[b] = int(n_tests)
    allc, _ = exchange_counts(counts, dist, device)
    per_bin = allc.sum(axis=0)
    prefix = np.concatenate([[0], np.cumsum(per_bin)[:-1]])
    parts = []
    for b, col_offset, records, _ in my_bins:
        rec = records.copy()
        if conf.bonf_dynamic:
            rec["bonf"] += int(prefix[b])
            rec = rec[rec["pvalue"] * rec["bonf"].astype(np.longdouble) < np.float32(conf.sig)]
        rec["col"] += int(col_offset)
        parts.append(rec)
    mine = np.concatenate(parts) if parts else np.zeros(0, _lib.INDEL_RECORD_DTYPE)
    allrecs = gather_records(mine, 0, dist, device)
    if allrecs is not None and len(allrecs):
        allrecs = allrecs[np.argsort(allrecs["col"], kind="stable")]
    total = int(per_bin.sum())
    if conf.bonf_dynamic:
        conf.c.bonf_indel = int(bonf_indel_start) + total
    conf.c.num_indel_tests += total
    return allrecs, total


def finish_bins(conf, my_bins, n_bins_total, dist=None, device=None):
    """Sharded step over call-parallel style bins (plan_regions): this rank ran the kernels of `my_bins` =
    [(bin_index, col_offset, sparse pvals, n_tested), ...], every bin as its own batch starting from the same
    conf.bonf_subst.  One all-gather of the per-bin tested-column counts gives every bin its exact running
    Bonferroni prefix (the bins before it in genome order, whoever ran them); records are finalised per bin,
    gathered, and put into genome order on rank 0.  conf ends up as after the single-process loop."""
    counts = np.zeros(int(n_bins_total), np.int64)
    for b, _, _, n_tested in my_bins:
        counts[b] = int(n_tested)
    allc, _ = exchange_counts(counts, dist, device)
    per_bin = allc.sum(axis=0)                       # every bin is owned by exactly one rank
    prefix = np.concatenate([[0], np.cumsum(per_bin)[:-1]])
    parts = []
    for b, col_offset, pvals, _ in my_bins:
        pv = rebase_bonferroni(pvals, int(prefix[b])) if conf.bonf_dynamic else pvals
        r = finalize_pvals(conf, pv, None)
        r["col"] += int(col_offset)
        parts.append(r)
    mine = np.concatenate(parts) if parts else np.zeros(0, _lib.SNV_RECORD_DTYPE)
    allrecs = gather_records(mine, 0, dist, device)
    if allrecs is not None and len(allrecs):
        allrecs = allrecs[np.argsort(allrecs["col"], kind="stable")]
    total = int(per_bin.sum())
    if total > 0:
        if conf.bonf_dynamic:
            conf.c.bonf_subst = (0 if conf.c.bonf_subst == 1 else conf.c.bonf_subst) + 3 * total
        conf.c.num_snv_tests += 3 * total
    return allrecs, total
