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

What a collective costs a rank whose GPU is running count kernels back to back (measured with a one-rank
communicator, profiles/NOTES.md round 5): every device-side piece of it -- the staging copies of a
host-resident count, the RCCL kernel -- waits for a wave slot like any other kernel, ~1 ms per step for
three blocking collectives.  Hence (i) `set_host_group`: the test counts, which are host integers on both
ends, travel over a host-side group (gloo) when the caller provides one, and (ii) the record gather can be
split into `gather_records_start` / `gather_records_wait` (fixed-capacity pieces, so that no second
all-gather of the piece sizes is needed; RCCL, asynchronous): a caller that pipelines steps collects step
k's records while step k + 1 runs and never blocks on the device in between.
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


# A host-side process group (gloo) for the counts, which are host integers on every rank: see the module docstring.
_HOST_GROUP = None


def set_host_group(group):
    """`group` = a torch.distributed group whose backend takes CPU tensors (gloo), spanning the same ranks as the
    default group, or None: exchange_counts then goes through it instead of staging 8 bytes per rank through the GPU."""
    global _HOST_GROUP
    _HOST_GROUP = group


def exchange_counts(local_counts, dist=None, device=None):
    """One all-gather of a small int64 vector per rank ({tested SNV columns, indel tests}).
    -> (array [world, len(local_counts)], exclusive prefix of this rank as an array)."""
    v = np.asarray(local_counts, np.int64).reshape(-1)
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE_COLLECTIVES):
        return v.reshape(1, -1), np.zeros_like(v)
    import torch
    ws, rank = dist.get_world_size(), dist.get_rank()
    if _HOST_GROUP is not None:
        mine = torch.from_numpy(v.copy())
        allc = torch.zeros(ws * len(v), dtype=torch.int64)
        dist.all_gather_into_tensor(allc, mine, group=_HOST_GROUP)
        allc = allc.numpy().reshape(ws, len(v))
        return allc, allc[:rank].sum(axis=0)
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


class PendingGather:
    """A record gather that has been started (gather_records_start) and not yet collected."""
    __slots__ = ("records", "work", "big", "mine", "stage", "host", "ev", "cap", "rdtype", "ws", "rank")


_GATHER_HDR = 16        # bytes in front of a piece's records: int64 record count, int64 reserved (keeps the records aligned)
_PINNED = {}            # size -> free pinned staging buffers (hipHostMalloc per step would cost more than the exchange)


def _pinned_get(nbytes):
    import torch
    free = _PINNED.setdefault(int(nbytes), [])
    return free.pop() if free else torch.empty(int(nbytes), dtype=torch.uint8).pin_memory()


def _pinned_put(t):
    if t is not None:
        _PINNED.setdefault(int(t.numel()), []).append(t)


_XSTREAM = {}           # device -> the stream the gather's copies and its communicator wait are queued on


def _exchange_stream(dev):
    """A high-priority stream of its own for the record gather's device-side pieces: the caller's current stream carries
    the blocking copies of its next step's results, which must not queue up behind them."""
    import torch
    key = (dev.type, dev.index)
    if key not in _XSTREAM:
        _XSTREAM[key] = torch.cuda.Stream(device=dev, priority=-1)
    return _XSTREAM[key]


def gather_records_start(records, col_offset, cap, dist=None, device=None):
    """First half of gather_records for callers that pipeline steps: every rank sends ONE piece of fixed capacity
    (`cap` records, the same number on every rank -- e.g. 3 x the largest candidate-column count of the step, which the
    count all-gather already made known to everybody; the piece starts with its own record count), so that no second
    all-gather of the sizes is needed, and nothing here waits for the device: on a GPU the piece goes up from a pinned
    buffer by an asynchronous copy, the collective is queued on the communicator's stream (RCCL), and rank 0's copy of
    the gathered pieces back into pinned memory is queued behind it with an event that gather_records_wait waits for.
    -> PendingGather.  `cap` too small for this rank's records raises before anything is sent."""
    rec = records.copy()
    rec["col"] += int(col_offset)
    h = PendingGather()
    h.records, h.work, h.big, h.mine, h.stage, h.host, h.ev = rec, None, None, None, None, None, None
    h.rdtype = rec.dtype
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE_COLLECTIVES):
        return h
    import torch
    h.ws, h.rank, h.cap = dist.get_world_size(), dist.get_rank(), max(int(cap), 1)
    if len(rec) > h.cap:
        raise ValueError("gather_records_start: %d records, capacity %d" % (len(rec), h.cap))
    dev = torch.device(device or "cpu")
    width = h.rdtype.itemsize
    piece = _GATHER_HDR + h.cap * width
    on_gpu = dev.type != "cpu"
    if on_gpu:
        h.stage = _pinned_get(piece)
        buf = h.stage.numpy()
        buf[8:_GATHER_HDR] = 0
    else:
        buf = np.zeros(piece, np.uint8)
    buf[:8] = np.array([len(rec)], np.int64).view(np.uint8)
    buf[_GATHER_HDR: _GATHER_HDR + len(rec) * width] = rec.view(np.uint8).reshape(-1)
    if not on_gpu:
        h.mine = torch.from_numpy(buf)
        pieces = None
        if h.rank == 0:
            h.big = torch.empty(h.ws * piece, dtype=torch.uint8)
            pieces = list(h.big.view(h.ws, piece).unbind(0))
        h.work = dist.gather(h.mine, pieces, dst=0, async_op=True)
        h.records = None
        return h
    with torch.cuda.stream(_exchange_stream(dev)):
        h.mine = h.stage.to(dev, non_blocking=True)
        pieces = None
        if h.rank == 0:
            h.big = torch.empty(h.ws * piece, dtype=torch.uint8, device=dev)    # one buffer, one copy back to the host
            pieces = list(h.big.view(h.ws, piece).unbind(0))
        h.work = dist.gather(h.mine, pieces, dst=0, async_op=True)
        h.work.wait()                   # (RCCL: the exchange STREAM waits for the collective, not the host)
        if h.rank == 0:
            h.host = _pinned_get(h.ws * piece)
            h.host.copy_(h.big, non_blocking=True)
        h.ev = torch.cuda.Event()
        h.ev.record()
    h.records = None
    return h


def gather_records_wait(h):
    """Second half: the records of all ranks in shard order on rank 0, None elsewhere (every rank waits for its own side
    of the collective, so that the buffers it handed over are free again)."""
    if h.work is None:
        return h.records
    if h.ev is not None:
        h.ev.synchronize()
    else:
        h.work.wait()
    _pinned_put(h.stage)
    h.stage = h.mine = None
    if h.rank != 0:
        return None
    width = h.rdtype.itemsize
    piece = _GATHER_HDR + h.cap * width
    host = (h.host.numpy() if h.host is not None else h.big.numpy()).reshape(h.ws, piece)
    parts = []
    for r in range(h.ws):
        n = int(host[r, :8].view(np.int64)[0])
        parts.append(host[r, _GATHER_HDR: _GATHER_HDR + n * width].view(h.rdtype))
    out = np.concatenate(parts)         # (a copy: the pinned buffer goes back to the pool)
    _pinned_put(h.host)
    h.big = h.host = None
    return out


def finish_shard_start(conf, pvals, n_tested_local, ref_base, col_offset, dist=None, device=None):
    """Host + exchange half of one sharded step up to the point where the records are on their way: ONE all-gather of
    {tested columns, candidate columns} per rank (exact running Bonferroni prefix; the largest candidate count bounds
    every rank's records: at most three per column), the exact emit test on this shard's records, and the gather to
    rank 0 STARTED.  -> (PendingGather for finish_shard_wait, total tested columns).  Updates conf like the reference's
    single-process loop would."""
    allc, prefix = exchange_counts([int(n_tested_local), len(pvals)], dist, device)
    if conf.bonf_dynamic:
        pvals = rebase_bonferroni(pvals, int(prefix[0]))
    recs = finalize_pvals(conf, pvals, ref_base)
    h = gather_records_start(recs, col_offset, 3 * int(allc[:, 1].max()), dist, device)
    total = int(allc[:, 0].sum())
    if total > 0:
        if conf.bonf_dynamic:
            conf.c.bonf_subst = (0 if conf.c.bonf_subst == 1 else conf.c.bonf_subst) + 3 * total
        conf.c.num_snv_tests += 3 * total
    return h, total


def finish_shard_wait(h):
    """-> the step's records on rank 0 (shard order), None elsewhere."""
    return gather_records_wait(h)


def finish_shard(conf, pvals, n_tested_local, ref_base, col_offset, dist=None, device=None):
    """Host + exchange half of one sharded step: exact running Bonferroni, emit test, gather.

    `pvals` are this shard's sparse device records (local Bonferroni factors, computed with the
    batch-start factor `conf.bonf_subst`, identical on every rank); returns (records on rank 0 or
    None, total tested columns).  Updates conf like the reference's single-process loop would.
    (= finish_shard_start + finish_shard_wait: one all-gather and one gather.)"""
    h, total = finish_shard_start(conf, pvals, n_tested_local, ref_base, col_offset, dist, device)
    return finish_shard_wait(h), total


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
