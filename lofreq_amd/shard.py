"""Region sharding of the column loop across GPUs (one process per GPU, torch.distributed).

Mirrors the reference's ``lofreq call-parallel`` model (src/scripts/lofreq2_call_pparallel.py:590-707):
contiguous genomic ranges, one per worker, no data-path exchange; what *is* exchanged is
  (1) each shard's number of tested columns -- the reference sums the per-shard
      "Number of substitution tests performed" log lines (:131-161, :685-690); here one all-gather
      of an int64 per rank, from which every rank also derives the exclusive prefix that turns its
      local running Bonferroni factor into the single-process one, and
  (2) the reported variants, gathered to rank 0 in shard order (the reference runs
      ``bcftools concat``, :164-185).
Over RCCL/xGMI on GPUs, gloo in the CPU tests -- both through the C ABI's lfq_shard_* functions (round 6: one implementation
of the exchange, shared with the C workers of integration/lofreq_amd_parallel.c; see "the transport" below).  Payloads are
tens of bytes to a few KB: latency-bound, so one collective of each kind and nothing else.

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

# ---- the transport: ONE implementation of the exchange, the C ABI's (lfq_shard_* in lofreq_amd/csrc/lfq_shard.hip) ------------
# What a C worker (integration/lofreq_amd_parallel.c) and this module run is the same code: lfq_shard_exchange_counts for the
# test counts, lfq_shard_gather_start / _wait for the records.  torch.distributed only (i) carries rank 0's ncclUniqueId to the
# other ranks, from which every rank makes the RCCL communicator the ABI takes (ncclCommInitRank through ctypes, the same
# librccl the process already holds), and (ii) as a gloo group serves as the host transport (lfq_shard_set_host_allgather).
_HOST_GROUP = None      # a gloo group for the counts (host integers on both ends): see the module docstring
_T = {"dist": None, "cb": None, "cb_group": None, "comm": None, "rccl": None, "ctx": None, "own_ctx": None}


def set_host_group(group):
    """`group` = a torch.distributed group whose backend takes CPU tensors (gloo), spanning the same ranks as the
    default group, or None: the test counts then travel through it (lfq_shard_set_host_allgather) instead of being staged
    through the GPU."""
    global _HOST_GROUP
    _HOST_GROUP = group
    _T["cb_group"] = None           # (re-registered at the next exchange)
    if group is None:
        if _T["cb"] is not None:
            _lib.load().lfq_shard_set_host_allgather(_lib.HOST_ALLGATHER_FN(), None)
            _T["cb"] = None
        if _T.get("shm"):
            _lib.load().lfq_shard_shm_close()
            _T["shm"] = False


def set_context(caller):
    """The SnvCaller whose context the RCCL path stages through (its device, its stream); without one the first exchange on
    a GPU makes a context of its own."""
    _T["ctx"] = caller


def shutdown():
    """Before dist.destroy_process_group(): the communicator and the host transport go."""
    L = _lib.load()
    if _T["cb"] is not None:
        L.lfq_shard_set_host_allgather(_lib.HOST_ALLGATHER_FN(), None)
    if _T.get("shm"):
        L.lfq_shard_shm_close()
    if _T["comm"] is not None and _T["rccl"] is not None:
        _T["rccl"].ncclCommDestroy(_T["comm"])
    if _T["own_ctx"] is not None:
        _T["own_ctx"].close()
    _T.update(dist=None, cb=None, cb_group=None, comm=None, own_ctx=None, shm=False, no_rccl=False)


def _distributed(dist):
    return dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE_COLLECTIVES)


def _host_transport(dist, group):
    """The ABI's host all-gather for the test counts: the library's shared-memory transport when every rank of `group` is on
    this node (lfq_shard_shm_open; LFQ_SHARD_HOST_TRANSPORT=gloo keeps it off), else `group` (gloo) itself through a callback."""
    if _T["cb_group"] is group and (_T["cb"] is not None or _T.get("shm")):
        return
    import ctypes as C
    import os
    import socket
    import torch
    L = _lib.load()
    ws, rank = dist.get_world_size(), dist.get_rank()
    if ws > 1 and os.environ.get("LFQ_SHARD_HOST_TRANSPORT", "shm") != "gloo":
        names = [None] * ws
        dist.all_gather_object(names, socket.gethostname(), group=group)
        if len(set(names)) == 1:
            name = ["/lofreq_amd.%d.%s" % (os.getuid(), os.urandom(8).hex())]       # new per run: rank 0 draws it
            dist.broadcast_object_list(name, src=0, group=group)
            ok = torch.tensor([1 if L.lfq_shard_shm_open(name[0].encode(), ws, rank) == 0 else 0])
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)                   # everybody has it mapped, or nobody uses it
            if int(ok.item()) == 1:
                if rank == 0:
                    L.lfq_shard_shm_unlink()                                         # nothing stays behind in /dev/shm
                _T["shm"], _T["cb"], _T["cb_group"] = True, None, group
                return
            L.lfq_shard_shm_close()

    def cb(user, world, rank, send, recv, nbytes):
        try:
            mine = torch.frombuffer((C.c_uint8 * nbytes).from_address(send), dtype=torch.uint8).clone()
            out = torch.empty(world * nbytes, dtype=torch.uint8)
            dist.all_gather_into_tensor(out, mine, group=group)
            C.memmove(recv, out.data_ptr(), world * nbytes)
            return 0
        except Exception as e:          # the C side turns this into LFQ_ERR_HIP
            __import__("sys").stderr.write("lofreq_amd.shard: host all-gather failed: %r\n" % (e,))
            return -1
    _T["cb"] = _lib.HOST_ALLGATHER_FN(cb)
    _T["cb_group"] = group
    _lib.load().lfq_shard_set_host_allgather(_T["cb"], None)


def _rccl_comm(dist, device):
    """-> (lfq_ctx handle, ncclComm_t) of this process: made once, from rank 0's ncclUniqueId"""
    import ctypes as C
    import os
    import torch
    if _T["comm"] is not None:
        return _T["ctx"].h, _T["comm"]
    dev = torch.device(device)
    if _T["ctx"] is None:
        from .caller import SnvCaller
        _T["own_ctx"] = _T["ctx"] = SnvCaller(dev.index or 0)
    rccl = None
    for mode in (os.RTLD_NOW | os.RTLD_NOLOAD, os.RTLD_NOW | os.RTLD_GLOBAL):     # the copy the process already has first
        for name in ("librccl.so", "librccl.so.1"):
            try:
                rccl = C.CDLL(name, mode=mode)
                break
            except OSError:
                continue
        if rccl is not None:
            break
    if rccl is None:
        raise RuntimeError("lofreq_amd.shard: no librccl in this process")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_byte * 128)]
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    ws, rank = dist.get_world_size(), dist.get_rank()
    uid = UniqueId()
    if rank == 0 and rccl.ncclGetUniqueId(C.byref(uid)) != 0:
        raise RuntimeError("ncclGetUniqueId failed")
    t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8)
    if _HOST_GROUP is not None:
        dist.broadcast(t, src=0, group=_HOST_GROUP)
    else:
        td = t.to(dev)
        dist.broadcast(td, src=0)
        t = td.cpu()
    C.memmove(C.byref(uid), t.numpy().ctypes.data, 128)
    comm = C.c_void_p()
    with torch.cuda.device(dev):
        if rccl.ncclCommInitRank(C.byref(comm), ws, uid, rank) != 0:
            raise RuntimeError("ncclCommInitRank failed")
    _T["rccl"], _T["comm"] = rccl, comm
    return _T["ctx"].h, comm


def _transport(dist, device, want_device):
    """-> (ctx handle or None, comm or None) for an exchange.  The host transport is registered whenever there is a host
    group or the default group itself takes host tensors (gloo); a communicator is made when the default group is RCCL and
    (a) no host group exists, or (b) the caller wants the device's road (`want_device`: the record gather)."""
    backend = dist.get_backend()
    if _HOST_GROUP is not None:
        _host_transport(dist, _HOST_GROUP)
    elif backend != "nccl":
        _host_transport(dist, None)
    if backend == "nccl" and (_HOST_GROUP is None or want_device) and not _T.get("no_rccl"):
        try:
            return _rccl_comm(dist, device)
        except Exception as e:
            # (every rank runs the same code on the same image: a failure to find or initialise RCCL is every rank's)
            if _HOST_GROUP is None:
                raise
            __import__("sys").stderr.write("lofreq_amd.shard: no RCCL communicator (%r): the records take the host transport too\n" % (e,))
            _T["no_rccl"] = True
    return None, None


def exchange_counts(local_counts, dist=None, device=None):
    """One all-gather of a small int64 vector per rank ({tested SNV columns, indel tests}), by
    lfq_shard_exchange_counts.  -> (array [world, len(local_counts)], exclusive prefix of this rank as an array)."""
    v = np.ascontiguousarray(np.asarray(local_counts, np.int64).reshape(-1))
    if not _distributed(dist):
        return v.reshape(1, -1), np.zeros_like(v)
    ws, rank = dist.get_world_size(), dist.get_rank()
    ctx, comm = _transport(dist, device, want_device=False)
    allc = np.zeros((ws, len(v)), np.int64)
    prefix = np.zeros(len(v), np.int64)
    _lib.check(_lib.load().lfq_shard_exchange_counts(ctx, comm, ws, rank, v.ctypes.data, len(v), allc.ctypes.data,
                                                     prefix.ctypes.data), "lfq_shard_exchange_counts")
    return allc, prefix


def exchange_test_counts(n_tested_local, dist=None, device=None):
    """All-gather the per-shard tested-column counts -> (counts per rank, exclusive prefix of this rank)."""
    allc, prefix = exchange_counts([n_tested_local], dist, device)
    return [int(x) for x in allc[:, 0]], int(prefix[0])


def rebase_bonferroni(pvals, prefix_tested):
    """Turn shard-local running Bonferroni factors into the single-process ones: every tested column
    of an earlier shard contributes 3 tests (lofreq_call.c:794-801)."""
    pvals = np.ascontiguousarray(pvals.copy())
    _lib.check(_lib.load().lfq_shard_rebase_bonferroni(pvals.ctypes.data, len(pvals), int(prefix_tested)),
               "lfq_shard_rebase_bonferroni")
    return pvals


def gather_records(records, col_offset, dist=None, device=None):
    """Gather reported variants to rank 0 in shard order; `col` becomes a global column index.
    Works for SNV and indel records (any structured dtype with a `col` field).  One all-gather of the record counts (every
    rank needs the largest one: the pieces are of equal size), then ONE gather of the fixed-size pieces."""
    if not _distributed(dist):
        rec = records.copy()
        rec["col"] += int(col_offset)
        return rec
    allc, _ = exchange_counts([len(records)], dist, device)
    return gather_records_wait(gather_records_start(records, col_offset, int(allc[:, 0].max()), dist, device))


class PendingGather:
    """A record gather that has been started (gather_records_start) and not yet collected."""
    __slots__ = ("records", "handle", "piece", "cap", "rdtype", "ws", "rank", "buf")


_GATHER_HDR = 16        # bytes in front of a piece's records: int64 record count, int64 reserved (keeps the records aligned)


def gather_records_start(records, col_offset, cap, dist=None, device=None):
    """First half of gather_records for callers that pipeline steps (lfq_shard_gather_start): every rank sends ONE piece
    of fixed capacity (`cap` records, the same number on every rank -- e.g. 3 x the largest candidate-column count of the
    step, which the count all-gather already made known to everybody; the piece starts with its own record count), so
    that no second all-gather of the sizes is needed, and nothing here waits for the device: with a communicator the
    piece goes up from a pinned copy, the collective and rank 0's copy back are queued on the handle's own high-priority
    stream with an event that gather_records_wait waits for.  -> PendingGather.  `cap` too small for this rank's records
    raises before anything is sent."""
    rec = records.copy()
    rec["col"] += int(col_offset)
    h = PendingGather()
    h.records, h.handle, h.buf = rec, None, None
    h.rdtype = rec.dtype
    if not _distributed(dist):
        return h
    import ctypes as C
    h.ws, h.rank, h.cap = dist.get_world_size(), dist.get_rank(), max(int(cap), 1)
    if len(rec) > h.cap:
        raise ValueError("gather_records_start: %d records, capacity %d" % (len(rec), h.cap))
    width = h.rdtype.itemsize
    h.piece = _GATHER_HDR + h.cap * width
    buf = np.zeros(h.piece, np.uint8)
    buf[:8] = np.array([len(rec)], np.int64).view(np.uint8)
    buf[_GATHER_HDR: _GATHER_HDR + len(rec) * width] = np.ascontiguousarray(rec).view(np.uint8).reshape(-1)
    ctx, comm = _transport(dist, device, want_device=True)
    handle = C.c_void_p()
    _lib.check(_lib.load().lfq_shard_gather_start(ctx, comm, h.ws, h.rank, buf.ctypes.data, h.piece, 1 if h.rank == 0 else 0,
                                                  C.byref(handle)), "lfq_shard_gather_start")
    h.handle = handle
    h.records = None
    return h


def gather_records_wait(h):
    """Second half: the records of all ranks in shard order on rank 0, None elsewhere (every rank waits for its own side
    of the collective, so that the buffers it handed over are free again)."""
    if h.handle is None:
        return h.records
    L = _lib.load()
    if h.rank != 0:
        _lib.check(L.lfq_shard_gather_wait(h.handle, None, 0), "lfq_shard_gather_wait")
        h.handle = None
        return None
    host = np.zeros((h.ws, h.piece), np.uint8)
    _lib.check(L.lfq_shard_gather_wait(h.handle, host.ctypes.data, host.nbytes), "lfq_shard_gather_wait")
    h.handle = None
    width = h.rdtype.itemsize
    parts = []
    for r in range(h.ws):
        n = int(host[r, :8].view(np.int64)[0])
        parts.append(host[r, _GATHER_HDR: _GATHER_HDR + n * width].view(h.rdtype))
    return np.concatenate(parts)


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
