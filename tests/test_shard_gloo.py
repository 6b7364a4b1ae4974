"""CPU tests of the N > 1 path: two processes, gloo backend, 127.0.0.1 rendezvous.  Checks that region
sharding + the test-count all-gather + record gather reproduce the single-process result exactly
(running Bonferroni factor included)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_pvals(la, rng, cols, bonf_local):
    """sparse records as the DP kernels would emit them for a shard (host-constructed: no GPU here)"""
    pv = np.zeros(len(cols), la.COL_PVALS_DTYPE)
    pv["col"] = cols
    pv["bonf"] = bonf_local
    pv["logp"][:, 0] = -rng.uniform(5, 60, len(cols))
    pv["status"][:, 0] = la.LFQ_PV_LOG
    pv["counts"]["alt_counts"][:, 0] = rng.integers(5, 50, len(cols))
    pv["counts"]["kmax"] = pv["counts"]["alt_counts"][:, 0]
    pv["counts"]["alt_raw_counts"][:, 0] = pv["counts"]["alt_counts"][:, 0] + 1
    pv["counts"]["alt_fw"][:, 0] = pv["counts"]["alt_counts"][:, 0] // 2
    pv["counts"]["ref_fw"] = 400
    pv["counts"]["ref_rv"] = 390
    pv["counts"]["coverage"] = 900
    return pv


def _scenario(la):
    rng = np.random.default_rng(123)
    ncols = 2000
    tested = rng.random(ncols) < 0.7
    prefix = np.cumsum(tested)                      # inclusive tested count
    cand = np.nonzero(tested & (rng.random(ncols) < 0.03))[0]
    ref = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, ncols)].copy()
    pv = _fake_pvals(la, rng, cand, 3 * prefix[cand])
    pv["ref_base"] = ref[cand]                      # device records carry their column's reference base
    return ncols, tested, prefix, ref, pv


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lofreq_amd as la
    from lofreq_amd import shard
    ncols, tested, prefix, ref, pv = _scenario(la)
    lo, hi = shard.shard_ranges(ncols, world)[rank]
    mine = pv[(pv["col"] >= lo) & (pv["col"] < hi)].copy()
    # what a shard sees locally: column indices and Bonferroni factors restart at its own origin
    before = int(tested[:lo].sum())
    mine["col"] -= lo
    mine["bonf"] -= 3 * before
    conf = la.VarcallConf()
    recs, total = shard.finish_shard(conf, mine, int(tested[lo:hi].sum()), ref[lo:hi], lo, dist, None)
    if rank == 0:
        np.save(out, recs.view(np.uint8))
        np.save(out + ".meta", np.array([total, conf.bonf_subst, conf.num_snv_tests]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_equals_single_process(tmp_path, world):
    import lofreq_amd as la
    from lofreq_amd import shard
    out = str(tmp_path / "recs.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out).view(la.SNV_RECORD_DTYPE)
    total, bonf, ntests = np.load(out + ".meta.npy")
    ncols, tested, prefix, ref, pv = _scenario(la)
    conf = la.VarcallConf()
    exp, total1 = shard.finish_shard(conf, pv, int(tested.sum()), ref, 0, None, None)
    assert total == total1 == int(tested.sum())
    assert bonf == conf.bonf_subst == 3 * int(tested.sum()) and ntests == conf.num_snv_tests
    assert len(got) == len(exp) and len(exp) > 10
    assert got.tobytes() == exp.tobytes()


def _lagged_worker(rank, world, port, out):
    """three steps pipelined as bench.py's sharded loop runs them: the counts over a host group of their own, every
    step's record gather started and collected one step later (the last one after the loop)"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lofreq_amd as la
    from lofreq_amd import shard
    shard.set_host_group(dist.new_group(backend="gloo"))
    ncols, tested, prefix, ref, pv = _scenario(la)
    lo, hi = shard.shard_ranges(ncols, world)[rank]
    before = int(tested[:lo].sum())
    outs, prev = [], None
    for step in range(3):
        # (step 1: this rank's shard has no candidate column at all; step 2: only rank 0's has any)
        sel = (pv["col"] >= lo) & (pv["col"] < hi) & ((step == 0) | ((step == 2) & (rank == 0)))
        mine = pv[sel].copy()
        mine["col"] -= lo
        mine["bonf"] -= 3 * before
        conf = la.VarcallConf()
        h, total = shard.finish_shard_start(conf, mine, int(tested[lo:hi].sum()), ref[lo:hi], lo, dist, None)
        if prev is not None:
            outs.append(shard.finish_shard_wait(prev))
        prev = h
        assert total == int(tested.sum()) and conf.bonf_subst == 3 * total
    outs.append(shard.finish_shard_wait(prev))
    if rank == 0:
        for i, r in enumerate(outs):
            np.save("%s.%d.npy" % (out, i), r.view(np.uint8))
    else:
        assert all(r is None for r in outs)
    shard.set_host_group(None)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_lagged_gather_equals_single_process(tmp_path, world):
    """finish_shard_start / finish_shard_wait (one all-gather over the host group, one asynchronous fixed-capacity
    gather, collected a step later) give the records of the one-process run, also for steps in which some or all
    shards have nothing to report"""
    import lofreq_amd as la
    from lofreq_amd import shard
    out = str(tmp_path / "lag")
    mp.spawn(_lagged_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    ncols, tested, prefix, ref, pv = _scenario(la)
    lo0, hi0 = shard.shard_ranges(ncols, world)[0]
    for step in range(3):
        got = np.load("%s.%d.npy" % (out, step)).view(la.SNV_RECORD_DTYPE)
        sel = np.ones(len(pv), bool) if step == 0 else (np.zeros(len(pv), bool) if step == 1 else
                                                        ((pv["col"] >= lo0) & (pv["col"] < hi0)))
        conf = la.VarcallConf()
        exp, _ = shard.finish_shard(conf, pv[sel], int(tested.sum()), ref, 0, None, None)
        assert got.tobytes() == exp.tobytes(), step
        assert (len(exp) > (10 if world < 8 else 3)) == (step != 1)


def _latency_worker(rank, world, port, out, transport):
    """what bench.py's sharded step does per step on the host side of the exchange, 300 times: the {tested columns, candidate
    columns} all-gather over the host group"""
    import time
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LFQ_SHARD_HOST_TRANSPORT"] = transport
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lofreq_amd import shard
    shard.set_host_group(dist.new_group(backend="gloo"))
    ts = []
    for i in range(350):
        t0 = time.perf_counter()
        allc, prefix = shard.exchange_counts([1000 + rank + i, 7 * rank], dist, None)
        ts.append(time.perf_counter() - t0)
        assert allc[:, 0].tolist() == [1000 + r + i for r in range(world)] and int(prefix[1]) == 7 * rank * (rank - 1) // 2
    if rank == 0:
        np.save(out, np.array(ts[50:]))
    shard.shutdown()
    dist.destroy_process_group()


def test_host_exchange_latency_at_eight_ranks(tmp_path):
    """the per-step test-count all-gather of eight ranks on one node costs the host <= 0.1 ms.  Through
    the library's shared-memory transport (lfq_shard_shm_open, what shard.set_host_group picks on one node); the same
    exchange over the gloo group itself (the fallback for ranks on several nodes) gives the same numbers, slower."""
    out = str(tmp_path / "lat.npy")
    mp.spawn(_latency_worker, args=(8, _free_port(), out, "shm"), nprocs=8, join=True)
    shm = np.load(out)
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("lofreq_amd.%d." % os.getuid())]     # unlinked once mapped
    mp.spawn(_latency_worker, args=(8, _free_port(), out, "gloo"), nprocs=8, join=True)
    gloo = np.load(out)
    print("host all-gather at 8 ranks: shm median %.4f ms (p90 %.4f), gloo median %.3f ms"
          % (1e3 * np.median(shm), 1e3 * np.percentile(shm, 90), 1e3 * np.median(gloo)))
    if (os.cpu_count() or 1) >= 8:                      # eight spinning ranks need a core each
        assert np.median(shm) <= 1e-4, np.median(shm)
    assert np.median(shm) < np.median(gloo)


def test_gather_capacity_is_checked():
    import lofreq_amd as la
    from lofreq_amd import shard
    recs = np.zeros(5, la.SNV_RECORD_DTYPE)
    h = shard.gather_records_start(recs, 7, 2)            # no communicator: nothing is sent, nothing to bound
    assert (shard.gather_records_wait(h)["col"] == 7).all()


def test_shard_ranges():
    from lofreq_amd import shard
    assert shard.shard_ranges(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert shard.shard_ranges(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]


def _indel_scenario(la):
    """emitted-candidate indel records of a 600-column run with their single-process running factors"""
    rng = np.random.default_rng(77)
    ncols = 600
    tests_per_col = rng.choice([0, 0, 1, 2, 3], ncols)
    first_test = np.concatenate([[0], np.cumsum(tests_per_col)])      # tests before column c
    recs = []
    for c in range(ncols):
        for e in range(tests_per_col[c]):
            if rng.random() < 0.5:
                r = np.zeros(1, la.INDEL_RECORD_DTYPE)
                r["col"], r["side"], r["event"] = c, e & 1, e
                r["bonf"] = 1 + first_test[c] + e + 1                 # bonf_indel after this test's increment
                # p-values straddling sig / bonf so that the exact factor matters
                r["pvalue"] = np.longdouble(0.01) / np.longdouble(rng.uniform(0.2, 3.0) * (first_test[c] + 2))
                r["qual"] = int(-10 * np.log10(float(r["pvalue"][0])))
                recs.append(r)
    return ncols, tests_per_col, first_test, np.concatenate(recs)


def _indel_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lofreq_amd as la
    from lofreq_amd import shard
    ncols, tpc, first, recs = _indel_scenario(la)
    lo, hi = shard.shard_ranges(ncols, world)[rank]
    mine = recs[(recs["col"] >= lo) & (recs["col"] < hi)].copy()
    mine["col"] -= lo
    mine["bonf"] -= int(first[lo])                  # the shard's local running factor
    # what the local emit test (local factor) would have passed on to us
    mine = mine[mine["pvalue"] * mine["bonf"].astype(np.longdouble) < np.float32(0.01)]
    conf = la.VarcallConf()
    n_local = int(tpc[lo:hi].sum())
    conf.c.bonf_indel += n_local                    # as lfq_call_indels_batch leaves it
    conf.c.num_indel_tests += n_local
    got, total = shard.finish_indel_shard(conf, 1, mine, n_local, lo, dist, None)
    if rank == 0:
        np.save(out, got.view(np.uint8))
        np.save(out + ".meta", np.array([total, conf.bonf_indel, conf.num_indel_tests]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_indels_equal_single_process(tmp_path, world):
    import lofreq_amd as la
    out = str(tmp_path / "irecs.npy")
    mp.spawn(_indel_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out).view(la.INDEL_RECORD_DTYPE)
    total, bonf, ntests = np.load(out + ".meta.npy")
    ncols, tpc, first, recs = _indel_scenario(la)
    exp = recs[recs["pvalue"] * recs["bonf"].astype(np.longdouble) < np.float32(0.01)]
    assert total == ntests == int(tpc.sum()) and bonf == 1 + int(tpc.sum())
    assert 5 < len(exp) < len(recs)
    assert len(got) == len(exp)
    for k in la.INDEL_RECORD_DTYPE.names:            # field-wise: the long double carries 6 padding bytes
        assert (got[k] == exp[k]).all(), k


def _indel_bins_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lofreq_amd as la
    from lofreq_amd import shard
    ncols, tpc, first, recs = _indel_scenario(la)
    # seven bins of unequal length, dealt round-robin: a rank's bins are NOT contiguous in the genome
    edges = np.linspace(0, ncols, 8).astype(int)
    edges[3] += 5
    mine = []
    for b in range(7):
        if b % world != rank:
            continue
        lo, hi = int(edges[b]), int(edges[b + 1])
        r = recs[(recs["col"] >= lo) & (recs["col"] < hi)].copy()
        r["col"] -= lo
        r["bonf"] -= int(first[lo])                 # every bin starts from bonf_indel 1
        r = r[r["pvalue"] * r["bonf"].astype(np.longdouble) < np.float32(0.01)]     # the bin's own emit test
        mine.append((b, lo, r, int(tpc[lo:hi].sum())))
    conf = la.VarcallConf()
    got, total = shard.finish_indel_bins(conf, 1, mine, 7, dist, None)
    if rank == 0:
        np.save(out, got.view(np.uint8))
        np.save(out + ".meta", np.array([total, conf.bonf_indel, conf.num_indel_tests]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_indel_bins_equal_single_process(tmp_path, world):
    import lofreq_amd as la
    out = str(tmp_path / "ibins.npy")
    mp.spawn(_indel_bins_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out).view(la.INDEL_RECORD_DTYPE)
    total, bonf, ntests = np.load(out + ".meta.npy")
    ncols, tpc, first, recs = _indel_scenario(la)
    exp = recs[recs["pvalue"] * recs["bonf"].astype(np.longdouble) < np.float32(0.01)]
    assert total == ntests == int(tpc.sum()) and bonf == 1 + int(tpc.sum())
    assert len(got) == len(exp) > 5
    for k in la.INDEL_RECORD_DTYPE.names:
        assert (got[k] == exp[k]).all(), k


# ---- call-parallel style bins (shard.plan_regions / finish_bins) --------------------------------------------

def _exome_like(rng, length=3_000_000, n_targets=1200):
    """BED-like targets with ragged depth (log-normal per target) and a few K-heavy positions"""
    depth = np.zeros(length)
    kest = np.zeros(length)
    regions, pos = [], 0
    for _ in range(n_targets):
        pos += int(rng.integers(150, 2500))
        ln = int(rng.integers(100, 2500))
        if pos + ln >= length:
            break
        regions.append(("chr1" if pos < length // 2 else "chr2", pos, pos + ln))
        depth[pos:pos + ln] = rng.lognormal(5.0, 1.0)
        pos += ln
    hot = rng.integers(0, length, 200)
    kest[hot] = rng.integers(5, 3000, 200)
    return regions, {"chr1": depth, "chr2": depth}, {"chr1": kest, "chr2": kest}


@pytest.mark.parametrize("world", [2, 4, 8])
def test_plan_regions_balances_ragged_targets(world):
    """SURVEY 8e / lofreq2_call_pparallel.py:590-613: >= 2 bins per worker, greedy split of the biggest, dealt longest
    first; balanced by sum of depth + a DP term to within 10 % of the mean on an exome-like BED"""
    from lofreq_amd import shard
    rng = np.random.default_rng(5)
    regions, depth, kest = _exome_like(rng)
    cost = shard.make_cost_fn(depth, kest)
    bins, owner = shard.plan_regions(regions, cost, world)
    # the bins tile the targets exactly, in genome order
    assert bins == sorted(bins, key=lambda b: (b[0] != "chr1", b[1]))
    covered = {}
    for c, b, e in bins:
        assert e > b
        covered.setdefault(c, []).append((b, e))
    want = {}
    for c, b, e in regions:
        want.setdefault(c, []).append((b, e))
    for c in want:
        merged = []
        for b, e in sorted(covered[c]):
            if merged and merged[-1][1] == b and not any(b == wb for wb, _ in want[c]):
                merged[-1] = (merged[-1][0], e)
            else:
                merged.append((b, e))
        assert merged == sorted(want[c]), c
    load = np.zeros(world)
    for b, o in zip(bins, owner):
        load[o] += cost(*b)
    assert load.max() <= 1.1 * load.mean(), (load.max() / load.mean())
    assert np.bincount(owner, minlength=world).min() >= shard.BIN_PER_THREAD
    # one uniform contig: the reference's own rule (split while the biggest is not below total / (2 * workers))
    bins, owner = shard.plan_regions([("synth", 0, 8_000_000)], lambda c, b, e: 1e4 * (e - b), world)
    assert len(bins) == 4 * world and sorted(np.bincount(owner)) == [4] * world
    assert all(e - b == 8_000_000 // (4 * world) for _, b, e in bins)


def _max_over_mean(bins, owner, cost, world):
    load = np.zeros(world)
    for b, o in zip(bins, owner):
        load[o] += cost(*b)
    return load.max() / load.mean(), np.bincount(owner, minlength=world)


def test_plan_regions_at_eight_workers_on_the_c4_and_c5_depth_profiles():
    """what an 8-GPU node gets of BASELINE configs[3] and [4] -- the heaviest GPU's cost within 10 % of
    the mean.  C4: one 4.6 Mb contig, 500x with the ripple a real BAM has (+- 30 %, a few 5-fold pile-ups) and insertion /
    deletion sites whose columns cost a DP each; C5: ragged exome targets (150 .. 3500 bp, the generator of bench.py's
    make_tile) over 64 Mb at 200x with log-normal depth per target.  Also: the uniform synthetic genomes bench.py runs."""
    from lofreq_amd import shard
    rng = np.random.default_rng(64)
    W = 8
    # C4 shape, 100-base windows
    nwin = 46000
    depth = 500.0 * (1.0 + 0.3 * np.sin(np.arange(nwin) / 700.0)) * rng.lognormal(0.0, 0.1, nwin)
    depth[rng.integers(0, nwin, 40)] *= 5.0
    kest = np.zeros(nwin)
    kest[rng.integers(0, nwin, 900)] = rng.integers(5, 150, 900)           # planted variants: K ~ AF x depth
    cost = shard.make_cost_fn({"ecoli": depth}, {"ecoli": kest / 100.0}, bin_size=100)
    bins, owner = shard.plan_regions([("ecoli", 0, 4_600_000)], cost, W)
    r, per = _max_over_mean(bins, owner, cost, W)
    assert r <= 1.10 and per.min() >= shard.BIN_PER_THREAD, (r, per)
    assert sum(e - b for _, b, e in bins) == 4_600_000
    # C5 shape
    targets, x = [], 300
    while x < 64_000_000 - 4000:
        l = int(rng.choice([150, 300, 600, 1200, 2000, 3500], p=[0.2, 0.25, 0.2, 0.15, 0.12, 0.08]))
        targets.append(("chrE", x, x + l))
        x += l + int(rng.integers(200, 1900))
    tdepth = {t: 200.0 * rng.lognormal(0.0, 0.6) for t in targets}
    starts = np.array([t[1] for t in targets])
    ends = np.array([t[2] for t in targets])
    dens = np.array([tdepth[t] for t in targets])

    def cost5(c, b, e):
        lo = np.clip(b, starts, ends)
        hi = np.clip(e, starts, ends)
        return float(((hi - lo) * dens).sum())
    bins, owner = shard.plan_regions(targets, cost5, W)
    r, per = _max_over_mean(bins, owner, cost5, W)
    assert r <= 1.10 and per.min() >= shard.BIN_PER_THREAD, (r, per)
    assert 25_000_000 < sum(e - b for _, b, e in bins) < 32_000_000      # ~29 Mb of targets, every base in exactly one bin
    # the uniform genomes of `bench.py --config C4 / C5 --gpus 8`: 32 equal bins, four per GPU, whatever N is
    for glen, d in ((4_600_000, 500), (64_000_000, 200)):
        glen = glen // 32 * 32
        bins, owner = shard.plan_regions([("synth", 0, glen)], lambda c, b, e: float(d) * (e - b), W, bins_per_worker=2)
        assert len(bins) == 32 and sorted(np.bincount(owner)) == [4] * W and len({e - b for _, b, e in bins}) == 1


def _bins_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lofreq_amd as la
    from lofreq_amd import shard
    ncols, tested, prefix, ref, pv = _scenario(la)
    # ragged cost: the second half of the genome is four times as expensive
    cost = lambda c, b, e: float((e - b) + 3 * max(0, e - max(b, ncols // 2)))
    bins, owner = shard.plan_regions([("g", 0, ncols)], cost, world)
    mine = []
    for i, ((_, lo, hi), o) in enumerate(zip(bins, owner)):
        if o != rank:
            continue
        p = pv[(pv["col"] >= lo) & (pv["col"] < hi)].copy()
        p["col"] -= lo                               # every bin is its own batch: local columns, local factors
        p["bonf"] -= 3 * int(tested[:lo].sum())
        mine.append((i, lo, p, int(tested[lo:hi].sum())))
    conf = la.VarcallConf()
    recs, total = shard.finish_bins(conf, mine, len(bins), dist, None)
    if rank == 0:
        np.save(out, recs.view(np.uint8))
        np.save(out + ".meta", np.array([total, conf.bonf_subst, conf.num_snv_tests, len(bins)]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bins_equal_single_process(tmp_path, world):
    """interleaved bins of several ranks: per-bin exact Bonferroni prefix + genome-order merge == one process"""
    import lofreq_amd as la
    from lofreq_amd import shard
    out = str(tmp_path / "brecs.npy")
    mp.spawn(_bins_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out).view(la.SNV_RECORD_DTYPE)
    total, bonf, ntests, nbins = np.load(out + ".meta.npy")
    ncols, tested, prefix, ref, pv = _scenario(la)
    conf = la.VarcallConf()
    exp, total1 = shard.finish_shard(conf, pv, int(tested.sum()), ref, 0, None, None)
    assert nbins >= 2 * world
    assert total == total1 and bonf == conf.bonf_subst and ntests == conf.num_snv_tests
    assert len(got) == len(exp) and got.tobytes() == exp.tobytes()
