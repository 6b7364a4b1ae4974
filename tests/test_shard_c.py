"""The C-level exchange of a sharded run (lfq_shard_*, include/lofreq_amd.h) against lofreq_amd/shard.py, which does the
same exchange on torch.distributed.  CPU: the arithmetic of one process (world 1, no communicator) and the argument
checks; GPU: the same calls through a real one-rank RCCL communicator (ncclAllGather on the device)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lofreq_amd import _lib, shard          # noqa: E402
import lofreq_amd as la                       # noqa: E402


def _fake_pvals(n, seed):
    rng = np.random.default_rng(seed)
    pv = np.zeros(n, _lib.COL_PVALS_DTYPE)
    pv["col"] = np.sort(rng.choice(100000, n, replace=False))
    pv["bonf"] = 3 * (1 + np.arange(n))
    return pv


def _fake_records(n, seed):
    rng = np.random.default_rng(seed)
    r = np.zeros(n, _lib.SNV_RECORD_DTYPE)
    r["col"] = np.sort(rng.choice(100000, n, replace=False))
    r["qual"] = rng.integers(0, 3000, n)
    r["dp"] = rng.integers(10, 10000, n)
    return r


def _exchange(L, ctx, comm, local):
    local = np.asarray(local, np.int64)
    allc = np.zeros((1, len(local)), np.int64)
    prefix = np.zeros(len(local), np.int64)
    _lib.check(L.lfq_shard_exchange_counts(ctx, comm, 1, 0, local.ctypes.data, len(local), allc.ctypes.data,
                                           prefix.ctypes.data), "lfq_shard_exchange_counts")
    return allc, prefix


def _run_world1(L, ctx, comm):
    allc, prefix = _exchange(L, ctx, comm, [1234, 56])
    want_all, want_prefix = shard.exchange_counts([1234, 56])
    assert np.array_equal(allc, want_all) and np.array_equal(prefix, want_prefix)
    pv = _fake_pvals(50, 1)
    mine = pv.copy()
    _lib.check(L.lfq_shard_rebase_bonferroni(mine.ctypes.data, len(mine), 777), "rebase")
    assert np.array_equal(mine["bonf"], shard.rebase_bonferroni(pv, 777)["bonf"])
    recs = _fake_records(40, 2)
    out = np.zeros(64, _lib.SNV_RECORD_DTYPE)
    n_out = C.c_int64(0)
    _lib.check(L.lfq_shard_gather_records(ctx, comm, 1, 0, recs.ctypes.data, len(recs), 5000, out.ctypes.data, len(out),
                                          C.byref(n_out)), "gather")
    want = shard.gather_records(recs, 5000)
    assert n_out.value == len(want) and out[: n_out.value].tobytes() == want.tobytes()
    # capacity too small: the count still comes back
    small = np.zeros(8, _lib.SNV_RECORD_DTYPE)
    rc = L.lfq_shard_gather_records(ctx, comm, 1, 0, recs.ctypes.data, len(recs), 0, small.ctypes.data, len(small),
                                    C.byref(n_out))
    assert rc == _lib.LFQ_ERR_CAPACITY and n_out.value == len(recs)


def test_world1_matches_shard_py():
    L = _lib.load()
    _run_world1(L, None, None)
    for start in (1, 3000):
        for dynamic in (0, 1):
            a, b = la.VarcallConf(), la.VarcallConf()
            a.c.bonf_subst = b.c.bonf_subst = start
            a.c.bonf_dynamic = b.c.bonf_dynamic = dynamic
            _lib.check(L.lfq_shard_advance_conf(C.byref(a.c), 4321), "advance")
            # what shard.finish_shard does to conf after the exchange
            if b.c.bonf_dynamic:
                b.c.bonf_subst = (0 if b.c.bonf_subst == 1 else b.c.bonf_subst) + 3 * 4321
            b.c.num_snv_tests += 3 * 4321
            assert (a.c.bonf_subst, a.c.num_snv_tests) == (b.c.bonf_subst, b.c.num_snv_tests)


def test_argument_checks():
    L = _lib.load()
    one = np.zeros(1, np.int64)
    assert L.lfq_shard_exchange_counts(None, None, 2, 0, one.ctypes.data, 1, one.ctypes.data, None) == -1  # world 2 needs a communicator
    assert L.lfq_shard_exchange_counts(None, None, 1, 1, one.ctypes.data, 1, one.ctypes.data, None) == -1
    assert L.lfq_shard_rebase_bonferroni(None, 3, 0) == -1
    assert L.lfq_shard_advance_conf(None, 1) == -1


class _NcclId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


@pytest.mark.gpu
def test_one_rank_rccl_communicator():
    """the same exchange through ncclAllGather: a communicator of one rank on cuda:0"""
    import torch  # noqa: F401  (brings librccl into the process)
    rccl = None
    for name in ("librccl.so", "librccl.so.1"):
        try:
            rccl = C.CDLL(name, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            continue
    if rccl is None:
        tl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        rccl = C.CDLL(tl, mode=C.RTLD_GLOBAL)
    uid = _NcclId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclId, C.c_int]
    caller = la.SnvCaller(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        _run_world1(_lib.load(), caller.h, comm)
    finally:
        rccl.ncclCommDestroy(comm)
        caller.close()
