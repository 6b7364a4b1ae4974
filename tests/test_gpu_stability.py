"""-m gpu: repeated calls of every batch entry point leave the device memory where it was (no leak in the per-call
staging), results stay identical from call to call, and invalid arguments come back as error codes."""
import ctypes as C

import numpy as np
import pytest
import torch

import golden_util as gu
import util

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def _free_bytes():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def test_repeated_calls_no_leak_and_deterministic(caller, oracle):
    import lofreq_amd as la
    fx, reads = gu.load_plpindel(gu.plpindel_fixtures()[0])
    ref = fx["genome"].encode()
    rng = np.random.default_rng(3)
    host = util.random_batch(rng, 400, 50, 600)
    first = None
    base = None
    for it in range(12):
        tags = la.baq_batch(caller, reads, ref, extended=True, idaq=True)
        sq, sqb = la.source_qual_batch(caller, reads, ref)
        cols, col_pos = la.pileup_indel_columns(caller, reads, ref, 0, len(ref))
        dt = la.pileup_snv_tracks(caller, reads, ref, 0, len(ref), lb=[t[0] for t in tags], sq=sqb)
        conf = la.VarcallConf()
        recs, _, st = caller.call_snvs(dt, conf)
        irecs, nt = la.call_indels(caller, cols, la.VarcallConf())
        recs2, _, _ = caller.call_snvs(util.to_pileup_batch(la, host), la.VarcallConf())
        det, _ = caller.uniq_detlim(util.to_pileup_batch(la, host), np.full(400, 0.05, np.float32))
        sig = (b"".join(t[0].tobytes() for t in tags), sq.tobytes(), recs.tobytes(), irecs["qual"].tobytes(),
               recs2.tobytes(), det.tobytes(), cols.sides[0]["rd_q"].tobytes())
        if first is None:
            first = sig
        assert sig == first, "call %d differs from the first" % it
        if it == 1:
            base = _free_bytes()            # after the pools have grown to this workload
    assert abs(_free_bytes() - base) < (64 << 20)


def test_invalid_arguments_are_error_codes(caller):
    from lofreq_amd import _lib
    L = _lib.load()
    vp = C.c_void_p
    assert L.lfq_source_qual_batch(caller.h, None, -1, 6, None, None, None) < 0
    assert L.lfq_pileup_indel_columns(caller.h, None, None, 0, 10, 0, None, None) < 0
    assert L.lfq_pileup_skip_snv_columns(caller.h, None, 5) < 0
    assert L.lfq_uniq_detlim_batch(caller.h, None, 0, None, None, None) < 0
    rd = _lib.PileupReads()
    rd.n_reads = -1
    out = C.POINTER(_lib.IndelColumnsC)()
    assert L.lfq_pileup_indel_columns(caller.h, C.byref(rd), None, 0, 10, 0, C.byref(out), None) < 0
    rd.n_reads = 0
    assert L.lfq_pileup_indel_columns(caller.h, C.byref(rd), None, 10, 0, 0, C.byref(out), None) < 0   # end < begin
    t = _lib.Tracks()
    t.ncols = 3
    af = np.array([0.1, 0.2, 0.1], np.float32)
    det = np.zeros(3, np.uint8)
    assert L.lfq_uniq_detlim_batch(caller.h, C.byref(t), 0, vp(af.ctypes.data), vp(det.ctypes.data), None) < 0   # no tracks
    one = np.zeros(32, np.uint8)
    off = np.array([0, 4, 8, 12], np.uint64)
    t.nt = t.bq = t.mq = vp(one.ctypes.data)
    t.col_off, t.ref_base = vp(off.ctypes.data), vp(np.frombuffer(b"ACG" + bytes(13), np.uint8).copy().ctypes.data)
    af[1] = np.nan                                              # a NaN AF is refused (an AF out of [0, 1] is RESET: test_gpu_uniq)
    assert L.lfq_uniq_detlim_batch(caller.h, C.byref(t), 0, vp(af.ctypes.data), vp(det.ctypes.data), None) < 0


def test_submit_collect_two_batches_in_flight(caller, oracle):
    """lfq_call_snvs_submit / _collect on two contexts: batch k+1 launched before batch k is finished on the host;
    records identical to the one-call route"""
    import lofreq_amd as la
    rng = np.random.default_rng(11)
    hosts = [util.random_batch(rng, 300, 300, 900, planted={7: 0.3, 100: 0.5, 200: 0.2}) for _ in range(4)]
    batches = [util.to_pileup_batch(la, h) for h in hosts]
    want = [caller.call_snvs(b, la.VarcallConf())[0] for b in batches]
    other = la.SnvCaller(0)
    ctx = [caller, other]
    confs = [la.VarcallConf() for _ in batches]
    got = [None] * len(batches)
    ctx[0].call_snvs_submit(batches[0], confs[0])
    for k in range(1, len(batches) + 1):
        if k < len(batches):
            ctx[k % 2].call_snvs_submit(batches[k], confs[k])
        got[k - 1], st = ctx[(k - 1) % 2].call_snvs_collect(confs[k - 1])
    for g, w, cf in zip(got, want, confs):
        assert g.tobytes() == w.tobytes()
        assert cf.num_snv_tests > 0
    assert sum(len(w) for w in want) > 0
    # collect without a submitted batch is an error, not a hang
    from lofreq_amd import _lib
    n = __import__("ctypes").c_int64(0)
    assert _lib.load().lfq_call_snvs_collect(caller.h, __import__("ctypes").byref(confs[0].c), None, 0,
                                             __import__("ctypes").byref(n), None, None) < 0
    other.close()


def test_submit_collect_dense_counts_without_strands(oracle):
    """lfq_set_dense_strand_counts(ctx, 0) + lfq_call_snvs_submit + lfq_call_snvs_collect(h_counts): the dense entries of a
    shallow, nt-packed batch are COMPLETE (every column, strand fields 0) -- the shared-wavefront count kernel skips the
    entries of untested columns only after lfq_set_dense_counts(ctx, 0), and then collect refuses h_counts instead of
    handing out what the array held before (the batch stays collectable without it).  All three gates give the same records."""
    import ctypes as C
    import lofreq_amd as la
    from lofreq_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(23)
    host = util.random_batch(rng, 500, 100, 700, alt_rate=0.0004, planted={7: 0.3, 100: 0.5, 200: 0.2}, ref_n_frac=0.05)
    ores, _ = util.run_oracle(oracle, host)
    assert 50 < int((ores["tested"] == 0).sum()) < 450                       # tested and untested columns side by side
    own = la.SnvCaller(0)
    try:
        batch = util.to_pileup_batch(la, host).packed()
        # a first batch that fills the context's dense array with OTHER columns' entries: stale data to be caught
        other = util.to_pileup_batch(la, util.random_batch(rng, 500, 100, 700, alt_rate=0.01)).packed()
        own.call_snvs(other, la.VarcallConf(), want_counts=True)
        own.set_dense_strand_counts(False)
        want = None
        for gate in ("tail", "end", "none"):
            own.set_batch_gate(gate)
            conf = la.VarcallConf()
            own.call_snvs_submit(batch, conf)
            rec = np.empty(3 * batch.ncols, dtype=_lib.SNV_RECORD_DTYPE)
            counts = np.full(batch.ncols, 0x5A, dtype=np.uint8).repeat(64).view(_lib.COL_COUNTS_DTYPE)
            n, st = C.c_int64(0), _lib.BatchStats()
            rc = L.lfq_call_snvs_collect(own.h, C.byref(conf.c), C.c_void_p(rec.ctypes.data), len(rec), C.byref(n),
                                         C.c_void_p(counts.ctypes.data), C.byref(st))
            own._sub = None
            assert rc == 0
            for f in ("n_err_probs", "alt_counts"):
                assert np.array_equal(counts[f], ores[f]), f
            assert np.array_equal(counts["tested"].astype(np.int32), ores["tested"])
            # record-only fields (strands, raw alt counts): 0 or the true value -- the library counts them where it has a
            # use for them itself (the heavy columns' strand-bias precompute), never anything else
            raw_ok = (counts["alt_raw_counts"] == ores["alt_raw_counts"]).all(axis=1)
            assert (raw_ok | (counts["alt_raw_counts"] == 0).all(axis=1)).all()
            light = counts["kmax"] < 12
            assert not counts["ref_fw"][light].any() and not counts["alt_fw"][light].any() and light.sum() > 300
            recs = rec[: n.value].copy()
            assert n.value > 0 and (want is None or recs.tobytes() == want.tobytes())
            want = recs
        # sparse entries: h_counts is refused, the batch is still there
        own.set_dense_counts(False)
        conf = la.VarcallConf()
        own.call_snvs_submit(batch, conf)
        n = C.c_int64(0)
        counts = np.zeros(batch.ncols, dtype=_lib.COL_COUNTS_DTYPE)
        rec = np.empty(3 * batch.ncols, dtype=_lib.SNV_RECORD_DTYPE)
        assert L.lfq_call_snvs_collect(own.h, C.byref(conf.c), C.c_void_p(rec.ctypes.data), len(rec), C.byref(n),
                                       C.c_void_p(counts.ctypes.data), None) == -1          # LFQ_ERR_INVALID
        recs, st = own.call_snvs_collect(conf, records_capacity=3 * batch.ncols)
        assert recs.tobytes() == want.tobytes()
    finally:
        own.close()


def test_private_stream_same_results(oracle):
    """lfq_set_private_stream (ABI 5): a context with a launch stream of its own gives what the device's shared stream gives, on
    every chain (BAQ + IDAQ, source quality, both pileups, SNV and indel calls, two batches in flight); switching back works; an
    unknown context is an error code."""
    import lofreq_amd as la
    from lofreq_amd import _lib
    L = _lib.load()
    assert L.lfq_set_private_stream(None, 1) < 0
    fx, reads = gu.load_plpindel(gu.plpindel_fixtures()[0])
    ref = fx["genome"].encode()
    rng = np.random.default_rng(11)
    host = util.random_batch(rng, 600, 50, 1500)

    def run(c):
        tags = la.baq_batch(c, reads, ref, extended=True, idaq=True)
        sq, sqb = la.source_qual_batch(c, reads, ref)
        cols, _ = la.pileup_indel_columns(c, reads, ref, 0, len(ref))
        dt = la.pileup_snv_tracks(c, reads, ref, 0, len(ref), lb=[t[0] for t in tags], sq=sqb)
        recs, _, _ = c.call_snvs(dt, la.VarcallConf())
        irecs, _ = la.call_indels(c, cols, la.VarcallConf())
        b = util.to_pileup_batch(la, host)
        conf_a, conf_b = la.VarcallConf(), la.VarcallConf()
        c.call_snvs_submit(b, conf_a)
        ra, _ = c.call_snvs_collect(conf_a)
        c.call_snvs_submit(b, conf_b)
        rb, _ = c.call_snvs_collect(conf_b)
        return (b"".join(t[0].tobytes() for t in tags), sq.tobytes(), recs.tobytes(), irecs["qual"].tobytes(), ra.tobytes(), rb.tobytes())

    shared, private = la.SnvCaller(0), la.SnvCaller(0)
    try:
        private.set_private_stream(True)
        want = run(shared)
        assert run(private) == want
        private.set_private_stream(False)
        assert run(private) == want
        private.set_private_stream(True)
        assert run(private) == want
    finally:
        shared.close()
        private.close()
