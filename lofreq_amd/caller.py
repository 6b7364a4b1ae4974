"""Host-side mirror of LoFreq's SNV calling interface for the MI355X path.

Names follow the reference: :class:`VarcallConf` is ``varcall_conf_t`` (snpcaller.h:38-63,
defaults snpcaller.c:627-651), :class:`PileupBatch` is a batch of ``plp_col_t`` columns packed as
byte tracks (plp.h:88-91), :meth:`SnvCaller.call_snvs` is the ``call_snvs()`` loop
(lofreq_call.c:735-879) and :func:`filter_records` the final ``lofreq filter`` step that
``lofreq call`` runs on its own output (lofreq_call.c:1506-1538).

All compute goes through the C ABI in ``liblofreq_amd.so`` (hand-written HIP kernels); PyTorch is
only used to own device memory for HBM-resident batches.
"""
import ctypes as C

import numpy as np

from . import _lib

NT4 = b"ACGTN"


class VarcallConf:
    """varcall_conf_t for the SNV path.  ``bonf_subst`` / ``num_snv_tests`` are mutated by calls."""

    def __init__(self, **overrides):
        self.c = _lib.Conf()
        _lib.load().lfq_conf_init(C.byref(self.c))
        for k, v in overrides.items():
            if not hasattr(self.c, k):
                raise AttributeError(k)
            setattr(self.c, k, v)

    def __getattr__(self, k):
        return getattr(self.__dict__["c"], k)

    def copy(self):
        o = VarcallConf()
        C.memmove(C.byref(o.c), C.byref(self.c), C.sizeof(_lib.Conf))
        return o


class PileupBatch:
    """A batch of pileup columns as packed byte tracks (struct-of-arrays + CSR offsets).

    Host batches hold numpy arrays; device batches hold torch uint8/int64 tensors resident in HBM.
    """

    def __init__(self, nt, bq, mq, col_off, ref_base, baq=None, sq=None, coverage_plp=None, num_bases=None,
                 on_device=False, max_col_obs=0, nt_packed=False):
        self.nt, self.bq, self.baq, self.mq, self.sq = nt, bq, baq, mq, sq
        self.col_off, self.ref_base = col_off, ref_base
        self.coverage_plp, self.num_bases = coverage_plp, num_bases
        self.on_device = on_device
        self.ncols = int(len(col_off) - 1)
        self.max_col_obs = int(max_col_obs)
        self.nt_packed = bool(nt_packed)         # LFQ_TRACKS_NT_PACKED

    @staticmethod
    def from_columns(columns, ref_bases, coverage_plp=None, num_bases=None):
        """columns: list of dicts with per-observation arrays nt (0..4), strand (0/1), bq, baq, mq[, sq];
        -1 in baq/sq means missing (encoded 255)."""
        nts, bqs, baqs, mqs, sqs, off = [], [], [], [], [], [0]
        has_baq = any(c.get("baq") is not None for c in columns)
        has_sq = any(c.get("sq") is not None for c in columns)
        for c in columns:
            n = len(c["nt"])
            strand = np.asarray(c.get("strand", np.zeros(n, np.int64)))
            nts.append((np.asarray(c["nt"]).astype(np.uint8) & 7) | (strand.astype(np.uint8) << 3))
            bqs.append(np.asarray(c["bq"]).astype(np.uint8))
            mqs.append(np.asarray(c["mq"]).astype(np.uint8))
            if has_baq:
                b = np.asarray(c["baq"] if c.get("baq") is not None else np.full(n, -1))
                baqs.append(np.where(b < 0, 255, np.minimum(b, 254)).astype(np.uint8))   # clamp like the C shim
            if has_sq:
                s = np.asarray(c["sq"] if c.get("sq") is not None else np.full(n, -1))
                sqs.append(np.where(s < 0, 255, np.minimum(s, 254)).astype(np.uint8))
            off.append(off[-1] + n)
        cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.uint8)
        rb = np.frombuffer(bytes(ref_bases), dtype=np.uint8) if isinstance(ref_bases, (bytes, bytearray, str)) \
            else np.asarray(ref_bases, dtype=np.uint8)
        if isinstance(ref_bases, str):
            rb = np.frombuffer(ref_bases.encode(), dtype=np.uint8)
        return PileupBatch(cat(nts), cat(bqs), cat(mqs), np.asarray(off, np.uint64), rb.copy(),
                           baq=cat(baqs) if has_baq else None, sq=cat(sqs) if has_sq else None,
                           coverage_plp=None if coverage_plp is None else np.asarray(coverage_plp, np.int32),
                           num_bases=None if num_bases is None else np.asarray(num_bases, np.int32))

    def packed(self):
        """the same host batch with its nt track in the LFQ_TRACKS_NT_PACKED layout (lfq_pack_nt_track): what a producer
        that packs while it fills sends instead -- half the nt bytes over PCIe, the 1.5-bytes-per-observation count kernel"""
        assert not self.on_device and not self.nt_packed
        nt = np.ascontiguousarray(self.nt, np.uint8)
        n = int(self.col_off[-1])
        out = np.zeros((n + 7) // 8 * 4 + 16, np.uint8)
        _lib.check(_lib.load().lfq_pack_nt_track(C.c_void_p(nt.ctypes.data), n, C.c_void_p(out.ctypes.data)), "lfq_pack_nt_track")
        b = PileupBatch(out, self.bq, self.mq, self.col_off, self.ref_base, baq=self.baq, sq=self.sq,
                        coverage_plp=self.coverage_plp, num_bases=self.num_bases, max_col_obs=self.max_col_obs, nt_packed=True)
        return b

    def _ptr(self, a):
        if a is None:
            return None
        if self.on_device:
            return C.c_void_p(a.data_ptr())
        return C.c_void_p(a.ctypes.data)

    def _tracks(self):
        if not self.on_device:
            for name in ("nt", "bq", "baq", "mq", "sq", "ref_base"):
                a = getattr(self, name)
                if a is not None:
                    setattr(self, name, np.ascontiguousarray(a, dtype=np.uint8))
            self.col_off = np.ascontiguousarray(self.col_off, dtype=np.uint64)
        t = _lib.Tracks()
        t.nt, t.bq, t.baq, t.mq, t.sq = (self._ptr(self.nt), self._ptr(self.bq), self._ptr(self.baq),
                                         self._ptr(self.mq), self._ptr(self.sq))
        t.col_off, t.ref_base = self._ptr(self.col_off), self._ptr(self.ref_base)
        t.coverage_plp, t.num_bases = self._ptr(self.coverage_plp), self._ptr(self.num_bases)
        t.ncols = self.ncols
        t.max_col_obs = self.max_col_obs
        t.flags = _lib.LFQ_TRACKS_NT_PACKED if self.nt_packed else 0
        return t


class SnvCaller:
    """One context per GPU (lfq_ctx).  Not thread-safe, like the reference's caller."""

    def __init__(self, device=0):
        self.L = _lib.load()
        h = C.c_void_p()
        _lib.check(self.L.lfq_create(C.byref(h), int(device)), "lfq_create")
        self.h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "h", None):
            for rs in list(getattr(self, "_readsets", ())):     # a read set is destroyed before its context
                rs.close()
            self.L.lfq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- layer 2: the call_snvs loop over one batch ------------------------------------------
    def call_snvs(self, batch, conf, want_counts=False, records_capacity=None):
        """-> (records[SNV_RECORD_DTYPE] in column order, counts or None, BatchStats).
        Mutates conf.bonf_subst / conf.num_snv_tests like the reference's per-column loop."""
        t = batch._tracks()
        cap = int(records_capacity if records_capacity is not None else max(3 * batch.ncols, 16))
        rec = np.empty(cap, dtype=_lib.SNV_RECORD_DTYPE)      # the library zero-fills what it writes
        n = C.c_int64(0)
        counts = np.zeros(batch.ncols, dtype=_lib.COL_COUNTS_DTYPE) if want_counts else None
        st = _lib.BatchStats()
        rc = self.L.lfq_call_snvs_batch(self.h, C.byref(conf.c), C.byref(t), 1 if batch.on_device else 0,
                                        C.c_void_p(rec.ctypes.data), cap, C.byref(n),
                                        C.c_void_p(counts.ctypes.data) if want_counts else None, C.byref(st))
        _lib.check(rc, "lfq_call_snvs_batch")
        return rec[: n.value].copy(), counts, st

    def set_dense_counts(self, on):
        """lfq_set_dense_counts: off = snv_batch_device may leave the dense entries of untested columns unwritten"""
        _lib.check(self.L.lfq_set_dense_counts(self.h, 1 if on else 0), "lfq_set_dense_counts")

    def baq_times(self):
        """lfq_last_baq_times: the BAQ kernels of this context's last read-set BAQ call -> dict (ms_kernels, n_launches, n_reads, n_bases)"""
        t = _lib.BaqTimes()
        _lib.check(self.L.lfq_last_baq_times(self.h, C.byref(t)), "lfq_last_baq_times")
        return {"ms_kernels": float(t.ms_kernels), "n_launches": int(t.n_launches), "n_reads": int(t.n_reads), "n_bases": int(t.n_bases)}

    GATES = {"tail": 0, "end": 1, "none": 2}       # LFQ_GATE_TAIL / _END / _NONE (include/lofreq_amd.h)

    def set_batch_gate(self, gate):
        """lfq_set_batch_gate: what this context's next count kernel waits for when the device's previous batch still runs"""
        _lib.check(self.L.lfq_set_batch_gate(self.h, self.GATES[gate]), "lfq_set_batch_gate")

    def set_private_stream(self, on):
        """lfq_set_private_stream: a launch stream of this context's own (one context per host thread) instead of the device's shared one"""
        _lib.check(self.L.lfq_set_private_stream(self.h, 1 if on else 0), "lfq_set_private_stream")

    def set_dense_strand_counts(self, on):
        """lfq_set_dense_strand_counts: off = strand counts only for the columns of the sparse output (layer 1, submit)"""
        _lib.check(self.L.lfq_set_dense_strand_counts(self.h, 1 if on else 0), "lfq_set_dense_strand_counts")

    def set_pileup_nt_packed(self, on):
        """lfq_set_pileup_nt_packed: layout of the nt track the device pileup hands out (default: packed nibbles)"""
        _lib.check(self.L.lfq_set_pileup_nt_packed(self.h, 1 if on else 0), "lfq_set_pileup_nt_packed")

    def set_pileup_unsorted(self, on):
        """lfq_set_pileup_unsorted: take reads that are not position-sorted (read-major kernels, no fixed order within a
        column) instead of refusing them like mpileup does"""
        _lib.check(self.L.lfq_set_pileup_unsorted(self.h, 1 if on else 0), "lfq_set_pileup_unsorted")

    def set_baq_hmm_params(self, gap_open=1e-5, gap_ext=0.4):
        """lfq_set_baq_hmm_params: kpa_ext_par_t.d / .e of the BAQ HMM (defaults: kpa_ext_par_lofreq_illumina,
        kprobaln_ext.c:50; a -DPACBIO_REALN build of the reference uses 0.1 / 0.4, :51)"""
        _lib.check(self.L.lfq_set_baq_hmm_params(self.h, float(gap_open), float(gap_ext)), "lfq_set_baq_hmm_params")

    def set_indel_arrays_on_host(self, on):
        """lfq_set_indel_arrays_on_host: off = the quality arrays of the indel columns stay on the device only"""
        _lib.check(self.L.lfq_set_indel_arrays_on_host(self.h, 1 if on else 0), "lfq_set_indel_arrays_on_host")

    def call_snvs_submit(self, batch, conf):
        """first half of call_snvs: launch the kernels of the batch and return (one batch in flight per context)"""
        t = batch._tracks()
        self._sub = (batch, t)                   # keep the arrays alive until collect
        _lib.check(self.L.lfq_call_snvs_submit(self.h, C.byref(conf.c), C.byref(t), 1 if batch.on_device else 0),
                   "lfq_call_snvs_submit")

    def call_snvs_wait(self):
        """block until the kernels of the submitted batch are done (lfq_call_snvs_wait)"""
        _lib.check(self.L.lfq_call_snvs_wait(self.h), "lfq_call_snvs_wait")

    def call_snvs_collect(self, conf, records_capacity=1 << 16):
        """second half: wait, finish on the host -> (records, BatchStats); mutates conf like call_snvs"""
        cap = int(records_capacity)
        rec = np.empty(cap, dtype=_lib.SNV_RECORD_DTYPE)
        n = C.c_int64(0)
        st = _lib.BatchStats()
        rc = self.L.lfq_call_snvs_collect(self.h, C.byref(conf.c), C.c_void_p(rec.ctypes.data), cap, C.byref(n), None,
                                          C.byref(st))
        self._sub = None
        _lib.check(rc, "lfq_call_snvs_collect")
        return rec[: n.value].copy(), st

    def uniq_detlim(self, batch, af):
        """`lofreq uniq --use-det-lim` (uniq_snv, lofreq_uniq.c:274-333) over a batch of columns: af[col] = the
        variant's allele frequency -> (detectable uint8 per column: the UNIQ condition, p-values longdouble)"""
        t = batch._tracks()
        af = np.ascontiguousarray(af, np.float32)
        assert len(af) == batch.ncols
        det = np.zeros(max(batch.ncols, 1), np.uint8)
        pv = np.zeros(max(batch.ncols, 1), np.longdouble)
        rc = self.L.lfq_uniq_detlim_batch(self.h, C.byref(t), 1 if batch.on_device else 0, C.c_void_p(af.ctypes.data),
                                          C.c_void_p(det.ctypes.data), C.c_void_p(pv.ctypes.data))
        _lib.check(rc, "lfq_uniq_detlim_batch")
        return det[: batch.ncols], pv[: batch.ncols]

    def uniq_binom(self, batch, af, alt_bases):
        """`lofreq uniq` default mode (uniq_snv's binomial branch, lofreq_uniq.c:335-393) over a batch of columns:
        af[col] / alt_bases[col] = the variant's allele frequency and alt nucleotide -> (UQ phred values int32, -1 = no
        UQ tag; p-values float64)"""
        t = batch._tracks()
        af = np.ascontiguousarray(af, np.float32)
        alt = np.frombuffer(alt_bases.encode() if isinstance(alt_bases, str) else bytes(alt_bases), np.uint8).copy()
        assert len(af) == batch.ncols == len(alt)
        uq = np.zeros(max(batch.ncols, 1), np.int32)
        pv = np.zeros(max(batch.ncols, 1), np.float64)
        rc = self.L.lfq_uniq_binom_batch(self.h, C.byref(t), 1 if batch.on_device else 0, C.c_void_p(af.ctypes.data),
                                         C.c_void_p(alt.ctypes.data), C.c_void_p(uq.ctypes.data), C.c_void_p(pv.ctypes.data))
        _lib.check(rc, "lfq_uniq_binom_batch")
        return uq[: batch.ncols], pv[: batch.ncols]

    # -- layer 1: kernels only, device-resident in and out --------------------------------------
    def snv_batch_device(self, batch, conf, d_counts, d_pvals, pvals_capacity, stream=None):
        assert batch.on_device
        t = batch._tracks()
        rc = self.L.lfq_snv_batch_device(self.h, C.byref(conf.c), C.byref(t), C.c_void_p(d_counts.data_ptr()),
                                         C.c_void_p(d_pvals.data_ptr()), int(pvals_capacity),
                                         C.c_void_p(stream) if stream else None)
        _lib.check(rc, "lfq_snv_batch_device")

    def batch_finish(self):
        st = _lib.BatchStats()
        _lib.check(self.L.lfq_batch_finish(self.h, C.byref(st)), "lfq_batch_finish")
        return st

    def kernel_times(self):
        kt = _lib.KernelTimes()
        _lib.check(self.L.lfq_last_kernel_times(self.h, C.byref(kt)))
        return {k: getattr(kt, k) for k, _ in kt._fields_}

    def dp_work(self):
        """lfq_last_dp_work: DP cells / rows the kernels of the last batch processed, class sizes, layout bytes"""
        w = _lib.DpWork()
        _lib.check(self.L.lfq_last_dp_work(self.h, C.byref(w)))
        return {k: int(getattr(w, k)) for k, _ in w._fields_}

    def synchronize(self):
        _lib.check(self.L.lfq_synchronize(self.h))

    # -- synthetic workload, generated directly in HBM -----------------------------------------
    def synth_batch(self, seed, depth, ncols, plant_period=997, col_begin=0, nt_packed=True):
        """the synthetic workload of include/lofreq_synth.h, generated in HBM; nt_packed: two observations per byte
        in the nt track (LFQ_TRACKS_NT_PACKED), the layout device-resident producers should use"""
        import torch
        dev = torch.device("cuda", self.device)
        n = ncols * depth
        pad = (n + 31) // 32 * 32 + 32
        nt = torch.empty(pad // 2 if nt_packed else pad, dtype=torch.uint8, device=dev)
        bq = torch.empty(pad, dtype=torch.uint8, device=dev)
        baq = torch.empty(pad, dtype=torch.uint8, device=dev)
        mq = torch.empty(pad, dtype=torch.uint8, device=dev)
        off = torch.empty(ncols + 1, dtype=torch.int64, device=dev)
        ref = torch.empty(ncols + 16, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        rc = self.L.lfq_synth_fill_device_layout(self.h, int(seed), int(depth), int(plant_period), int(col_begin),
                                                 int(ncols), C.c_void_p(nt.data_ptr()), C.c_void_p(bq.data_ptr()),
                                                 C.c_void_p(baq.data_ptr()), C.c_void_p(mq.data_ptr()),
                                                 C.c_void_p(off.data_ptr()), C.c_void_p(ref.data_ptr()),
                                                 1 if nt_packed else 0, None)
        _lib.check(rc, "lfq_synth_fill_device_layout")
        self.synchronize()
        b = PileupBatch(nt, bq, mq, off, ref, baq=baq, on_device=True, max_col_obs=depth, nt_packed=nt_packed)
        b.ncols = ncols
        return b


def pvalue_from_log(logp, status):
    """expl() + the reference's clamp (snpcaller.c:1047-1059), as np.longdouble."""
    pv = np.zeros(1, dtype=_lib.COL_PVALS_DTYPE)
    pv["logp"][0, 0] = logp
    pv["status"][0, 0] = status
    pv["counts"]["alt_counts"][0, 0] = 1
    pv["counts"]["coverage"][0] = 1
    pv["bonf"][0] = 0  # emit test passes for any finite p: p*0 < sig
    conf = VarcallConf(sig=1.0)
    rec = np.zeros(4, dtype=_lib.SNV_RECORD_DTYPE)
    n = C.c_int64(0)
    ref = np.frombuffer(b"A", dtype=np.uint8)
    _lib.check(_lib.load().lfq_finalize_pvals(C.byref(conf.c), C.c_void_p(pv.ctypes.data), 1, None,
                                              C.c_void_p(ref.ctypes.data) if ref is not None else None,
                                              C.c_void_p(rec.ctypes.data), 4,
                                              C.byref(n)))
    if n.value == 0:
        return np.finfo(np.longdouble).max
    return rec["pvalue"][0]


def finalize_pvals(conf, pvals, ref_base=None, coverage_plp=None):
    """Host finishing step on sparse device records -> reported SNVs (column order).  `ref_base` may be None:
    every record carries its column's reference base."""
    pvals = np.ascontiguousarray(pvals, dtype=_lib.COL_PVALS_DTYPE)
    rec = np.zeros(max(3 * len(pvals), 4), dtype=_lib.SNV_RECORD_DTYPE)
    n = C.c_int64(0)
    ref = None if ref_base is None else np.ascontiguousarray(ref_base, dtype=np.uint8)
    cov = None if coverage_plp is None else np.ascontiguousarray(coverage_plp, dtype=np.int32)
    _lib.check(_lib.load().lfq_finalize_pvals(C.byref(conf.c), C.c_void_p(pvals.ctypes.data), len(pvals),
                                              C.c_void_p(cov.ctypes.data) if cov is not None else None,
                                              C.c_void_p(ref.ctypes.data) if ref is not None else None,
                                              C.c_void_p(rec.ctypes.data),
                                              len(rec), C.byref(n)))
    return rec[: n.value].copy()


def format_vcf_record(rec, chrom, pos0, filter_str=None):
    """One VCF line, byte-identical to vcf_write_var (vcf.c:469-497) at HEAD (with ;HQA=)."""
    buf = C.create_string_buffer(512)
    r = np.ascontiguousarray(np.asarray(rec).reshape(1), dtype=_lib.SNV_RECORD_DTYPE)
    n = _lib.load().lfq_format_snv_record(buf, 512, chrom.encode(), int(pos0), C.c_void_p(r.ctypes.data),
                                          filter_str.encode() if filter_str else None)
    return buf.raw[:n].decode()


def format_vcf(records, chrom, pos0=None, keep=None, filter_str=None):
    """All (kept) records of a batch as VCF text in one C call (vcf_write_var, vcf.c:469-497)."""
    r = np.ascontiguousarray(records, dtype=_lib.SNV_RECORD_DTYPE)
    if len(r) == 0:
        return ""
    p = None if pos0 is None else np.ascontiguousarray(pos0, dtype=np.int64)
    k = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
    args = (chrom.encode(), C.c_void_p(p.ctypes.data) if p is not None else None, C.c_void_p(r.ctypes.data), len(r),
            C.c_void_p(k.ctypes.data) if k is not None else None, filter_str.encode() if filter_str else None)
    L = _lib.load()
    buf = C.create_string_buffer((len(chrom) + 128) * len(r) + 64)     # a line is about len(chrom) + 95 bytes
    n = L.lfq_format_vcf(buf, len(buf), *args)
    if n > len(buf):                             # the C call returns the size it needs: allocate that and repeat
        buf = C.create_string_buffer(int(n) + 1)
        n = L.lfq_format_vcf(buf, len(buf), *args)
    if n < 0 or n > len(buf):
        raise RuntimeError("lfq_format_vcf failed (%d)" % n)
    return buf.raw[:n].decode()


def binom_cdf(n, k, pr):
    """lfq_binom_cdf: (P(X <= k), X ~ Binomial(n, pr); cdfbin's status code)"""
    st = C.c_int(0)
    p = _lib.load().lfq_binom_cdf(int(n), int(k), float(pr), C.byref(st))
    return p, st.value


def uniq_mtc(uq, mtc_type="fdr", alpha=0.001, ntests=0):
    """apply_uniq_filter_mtc (lofreq_uniq.c:140-206) on UQ phred values -> boolean PASS mask"""
    uq = np.ascontiguousarray(uq, np.int32)
    out = np.zeros(max(len(uq), 1), np.uint8)
    _lib.check(_lib.load().lfq_uniq_mtc(C.c_void_p(uq.ctypes.data), len(uq), {"bonf": 1, "holm": 2, "fdr": 3}[mtc_type],
                                        float(alpha), int(ntests), C.c_void_p(out.ctypes.data)), "lfq_uniq_mtc")
    return out[: len(uq)].astype(bool)


def snvqual_thresh(sig, bonf_subst):
    return _lib.load().lfq_snvqual_thresh(C.c_float(sig), int(bonf_subst))


def filter_records(records, snvqual_threshold, apply_defaults=True):
    """`lofreq filter` as run by `lofreq call`: boolean keep mask."""
    r = np.ascontiguousarray(records, dtype=_lib.SNV_RECORD_DTYPE)
    keep = np.zeros(max(len(r), 1), dtype=np.uint8)
    _lib.check(_lib.load().lfq_filter_records(C.c_void_p(r.ctypes.data), len(r), int(snvqual_threshold),
                                              1 if apply_defaults else 0, C.c_void_p(keep.ctypes.data)))
    return keep[: len(r)].astype(bool)


def write_vcf_header(source="lofreq_amd call", reference=None):
    """vcf_write_new_header (vcf.c:649-676) minus the volatile ##fileDate line."""
    lines = ["##fileformat=VCFv4.0"]
    if source:
        lines.append("##source=%s" % source)
    if reference:
        lines.append("##reference=%s" % reference)
    lines += [
        '##INFO=<ID=DP,Number=1,Type=Integer,Description="Raw Depth">',
        '##INFO=<ID=AF,Number=1,Type=Float,Description="Allele Frequency">',
        '##INFO=<ID=SB,Number=1,Type=Integer,Description="Phred-scaled strand bias at this position">',
        '##INFO=<ID=DP4,Number=4,Type=Integer,Description="Counts for ref-forward bases, ref-reverse, '
        'alt-forward and alt-reverse bases">',
        '##INFO=<ID=HQA,Number=1,Type=Integer,Description="Count of high quality alt bases supporting SNP call">',
        '##INFO=<ID=INDEL,Number=0,Type=Flag,Description="Indicates that the variant is an INDEL.">',
        '##INFO=<ID=CONSVAR,Number=0,Type=Flag,Description="Indicates that the variant is a consensus variant '
        '(as opposed to a low frequency variant).">',
        '##INFO=<ID=HRUN,Number=1,Type=Integer,Description="Homopolymer length to the right of report indel '
        'position">',
        "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO",
    ]
    return "\n".join(lines) + "\n"
