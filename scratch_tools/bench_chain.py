"""reads -> BAQ/IDAQ -> both pileups -> SNV + indel calls, everything through the C ABI, timed per stage.
usage: python scratch_tools/bench_chain.py [n_reads] [genome_len]"""
import sys, os, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, root)
import numpy as np, ctypes as C
import lofreq_amd as la
from lofreq_amd import _lib
from lofreq_amd.pileup import DeviceTracks

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
glen = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
rl = 150
rng = np.random.default_rng(3)
genome = rng.integers(0, 4, glen).astype(np.uint8)
gen_ascii = np.frombuffer(b"ACGT", np.uint8)[genome].tobytes()
pos = np.sort(rng.integers(0, glen - rl - 20, n)).astype(np.int32)
# every read: 150 bases; 4 % of the reads carry one 1-3 bp insertion or deletion in the middle third
has = rng.random(n) < 0.04
kind = rng.random(n) < 0.5
ilen = rng.integers(1, 4, n)
cut = rng.integers(50, 100, n)
base = genome[(pos[:, None] + np.arange(rl + 4)[None, :])]
seq = np.empty((n, rl), np.uint8)
cig = []
cig_off = np.zeros(n + 1, np.int64)
plain = ~has
seq[plain] = base[plain, :rl]
for i in np.nonzero(has)[0]:
    c, k = int(cut[i]), int(ilen[i])
    if kind[i]:      # insertion: c ref bases, k random bases, rest
        seq[i, :c] = base[i, :c]; seq[i, c:c + k] = rng.integers(0, 4, k); seq[i, c + k:] = base[i, c:rl - k]
    else:            # deletion
        seq[i, :c] = base[i, :c]; seq[i, c:] = base[i, c + k:rl + k]
cigs = np.zeros((n, 3), np.uint32); ncig = np.ones(n, np.int64)
cigs[:, 0] = (rl << 4)
ii = np.nonzero(has)[0]
cigs[ii, 0] = (cut[ii].astype(np.uint32) << 4)
cigs[ii, 1] = (ilen[ii].astype(np.uint32) << 4) | np.where(kind[ii], 1, 2).astype(np.uint32)
cigs[ii, 2] = ((rl - cut[ii] - np.where(kind[ii], ilen[ii], 0)).astype(np.uint32) << 4)
ncig[ii] = 3
cig_off[1:] = np.cumsum(ncig)
mask = np.arange(3)[None, :] < ncig[:, None]
cig = np.ascontiguousarray(cigs[mask])
mism = rng.random(seq.shape) < 0.003
seq[mism] = (seq[mism] + 1) % 4
qual = np.clip(np.round(rng.normal(34, 5, seq.shape)), 2, 41).astype(np.uint8)
seq_off = np.arange(n + 1, dtype=np.int64) * rl
seqf = np.ascontiguousarray(seq.reshape(-1)); qualf = np.ascontiguousarray(qual.reshape(-1))
bi = rng.integers(33 + 30, 33 + 50, n * rl).astype(np.uint8); bd = rng.integers(33 + 30, 33 + 50, n * rl).astype(np.uint8)
mapq = np.full(n, 60, np.uint8); rev = (rng.random(n) < 0.5).astype(np.uint8)
print("reads: %d x %d bp over %d bp (depth %.0f), %d with an indel" % (n, rl, glen, n * rl / glen, int(has.sum())), flush=True)

caller = la.SnvCaller(0)
L = _lib.load()
rd = _lib.BaqReads(); rd.n_reads = n; rd.pos = pos.ctypes.data; rd.cigar_off = cig_off.ctypes.data; rd.cigar = cig.ctypes.data
rd.seq_off = seq_off.ctypes.data; rd.seq = seqf.ctypes.data; rd.qual = qualf.ctypes.data
rd.ref = C.cast(C.c_char_p(gen_ascii), C.c_void_p); rd.ref_len = glen
lb = np.zeros(n * rl, np.uint8); ai = np.zeros(n * rl, np.uint8); ad = np.zeros(n * rl, np.uint8); fl = np.zeros(n, np.uint8)
pr = _lib.PileupReads(); pr.n_reads = n; pr.pos = pos.ctypes.data; pr.cigar_off = cig_off.ctypes.data; pr.cigar = cig.ctypes.data
pr.seq_off = seq_off.ctypes.data; pr.seq = seqf.ctypes.data; pr.qual = qualf.ctypes.data; pr.baq = lb.ctypes.data
pr.mapq = mapq.ctypes.data; pr.reverse = rev.ctypes.data; pr.ref = rd.ref; pr.ref_len = glen
tg = _lib.PileupIndelTags(); tg.bi = bi.ctypes.data; tg.bd = bd.ctypes.data; tg.ai = ai.ctypes.data; tg.ad = ad.ctypes.data
col_pos = np.zeros(glen, np.int64)
vp = C.c_void_p
L.lfq_set_indel_arrays_on_host(caller.h, 0)     # the quality arrays of the indel columns stay in HBM
for it in range(3):
    # ---- resident read set: one upload, everything else on the device copy
    T = [time.perf_counter()]
    h = vp()
    tg2 = _lib.PileupIndelTags(); tg2.bi = bi.ctypes.data; tg2.bd = bd.ctypes.data
    pr.baq = None
    assert L.lfq_readset_create(caller.h, C.byref(pr), C.byref(tg2), C.byref(h)) == 0
    T.append(time.perf_counter())
    assert L.lfq_readset_baq(caller.h, h, 1, 1) == 0
    T.append(time.perf_counter())
    outp = C.POINTER(_lib.IndelColumnsC)()
    assert L.lfq_readset_pileup_indels(caller.h, h, 0, glen, 0, C.byref(outp), col_pos.ctypes.data) == 0
    T.append(time.perf_counter())
    conf = la.VarcallConf(flag=la.LFQ_USE_BAQ | la.LFQ_USE_MQ | la.LFQ_USE_IDAQ)
    cap = 1 << 20
    rec = np.zeros(cap, dtype=_lib.INDEL_RECORD_DTYPE); nrec = C.c_int64(0); nt = C.c_int64(0)
    assert L.lfq_call_indels_batch(caller.h, C.byref(conf.c), outp, rec.ctypes.data, cap, C.byref(nrec), C.byref(nt)) == 0
    T.append(time.perf_counter())
    cons = np.frombuffer(C.string_at(outp.contents.cons_indel, outp.contents.ncols), np.uint8).copy()
    t = _lib.Tracks()
    assert L.lfq_readset_pileup_snv(caller.h, h, 0, glen, 3, C.byref(t), col_pos.ctypes.data) == 0
    assert L.lfq_pileup_skip_snv_columns(caller.h, cons.ctypes.data, len(cons)) == 0
    T.append(time.perf_counter())
    recs, _, st = caller.call_snvs(DeviceTracks(t, col_pos[: t.ncols]), conf, records_capacity=1 << 18)
    T.append(time.perf_counter())
    L.lfq_readset_destroy(h)
    d = [T[i + 1] - T[i] for i in range(6)]
    print("resident: upload %.3f s | BAQ+IDAQ %.3f s | indel pileup %.3f s | indel calls %.3f s (%d tests) | SNV pileup %.3f s | "
          "SNV calls %.3f s (%d tested columns, %d records) | total %.3f s = %.2f M reads/s"
          % (d[0], d[1], d[2], d[3], nt.value, d[4], d[5], st.n_tested, len(recs), sum(d), n / sum(d) / 1e6), flush=True)
pr.baq = lb.ctypes.data
L.lfq_set_indel_arrays_on_host(caller.h, 1)
for it in range(2):
    T = [time.perf_counter()]
    assert L.lfq_baq_idaq_batch(caller.h, C.byref(rd), 1, lb.ctypes.data, ai.ctypes.data, ad.ctypes.data, fl.ctypes.data) == 0
    T.append(time.perf_counter())
    flags = (fl << 2) | 3; flags = np.ascontiguousarray(flags, np.uint8); tg.tag_flags = flags.ctypes.data
    outp = C.POINTER(_lib.IndelColumnsC)()
    assert L.lfq_pileup_indel_columns(caller.h, C.byref(pr), C.byref(tg), 0, glen, 0, C.byref(outp), col_pos.ctypes.data) == 0
    T.append(time.perf_counter())
    conf = la.VarcallConf(flag=la.LFQ_USE_BAQ | la.LFQ_USE_MQ | la.LFQ_USE_IDAQ)
    cap = 1 << 20
    rec = np.zeros(cap, dtype=_lib.INDEL_RECORD_DTYPE); nrec = C.c_int64(0); nt = C.c_int64(0)
    assert L.lfq_call_indels_batch(caller.h, C.byref(conf.c), outp, rec.ctypes.data, cap, C.byref(nrec), C.byref(nt)) == 0
    T.append(time.perf_counter())
    cons = np.frombuffer(C.string_at(outp.contents.cons_indel, outp.contents.ncols), np.uint8).copy()
    t = _lib.Tracks()
    assert L.lfq_pileup_snv_tracks(caller.h, C.byref(pr), 0, glen, 3, C.byref(t), col_pos.ctypes.data) == 0
    assert L.lfq_pileup_skip_snv_columns(caller.h, cons.ctypes.data, len(cons)) == 0
    T.append(time.perf_counter())
    dtk = DeviceTracks(t, col_pos[: t.ncols])
    recs, _, st = caller.call_snvs(dtk, conf, records_capacity=1 << 18)
    T.append(time.perf_counter())
    d = [T[i + 1] - T[i] for i in range(5)]
    print("BAQ+IDAQ %.3f s | indel pileup %.3f s | indel calls %.3f s (%d tests, %d records) | SNV pileup %.3f s | "
          "SNV calls %.3f s (%d tested columns, %d records) | total %.3f s = %.2f M reads/s, %.2f M columns/s"
          % (d[0], d[1], d[2], nt.value, nrec.value, d[3], d[4], st.n_tested, len(recs), sum(d), n / sum(d) / 1e6,
             t.ncols / sum(d) / 1e6), flush=True)
