import sys, os, time
root=os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root,"oracle"))
import numpy as np, ctypes as C
import lofreq_amd as la
from lofreq_amd import _lib
import pyoracle as orc
n=int(sys.argv[1]) if len(sys.argv)>1 else 400000
rl=150
rng=np.random.default_rng(1)
glen=2_000_000
genome=rng.integers(0,4,glen).astype(np.uint8)
gen_ascii=np.frombuffer(b"ACGT",np.uint8)[genome].tobytes()
pos=np.sort(rng.integers(0,glen-rl-10,n)).astype(np.int32)
seq=genome[(pos[:,None]+np.arange(rl)[None,:])].astype(np.uint8)
mism=rng.random(seq.shape)<0.005
seq[mism]=(seq[mism]+1)%4
qual=np.clip(np.round(rng.normal(33,6,seq.shape)),2,41).astype(np.uint8)
cig=np.full(n,(rl<<4)|0,np.uint32)
cig_off=np.arange(n+1,dtype=np.int64); seq_off=np.arange(n+1,dtype=np.int64)*rl
out=np.zeros(n*rl,np.uint8)
caller=la.SnvCaller(0)
rd=_lib.BaqReads(); rd.n_reads=n; rd.pos=pos.ctypes.data; rd.cigar_off=cig_off.ctypes.data; rd.cigar=cig.ctypes.data
rd.seq_off=seq_off.ctypes.data; seqf=np.ascontiguousarray(seq.reshape(-1)); qualf=np.ascontiguousarray(qual.reshape(-1))
rd.seq=seqf.ctypes.data; rd.qual=qualf.ctypes.data; rd.ref=C.cast(C.c_char_p(gen_ascii),C.c_void_p); rd.ref_len=glen
L=_lib.load()
for it in range(3):
    t0=time.perf_counter(); rc=L.lfq_baq_batch(caller.h, C.byref(rd), 1, out.ctypes.data); dt=time.perf_counter()-t0
    print("gpu: %d reads x %d bp in %.3f s -> %.2f M reads/s (host buffers in and out, allocations included), rc %d"%(n,rl,dt,n/dt/1e6,rc))
# cpu oracle on a sample
m=2000
t0=time.perf_counter()
for i in range(m):
    o=orc.baq_read(int(pos[i]), [("M",rl)], seq[i], qual[i], gen_ascii, True)
dt=time.perf_counter()-t0
print("cpu oracle (1 thread, via ctypes): %d reads in %.2f s -> %.0f reads/s"%(m,dt,m/dt))
assert o.tobytes()==out[(m-1)*rl:m*rl].tobytes()
print("last sample read identical to the GPU result")

# ---- device-side pileup of the same reads (lb from the GPU BAQ), then the call, tracks resident in HBM
lb = out
mapq = np.full(n, 60, np.uint8); rev = (rng.random(n) < 0.5).astype(np.uint8)
pr = _lib.PileupReads(); pr.n_reads = n; pr.pos = pos.ctypes.data; pr.cigar_off = cig_off.ctypes.data; pr.cigar = cig.ctypes.data
pr.seq_off = seq_off.ctypes.data; pr.seq = seqf.ctypes.data; pr.qual = qualf.ctypes.data; pr.baq = lb.ctypes.data
pr.mapq = mapq.ctypes.data; pr.reverse = rev.ctypes.data; pr.ref = C.cast(C.c_char_p(gen_ascii), C.c_void_p); pr.ref_len = glen
t = _lib.Tracks(); col_pos = np.zeros(glen, np.int64)
for it in range(3):
    t0 = time.perf_counter(); rc = L.lfq_pileup_snv_tracks(caller.h, C.byref(pr), 0, glen, 3, C.byref(t), col_pos.ctypes.data); dt = time.perf_counter() - t0
    print("pileup: %d reads -> %d columns, %.1f M observations in %.3f s (%.1f M reads/s, host buffers in), rc %d" % (n, t.ncols, n * rl / 1e6, dt, n / dt / 1e6, rc))
from lofreq_amd.pileup import DeviceTracks
dtk = DeviceTracks(t, col_pos[: t.ncols].copy())
conf = la.VarcallConf()
t0 = time.perf_counter(); recs, _, st = caller.call_snvs(dtk, conf, records_capacity=1 << 16); dt = time.perf_counter() - t0
print("call on the resident tracks: %d columns, %d tested, %d records in %.4f s" % (dtk.ncols, st.n_tested, len(recs), dt))

# ---- source quality of the same reads (lofreq call -s), then the indel fields of the pileup
sq = np.zeros(n, np.int32); sqb = np.zeros(n, np.uint8)
for it in range(3):
    t0 = time.perf_counter(); rc = L.lfq_source_qual_batch(caller.h, C.byref(rd), -1, 6, None, sq.ctypes.data, sqb.ctypes.data); dt = time.perf_counter() - t0
    print("source quality: %d reads in %.3f s (%.1f M reads/s, host buffers in and out), rc %d; %d reads ran the DP" % (n, dt, n / dt / 1e6, rc, int(((sq >= 0) & (sq < 49314)).sum())))
m = 3000
t0 = time.perf_counter()
for i in range(m):
    o = orc.source_qual(int(pos[i]), [("M", rl)], seq[i], qual[i], gen_ascii)
    assert o == sq[i], (i, o, sq[i])
dt = time.perf_counter() - t0
print("cpu oracle source quality (1 thread, via ctypes): %d reads in %.2f s -> %.0f reads/s, identical" % (m, dt, m / dt))
bi = rng.integers(33 + 25, 33 + 50, n * rl).astype(np.uint8); bd = rng.integers(33 + 25, 33 + 50, n * rl).astype(np.uint8)
tg = _lib.PileupIndelTags(); tg.bi = bi.ctypes.data; tg.bd = bd.ctypes.data
outp = C.POINTER(_lib.IndelColumnsC)()
for it in range(3):
    t0 = time.perf_counter(); rc = L.lfq_pileup_indel_columns(caller.h, C.byref(pr), C.byref(tg), 0, glen, 0, C.byref(outp), col_pos.ctypes.data); dt = time.perf_counter() - t0
    print("indel fields of the pileup: %d reads -> %d columns in %.3f s (%.1f M reads/s), rc %d" % (n, outp.contents.ncols, dt, n / dt / 1e6, rc))
