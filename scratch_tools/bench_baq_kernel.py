import sys, os, time
root=os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, root)
import numpy as np, ctypes as C
from lofreq_amd import _lib
if len(sys.argv) > 2:
    _lib.LIB_PATH = sys.argv[2]
import lofreq_amd as la
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
for it in range(4):
    t0=time.perf_counter(); rc=L.lfq_baq_batch(caller.h, C.byref(rd), 1, out.ctypes.data); dt=time.perf_counter()-t0
    print("%s: %d reads in %.4f s"%(sys.argv[2] if len(sys.argv)>2 else "default", n, dt), flush=True)
