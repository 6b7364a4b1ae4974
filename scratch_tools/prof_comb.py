import sys, os, ctypes as C
root=os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, root)
import torch, numpy as np
from lofreq_amd import _lib
_lib.LIB_PATH=os.path.join(root,"scratch_tools","liblofreq_amd_prof.so")
import lofreq_amd as la
SEED = 0x9E3779B97F4A7C15 ^ (3 << 32)
dev=torch.device("cuda",0); caller=la.SnvCaller(0)
ncols=1000000; depth=10000
batch=caller.synth_batch(SEED, depth, ncols, plant_period=997)
d_counts=torch.zeros(ncols*64,dtype=torch.uint8,device=dev); d_pvals=torch.zeros(ncols*128,dtype=torch.uint8,device=dev)
torch.cuda.synchronize()
L=_lib.load(); L.lfq_debug_counters.argtypes=[C.c_void_p, C.POINTER(C.c_int32)]
os.environ["LFQ_DEBUG_SKIP"]=sys.argv[1] if len(sys.argv)>1 else "light,mid"
for it in range(3):
    conf=la.VarcallConf(); caller.snv_batch_device(batch, conf, d_counts, d_pvals, ncols); st=caller.batch_finish()
cnt=(C.c_int32*16)(); L.lfq_debug_counters(caller.h, cnt); c=list(cnt)
n=max(c[14],1)
print("records", c[14], "per record us: load_b %.1f conv %.1f scan %.1f tail+write %.1f | total wall %.1f us, cycles %.0f -> %.2f GHz" % (
    c[8]/n/100, c[9]/n/100, c[10]/n/100, c[11]/n/100, c[13]/n/100, c[12]*256/n, (c[12]*256/n)/(c[13]/n*10)))
print("emit %.1f us, whole record %.1f us" % (c[5]/n/100, c[6]/n/100)); print(caller.kernel_times())
