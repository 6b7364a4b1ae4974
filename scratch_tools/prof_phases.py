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
for it in range(2):
    conf=la.VarcallConf(); caller.snv_batch_device(batch, conf, d_counts, d_pvals, ncols); st=caller.batch_finish()
cnt=(C.c_int32*16)(); L.lfq_debug_counters(caller.h, cnt); c=list(cnt)
print("counters", c); print("stage %.1f Mticks, load-issue %.1f, rows %.1f, n_rows %d -> %.1f ticks/row; stage ticks/chunk ~ %.0f"%(c[8]*256/1e6,c[9]*256/1e6,c[10]*256/1e6,c[11], c[10]*256/max(c[11],1), c[8]*256/max(c[11]/64,1)))
print(caller.kernel_times())
