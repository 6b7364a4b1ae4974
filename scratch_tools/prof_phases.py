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
for skip in ["mid,big", "light,big", "light,mid", ""]:
    os.environ["LFQ_DEBUG_SKIP"]=skip
    for it in range(3):
        conf=la.VarcallConf(); caller.snv_batch_device(batch, conf, d_counts, d_pvals, ncols); st=caller.batch_finish()
    cnt=(C.c_int32*16)(); L.lfq_debug_counters(caller.h, cnt); c=list(cnt)
    print("skip=%-10s"%skip, "counters", c)
    print("   stage %.1f Mticks, load-issue %.1f, rows %.1f, n_rows %d -> %.1f ticks/row"%(c[8]*256/1e6,c[9]*256/1e6,c[10]*256/1e6,c[11], c[10]*256/max(c[11],1)))
    print("  ", {k:round(v,3) for k,v in caller.kernel_times().items()})
cnt=d_counts.cpu().numpy().view(la.COL_COUNTS_DTYPE)
k=cnt["kmax"]
for lo,hi in [(1,2),(2,4),(4,8),(8,16),(16,64),(64,250),(250,100000)]:
    print("K[%d,%d): %d"%(lo,hi,((k>=lo)&(k<hi)&(cnt["tested"]==1)).sum()))
