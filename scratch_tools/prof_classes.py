import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import lofreq_amd as la
SEED = 0x9E3779B97F4A7C15 ^ (3 << 32)
dev=torch.device("cuda",0)
caller=la.SnvCaller(0)
for ncols,pp in [(4096,1),(1024,1),(256,1)]:
    depth=10000
    batch=caller.synth_batch(SEED, depth, ncols, plant_period=pp)
    d_counts=torch.zeros(ncols*64,dtype=torch.uint8,device=dev); d_pvals=torch.zeros(ncols*128,dtype=torch.uint8,device=dev)
    torch.cuda.synchronize()
    for it in range(2):
        conf=la.VarcallConf()
        caller.snv_batch_device(batch, conf, d_counts, d_pvals, ncols)
        st=caller.batch_finish()
    cnt=d_counts.cpu().numpy().view(la.COL_COUNTS_DTYPE)
    k=cnt["kmax"]
    print(ncols, "classes: light %d mid %d big %d"%(((k>0)&(k<64)).sum(), ((k>=64)&(k<505)).sum(), (k>=505).sum()), "n_pvals",st.n_pvals, {a:round(b,3) for a,b in caller.kernel_times().items()})
