import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import lofreq_amd as la
from lofreq_amd import shard
SEED = 0x9E3779B97F4A7C15 ^ (3 << 32)
dev=torch.device("cuda",0)
caller=la.SnvCaller(0)
ncols=1000000; depth=10000
batch=caller.synth_batch(SEED, depth, ncols, plant_period=997)
d_counts=torch.zeros(ncols*64,dtype=torch.uint8,device=dev); d_pvals=torch.zeros(ncols*128,dtype=torch.uint8,device=dev)
torch.cuda.synchronize()
for it in range(3):
    T=[time.perf_counter()]
    conf=la.VarcallConf()
    caller.snv_batch_device(batch, conf, d_counts, d_pvals, ncols); T.append(time.perf_counter())
    st=caller.batch_finish(); T.append(time.perf_counter())
    pv=d_pvals[:st.n_pvals*128].cpu().numpy().view(la.COL_PVALS_DTYPE); T.append(time.perf_counter())
    T.append(time.perf_counter())
    recs=la.finalize_pvals(conf, pv, None); T.append(time.perf_counter())
    thr=la.snvqual_thresh(conf.sig, 3*st.n_tested); keep=la.filter_records(recs, thr, apply_defaults=False); T.append(time.perf_counter())
    text=la.format_vcf(recs,"synth",keep=keep,filter_str="PASS"); T.append(time.perf_counter())
    print("launch %.2f finish(sync) %.2f d2h_pvals %.2f d2h_ref %.2f finalize %.2f filter %.2f format %.2f | n_pvals %d recs %d" % tuple([1e3*(T[i+1]-T[i]) for i in range(7)]+[st.n_pvals,len(recs)]), caller.kernel_times())
st_counts=np.bincount(pv["status"].reshape(-1),minlength=4); print("status hist",st_counts, "kmax>=505:", int((pv["counts"]["kmax"]>=505).sum()))
