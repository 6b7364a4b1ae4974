import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import lofreq_amd as la
SEED = 0x9E3779B97F4A7C15 ^ (3 << 32)
caller = la.SnvCaller(0)
batch = caller.synth_batch(SEED, 10000, 1000000, plant_period=997)
acc = np.zeros(6)
K = 30
for it in range(K + 5):
    T = [time.perf_counter()]
    conf = la.VarcallConf(); T.append(time.perf_counter())
    recs, _, st = caller.call_snvs(batch, conf, records_capacity=1 << 16); T.append(time.perf_counter())
    thr = la.snvqual_thresh(conf.sig, conf.bonf_subst); T.append(time.perf_counter())
    keep = la.filter_records(recs, thr, apply_defaults=False); T.append(time.perf_counter())
    text = la.format_vcf(recs, "synth", keep=keep, filter_str="PASS"); T.append(time.perf_counter())
    kt = caller.kernel_times(); T.append(time.perf_counter())
    if it >= 5:
        acc += np.diff(T)
print("per step ms: conf %.3f  call_snvs %.3f  thresh %.3f  filter %.3f  format %.3f  kernel_times %.3f  (kernels %.3f)" % (*(1e3 * acc / K), kt["ms_total"]))
