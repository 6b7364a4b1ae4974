"""experiment: two contexts, batch k+1 submitted before batch k is finished on the host (software pipeline, depth 2)"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import lofreq_amd as la
SEED = 0x9E3779B97F4A7C15 ^ (3 << 32)
dev = torch.device("cuda", 0)
ncols, depth = 1000000, 10000
callers = [la.SnvCaller(0), la.SnvCaller(0)]
batch = callers[0].synth_batch(SEED, depth, ncols, plant_period=997)
bufs = [(torch.zeros(ncols * 64, dtype=torch.uint8, device=dev), torch.zeros(ncols * 128, dtype=torch.uint8, device=dev)) for _ in range(2)]
torch.cuda.synchronize()

def submit(k):
    conf = la.VarcallConf()
    callers[k % 2].snv_batch_device(batch, conf, bufs[k % 2][0], bufs[k % 2][1], ncols)
    return conf

def collect(k, conf):
    c = callers[k % 2]
    st = c.batch_finish()
    pv = bufs[k % 2][1][: st.n_pvals * 128].cpu().numpy().view(la.COL_PVALS_DTYPE)
    recs = la.finalize_pvals(conf, pv, None)
    conf.c.bonf_subst = 3 * st.n_tested
    thr = la.snvqual_thresh(conf.sig, conf.bonf_subst)
    keep = la.filter_records(recs, thr, apply_defaults=False)
    return la.format_vcf(recs, "synth", keep=keep, filter_str="PASS")

for mode in ("sequential", "pipelined", "sequential", "pipelined"):
    K = 40
    for w in range(5):
        collect(w, submit(w))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "sequential":
        for k in range(K):
            text = collect(k, submit(k))
    else:
        prev = submit(0)
        for k in range(1, K):
            cur = submit(k)
            text = collect(k - 1, prev)
            prev = cur
        text = collect(K - 1, prev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s: %.3f ms/step, %d VCF lines" % (mode, 1e3 * dt / K, text.count("\n")), flush=True)
