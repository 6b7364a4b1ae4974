#!/usr/bin/env python
"""bench.py -- pileup columns/sec of the per-column SNV calling path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path (count -> running-Bonferroni scan -> Poisson-binomial DP -> host
emit test / VCF records) over one batch of synthetic pileup columns that are ALREADY RESIDENT IN HBM
(generated on the device by the workload spec of include/lofreq_synth.h).  Default workload =
BASELINE.json configs[2] ("C3"): synthetic 1 Mb genome, 10000x ultra-deep, SNV-only,
--no-default-filter, dynamic Bonferroni -- the configuration the metric "pileup columns/sec at depth
10000" is quoted on; it fits one GPU (40 GB of tracks).  With N > 1 every rank owns its own 1 Mb
region shard (weak scaling, the reference's call-parallel model) and the only exchange is the tested
-column count all-gather + the record gather of lofreq_amd/shard.py.

Prints ONE JSON line on rank 0 (see the repository prompt for the contract) with two extra objects:
`roofline` (HBM roofline of the dominant kernel from HIP-event timings taken inside the C library on
the stream the kernels run on) and `cpu_baseline` (the oracle, single thread, on a bounded sample of
the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
SEED = 0x9E3779B97F4A7C15 ^ (3 << 32)   # SURVEY 8d seed formula, config id 3


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--depth", type=int, default=10000)
    ap.add_argument("--cols", type=int, default=1000000, help="columns per GPU (region shard)")
    ap.add_argument("--plant-period", type=int, default=997)
    ap.add_argument("--cpu-sample-cols", type=int, default=8000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nt-bytes", action="store_true",
                    help="one byte per observation in the nt track instead of the packed layout (LFQ_TRACKS_NT_PACKED)")
    ap.add_argument("--pipeline", action="store_true",
                    help="N=1 only: two contexts, batch k+1 is submitted (lfq_call_snvs_submit) before batch k is "
                         "collected; every step still does all of its work inside the timed region")
    ap.add_argument("--shard-path", action="store_true",
                    help="use the layer-1 + shard-exchange step (what N > 1 runs) even at N = 1")
    return ap.parse_args()


def cpu_baseline(depth, plant_period, sample_cols):
    """Oracle (CPU restatement of the reference algorithm), one thread, first `sample_cols` columns of
    the same workload.  Returns (dict for the JSON line, oracle results for the concordance check)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc
    orc.build()
    host = orc.synth_fill(SEED, depth, plant_period, 0, sample_cols)
    conf = orc.default_conf()
    t0 = time.perf_counter()
    res, tm = orc.call_batch(host["nt"], host["bq"], host["baq"], host["mq"], None, host["col_off"],
                             host["ref_base"], conf, timing=True)
    dt = time.perf_counter() - t0
    out = {
        "value": sample_cols / dt, "unit": "columns/s", "cores": 1, "kind": "port",
        "sample": "first %d columns of the same workload (depth %d, planted SNV every %d columns), "
                  "%.1f s wall; merge %.1f s / sort %.1f s / DP %.1f s; host cpus available: %d"
                  % (sample_cols, depth, plant_period, dt, tm.t_merge, tm.t_sort, tm.t_dp, os.cpu_count()),
    }
    return out, res


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)

    import lofreq_amd as la
    from lofreq_amd import shard

    caller = la.SnvCaller(local_rank)
    caller.set_dense_strand_counts(False)         # DP4 only for the columns that emit (what layer 2 does by itself)
    ncols, depth = args.cols, args.depth
    col_begin = rank * ncols                      # this rank's region shard
    batch = caller.synth_batch(SEED, depth, ncols, plant_period=args.plant_period, col_begin=col_begin,
                               nt_packed=not args.nt_bytes)
    d_counts = torch.zeros(ncols * 64, dtype=torch.uint8, device=dev)
    pv_cap = ncols
    d_pvals = torch.zeros(pv_cap * 128, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)

    def step():
        """One pass: kernels, sparse results to the host, exact emit test, exchange, VCF text."""
        conf = la.VarcallConf()                   # default sig, dynamic Bonferroni from 1
        if world == 1 and not args.shard_path:
            # layer 2 of the C ABI (lfq_call_snvs_batch): the whole call_snvs loop over the batch in one call
            recs, _, st = caller.call_snvs(batch, conf, records_capacity=1 << 16)
        else:
            # layer 1 + the shard exchange: the running Bonferroni factor needs every rank's tested-column count
            caller.snv_batch_device(batch, conf, d_counts, d_pvals, pv_cap)
            st = caller.batch_finish()
            pv = d_pvals[: st.n_pvals * 128].cpu().numpy().view(la.COL_PVALS_DTYPE)
            recs, total = shard.finish_shard(conf, pv, st.n_tested, None, col_begin,      # records carry their ref base
                                             dist if world > 1 else None, dev)
        text = None
        if rank == 0:
            # --no-default-filter + dynamic Bonferroni: QUAL threshold from the final factor
            thr = la.snvqual_thresh(conf.sig, conf.bonf_subst)
            keep = la.filter_records(recs, thr, apply_defaults=False)
            text = la.format_vcf(recs, "synth", keep=keep, filter_str="PASS")
        return conf, st, recs, text, caller.kernel_times()

    pipelined = args.pipeline and world == 1 and not args.shard_path
    if pipelined:
        callers = [caller, la.SnvCaller(local_rank)]
        callers[1].set_dense_strand_counts(False)

        def submit(k):
            conf = la.VarcallConf()
            callers[k % 2].call_snvs_submit(batch, conf)
            return conf

        def collect(k, conf):
            recs, st = callers[k % 2].call_snvs_collect(conf, records_capacity=1 << 16)
            thr = la.snvqual_thresh(conf.sig, conf.bonf_subst)
            keep = la.filter_records(recs, thr, apply_defaults=False)
            text = la.format_vcf(recs, "synth", keep=keep, filter_str="PASS")
            return conf, st, recs, text, callers[k % 2].kernel_times()

    for _ in range(args.warmup):
        step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    kt_acc = None
    barrier()
    t0 = time.perf_counter()
    if pipelined:
        pending = submit(0)
        for k in range(1, args.steps + 1):
            nxt = submit(k) if k < args.steps else None
            conf, st, recs, text, kt = collect(k - 1, pending)
            kt_acc = kt if kt_acc is None else {x: kt_acc[x] + kt[x] for x in kt}
            pending = nxt
    else:
        for _ in range(args.steps):
            conf, st, recs, text, kt = step()
            kt_acc = kt if kt_acc is None else {k: kt_acc[k] + kt[k] for k in kt}
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        steps = max(args.steps, 1)
        ms_per_step = 1e3 * elapsed / steps
        total_cols = ncols * world
        value = total_cols * steps / elapsed
        kt = {k: v / steps for k, v in kt_acc.items()}
        n_launch = max(int(round(kt["n_segments"])), 1)       # count-kernel launches per step
        # Dominant kernel = the one with the largest summed duration per step.  (The three DP kernels run
        # CONCURRENTLY with each other; their individual durations overlap and are not additive.)
        # Algorithmic bytes per SURVEY 8(d): 4*depth + 80 per column; one launch covers ncols/n_launch columns.
        alg_bytes = ncols * (4.0 * depth + 80.0) / n_launch
        # ms_dp_light/mid/big are the spans of the three DP stream chains (quad+retry | mid+segments+fold |
        # prep+segments+fold), which overlap; the count kernel is one launch on its own.
        cands = {"lfq_count_kernel": kt["ms_count"] / n_launch, "dp chain: lfq_dp_quad_kernel<8>+retry": kt["ms_dp_light"] / n_launch,
                 "dp chain: mid class": kt["ms_dp_mid"] / n_launch, "dp chain: big class": kt["ms_dp_big"] / n_launch}
        dom = max(cands, key=cands.get)
        dom_ms = cands[dom]
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                tj = json.load(open(pmc))
                # the instantiation bench.py runs: <packed nt, strand planes>; dense strand counts are switched off above
                key = dom + ("<false, false>" if args.nt_bytes else "<true, false>") if dom == "lfq_count_kernel" else dom
                traffic = tj.get(key)
            except Exception:
                traffic = None
        line = {
            "metric": "pileup columns/sec at depth %d" % depth,
            "value": value, "unit": "columns/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "C3: synthetic 1 Mb genome per GPU, uniform %dx depth, SNV-only, "
                            "--no-default-filter, dynamic Bonferroni (BASELINE.json configs[2])" % depth,
                "columns_per_gpu": ncols, "depth": depth, "planted_snv_period": args.plant_period,
                "sharding": "region shard per GPU, test-count all-gather + record gather (RCCL)",
                "records_per_step": int(len(recs)), "tested_columns_rank0": int(st.n_tested),
                "pipeline_depth": 2 if pipelined else 1,
                "nt_layout": "bytes" if args.nt_bytes else "packed nibbles (LFQ_TRACKS_NT_PACKED)",
                "kernel_ms": kt,
            },
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": dom_ms,
                # frac prices SURVEY 8(d)'s 4 bytes per observation; the kernel only has to READ 2 of them, so
                # frac can exceed 1.  traffic_frac is the real HBM utilisation: measured bytes / duration / peak.
                "traffic_frac": (traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and dom_ms > 0 else None,
                "count_kernel": {
                    "avg_launch_ms": kt["ms_count"] / n_launch,
                    "achieved": alg_bytes / (kt["ms_count"] / n_launch * 1e-3) / 1e9 if kt["ms_count"] > 0 else 0.0,
                    "note": "streaming kernel; reads only the nt+bq tracks (1.5 of the 4 algorithmic bytes per "
                            "observation with the packed nt layout, 2 with bytes) in the default filter configuration"},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = min(args.cpu_sample_cols, ncols)
            base, ores = cpu_baseline(depth, args.plant_period, sample)
            line["cpu_baseline"] = base
            # VCF concordance on the sample: same records, same QUAL, from the full GPU run
            exp = [(c, int(ores["qual"][c, a])) for c in range(sample) for a in range(3) if ores["emitted"][c, a]]
            got = [(int(r["col"]), int(r["qual"])) for r in recs if r["col"] < sample]
            line["config"]["vcf_concordance"] = {"sample_columns": sample, "reference_records": len(exp),
                                                 "gpu_records": len(got), "identical": exp == got}
            line["config"]["speedup_vs_cpu_1thread"] = value / base["value"]
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    caller.close()


if __name__ == "__main__":
    main()
