#!/usr/bin/env python
"""bench.py -- pileup columns/sec of the per-column SNV calling path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config C3|C2|C4|C5] [--mode resident|host-abi|chain|baq]

A step is one pass of the hot path (count -> running-Bonferroni scan -> Poisson-binomial DP -> host emit test /
filter / VCF records) over one batch of synthetic pileup columns that are ALREADY RESIDENT IN HBM (generated on
the device by the workload spec of include/lofreq_synth.h).  Default workload = BASELINE.json configs[2] ("C3"):
synthetic 1 Mb genome, 10000x ultra-deep, SNV-only, --no-default-filter, dynamic Bonferroni -- the configuration
the metric "pileup columns/sec at depth 10000" is quoted on; it fits one GPU (35 GB of tracks).  `--config C2` is
configs[1] (1000x, default filter applied).  With N > 1 every rank owns its own 1 Mb region shard (weak scaling,
the reference's call-parallel model) or, with `--scaling strong`, its bins of ONE 1 Mb genome cut and dealt by
lofreq_amd.shard.plan_regions; the only exchange is the tested-column count all-gather + the record gather of
lofreq_amd/shard.py.

Prints ONE JSON line on rank 0 with these extra objects:
  roofline      HBM roofline of the dominant kernel: bytes the launched instantiation moves (from the layout: the
                library reports them per batch) / its HIP-event duration (events on the stream the kernel runs on,
                taken inside the C library) / 8 TB/s.  `traffic` = HBM bytes per launch from rocprofv3 PMC passes
                that THIS run spawns on the same workload (FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE
                doubled for 16 B/lane streaming reads per MI355X_MICROARCH.md), or null when rocprofv3 is not usable.
                `algorithmic_8d` keeps SURVEY 8(d)'s 4*depth+80 bytes per column figure (the kernel has to read only
                1.5 of those 4 bytes, so that ratio can exceed 1 and is not called a fraction).
  dp            secondary figure: recurrence cells processed per step (device counters), cells/s over
                the DP span, VALU-busy of the DP kernels from an SQ_INSTS_VALU pass.
  cpu_baseline  the oracle (CPU restatement of the reference algorithm), one pinned core, bounded sample.
  config.secondary   (N = 1) the two paths a caller outside this benchmark takes: `host_abi` = host buffers through
                lfq_call_snvs_batch(tracks_on_device = 0), PCIe included; `chain` = reads -> BAQ -> pileup -> calls.
"""
import argparse
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured for a float4 copy
N_SIMD = 1024              # 256 CUs x 4 SIMDs
CLK_HZ = 2.4e9
CONFIGS = {
    # name: (BASELINE.json index, depth, columns per GPU, default filter, CPU sample columns)
    "C3": (2, 10000, 1000000, False, 8000),
    "C2": (1, 1000, 1000000, True, 60000),
}


def seed_of(config_id):
    return 0x9E3779B97F4A7C15 ^ (config_id << 32)      # SURVEY 8d seed formula


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100,
                    help="untimed steps before the timed ones (default 100 = 0.3 s: the first 200-step block after 10 was 1 %% slower than the blocks behind it)")
    ap.add_argument("--config", choices=["C1"] + sorted(CONFIGS) + sorted(GENOME_CONFIGS), default="C3",
                    help="C3 (default, the metric) / C2: resident synthetic pileups; C4 / C5: a whole genome of reads through the "
                         "reads -> VCF chain, region- / BED-sharded over --gpus (strong scaling)")
    ap.add_argument("--vcf-out", default=None, help="C4 / C5: rank 0 writes the VCF text of the last step here")
    ap.add_argument("--genome-scale", type=float, default=1.0, help="C4 / C5: a smaller genome of the same shape (tests)")
    ap.add_argument("--distinct-bins", type=int, default=4,
                    help="C4 / C5: distinct sets of reads (generated from different seeds) that the genome's bins cycle through; "
                         "every bin its own set = the number of bins (32: minutes of generation and ~10 GB of pinned host memory)")
    ap.add_argument("--upload-reads", action="store_true",
                    help="C4 / C5: the timed steps upload every bin's reads from (pinned) host memory again -- the PCIe-inclusive "
                         "rate; default: the read sets are resident in HBM when the timed region starts (uploaded in the "
                         "warm-up), and the PCIe-inclusive rate of a few extra steps is reported beside `value`")
    ap.add_argument("--no-upload-rate", action="store_true", help="C4 / C5: skip the extra steps that time the PCIe-inclusive rate")
    ap.add_argument("--host-threads", type=int, default=6,
                    help="C4 / C5: host threads per rank, one context each, that work through the rank's bins (the host part "
                         "of one bin -- CIGAR geometry, event tables, test descriptors -- then runs under the kernels of another)")
    ap.add_argument("--idaq", action="store_true", help="--mode baq: also the indel alignment qualities (ai / ad)")
    ap.add_argument("--mode", choices=["resident", "host-abi", "chain", "baq"], default="resident",
                    help="resident (default, the metric): tracks in HBM; host-abi: host buffers through the C ABI; "
                         "chain: reads -> BAQ -> pileup -> calls on a resident read set")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--cols", type=int, default=None, help="columns per GPU (region shard)")
    ap.add_argument("--plant-period", type=int, default=997)
    ap.add_argument("--cpu-sample-cols", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (roofline.traffic = null)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the host-abi / chain measurements of config.secondary")
    ap.add_argument("--nt-bytes", action="store_true",
                    help="one byte per observation in the nt track instead of the packed layout (LFQ_TRACKS_NT_PACKED)")
    ap.add_argument("--in-flight", type=int, default=0, choices=[0, 1, 2, 3, 4, 6, 8],
                    help="pipelined loop: batches submitted and not yet waited for (1: step k + 1 is submitted when the kernels "
                         "of step k are done; 2-4: that many are queued on the device, --gate says what a queued batch's count "
                         "kernel waits for; 0 = default: the warm-up times the candidates and keeps the fastest)")
    ap.add_argument("--gate", choices=["auto", "tail", "end", "none"], default="auto",
                    help="with two batches in flight: what the second one's count kernel waits for on the device "
                         "(lfq_set_batch_gate): the first one's row-bound DP kernels (tail), all of its kernels (end: batch "
                         "after batch with no host round trip between them), nothing (none); auto = timed in the warm-up")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="N = 1: one context, every step waits for its own host finish before the next batch is launched "
                         "(default: two contexts; the host finish of step k runs under the kernels of step k + 1)")
    ap.add_argument("--submit-thread", action="store_true",
                    help="pipelined loop: a submit thread beside the finishing thread instead of submit and finish in turn on "
                         "one thread (measured in round 6: C2 0.573 against 0.542 ms per step, C3 the same -- the device, "
                         "not the host, is what paces both -- so it is off by default; profiles/r06_host_threads.md)")
    ap.add_argument("--shard-path", action="store_true",
                    help="use the layer-1 + shard-exchange step (what N > 1 runs) even at N = 1")
    ap.add_argument("--workers", type=int, default=1,
                    help="--mode chain: region workers (processes) sharing the GPU, as call-parallel runs one per bin")
    ap.add_argument("--overlap-regions", action="store_true",
                    help="--mode chain, one worker: start region k + 1 (upload, BAQ kernels) before the pileups and calls of "
                         "region k, as integration/lofreq_amd_region.c does")
    ap.add_argument("--pageable", action="store_true",
                    help="--mode chain, one worker: the caller's read arrays in pageable memory (default: the per-base arrays "
                         "pinned, as the region binding keeps them)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed blocks of --steps steps: the first one is the reported value, all of them go into `repeats`")
    ap.add_argument("--no-full-check", action="store_true",
                    help="N = 1: skip the whole-batch oracle run on all host cores (config.vcf_concordance then covers the "
                         "cpu_baseline sample only)")
    ap.add_argument("--full-check-procs", type=int, default=None, help="oracle processes of the whole-batch check")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="C3, N = 1: skip the short C2 / C4 / C5 child runs whose scalars (c2_*, c4_*, c5_*) the line's config carries")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seed, depth, plant_period, sample_cols, default_filter):
    """Oracle (CPU restatement of the reference algorithm), one thread pinned to one core, first `sample_cols`
    columns of the same workload.  Returns (dict for the JSON line, oracle results for the concordance check)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc
    orc.build()
    host = orc.synth_fill(seed, depth, plant_period, 0, sample_cols)
    conf = orc.default_conf()
    pinned = None
    old_aff = None
    try:                                            # taskset -c <first allowed core>, in-process
        old_aff = os.sched_getaffinity(0)
        pinned = min(old_aff)
        os.sched_setaffinity(0, {pinned})
    except (AttributeError, OSError):
        pinned = None
    t0 = time.perf_counter()
    res, tm = orc.call_batch(host["nt"], host["bq"], host["baq"], host["mq"], None, host["col_off"],
                             host["ref_base"], conf, timing=True)
    dt = time.perf_counter() - t0
    if old_aff is not None:
        try:
            os.sched_setaffinity(0, old_aff)
        except OSError:
            pass
    # SURVEY 8d's cell count in the REFERENCE's row order (probabilities sorted ascending, its own pruning row)
    k = np.max(res["alt_counts"], axis=1).astype(np.int64)
    n = res["dp_rows"].astype(np.int64)
    cells_ref = int(np.sum(np.where(n <= k, n * (n + 1) // 2, k * (k + 1) // 2 + (n - k) * k)))
    out = {
        "value": sample_cols / dt, "unit": "columns/s", "cores": 1, "kind": "port",
        "cpu_model": cpu_model(), "host_cpus": os.cpu_count(), "pinned_to_core": pinned,
        "sample": "first %d columns of the same workload (depth %d, planted SNV every %d columns), %.1f s wall on "
                  "one pinned core; merge %.1f s / sort %.1f s / DP %.1f s"
                  % (sample_cols, depth, plant_period, dt, tm.t_merge, tm.t_sort, tm.t_dp),
        "dp_cells_reference_order": cells_ref,
        "dp_cells_per_s": cells_ref / tm.t_dp if tm.t_dp > 0 else None,
    }
    try:
        out["reference_cli"] = reference_cli_baseline(depth, default_filter)
    except Exception as e:              # the reference's binary is an extra: the port's figure above stands without it
        out["reference_cli"] = {"error": repr(e)[:200]}
    return out, res


def reference_cli_baseline(depth, default_filter, glen=1300):
    """`lofreq call` of the REFERENCE ITSELF on this box's host, one pinned core: the prebuilt 2.1.4 binary of the reference
    tree's dist/ tarball, unpacked into oracle/_ref by `make -C oracle ref` in the build container (git-ignored, travels with
    the other built files).  north_star's target is stated against this: "the single-thread CPU `lofreq call` rate in
    pileup-columns/sec" at the config's depth.  The whole CLI -- SAM parsing, on-the-fly BAQ, pileup, calls, VCF -- on a bounded
    sample (seeded reads of tests/golden_reads.py over `glen` bases at `depth`x), not only the column path the port above times:
    reported beside it, never as the headline's denominator.  -> dict or None (no binary here)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "bin", "lofreq")
    if not os.path.exists(exe):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_reads as gr
    R = gr.make(seed=900 + depth, glen=glen, depth_lo=depth, depth_hi=depth, min_q=6, snv_every=97, mapq_mix=False)
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        open(os.path.join(tmp, "t.fa"), "w").write(">chr1\n" + R["ref"].decode() + "\n")
        gr.write_sam(R, os.path.join(tmp, "t.sam"))
        subprocess.check_call([exe, "faidx", "t.fa"], cwd=tmp)
        env = dict(os.environ, PATH=os.path.dirname(exe) + ":" + os.environ.get("PATH", ""))
        try:
            core = min(os.sched_getaffinity(0))
            pin = lambda: os.sched_setaffinity(0, {core})
        except (AttributeError, OSError):
            core, pin = None, None
        args = [exe, "call", "-f", "t.fa", "-o", "out.vcf"] + ([] if default_filter else ["--no-default-filter"]) + ["t.sam"]
        t0 = time.perf_counter()
        p = subprocess.run(args, cwd=tmp, env=env, capture_output=True, text=True, timeout=300, preexec_fn=pin)
        dt = time.perf_counter() - t0
        if p.returncode != 0:
            raise RuntimeError(p.stderr[-200:])
        n_vcf = sum(1 for l in open(os.path.join(tmp, "out.vcf")) if not l.startswith("#"))
    return {"value": glen / dt, "unit": "columns/s", "cores": 1, "kind": "reference", "pinned_to_core": core,
            "binary": "lofreq 2.1.4 (the reference tree's dist/lofreq_star-2.1.4_linux-x86-64.tgz, oracle/_ref/bin/lofreq)",
            "command": "lofreq call%s -f t.fa -o out.vcf t.sam" % ("" if default_filter else " --no-default-filter"),
            "sample": "%d reads x 150 bp over %d bases at %dx (tests/golden_reads.py, an SNV site every 97 bases): %.1f s wall "
                      "on one pinned core, %d VCF lines; SAM parsing + BAQ + pileup + calls (the whole CLI)"
                      % (R["n"], glen, depth, dt, n_vcf)}


# ---------------------------------------------------------------------------------------------------------
# rocprofv3 counter passes, spawned by the run itself (rank 0, N = 1)
# ---------------------------------------------------------------------------------------------------------

def _short_kernel(name):
    return name.replace("void ", "").split("(")[0]


def pmc_pass(counter, child_args, timeout_s=240):
    """One `rocprofv3 --pmc <counter> --kernel-trace` pass over a 2-step child run of this script.
    -> {kernel: (sum, launches)} or None.  One counter per pass: FETCH_SIZE + WRITE_SIZE together exceed the
    TCC slots of gfx950 (MI355X_MICROARCH.md "rocprofv3 PMC slots")."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out_dir = tempfile.mkdtemp(prefix="lfq_pmc_", dir="/tmp")
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    env["LFQ_SINGLE_STREAM"] = "1"      # counter collection serialises dispatches; cross-stream waits do not get along with it
    cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", out_dir, "-o", "pmc", "--",
           sys.executable, os.path.abspath(__file__), "--pmc-child"] + child_args
    try:
        p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                             start_new_session=True)
        try:
            p.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, 9)         # the exact process group started above
            p.wait()
            return None
        dbs = glob.glob(os.path.join(out_dir, "**", "*.db"), recursive=True)
        if p.returncode != 0 or not dbs:
            return None
        res = {}
        db = sqlite3.connect(dbs[0])
        for name, cname, val, n in db.execute(
                "select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                "group by kernel_name, counter_name"):
            if cname == counter:
                res[_short_kernel(name)] = (float(val), int(n))
        db.close()
        return res
    except Exception:
        return None
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def live_pmc(child_args, count_kernel):
    """HBM traffic of the count kernel and VALU instructions of the DP kernels, from three counter passes."""
    out = {"traffic": None, "fetch_kib": None, "write_kib": None, "valu_insts_dp": None, "valu_by_kernel": None,
           "source": "rocprofv3 --pmc passes spawned by this run (2-step child of the same command, LFQ_SINGLE_STREAM=1)"}
    f = pmc_pass("FETCH_SIZE", child_args)
    w = pmc_pass("WRITE_SIZE", child_args) if f is not None else None
    v = pmc_pass("SQ_INSTS_VALU", child_args) if f is not None else None
    if f and count_kernel not in f:
        # the instantiation's template arguments as the runtime names them (the caller knows the family, not the spelling)
        fam = [k for k in f if k.startswith(count_kernel.split("<")[0] + "<")]
        if len(fam) == 1:
            count_kernel = fam[0]
    out["kernel"] = count_kernel
    if f and w and count_kernel in f and count_kernel in w:
        fk, n = f[count_kernel]
        wk, _ = w[count_kernel]
        out["fetch_kib"] = fk / n
        out["write_kib"] = wk / n
        # FETCH_SIZE counts 64 B per 128 B request of a 16 B/lane streaming read on gfx950: doubled (guide, HBM section)
        out["traffic"] = (2.0 * fk + wk) * 1024.0 / n
    if v:
        dp = {k: x for k, x in v.items() if k.startswith("lfq_dp_") or k.startswith("lfq_strand_")}
        # per step: the pass ran as many steps as it launched count kernels (the screen kernel's variant may differ
        # between the first step of a context and the later ones, so no single DP kernel counts the steps)
        steps = max(v[count_kernel][1], 1) if count_kernel in v else 2
        out["valu_insts_dp"] = sum(x[0] for x in dp.values()) / steps
        out["valu_by_kernel"] = {k: x[0] / steps for k, x in dp.items()}
        if count_kernel in v:
            out["valu_insts_count"] = v[count_kernel][0] / v[count_kernel][1]
    return out


# ---------------------------------------------------------------------------------------------------------
# secondary paths (N = 1): host-buffer ABI, reads -> VCF chain
# ---------------------------------------------------------------------------------------------------------

def bench_host_abi(caller, la, seed, depth, ncols, plant_period, steps):
    """Host buffers through the C ABI, the way the plp_proc_func shim hands them over (integration/lofreq_amd_shim.c): the
    tracks in PINNED host memory (lfq_host_alloc), nt nibble-packed, two batches in flight -- lfq_call_snvs_submit of
    batch k + 1 (its DMA uploads) before lfq_call_snvs_collect of batch k, on two contexts, every batch an independent
    region with a conf of its own.  PCIe upload included; bytes = the tracks as sent + headers."""
    import torch
    # the workload of include/lofreq_synth.h, generated on the device in the byte layout and copied to host arrays
    dev_b = caller.synth_batch(seed, depth, ncols, plant_period=plant_period, nt_packed=False)
    n = ncols * depth

    def pinned(t):
        h = torch.empty(t.numel(), dtype=t.dtype).pin_memory()
        h.copy_(t.cpu())
        return h

    col_off = dev_b.col_off.cpu().numpy().astype(np.uint64)
    ref = dev_b.ref_base[:ncols].cpu().numpy().copy()
    sets = []
    for _ in range(2):
        plain = la.PileupBatch(dev_b.nt[:n].cpu().numpy(), dev_b.bq[:n].cpu().numpy(), dev_b.mq[:n].cpu().numpy(), col_off, ref,
                               baq=dev_b.baq[:n].cpu().numpy(), max_col_obs=depth).packed()
        keep = {k: pinned(torch.from_numpy(np.ascontiguousarray(getattr(plain, k)))) for k in ("nt", "bq", "mq", "baq")}
        b = la.PileupBatch(keep["nt"].numpy(), keep["bq"].numpy(), keep["mq"].numpy(), col_off, ref, baq=keep["baq"].numpy(),
                           max_col_obs=depth, nt_packed=True)
        b._pinned = keep
        sets.append(b)
    del dev_b
    n_obs = int(col_off[-1])
    callers = [caller, la.SnvCaller(caller.device)]
    confs = [None, None]

    def submit(k):
        confs[k % 2] = la.VarcallConf()
        callers[k % 2].call_snvs_submit(sets[k % 2], confs[k % 2])

    def collect(k):
        return callers[k % 2].call_snvs_collect(confs[k % 2], records_capacity=1 << 16)

    submit(0)
    collect(0)                                       # warm-up: staging allocations of both contexts
    submit(1)
    collect(1)
    t0 = time.perf_counter()
    submit(0)
    for k in range(steps):
        if k + 1 < steps:
            submit(k + 1)
        recs, st = collect(k)
    dt = (time.perf_counter() - t0) / steps
    callers[1].close()
    # the link's own rate for comparison: a plain pinned host -> device copy of 256 MiB
    hp = torch.empty(1 << 28, dtype=torch.uint8).pin_memory()
    dd = torch.empty(1 << 28, dtype=torch.uint8, device=torch.device("cuda", caller.device))
    dd.copy_(hp, non_blocking=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(4):
        dd.copy_(hp, non_blocking=True)
    torch.cuda.synchronize()
    link = 4 * (1 << 28) / (time.perf_counter() - t1) / 1e9
    del hp, dd
    byt = 3.5 * n_obs + ncols * 9.0
    return {"columns_per_s": ncols / dt, "ms_per_batch": dt * 1e3, "columns_per_batch": ncols, "depth": depth,
            "host_bytes_per_batch": byt, "effective_GBps": byt / dt / 1e9, "pcie_peak_GBps": 63.0,
            "frac_of_pcie": byt / dt / 1e9 / 63.0, "pcie_measured_copy_GBps": link, "frac_of_measured_copy": byt / dt / 1e9 / link,
            "records": int(len(recs)), "nt_layout": "packed nibbles (host-packed)",
            "note": "pinned host arrays in (lfq_host_alloc: what the shim's buffers are), VCF records out; two batches in "
                    "flight: the upload of batch k + 1 under the kernels of batch k; upload + kernels + host finish per batch"}


def make_reads(n, glen, rl=150, seed=3, indel_frac=0.04):
    """Position-sorted synthetic reads over a random genome: 0.3 % mismatches, `indel_frac` of the reads with one
    1-3 bp insertion or deletion, BI / BD tags."""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, glen).astype(np.uint8)
    gen_ascii = np.frombuffer(b"ACGT", np.uint8)[genome].tobytes()
    pos = np.sort(rng.integers(0, glen - rl - 20, n)).astype(np.int32)
    has = rng.random(n) < indel_frac
    kind = rng.random(n) < 0.5
    ilen = rng.integers(1, 4, n)
    cut = rng.integers(50, 100, n)
    base = genome[(pos[:, None] + np.arange(rl + 4)[None, :])]
    seq = np.empty((n, rl), np.uint8)
    plain = ~has
    seq[plain] = base[plain, :rl]
    for i in np.nonzero(has)[0]:
        c, k = int(cut[i]), int(ilen[i])
        if kind[i]:
            seq[i, :c] = base[i, :c]
            seq[i, c:c + k] = rng.integers(0, 4, k)
            seq[i, c + k:] = base[i, c:rl - k]
        else:
            seq[i, :c] = base[i, :c]
            seq[i, c:] = base[i, c + k:rl + k]
    cigs = np.zeros((n, 3), np.uint32)
    ncig = np.ones(n, np.int64)
    cigs[:, 0] = (rl << 4)
    ii = np.nonzero(has)[0]
    cigs[ii, 0] = (cut[ii].astype(np.uint32) << 4)
    cigs[ii, 1] = (ilen[ii].astype(np.uint32) << 4) | np.where(kind[ii], 1, 2).astype(np.uint32)
    cigs[ii, 2] = ((rl - cut[ii] - np.where(kind[ii], ilen[ii], 0)).astype(np.uint32) << 4)
    ncig[ii] = 3
    cig_off = np.zeros(n + 1, np.int64)
    cig_off[1:] = np.cumsum(ncig)
    cig = np.ascontiguousarray(cigs[np.arange(3)[None, :] < ncig[:, None]])
    mism = rng.random(seq.shape) < 0.003
    seq[mism] = (seq[mism] + 1) % 4
    qual = np.clip(np.round(rng.normal(34, 5, seq.shape)), 2, 41).astype(np.uint8)
    return {
        "n": n, "rl": rl, "glen": glen, "ref": gen_ascii, "pos": pos, "cig_off": cig_off, "cig": cig,
        "seq_off": np.arange(n + 1, dtype=np.int64) * rl, "seq": np.ascontiguousarray(seq.reshape(-1)),
        "qual": np.ascontiguousarray(qual.reshape(-1)),
        "bi": rng.integers(33 + 30, 33 + 50, n * rl).astype(np.uint8),
        "bd": rng.integers(33 + 30, 33 + 50, n * rl).astype(np.uint8),
        "mapq": np.full(n, 60, np.uint8), "rev": (rng.random(n) < 0.5).astype(np.uint8), "n_indel_reads": int(has.sum()),
        "flags": np.full(max(n, 1), 3, np.uint8),       # every read carries BI and BD (bits 0 / 1 of lfq_pileup_indel_tags.tag_flags)
    }


def bench_baq(caller, la, n_reads, glen, iters, want_idaq=False):
    """the BAQ (kpa_ext_glocal, bam_md_ext.c:260-491) step alone on a resident read set: `lofreq alnqual` / the
    on-the-fly BAQ of `lofreq call`, per 400 K x 150 bp reads (the unit VERDICT r01 item 4 is quoted on)"""
    import ctypes as C
    from lofreq_amd import _lib
    R = make_reads(n_reads, glen, indel_frac=0.04 if want_idaq else 0.0)
    L = _lib.load()
    pr = _lib.PileupReads()
    pr.n_reads = R["n"]
    pr.pos, pr.cigar_off, pr.cigar = R["pos"].ctypes.data, R["cig_off"].ctypes.data, R["cig"].ctypes.data
    pr.seq_off, pr.seq, pr.qual = R["seq_off"].ctypes.data, R["seq"].ctypes.data, R["qual"].ctypes.data
    pr.baq = None
    pr.mapq, pr.reverse = R["mapq"].ctypes.data, R["rev"].ctypes.data
    pr.ref = C.cast(C.c_char_p(R["ref"]), C.c_void_p)
    pr.ref_len = glen
    tg = _lib.PileupIndelTags()
    tg.bi, tg.bd = R["bi"].ctypes.data, R["bd"].ctypes.data
    h = C.c_void_p()
    _lib.check(L.lfq_readset_create(caller.h, C.byref(pr), C.byref(tg), C.byref(h)), "lfq_readset_create")
    best = None
    for _ in range(iters + 1):
        t0 = time.perf_counter()
        _lib.check(L.lfq_readset_baq(caller.h, h, 1, 1 if want_idaq else 0), "lfq_readset_baq")
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    L.lfq_readset_destroy(h)
    n_bases = int(R["seq_off"][-1])
    return {"reads": R["n"], "read_len": 150, "idaq": bool(want_idaq), "s_call": best, "reads_per_s": R["n"] / best,
            # seq + qual + one reference window in, lb out
            "algorithmic_bytes": 3 * n_bases + R["n"] * (150 + 14),
            "note": "wall time of lfq_readset_baq (host geometry of the CIGARs + kernel + sync) on a resident read "
                    "set; kernel time alone: profiles/r03_baq_rocprof_stats.md"}


def bench_chain(caller, la, n_reads, glen, iters, call_indels=True, start_barrier=None, overlap=False, pinned=True):
    """reads -> BAQ (+ IDAQ) -> device pileup(s) -> SNV (+ indel) calls on a resident read set: the reference's
    `lofreq call [--call-indels]` with BAQ on (BASELINE.md end-to-end rows), everything after BAM decoding."""
    import ctypes as C
    from lofreq_amd import _lib
    from lofreq_amd.pileup import DeviceTracks
    R = make_reads(n_reads, glen)
    if pinned:
        # what integration/lofreq_amd_region.c does with its per-base arrays (lfq_host_alloc): the copies of
        # lfq_readset_create are then DMA transfers queued at once, and the BAQ kernels -- not this thread -- wait for them
        import torch
        keep = {}
        for k in ("seq", "qual", "bi", "bd", "pos", "cig_off", "cig", "seq_off", "mapq", "rev"):
            keep[k] = torch.from_numpy(R[k]).pin_memory()
            R[k] = keep[k].numpy()
        R["_pinned"] = keep
    L = _lib.load()
    vp = C.c_void_p
    pr = _lib.PileupReads()
    pr.n_reads = R["n"]
    pr.pos, pr.cigar_off, pr.cigar = R["pos"].ctypes.data, R["cig_off"].ctypes.data, R["cig"].ctypes.data
    pr.seq_off, pr.seq, pr.qual = R["seq_off"].ctypes.data, R["seq"].ctypes.data, R["qual"].ctypes.data
    pr.baq = None
    pr.mapq, pr.reverse = R["mapq"].ctypes.data, R["rev"].ctypes.data
    pr.ref = C.cast(C.c_char_p(R["ref"]), C.c_void_p)
    pr.ref_len = glen
    tg = _lib.PileupIndelTags()
    tg.bi, tg.bd = R["bi"].ctypes.data, R["bd"].ctypes.data
    col_pos = np.zeros(glen, np.int64)
    col_pos_s = np.zeros(glen, np.int64)
    L.lfq_set_indel_arrays_on_host(caller.h, 0)
    best = None
    wall = [0.0, 0.0]
    totals = []
    state = {}

    def start():
        """region start: the read set goes up, its BAQ (+ IDAQ) kernels are queued (asynchronous)"""
        T = [time.perf_counter()]
        h = vp()
        _lib.check(L.lfq_readset_create(caller.h, C.byref(pr), C.byref(tg), C.byref(h)), "lfq_readset_create")
        T.append(time.perf_counter())
        _lib.check(L.lfq_readset_baq(caller.h, h, 1, 1 if call_indels else 0), "lfq_readset_baq")
        T.append(time.perf_counter())
        return h, T

    def finish(h, T):
        """pileups, calls, records of a started region"""
        T = T + [time.perf_counter()]               # (overlapped mode: the start of the next region sits in between)
        gap = T[3] - T[2]
        conf = la.VarcallConf(flag=la.LFQ_USE_BAQ | la.LFQ_USE_MQ | la.LFQ_USE_IDAQ)
        n_tests = C.c_int64(0)
        nrec = C.c_int64(0)
        cons = None
        t = _lib.Tracks()

        def snv_pileup():
            # between the indel pileup and the indel tests: the call returns when the scatter pass is queued, which then
            # runs under the host part of the tests (gates, packing)
            t0_ = time.perf_counter()
            _lib.check(L.lfq_readset_pileup_snv(caller.h, h, 0, glen, 3, C.byref(t), col_pos_s.ctypes.data),
                       "lfq_readset_pileup_snv")
            return time.perf_counter() - t0_
        if call_indels:
            outp = C.POINTER(_lib.IndelColumnsC)()
            _lib.check(L.lfq_readset_pileup_indels(caller.h, h, 0, glen, 0, C.byref(outp), col_pos.ctypes.data),
                       "lfq_readset_pileup_indels")
            T.append(time.perf_counter())
            t_snv = snv_pileup()
            cap = 1 << 20
            rec = np.zeros(cap, dtype=_lib.INDEL_RECORD_DTYPE)
            _lib.check(L.lfq_call_indels_batch(caller.h, C.byref(conf.c), outp, rec.ctypes.data, cap, C.byref(nrec),
                                               C.byref(n_tests)), "lfq_call_indels_batch")
            cons = np.frombuffer(C.string_at(outp.contents.cons_indel, outp.contents.ncols), np.uint8).copy()
        else:
            T.append(time.perf_counter())
            t_snv = snv_pileup()
        col_pos_snv = col_pos_s[: t.ncols]
        T.append(time.perf_counter())
        if cons is not None:
            _lib.check(L.lfq_pileup_skip_snv_columns(caller.h, cons.ctypes.data, len(cons)), "skip")
        T.append(time.perf_counter())
        recs, _, st = caller.call_snvs(DeviceTracks(t, col_pos_snv), conf, records_capacity=1 << 18)
        T.append(time.perf_counter())
        L.lfq_readset_destroy(h)
        d = [T[1] - T[0], T[2] - T[1]] + [T[i + 1] - T[i] for i in range(3, 7)]
        d[3] -= t_snv                                # (the SNV pileup's call sits inside the interval of the indel tests)
        d[4] += t_snv
        state.update(t=t, n_tests=n_tests, nrec=nrec, recs=recs)
        return d, gap

    pending = None
    t_prev = None
    for it in range(iters + 2 + (1 if overlap else 0)):     # two warm-up regions: the grow-only buffers settle in the second
        if it == 2:
            if start_barrier is not None:
                start_barrier.wait(timeout=300)     # region workers: every process starts its timed regions together
            wall[0] = time.time()
        if overlap:
            # what integration/lofreq_amd_region.c does: region k + 1 is started (upload and BAQ kernels queued) before the
            # pileups and calls of region k, whose host parts then run under those kernels; a region's time = the interval
            # between two region ends
            nxt = start() if it < iters + 2 else None
            if pending is None:
                pending = nxt
                t_prev = time.perf_counter()
                continue
            d, _ = finish(*pending)
            pending = nxt
            now = time.perf_counter()
            tot = now - t_prev
            t_prev = now
        else:
            d, _ = finish(*start())
            tot = sum(d)
        wall[1] = time.time()
        if it >= 2 + (1 if overlap else 0):
            totals.append(tot)
        if it >= 2 and (best is None or tot < best["s_total"]):
            t = state["t"]
            best = {"s_total": tot, "s_upload": d[0], "s_baq": d[1], "s_indel_pileup": d[2], "s_indel_calls": d[3],
                    "s_snv_pileup": d[4], "s_snv_calls": d[5], "columns": int(t.ncols), "indel_tests": int(state["n_tests"].value),
                    "snv_records": int(len(state["recs"])), "indel_records": int(state["nrec"].value)}
    t = state["t"]
    L.lfq_set_indel_arrays_on_host(caller.h, 1)
    best.update({"s_mean": sum(totals) / max(len(totals), 1), "iterations": len(totals), "s_each": [round(t, 5) for t in totals[:16]],
                 "wall_begin": wall[0], "wall_end": wall[1]})
    best.update({"nt_layout": "packed nibbles (written by the device pileup)" if (t.flags & 1) else "bytes",
                 "reads": n_reads, "read_len": R["rl"], "genome_len": glen, "depth": n_reads * R["rl"] / glen,
                 "reads_per_s": n_reads / best["s_total"], "columns_per_s": best["columns"] / best["s_total"],
                 "call_indels": bool(call_indels),
                 "regions_overlapped": bool(overlap),
                 "caller_arrays": "pinned" if pinned else "pageable",
                 "note": "resident read set; BAM decoding (htslib, CPU) not included; reference end-to-end rows "
                         "(BASELINE.md 2): 6736 cols/s without BAQ, 1334 cols/s with BAQ, one CPU thread"})
    return best


# ---- BASELINE.json configs[3] / configs[4]: a whole genome of reads, region-sharded ---------------------------------------

GENOME_CONFIGS = {
    # C4: E. coli-sized genome, 500x, SNV + --call-indels, default filter; bins = plan_regions over the contig
    "C4": dict(idx=3, genome=4_600_000, depth=500, call_indels=True, targets=False, bins=32,
               what="synthetic 4.6 Mb genome, 500x reads (150 bp), SNV + --call-indels, BAQ / IDAQ on, default filter, "
                    "dynamic Bonferroni, region-sharded (BASELINE.json configs[3])"),
    # C5: exome-like ragged BED targets (~45 % of a 64 Mb window = 29 Mb), 200x on the targets, SNVs only
    "C5": dict(idx=4, genome=64_000_000, depth=200, call_indels=False, targets=True, bins=32,
               what="synthetic exome: ragged BED targets (150..3500 bp) over a 64 Mb window, 200x reads on the targets, "
                    "SNV-only, BAQ on, default filter, Bonferroni summed over all bins (call-parallel's \"auto\": "
                    "lofreq2_call_pparallel.py:131-185, 685-707), BED-sharded (BASELINE.json configs[4])"),
}


def make_tile(cfg, tile_len, seed=11):
    """the reads of ONE bin (every bin of the synthetic genome carries the same reads: the work is a genome's, the host
    memory a bin's); planted SNVs every 997th position at 0.5 / 1 / 5 / 50 %; C5: only reads that overlap a target"""
    rl = 150
    n = int(tile_len * cfg["depth"] / rl)
    rng = np.random.default_rng(seed + 1)
    target = None
    if cfg["targets"]:
        # exome-like: reads only where they overlap a target (what `-l bed` fetches); no indels; cheap generators
        target = np.zeros(tile_len, np.uint8)
        x = 300
        while x < tile_len - 4000:
            l = int(rng.choice([150, 300, 600, 1200, 2000, 3500], p=[0.2, 0.25, 0.2, 0.15, 0.12, 0.08]))
            target[x:x + l] = 1
            x += l + int(rng.integers(200, 1900))
        cs = np.concatenate([[0], np.cumsum(target, dtype=np.int64)])
        pos = np.sort(rng.integers(0, tile_len - rl - 20, n)).astype(np.int32)
        pos = np.ascontiguousarray(pos[(cs[pos + rl] - cs[pos]) > 0])
        n = len(pos)
        genome = rng.integers(0, 4, tile_len).astype(np.uint8)
        seq = genome[pos[:, None] + np.arange(rl, dtype=np.int32)[None, :]]
        mism = rng.random(seq.shape, dtype=np.float32) < 0.003
        seq[mism] = (seq[mism] + 1) % 4
        R = {"n": n, "rl": rl, "glen": tile_len, "ref": np.frombuffer(b"ACGT", np.uint8)[genome].tobytes(), "pos": pos,
             "cig_off": np.arange(n + 1, dtype=np.int64), "cig": np.full(n, rl << 4, np.uint32),
             "seq_off": np.arange(n + 1, dtype=np.int64) * rl, "seq": seq.reshape(-1),
             "qual": rng.integers(28, 42, n * rl, dtype=np.uint8), "mapq": np.full(n, 60, np.uint8),
             "rev": (rng.random(n) < 0.5).astype(np.uint8)}
    else:
        R = make_reads(n, tile_len, rl=rl, seed=seed, indel_frac=0.04 if cfg["call_indels"] else 0.0)
    pos, seq = R["pos"], R["seq"].reshape(R["n"], rl)
    plain = (R["cig_off"][1:] - R["cig_off"][:-1]) == 1
    for k, p in enumerate(range(500, tile_len - rl - 20, 997)):
        af = (0.005, 0.01, 0.05, 0.5)[k % 4]
        lo, hi = np.searchsorted(pos, p - rl + 1), np.searchsorted(pos, p, side="right")
        idx = np.arange(lo, hi)
        idx = idx[plain[idx] & (rng.random(len(idx)) < af)]
        seq[idx, p - pos[idx]] = (seq[idx, p - pos[idx]] + 1 + k % 3) % 4
    if cfg["call_indels"]:
        # planted indels: every 4999th position an insertion or a deletion of 1..3 bases at 5 % or 30 % of the reads
        gcode = (np.frombuffer(R["ref"], np.uint8) >> 1) & 3                     # 'A','C','G','T' -> 0, 1, 3, 2 ...
        gcode = np.where(gcode == 3, 2, np.where(gcode == 2, 3, gcode)).astype(np.uint8)     # ... -> 0, 1, 2, 3
        ncig = (R["cig_off"][1:] - R["cig_off"][:-1]).astype(np.int64)
        cigs = np.zeros((R["n"], 3), np.uint32)
        cigs[np.arange(3)[None, :] < ncig[:, None]] = R["cig"]
        ar = np.arange(rl)
        for k, p in enumerate(range(2500, tile_len - 2 * rl, 4999)):
            ln, is_ins, af = 1 + k % 3, (k & 1) == 0, (0.05, 0.3)[(k >> 1) & 1]
            lo, hi = np.searchsorted(pos, p - rl + 26), np.searchsorted(pos, p - 20, side="right")
            idx = np.arange(lo, hi)
            idx = idx[(ncig[idx] == 1) & (rng.random(len(idx)) < af)]
            if not len(idx):
                continue
            c = (p - pos[idx] + 1).astype(np.int64)                               # bases of the read in front of the event
            src = pos[idx, None] + ar[None, :]
            if is_ins:
                after = ar[None, :] >= (c + ln)[:, None]
                s_new = gcode[np.where(after, src - ln, src)]
                inside = (ar[None, :] >= c[:, None]) & ~after
                s_new[inside] = ((ar[None, :] - c[:, None] + k) % 4).astype(np.uint8)[inside]
                cigs[idx, 1] = (ln << 4) | 1
                cigs[idx, 2] = ((rl - c - ln).astype(np.uint32) << 4)
            else:
                s_new = gcode[np.where(ar[None, :] >= c[:, None], src + ln, src)]
                cigs[idx, 1] = (ln << 4) | 2
                cigs[idx, 2] = ((rl - c).astype(np.uint32) << 4)
            cigs[idx, 0] = c.astype(np.uint32) << 4
            ncig[idx] = 3
            seq[idx] = s_new
        R["cig_off"] = np.concatenate([[0], np.cumsum(ncig)]).astype(np.int64)
        R["cig"] = np.ascontiguousarray(cigs[np.arange(3)[None, :] < ncig[:, None]])
        R["n_indel_reads"] = int((ncig == 3).sum())
    R["seq"] = np.ascontiguousarray(seq.reshape(-1))
    R["target"] = target
    return R


def genome_baq_roofline(args, cfg_name, cfg, caller, la, R, dev):
    """roofline of the dominant kernel of a reads -> VCF run, lfq_baq_reg_kernel: one bin's BAQ (+ IDAQ) call on a resident
    read set (uploads done), HIP events around its kernels on the stream they are launched on (lfq_last_baq_times), the
    traffic and instruction counters from rocprofv3 --pmc passes over a child of this command.  The kernel is bound by what it
    issues (FP64, one wavefront per SIMD), not by HBM: `frac` (algorithmic bytes against the HBM peak) is small by nature;
    `issue` says how busy the vector units were."""
    import torch
    rs = la.ReadSet.from_arrays(caller, R)
    rs.baq(extended=True, idaq=cfg["call_indels"])
    torch.cuda.synchronize(dev)
    ms, t = [], None
    for _ in range(3):
        rs.baq(extended=True, idaq=cfg["call_indels"])
        t = caller.baq_times()
        ms.append(t["ms_kernels"])
    rs.close()
    call_ms = float(np.median(ms))
    n_indel = int(R.get("n_indel_reads", 0)) if cfg["call_indels"] else 0
    # seq + qual in, one reference window per read in (read length + 2 x 7 band), lb out; ai / ad out for the reads with an indel
    alg = 3 * t["n_bases"] + t["n_reads"] * (150 + 14) + 2 * 150 * n_indel
    achieved = alg / (call_ms * 1e-3) / 1e9
    pmc = None
    if not args.no_pmc:
        child = ["--config", cfg_name, "--genome-scale", str(args.genome_scale)]
        f = pmc_pass("FETCH_SIZE", child)
        w = pmc_pass("WRITE_SIZE", child) if f is not None else None
        v = pmc_pass("SQ_INSTS_VALU", child) if f is not None else None
        if f and w and v:
            baq = lambda d: sum(x[0] for k, x in d.items() if k.startswith("lfq_baq_"))
            calls = 2.0                             # the child runs two calls
            pmc = {"fetch_kib": baq(f) / calls, "write_kib": baq(w) / calls, "valu_insts": baq(v) / calls,
                   "source": "rocprofv3 --pmc passes spawned by this run (two BAQ calls of one bin; all lfq_baq_* kernels of a call summed)"}
    traffic = (2.0 * pmc["fetch_kib"] + pmc["write_kib"]) * 1024.0 if pmc else None
    return {
        "bound": "hbm", "kernel": "lfq_baq_reg_kernel (all instantiations of one bin's call: %d launches)" % t["n_launches"],
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
        "bytes_per_launch": alg, "launch": "one lfq_readset_baq call = one bin = %d reads" % t["n_reads"],
        "avg_launch_ms": call_ms, "ms_of_the_three_calls": ms,
        "traffic_frac": (traffic / (call_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
        "traffic_over_algorithmic_bytes": (traffic / alg) if traffic else None,
        "issue": ({"valu_wave_instructions": pmc["valu_insts"],
                   "frac_of_issue_slots": pmc["valu_insts"] / (N_SIMD * CLK_HZ / 4.0 * call_ms * 1e-3),
                   "note": "SQ_INSTS_VALU / (1024 SIMDs x clock / 4 x the call's kernel time): the kernel's real bound "
                           "(FP64 recurrence, -ffp-contract=off, one wavefront per SIMD: DESIGN 6b)"} if pmc else None),
        "pmc": pmc,
        "note": "bound by FP64 issue at one wavefront per SIMD, not by HBM; reads / s in the kernel: %.3g" % (t["n_reads"] / (call_ms * 1e-3)),
    }


def genome_cpu_baseline(cfg, sample_len):
    """the oracle's reads -> VCF chain (tests/oracle_chain.py: BAQ / IDAQ per read, compile_plp_col, call_indels + call_snvs,
    the epilogue of main_call) on ONE pinned core over a sample of the same workload: the first `sample_len` bases of a bin"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_chain as oc
    import pyoracle as orc
    orc.build()
    pinned = None
    old_aff = None
    try:
        old_aff = os.sched_getaffinity(0)
        pinned = sorted(old_aff)[0]
        os.sched_setaffinity(0, {pinned})
    except (AttributeError, OSError):
        pass
    try:
        R = make_tile(cfg, sample_len)
        P = dict(R)
        t0 = time.perf_counter()
        orc.baq_idaq_reads(P, extended=True, idaq=cfg["call_indels"], procs=1)
        t1 = time.perf_counter()
        kw = dict(flag=1 | 2 | (8 if cfg["call_indels"] else 0))
        if cfg["targets"]:
            tg = R["target"]
            edges = np.flatnonzero(np.diff(np.concatenate([[0], tg, [0]])))
            targets = [("synth", int(a), int(b)) for a, b in zip(edges[0::2], edges[1::2])]
            out = oc.call_targets(orc, P, R["ref"], targets, kw)
            ncols = int(tg.sum())
        else:
            out = oc.call_region(orc, P, R["ref"], 0, sample_len, kw, call_indels=cfg["call_indels"], no_default_filter=False)
            ncols = len(out["col_pos"])
        t2 = time.perf_counter()
    finally:
        if old_aff is not None:
            try:
                os.sched_setaffinity(0, old_aff)
            except OSError:
                pass
    return R, out["lines"], {"value": ncols / (t2 - t0), "unit": "columns/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
            "host_cpus": os.cpu_count(), "pinned_to_core": pinned,
            "sample": "the first %d bases of a bin of the same workload (%d reads, %d called columns): %.1f s on one pinned core; "
                      "BAQ%s %.1f s / pileup + calls %.1f s"
                      % (sample_len, int(R["n"]), ncols, t2 - t0, " + IDAQ" if cfg["call_indels"] else "", t1 - t0, t2 - t1),
            "reads_per_s": int(R["n"]) / (t2 - t0), "vcf_lines_of_the_sample": len(out["lines"])}


def genome_sample_device_lines(cfg, caller, la, R, sample_len, chrom="chr1"):
    """The device's reads -> VCF chain on the cpu_baseline's sample (the first `sample_len` bases of a bin as a region of its
    own, or its targets in genome order with ONE running Bonferroni factor: `lofreq call -l bed`), so that the VCF lines the
    oracle wrote while it was timed are compared with the device's: the concordance flag of a C4 / C5 line."""
    gs = lambda l: l.split(";HQA=")[0]
    flag = la.LFQ_USE_BAQ | la.LFQ_USE_MQ | (la.LFQ_USE_IDAQ if cfg["call_indels"] else 0)
    from lofreq_amd import _lib
    caller.set_dense_strand_counts(True)
    caller.set_dense_counts(True)
    _lib.load().lfq_set_indel_arrays_on_host(caller.h, 1)      # (the timed steps keep a bin's indel arrays in HBM)
    rs = la.ReadSet.from_arrays(caller, R)
    rs.baq(extended=True, idaq=cfg["call_indels"])
    conf = la.VarcallConf(flag=flag)
    lines = []
    if cfg["targets"]:
        tg = R["target"]
        edges = np.flatnonzero(np.diff(np.concatenate([[0], tg, [0]])))
        recs_all = []
        for b, e in zip(edges[0::2], edges[1::2]):
            dt = rs.pileup_snv(int(b), int(e))
            recs, _, _ = caller.call_snvs(dt, conf)
            recs_all += [(int(dt.col_pos[int(r["col"])]), r) for r in recs]
        thr = la.snvqual_thresh(conf.sig, conf.bonf_subst)
        for p0, r in recs_all:
            if la.filter_records(np.array([r]), thr, apply_defaults=False)[0]:
                lines.append((p0, 1, la.format_vcf(np.array([r]), chrom, pos0=np.array([p0]), filter_str="PASS").rstrip("\n")))
    else:
        if cfg["call_indels"]:
            cols, col_pos = rs.pileup_indels(0, sample_len)
            irecs, _ = la.call_indels(caller, cols, conf)
            ikeep = la.filter_indel_records(irecs, la.snvqual_thresh(conf.sig, conf.bonf_indel), apply_defaults=True)
            for r, k in zip(irecs, ikeep):
                if k:
                    p0 = int(col_pos[int(r["col"])])
                    lines.append((p0, 0, la.format_indel_record(chrom, p0, cols, r, "PASS").rstrip("\n")))
        dt = rs.pileup_snv(0, sample_len)
        if cfg["call_indels"]:
            la.skip_snv_columns(caller, cols.cons_indel)
        recs, _, _ = caller.call_snvs(dt, conf)
        keep = la.filter_records(recs, la.snvqual_thresh(conf.sig, conf.bonf_subst), apply_defaults=True)
        for r, k in zip(recs, keep):
            if k:
                p0 = int(dt.col_pos[int(r["col"])])
                lines.append((p0, 1, la.format_vcf(np.array([r]), chrom, pos0=np.array([p0]), filter_str="PASS").rstrip("\n")))
    rs.close()
    caller.set_dense_strand_counts(False)
    caller.set_dense_counts(False)
    return [gs(l[2]) for l in sorted(lines, key=lambda t: (t[0], t[1]))]


def bench_genome(args, cfg_name, caller, la, shard, dist, world, rank, dev, xdev, comm_ranks):
    """One step = the whole synthetic genome: every bin (plan_regions, dealt to the ranks) goes reads -> BAQ (+ IDAQ) ->
    device pileup(s) -> SNV (+ indel) tests on its owner's GPU; then the shard exchange (per-bin test counts ->
    exact Bonferroni prefix per bin, records to rank 0: shard.finish_bins / finish_indel_bins) and the VCF text of the whole
    genome on rank 0.  Strong scaling: the genome is the same whatever N is."""
    import ctypes as C
    import torch
    from lofreq_amd import _lib
    from lofreq_amd.pileup import DeviceTracks
    cfg = GENOME_CONFIGS[cfg_name]
    glen, nb = int(round(cfg["genome"] * args.genome_scale / cfg["bins"])) * cfg["bins"], cfg["bins"]
    # call-parallel's cut (lofreq2_call_pparallel.py:590-613) with the number of bins an 8-GPU node gets (>= 2 per GPU,
    # strictly below total / 16), whatever N is: the bins -- and with them the output -- do not depend on N
    bins, owner = shard.plan_regions([("synth", 0, glen)], lambda c, b, e: float(cfg["depth"]) * (e - b), world,
                                     bins_per_worker=max(1, 16 // world))
    assert len(bins) == nb and len({e - b for _, b, e in bins}) == 1, (len(bins), nb)
    tile_len = bins[0][2] - bins[0][1]
    # the reads: `--distinct-bins` sets from different seeds, bin i takes set i mod that many (by its index in the genome, so
    # that the output does not depend on which rank owns it)
    n_tiles = max(1, min(args.distinct_bins, nb))
    tiles, keep_all = [], []
    for k in range(n_tiles):
        Rk = make_tile(cfg, tile_len, seed=11 + 101 * k)
        keep = {}
        for key in ("seq", "qual", "bi", "bd", "pos", "cig_off", "cig", "seq_off", "mapq", "rev"):      # pinned, as the region binding keeps them
            if key in Rk:
                keep[key] = torch.from_numpy(Rk[key]).pin_memory()
                Rk[key] = keep[key].numpy()
        keep_all.append(keep)
        tiles.append(Rk)
    R = tiles[0]
    L = _lib.load()
    flag = la.LFQ_USE_BAQ | la.LFQ_USE_MQ | (la.LFQ_USE_IDAQ if cfg["call_indels"] else 0)
    my = [(i, b, e) for i, ((_, b, e), o) in enumerate(zip(bins, owner)) if o == rank]
    cap = tile_len
    n_thr = max(1, min(args.host_threads, len(my)))
    callers = [caller] + [la.SnvCaller(dev.index) for _ in range(n_thr - 1)]
    outs = []
    for c_ in callers:
        c_.set_dense_strand_counts(False)
        c_.set_dense_counts(False)                  # only the sparse output of a bin is read
        L.lfq_set_indel_arrays_on_host(c_.h, 0)
        if n_thr > 1 and not os.environ.get("LFQ_BENCH_SHARED_STREAM"):
            # a host thread per context: a thread's waits cover its own bins only, one thread's BAQ kernels run beside
            # another's pileups (the device's shared launch stream ran the four threads' launches in one line)
            c_.set_private_stream(True)
        outs.append((torch.zeros(cap * 64, dtype=torch.uint8, device=dev), torch.zeros(cap * 128, dtype=torch.uint8, device=dev)))
    if args.pmc_child:              # counter passes: two BAQ (+ IDAQ) calls of one bin, nothing else
        rs = la.ReadSet.from_arrays(caller, R)
        for _ in range(2):
            rs.baq(extended=True, idaq=cfg["call_indels"])
        torch.cuda.synchronize(dev)
        rs.close()
        return None

    # Read sets resident in HBM (the default): a host thread keeps, per set of reads it meets, TWO resident copies and takes them in
    # turn -- the BAQ of its next bin is queued while the pileups of the bin before still read theirs.  Nothing computed is kept
    # between bins: every bin runs BAQ / IDAQ, both pileups and the tests on the raw reads again.  `upload["on"]`: a fresh
    # lfq_readset_create (pinned host arrays -> HBM) per bin instead, what a caller that streams a BAM pays on top.
    upload = {"on": bool(args.upload_reads)}
    pools = [dict() for _ in callers]

    def start(caller, i, t=0, seq=0):
        if upload["on"]:
            rs = la.ReadSet.from_arrays(caller, tiles[i % n_tiles])
        else:
            key = (i % n_tiles, seq & 1)
            rs = pools[t].get(key)
            if rs is None:
                rs = pools[t][key] = la.ReadSet.from_arrays(caller, tiles[i % n_tiles])
        rs.baq(extended=True, idaq=cfg["call_indels"])
        return rs

    phase_ms = {}

    def lap(name, t_prev):
        t_now = time.perf_counter()
        if trace is not None:
            phase_ms[name] = phase_ms.get(name, 0.0) + 1e3 * (t_now - t_prev)
        return t_now

    def finish(caller, d_counts, d_pvals, rs, i, b):
        """pileups + tests of a started bin -> (snv entry for finish_bins, indel entry or None, indel lines' makings)"""
        ient = None
        tp = time.perf_counter()
        skip = None
        target = tiles[i % n_tiles]["target"]
        if cfg["call_indels"]:
            outp = C.POINTER(_lib.IndelColumnsC)()
            col_pos = np.zeros(tile_len, np.int64)
            _lib.check(L.lfq_readset_pileup_indels(caller.h, rs.h, 0, tile_len, 0, C.byref(outp), col_pos.ctypes.data),
                       "lfq_readset_pileup_indels")
            tp = lap("pileup_indels", tp)
            dt = rs.pileup_snv(0, tile_len)
            tp = lap("pileup_snv", tp)
            ci = la.VarcallConf(flag=flag)
            rec = np.zeros(1 << 18, dtype=_lib.INDEL_RECORD_DTYPE)
            nrec, nt = C.c_int64(0), C.c_int64(0)
            _lib.check(L.lfq_call_indels_batch(caller.h, C.byref(ci.c), outp, rec.ctypes.data, len(rec), C.byref(nrec), C.byref(nt)),
                       "lfq_call_indels_batch")
            tp = lap("call_indels", tp)
            oc = outp.contents
            rec = rec[: nrec.value].copy()
            # what the VCF line of a record needs besides the record: reference base and the event's key
            lines = []
            rb = np.frombuffer(C.string_at(oc.ref_base, oc.ncols), np.uint8)
            for r in rec:
                sd = oc.side[int(r["side"])]
                ko = np.frombuffer(C.string_at(sd.key_off + 8 * int(r["event"]), 16), np.int64)
                key = C.string_at(sd.key_chars + int(ko[0]), int(ko[1] - ko[0])).decode()
                base = chr(int(rb[int(r["col"])]))
                lines.append((base + key, base) if r["side"] == 0 else (base, base + key))     # (alt, ref) / (alt, ref)
            rec["col"] = col_pos[rec["col"]]                      # column of the bin -> offset in the bin
            ient = (i, b, rec, int(nt.value), lines)
            skip = np.frombuffer(C.string_at(oc.cons_indel, oc.ncols), np.uint8).copy()
            tp = lap("indel python", tp)
        else:
            dt = rs.pileup_snv(0, tile_len)
            tp = lap("pileup_snv", tp)
        if target is not None:
            off = 1 - target[dt.col_pos]                          # `-l bed`: the columns outside the targets are not called
            skip = off if skip is None else (skip | off)
        if skip is not None:
            la.skip_snv_columns(caller, skip)
        conf = la.VarcallConf(flag=flag)
        n = int(dt.ncols)
        tp = lap("skip columns", tp)
        caller.snv_batch_device(dt, conf, d_counts, d_pvals, cap)
        st = caller.batch_finish()
        tp = lap("snv calls", tp)
        pv = d_pvals[: st.n_pvals * 128].cpu().numpy().view(la.COL_PVALS_DTYPE).copy()
        pv["col"] = dt.col_pos[pv["col"]]
        ncalled = n if target is None else int(target[dt.col_pos].sum())
        if not any(rs is x for pool in pools for x in pool.values()):
            rs.close()
        lap("records d2h", tp)
        return (i, b, pv, int(st.n_tested)), ient, ncalled

    def work(t, res):
        """host thread t: its bins one after the other on its own context; bin k + 1 is started (upload, BAQ kernels queued)
        before bin k is finished"""
        try:
            c_, (dc, dp) = callers[t], outs[t]
            pending = None
            for seq, (i, b, e) in enumerate(my[t::n_thr]):
                t_s = time.perf_counter()
                rs = start(c_, i, t, seq)
                lap("start (readset + BAQ launch)", t_s)
                if pending is not None:
                    res.append(finish(c_, dc, dp, *pending))
                pending = (rs, i, b)
            if pending is not None:
                res.append(finish(c_, dc, dp, *pending))
        except BaseException as ex:                  # the step must not hang on a dead thread
            res.append(ex)

    trace = [] if os.environ.get("LFQ_BENCH_TRACE_STEPS") else None

    def step():
        import threading
        ts = [time.perf_counter()]
        results = [[] for _ in range(n_thr)]
        th = [threading.Thread(target=work, args=(t, results[t])) for t in range(1, n_thr)]
        for x in th:
            x.start()
        work(0, results[0])
        for x in th:
            x.join()
        ts.append(time.perf_counter())
        snv, ind, ncols = [], [], 0
        for s_, i_, n_ in sorted((r for res in results for r in ([r_ for r_ in res if not isinstance(r_, BaseException)])), key=lambda r: r[0][0]):
            snv.append(s_); ncols += n_
            if i_ is not None:
                ind.append(i_)
        for res in results:
            for r_ in res:
                if isinstance(r_, BaseException):
                    raise r_
        conf = la.VarcallConf(flag=flag)
        recs, total = shard.finish_bins(conf, snv, nb, dist if world > 1 else None, xdev)
        ts.append(time.perf_counter())
        text = None
        itext = []
        if cfg["call_indels"]:
            # the exact factor of every indel test needs every bin's count; the surviving records are formatted where
            # their event keys live (the owner) and travel as text, like the per-bin VCFs call-parallel concatenates
            counts = np.zeros(nb, np.int64)
            for i, _, _, nt, _ in ind:
                counts[i] = nt
            allc, _ = shard.exchange_counts(counts, dist if world > 1 else None, xdev)
            per_bin = allc.sum(axis=0)
            prefix = np.concatenate([[0], np.cumsum(per_bin)[:-1]])
            tot_i = int(per_bin.sum())
            conf.c.bonf_indel = 1 + tot_i
            conf.c.num_indel_tests += tot_i
            thr_i = la.snvqual_thresh(conf.sig, conf.bonf_indel)
            for i, b, rec, nt, ra in ind:
                rec = rec.copy()
                rec["bonf"] += int(prefix[i])
                ok = rec["pvalue"] * rec["bonf"].astype(np.longdouble) < np.float32(conf.sig)
                ok &= la.filter_indel_records(rec, thr_i, apply_defaults=True).astype(bool)
                for r, (alt, ref), k_ in zip(rec, ra, ok):
                    if k_:
                        p0 = b + int(r["col"])
                        buf = C.create_string_buffer(1024 + len(ref) + len(alt))
                        L.lfq_format_indel_record(buf, len(buf), b"synth", p0, ref.encode(), alt.encode(), int(r["qual"]), int(r["dp"]),
                                                  C.c_float(float(r["af"])), int(r["sb"]), int(r["ref_fw"]), int(r["ref_rv"]),
                                                  int(r["alt_fw"]), int(r["alt_rv"]), int(r["hrun"]), b"PASS")
                        itext.append((p0, buf.value.decode()))
            if world > 1:
                gathered = [None] * world if rank == 0 else None
                dist.gather_object(itext, gathered, dst=0)
                if rank == 0:
                    itext = [x for part in gathered for x in part]
        ts.append(time.perf_counter())
        if rank == 0:
            thr = la.snvqual_thresh(conf.sig, conf.bonf_subst)
            keepm = la.filter_records(recs, thr, apply_defaults=True)
            stext = la.format_vcf(recs, "synth", keep=keepm, filter_str="PASS").splitlines(True)
            spos = [int(r["col"]) for r, k_ in zip(recs, keepm) if k_]
            merged = sorted([(p, 0, t) for p, t in itext] + [(p, 1, t) for p, t in zip(spos, stext)], key=lambda x: (x[0], x[1]))
            text = "".join(t for _, _, t in merged)
        if trace is not None:
            ts.append(time.perf_counter())
            trace.append([1e3 * (b_ - a_) for a_, b_ in zip(ts, ts[1:])])
            sys.stderr.write("[genome step] bins (threads) %.1f  finish_bins %.1f  indel exchange + text %.1f  SNV text + merge %.1f ms\n" % tuple(trace[-1]))
            sys.stderr.write("   per bin, ms (sum over the threads / bins): " + "  ".join("%s %.2f" % (k, v / nb) for k, v in phase_ms.items()) + "\n")
            phase_ms.clear()
        return conf, ncols, text, (len(recs) if recs is not None else 0, len(itext))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 1)):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        conf, ncols_mine, text, nrecs = step()
    barrier()
    dt = time.perf_counter() - t0
    dt_upload = None
    if not upload["on"] and not args.no_upload_rate:
        # the PCIe-inclusive rate beside it: the same steps with every bin's reads uploaded again (never `value`)
        upload["on"] = True
        step()
        barrier()
        t1 = time.perf_counter()
        n_up = max(1, min(args.steps, 3))
        for _ in range(n_up):
            _, _, text_up, _ = step()
        barrier()
        dt_upload = (time.perf_counter() - t1) / n_up
        upload["on"] = False
        if rank == 0 and text_up != text:
            raise RuntimeError("resident and uploaded read sets give different VCFs")
    for pool in pools:
        for rs_ in pool.values():
            rs_.close()
        pool.clear()
    tot_cols = torch.tensor([float(ncols_mine)], dtype=torch.float64, device=xdev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=xdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot_cols)
    dt, called = float(tmax.item()), int(tot_cols.item())
    roof = base = concord = None
    if rank == 0 and world == 1:
        try:
            roof = genome_baq_roofline(args, cfg_name, cfg, caller, la, R, dev)
        except Exception as e:          # the measured line survives a failed counter pass
            roof = {"error": repr(e)}
        if not args.no_cpu_baseline:
            try:
                sample_len = min(tile_len, args.cpu_sample_cols or (60000 if cfg["call_indels"] else 400000))
                Rs, olines, base = genome_cpu_baseline(cfg, sample_len)
                # the lines the oracle wrote while it was timed against the device chain on the same sample
                gs = lambda l: l.split(";HQA=")[0]
                dlines = genome_sample_device_lines(cfg, caller, la, Rs, sample_len)
                ol = [gs(l) for l in olines]
                concord = {"sample": "the cpu_baseline's sample as a run of its own", "oracle_lines": len(olines),
                           "device_lines": len(dlines), "identical": ol == dlines,
                           "only_oracle": [l for l in ol if l not in set(dlines)][:6],
                           "only_device": [l for l in dlines if l not in set(ol)][:6]}
            except Exception as e:
                base = {"error": repr(e)}
    L.lfq_set_indel_arrays_on_host(caller.h, 1)
    for c_ in callers[1:]:
        c_.close()
    if rank != 0:
        return None
    import hashlib
    return {
        "metric": "pileup columns/sec at depth %d, reads -> VCF (BAQ + device pileup + calls)" % cfg["depth"],
        "value": called * args.steps / dt, "unit": "columns/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic (%d distinct sets of reads, generated from different seeds, cycled over the genome's %d bins)" % (n_tiles, nb),
        "roofline": roof, "cpu_baseline": base,
        "config": {"workload": "%s: %s" % (cfg_name, cfg["what"]), "genome_len": glen, "called_columns": called, "bins": nb,
                   "bins_rank0": len(my), "host_threads_per_rank": n_thr, "bin_len": tile_len, "reads_per_bin": int(R["n"]), "reads_per_step": sum(int(tiles[i % n_tiles]["n"]) for i in range(nb)),
                   "distinct_bins": n_tiles,
                   "reads": ("uploaded from pinned host memory for every bin inside the timed region (--upload-reads)" if args.upload_reads else
                             "resident in HBM when the timed region starts (two copies per host thread and set, uploaded in the warm-up); "
                             "every bin runs BAQ, pileups and tests on the raw reads again"),
                   "ms_per_step_with_upload": (1e3 * dt_upload if dt_upload is not None else None),
                   "value_with_upload": (called / dt_upload if dt_upload else None),
                   "rccl_ranks": comm_ranks, "exchange_backend": (dist.get_backend() if world > 1 else None),
                   "snv_tests": int(conf.num_snv_tests), "indel_tests": int(conf.num_indel_tests),
                   "snv_records_before_filter": nrecs[0], "indel_records": nrecs[1],
                   "vcf_lines": text.count("\n"), "vcf_sha256": hashlib.sha256(text.encode()).hexdigest(),
                   "vcf_concordance": concord, "vcf_identical": (concord or {}).get("identical"),
                   "records_compared": (concord or {}).get("oracle_lines"),
                   "speedup_vs_cpu_1thread": (called * args.steps / dt / base["value"]) if (base and base.get("value")) else None,
                   "note": "1 step = the whole genome; the dominant kernel is lfq_baq_reg_kernel, FP64-issue-bound (DESIGN 6b): "
                           "`roofline` is one bin's BAQ call on a resident read set"},
    }, text


def other_configs(budget_s=420.0):
    """BASELINE.json configs[1], [3], [4] (C2, C4, C5) as short runs of this same script in child processes, after the C3
    line's own work: their figures go into the C3 line's `config` as SCALARS (c2_*, c4_*, c5_*), so that one driver run
    witnesses every config with its concordance flag.  A child that fails or runs out of time leaves `<cfg>_error`."""
    runs = [("c1", ["--config", "C1", "--steps", "2", "--warmup", "1"], 120),
            ("c2", ["--config", "C2", "--steps", "60", "--warmup", "5", "--no-secondary", "--no-pmc", "--no-other-configs"], 150),
            ("c4", ["--config", "C4", "--steps", "2", "--warmup", "1", "--no-pmc", "--cpu-sample-cols", "30000"], 200),
            ("c5", ["--config", "C5", "--steps", "2", "--warmup", "1", "--no-pmc", "--cpu-sample-cols", "150000"], 200)]
    out = {}
    t_start = time.perf_counter()
    for name, argv, limit in runs:
        left = budget_s - (time.perf_counter() - t_start)
        if left < 30:
            out[name + "_error"] = "skipped: the run's time budget for the other configs is spent"
            continue
        t0 = time.perf_counter()
        p = None
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, capture_output=True, text=True,
                               timeout=min(limit, left), cwd=ROOT)
            ln = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        except Exception as e:
            out[name + "_error"] = (repr(e) + (" | stderr: " + p.stderr[-400:] if p is not None else ""))[:700]
            continue
        cfg, roof, base = ln.get("config") or {}, ln.get("roofline") or {}, ln.get("cpu_baseline") or {}
        out[name + "_workload"] = cfg.get("workload")
        out[name + "_columns_per_s"] = ln.get("value")
        out[name + ("_ms_per_step" if name == "c2" else "_ms_per_genome")] = ln.get("ms_per_step")
        out[name + "_steps"] = ln.get("steps")
        out[name + "_vcf_identical"] = cfg.get("vcf_identical")
        out[name + "_records_compared"] = cfg.get("records_compared")
        out[name + "_roofline_frac"] = roof.get("frac")
        out[name + "_roofline_kernel"] = roof.get("kernel")
        out[name + "_cpu_baseline_columns_per_s"] = base.get("value")
        out[name + "_child_seconds"] = round(time.perf_counter() - t0, 1)
        if name == "c1":
            out["c1_reference"] = cfg.get("reference")
            for k in ("c1_roofline_frac", "c1_roofline_kernel", "c1_cpu_baseline_columns_per_s"):
                out.pop(k, None)
        elif name == "c2":
            out["c2_roofline_kernel_alone_frac"] = (roof.get("kernel_alone") or {}).get("frac")
            out["c2_step_frac_of_hbm_peak"] = (roof.get("step") or {}).get("frac")
            out["c2_columns_compared"] = cfg.get("columns_compared")
        else:
            out[name + "_ms_per_genome_with_upload"] = cfg.get("ms_per_step_with_upload")
            out[name + "_vcf_sha256"] = cfg.get("vcf_sha256")
            out[name + "_vcf_lines"] = cfg.get("vcf_lines")
            out[name + "_baq_call_ms"] = roof.get("avg_launch_ms")
    return out


def _rccl_options(dist):
    """The step's exchange is a few 8-byte .. kilobyte collectives between count kernels that keep every wave slot of the
    device taken: on a stream of normal priority an RCCL workgroup waits for a slot like a DP workgroup would (measured
    with a one-rank communicator, LFQ_BENCH_FORCE_DIST=1: profiles/NOTES.md) -- the communicator's stream gets the
    priority the DP streams have.  LFQ_BENCH_RCCL_PRIO=0 for the A/B run."""
    if os.environ.get("LFQ_BENCH_RCCL_PRIO", "1") == "0":
        return None
    opts = dist.ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = True
    return opts


def _chain_worker(idx, iters, barrier, queue):
    """one region worker of `--mode chain --workers W`: its own process, context and read set on GPU 0"""
    try:
        import torch
        torch.cuda.set_device(0)
        import lofreq_amd as la
        caller = la.SnvCaller(0)
        caller.set_dense_strand_counts(False)
        res = bench_chain(caller, la, 2000000, 1000000, iters, start_barrier=barrier)
        caller.close()
        queue.put((idx, res))
    except BaseException as e:                       # the parent must not wait for ever
        try:
            barrier.abort()
        except Exception:
            pass
        queue.put((idx, {"error": repr(e)}))


def chain_workers(n_workers, iters):
    """W region workers on one GPU, the way `lofreq call-parallel` runs one process per bin
    (lofreq2_call_pparallel.py:590-707): every process owns a context and works through its regions; the host part of one
    worker's region runs under the kernels of another's.  Aggregate = regions of all workers / (last end - first start)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    barrier = ctx.Barrier(n_workers)
    queue = ctx.Queue()
    procs = [ctx.Process(target=_chain_worker, args=(i, iters, barrier, queue)) for i in range(n_workers)]
    for p in procs:
        p.start()
    import queue as pyqueue
    results = []
    deadline = time.time() + 420
    while len(results) < n_workers and time.time() < deadline:
        try:
            results.append(queue.get(timeout=1.0))
        except pyqueue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):      # a worker died without reporting
                break
    if len(results) < n_workers or any("error" in r for _, r in results):
        try:
            barrier.abort()
        except Exception:
            pass
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=30)
        bad = [r["error"] for _, r in results if "error" in r]
        raise RuntimeError("chain worker failed: %s" % (bad[0] if bad else "no result (exit codes %s)" % [p.exitcode for p in procs]))
    for p in procs:
        p.join(timeout=60)
    res = [r for _, r in sorted(results, key=lambda x: x[0])]
    span = max(r["wall_end"] for r in res) - min(r["wall_begin"] for r in res)
    regions = sum(r["iterations"] for r in res)
    return {"workers": n_workers, "regions": regions, "s_span": span, "s_per_region": span / regions,
            "reads_per_s": regions * res[0]["reads"] / span, "columns_per_s": sum(r["columns"] * r["iterations"] for r in res) / span,
            "s_mean_per_worker": [r["s_mean"] for r in res], "columns": res[0]["columns"], "reads": res[0]["reads"],
            "indel_tests": res[0]["indel_tests"], "snv_records": res[0]["snv_records"],
            "note": "W processes x one context each on one GPU, each working through its own regions (same synthetic region); "
                    "aggregate over the span from the first timed start to the last end"}


def full_check(args, caller, la, batch, d_counts, d_pvals, pv_cap, seed, depth, ncols, default_filter, recs, text):
    """The batch of the timed steps, whole, against the oracle on all host cores (oracle/full_check.py): the dense
    integer outputs of the SAME count-kernel instantiation the timed steps ran (layer 1 on the resident batch), the
    records and the VCF text of the last timed step.  On a host too small to finish in about a minute: every planted
    column plus a stride."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import full_check as fc
    import pyoracle as orc
    orc.build()
    conf = la.VarcallConf()
    caller.set_dense_counts(True)                 # every column's entry is compared, not only the tested ones'
    caller.snv_batch_device(batch, conf, d_counts, d_pvals, pv_cap)
    st = caller.batch_finish()
    caller.set_dense_counts(False)
    counts = d_counts[: ncols * 64].cpu().numpy().view(la.COL_COUNTS_DTYPE).copy()
    procs = args.full_check_procs or fc.default_procs()
    est_s = ncols * depth * 1.7e-7 / procs              # ~1.6 ms per 10 000x column on one core of the bench hosts
    columns = None
    if est_s > 75.0:
        stride = int(est_s / 45.0) + 1
        planted = np.arange(0, ncols, max(args.plant_period, 1))
        columns = np.union1d(planted, np.arange(0, ncols, stride))
    out = fc.check_batch(orc, seed, depth, args.plant_period, ncols, counts, recs, gpu_vcf_text=text,
                         default_filter=default_filter, procs=procs, columns=columns, lazy_raw=True)
    out["scope"] = "every column of the timed batch" if columns is None else \
        "planted columns + every %d-th column (host too small for all of them)" % stride
    out["tested_columns"] = int(st.n_tested)
    return out


def spawn_ranks(n):
    """re-exec this command under torch.distributed.run with n ranks on this node; -> exit code"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs across processes on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def bench_c1(args):
    """BASELINE.json configs[0] as far as it can exist here (the denv2 BAM is not in the reference tree): the C1-SHAPED run of the
    reference's own 2.1.4 binary that tests/golden/big_c1_default.json holds (10.7 kb, 1 000 - 5 000x, `lofreq call` defaults:
    on-the-fly extended BAQ, dynamic Bonferroni, the `lofreq filter` epilogue; oracle/make_golden.py --big-only).  The reads are
    regenerated (tests/golden_reads.py, SHA-256 of the SAM text checked), go through the device's reads -> VCF chain, and the VCF
    lines are compared with the binary's byte for byte (;HQA= is HEAD-only and stripped).  A parity line with a clock on it."""
    import hashlib
    import torch
    import lofreq_amd as la
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_reads as gr
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "big_c1_default.json")))
    R = gr.make_from_fixture(fx)
    sha_ok = gr.sam_sha256(R) == fx["sam_sha256"]
    caller = la.SnvCaller(0)
    cfg = {"call_indels": False, "targets": False}
    times = []
    lines = None
    for _ in range(max(args.warmup if args.warmup != 100 else 1, 1) + max(min(args.steps, 5), 1)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lines = genome_sample_device_lines(cfg, caller, la, R, R["glen"])
        times.append(time.perf_counter() - t0)
    caller.close()
    n_warm = max(args.warmup if args.warmup != 100 else 1, 1)
    t = float(np.mean(times[n_warm:]))
    same = lines == fx["vcf"]
    return {"metric": "pileup columns/sec, reads -> VCF, C1 shape (10.7 kb, 1 000 - 5 000x, lofreq call defaults)",
            "value": R["glen"] / t, "unit": "columns/s", "n_gpus": 1, "steps": len(times) - n_warm, "warmup": n_warm,
            "ms_per_step": 1e3 * t, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "seeded reads of tests/golden_reads.py; expected output written by the reference's lofreq 2.1.4 binary",
            "config": {"workload": "C1 shape: BASELINE.json configs[0] at its own size with seeded reads (the denv2 BAM is not in the "
                                   "reference tree); 1 step = upload + BAQ + pileup + calls + filter + VCF text of the whole genome",
                       "reads": int(R["n"]), "sam_sha256_matches_fixture": sha_ok,
                       "vcf_identical": bool(same and sha_ok), "records_compared": len(fx["vcf"]),
                       "reference": fx["reference_binary"], "vcf_sha256": hashlib.sha256("\n".join(lines).encode()).hexdigest(),
                       "reference_binary_seconds_in_the_build_container": fx.get("binary_seconds_in_the_build_container")}}


def main():
    args = parse_args()
    if args.config == "C1":
        print(json.dumps(bench_c1(args)))
        return
    genome_cfg = args.config in GENOME_CONFIGS
    if genome_cfg:
        if args.steps == 200:                              # the defaults are C3's: a genome step takes ~0.3-0.6 s
            args.steps = 5
        if args.warmup == 100:
            args.warmup = 1
        args.scaling = "strong"
    cfg_idx, cfg_depth, cfg_cols, cfg_filter, cfg_sample = CONFIGS["C3" if genome_cfg else args.config]
    depth = args.depth or cfg_depth
    ncols = args.cols or cfg_cols
    seed = seed_of(3 if args.config == "C3" else 2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: launch the N ranks (one process per GPU) the way the driver does --
        # the reference's parallel wrapper forks its own workers too (lofreq2_call_pparallel.py:590-667)
        raise SystemExit(spawn_ranks(args.gpus))
    if genome_cfg:
        # C4 / C5: four host threads, a context each, every context with a launch stream of its own (lfq_set_private_stream):
        # with the three shared DP streams and the upload streams that is more streams than the runtime's default four
        # hardware queues, and streams that share a queue wait for each other's launches (C4 286-302 ms per genome with four
        # queues, 270-273 with eight; the shared stream: 288-297 / 292-305).  Read by the HIP runtime when it starts.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or without a launcher)"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # LFQ_BENCH_ONE_GPU=1 (debugging the N > 1 loop on a one-GPU box): every rank on cuda:0, exchange over gloo
    one_gpu = os.environ.get("LFQ_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    xdev = "cpu" if one_gpu else dev              # where the exchanged tensors live
    comm_ranks = 1
    # LFQ_BENCH_FORCE_DIST=1 (with --shard-path at N = 1): a one-rank RCCL communicator and every collective of the N > 1
    # step inside the timed region -- what the exchange costs a rank per step, measurable on a one-GPU box
    force_dist = world == 1 and os.environ.get("LFQ_BENCH_FORCE_DIST") == "1" and not genome_cfg
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ["LFQ_SHARD_FORCE_COLLECTIVES"] = "1"       # (read when lofreq_amd.shard is imported, below)
        dist.init_process_group(backend="nccl", device_id=dev, rank=0, world_size=1, pg_options=_rccl_options(dist))
    if world > 1:
        if not one_gpu and torch.cuda.device_count() < world:
            raise SystemExit("bench.py: %d ranks but %d visible GPUs (LFQ_BENCH_ONE_GPU=1 puts every rank on cuda:0 "
                             "with the exchange over gloo)" % (world, torch.cuda.device_count()))
        if one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev, pg_options=_rccl_options(dist))
        # the communicator's own idea of the job: a sum of ones over its ranks (RCCL all-reduce over xGMI on GPUs)
        ones = torch.ones(1, dtype=torch.int64, device=xdev)
        dist.all_reduce(ones)
        comm_ranks = int(ones.item())
        if comm_ranks != world:
            raise SystemExit("bench.py: communicator spans %d ranks, expected %d" % (comm_ranks, world))

    import lofreq_amd as la
    from lofreq_amd import shard

    # The per-step test counts are host integers on every rank: they travel over a host-side group (gloo) instead of
    # being staged through a GPU whose wave slots the count kernels hold (profiles/NOTES.md: three blocking RCCL
    # collectives cost a rank ~1 ms per step); the records go to rank 0 by ONE asynchronous RCCL gather per step.
    # LFQ_BENCH_EXCHANGE=rccl: everything through the RCCL communicator (the A/B run).
    exchange = {"counts": None, "records": None}
    if (world > 1 or force_dist) and dist.get_backend() == "nccl":
        exchange = {"counts": "lfq_shard_exchange_counts over the RCCL communicator",
                    "records": "lfq_shard_gather_start / _wait: ncclAllGather of fixed-capacity pieces on a stream of its own, collected one step later"}
        if os.environ.get("LFQ_BENCH_EXCHANGE", "host") != "rccl":
            try:
                import socket
                try:
                    socket.gethostbyname(socket.gethostname())
                except OSError:
                    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # one node: the loopback interface will do
                shard.set_host_group(dist.new_group(backend="gloo"))
                exchange["counts"] = ("lfq_shard_exchange_counts over the library's shared-memory host transport (one node; "
                                      "a gloo group otherwise)")
            except Exception as e:                                         # no host transport: RCCL for the counts too
                sys.stderr.write("bench.py: no gloo group for the test counts (%r); using RCCL\n" % (e,))
    elif world > 1:
        exchange = {"counts": "lfq_shard_exchange_counts, host transport", "records": "lfq_shard_gather_start / _wait, host transport"}

    caller = la.SnvCaller(local_rank)
    shard.set_context(caller)                     # the context the exchange's RCCL road stages through (lfq_shard_*)
    caller.set_dense_strand_counts(False)         # DP4 only for the columns that emit (what layer 2 does by itself)
    caller.set_dense_counts(False)                # ... and dense entries only for the tested columns (likewise)

    if genome_cfg:
        out = bench_genome(args, args.config, caller, la, shard, dist, world, rank, dev, xdev, comm_ranks)
        if rank == 0 and out is not None:
            line, text = out
            if args.vcf_out:
                open(args.vcf_out, "w").write(text)
            print(json.dumps(line))
        if world > 1:
            shard.shutdown()
            dist.destroy_process_group()
        caller.close()
        return
    if args.mode in ("chain", "baq", "host-abi") and world > 1:
        raise SystemExit("bench.py: --mode %s measures one GPU; the sharded reads -> VCF runs are --config C4 / C5" % args.mode)
    if args.mode == "chain":
        iters = max(args.steps // 100, 2)
        if args.workers > 1:
            caller.close()
            res = chain_workers(args.workers, iters)
            per_region = res["s_per_region"]
        else:
            res = bench_chain(caller, la, 2000000, 1000000, iters, overlap=args.overlap_regions, pinned=not args.pageable)
            caller.close()
            per_region = res["s_mean"]               # mean over the timed regions (s_total: the fastest one, by step)
            res["columns_per_s_best"] = res["columns_per_s"]
            res["columns_per_s"] = res["columns"] / per_region
            res["reads_per_s"] = res["reads"] / per_region
        line = {"metric": "pileup columns/sec, reads -> VCF chain (BAQ + device pileup + SNV and indel calls)",
                "value": res["columns_per_s"], "unit": "columns/s", "n_gpus": 1, "steps": iters,
                "warmup": 2, "ms_per_step": per_region * 1e3, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": "chain: regions of 2 M reads x 150 bp over 1 Mb (depth 300), --call-indels, BAQ on; "
                                       "1 step = 1 region, %d region worker(s)%s"
                                       % (args.workers, ", regions overlapped" if args.overlap_regions else ""), **res}}
        print(json.dumps(line))
        return
    if args.mode == "baq":
        res = bench_baq(caller, la, 400000, 2000000, max(args.steps // 20, 3), want_idaq=bool(args.idaq))
        line = {"metric": "reads/sec through BAQ (kpa_ext_glocal per read), resident read set", "value": res["reads_per_s"],
                "unit": "reads/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["s_call"] * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": "baq: 400 K reads x 150 bp over 2 Mb", **res}}
        print(json.dumps(line))
        caller.close()
        return
    if args.mode == "host-abi":
        res = bench_host_abi(caller, la, seed_of(2), 1000, 200000, args.plant_period, max(args.steps // 20, 3))
        line = {"metric": "pileup columns/sec at depth 1000, host buffers through the C ABI", "value": res["columns_per_s"],
                "unit": "columns/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": res["ms_per_batch"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic",
                "config": {"workload": "host-abi: C2-shaped batches (200 k columns x 1000x) from host memory", **res}}
        print(json.dumps(line))
        caller.close()
        return

    # ---- this rank's columns ----
    my_bins = None
    if args.scaling == "strong" and world > 1:
        # ONE genome of `--cols` columns whatever N is (a fixed 8 Mb genome would not fit one GPU at 10000x: 280 GB of
        # tracks), cut the way call-parallel cuts it (lofreq2_call_pparallel.py:590-613): >= 2 bins per GPU, split
        # greedily by cost (uniform depth here: by length), dealt longest first
        total = ncols
        bins, owner = shard.plan_regions([("synth", 0, total)], lambda c, b, e: float(depth) * (e - b), world)
        my_bins = [(i, b, e) for i, ((_, b, e), o) in enumerate(zip(bins, owner)) if o == rank]
        n_bins_total = len(bins)
        my_cols = sum(e - b for _, b, e in my_bins)
        col_begin = my_bins[0][1]
        batches = [caller.synth_batch(seed, depth, e - b, plant_period=args.plant_period, col_begin=b,
                                      nt_packed=not args.nt_bytes) for _, b, e in my_bins]
        batch = batches[0]
        cap_cols = max(e - b for _, b, e in my_bins)
    else:
        col_begin, my_cols = rank * ncols, ncols
        batch = caller.synth_batch(seed, depth, my_cols, plant_period=args.plant_period, col_begin=col_begin,
                                   nt_packed=not args.nt_bytes)
        cap_cols = my_cols
    d_counts = torch.zeros(cap_cols * 64, dtype=torch.uint8, device=dev)
    pv_cap = cap_cols
    d_pvals = torch.zeros(pv_cap * 128, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)

    def step():
        """One pass: kernels, sparse results to the host, exact emit test, exchange, filter, VCF text."""
        conf = la.VarcallConf()                   # default sig, dynamic Bonferroni from 1
        if my_bins is not None:
            # strong scaling: every bin of this rank is a batch of its own; one exchange for all of them
            done = []
            for (i, b, e), bt in zip(my_bins, batches):
                caller.snv_batch_device(bt, conf, d_counts, d_pvals, pv_cap)
                st = caller.batch_finish()
                done.append((i, b, d_pvals[: st.n_pvals * 128].cpu().numpy().view(la.COL_PVALS_DTYPE).copy(),
                             int(st.n_tested)))
            recs, total = shard.finish_bins(conf, done, n_bins_total, dist, xdev)
        elif world == 1 and not args.shard_path:
            # layer 2 of the C ABI (lfq_call_snvs_batch): the whole call_snvs loop over the batch in one call
            recs, _, st = caller.call_snvs(batch, conf, records_capacity=1 << 16)
        else:
            # layer 1 + the shard exchange: the running Bonferroni factor needs every rank's tested-column count
            caller.snv_batch_device(batch, conf, d_counts, d_pvals, pv_cap)
            st = caller.batch_finish()
            pv = d_pvals[: st.n_pvals * 128].cpu().numpy().view(la.COL_PVALS_DTYPE)
            recs, total = shard.finish_shard(conf, pv, st.n_tested, None, col_begin,      # records carry their ref base
                                             dist if (world > 1 or force_dist) else None, xdev)
        text = None
        if rank == 0:
            # QUAL threshold from the final dynamic Bonferroni factor (lofreq_call.c:1519-1538), then `lofreq filter`
            # with (C2) or without (C3, --no-default-filter) its defaults
            thr = la.snvqual_thresh(conf.sig, conf.bonf_subst)
            keep = la.filter_records(recs, thr, apply_defaults=cfg_filter)
            text = la.format_vcf(recs, "synth", keep=keep, filter_str="PASS")
        return conf, st, recs, text, caller.kernel_times()

    if args.pmc_child:
        for _ in range(2):
            step()
        caller.close()
        return

    for _ in range(args.warmup):
        step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # N = 1: the layer-2 call in its two halves on two contexts (lfq_call_snvs_submit / _wait / _collect): the host
    # finish of step k (sparse D2H, exact emit test, strand bias, filter, VCF text) runs under the kernels of step k + 1.
    # Every step still does all of its work inside the timed region.  Whether step k + 1 is submitted after the kernels
    # of step k are done (one batch in flight) or before (two; its count kernel then gated on the device), and whether
    # the two steps' kernels ever run at the same time (gates "tail" / "none") or not ("end"), is in_flight below.
    pipelined = my_bins is None and not args.no_pipeline
    layer2 = world == 1 and not args.shard_path
    if pipelined:
        # one context per batch in flight: four in the automatic mode
        NCTX = max(args.in_flight, 2) if args.in_flight else 8
        callers = [caller] + [la.SnvCaller(local_rank) for _ in range(NCTX - 1)]
        for c_ in callers[1:]:
            c_.set_dense_strand_counts(False)
            c_.set_dense_counts(False)
        # the sharded step (layer 1 + exchange) pipelines the same way: every context has its own device-side outputs
        out_bufs = [(d_counts, d_pvals)] + [(torch.zeros_like(d_counts), torch.zeros_like(d_pvals)) for _ in range(NCTX - 1)]

        def submit(k):
            conf = la.VarcallConf()
            if layer2:
                callers[k % NCTX].call_snvs_submit(batch, conf)
            else:
                dc, dp = out_bufs[k % NCTX]
                callers[k % NCTX].snv_batch_device(batch, conf, dc, dp, pv_cap)
            return conf

        def finish_start(k, conf):
            """Step k up to the point where its records are known (layer 2) or on their way to rank 0 (sharded step)."""
            if layer2:
                recs, st = callers[k % NCTX].call_snvs_collect(conf, records_capacity=1 << 16)
                return conf, st, recs, callers[k % NCTX].kernel_times()
            # host + exchange half of the sharded step, under the kernels of the next ones: the running Bonferroni
            # factor needs every rank's tested-column count (one all-gather), rank 0 gets everybody's records (one gather,
            # started here and collected by finish_end one step later: the host never waits for the device in between)
            st = callers[k % NCTX].batch_finish()
            pv = out_bufs[k % NCTX][1][: st.n_pvals * 128].cpu().numpy().view(la.COL_PVALS_DTYPE)
            h, _total = shard.finish_shard_start(conf, pv, st.n_tested, None, col_begin,
                                                 dist if (world > 1 or force_dist) else None, xdev)
            return conf, st, h, callers[k % NCTX].kernel_times()

        def finish_end(item):
            conf, st, recs, kt_ = item
            if not layer2:
                recs = shard.finish_shard_wait(recs)
            text = None
            if rank == 0:
                thr = la.snvqual_thresh(conf.sig, conf.bonf_subst)
                keep = la.filter_records(recs, thr, apply_defaults=cfg_filter)
                text = la.format_vcf(recs, "synth", keep=keep, filter_str="PASS")
            return conf, st, recs, text, kt_

        # the sharded step's VCF of step k is written while step k + 1 runs (every step's inside the timed region: the last
        # one is collected before run_steps returns)
        lagged = not layer2 and os.environ.get("LFQ_BENCH_EXCHANGE_LAG", "1") != "0"

        def finish(k, conf, prev):
            """-> (what the step before left to collect or None, the finished step or None)"""
            cur = finish_start(k, conf)
            if not lagged:
                return None, finish_end(cur)
            return cur, (finish_end(prev) if prev is not None else None)

        def wait(k):
            if layer2:
                callers[k % NCTX].call_snvs_wait()
            # (layer 1: lfq_batch_finish waits for the batch's event itself)

        # how the loop keeps the device fed: `n` batches in flight and, with two, what the second one's count kernel waits for
        # on the device (lfq_set_batch_gate)
        in_flight = {"n": args.in_flight or 1, "gate": args.gate if args.gate != "auto" else "tail",
                     "note": "as given" if args.in_flight else None}

        # LFQ_BENCH_TRACE_STEPS=1: the host's side of every step of the timed blocks (wait / finish / submit, seconds) and the
        # batch's kernel time go to stderr afterwards -- where a slow step lost its time
        step_trace = [] if os.environ.get("LFQ_BENCH_TRACE_STEPS") else None
        alone_kt = {}           # kernel times of a warm-up block in which the batches' kernels did not overlap

        def set_mode(n, gate):
            in_flight["n"], in_flight["gate"] = n, gate
            for c_ in callers:
                c_.set_batch_gate(gate)

        set_mode(in_flight["n"], in_flight["gate"])

        # The loop's two host halves on two threads: a SUBMIT thread launches batch k + depth as soon as the context it takes
        # is free, the main thread waits for batch k and finishes it (sparse D2H, exact emit test, strand bias, filter, VCF
        # text).  The library's calls release the GIL (ctypes), contexts are thread-compatible (one per batch), so a step's
        # host cost becomes max(submit, finish) instead of their sum -- what paces the 1000x batches (0.17 + 0.33 ms of host
        # against a count kernel of 0.3 ms) -- and a finish that loses the CPU for a few milliseconds no longer delays the
        # launches behind it.  Every step is still submitted, waited for and finished inside the timed region.
        # Opt-in (`--submit-thread`): measured, the device paces both shapes and the extra thread buys nothing.
        import queue
        import threading
        sub = {"cmd": queue.SimpleQueue(), "done": queue.SimpleQueue(), "thread": None}

        def submit_loop():
            while True:
                k = sub["cmd"].get()
                if k is None:
                    return
                try:
                    t0_ = time.perf_counter()
                    conf_ = submit(k)
                    sub["done"].put((k, conf_, time.perf_counter() - t0_, None))
                except BaseException as e:          # handed to the main thread, which raises it
                    sub["done"].put((k, None, 0.0, e))

        two_threads = bool(args.submit_thread)
        if two_threads:
            sub["thread"] = threading.Thread(target=submit_loop, name="lfq-bench-submit", daemon=True)
            sub["thread"].start()

        def run_steps(n):
            acc = None
            out = None
            if in_flight["n"] >= 2 and two_threads:
                depth = in_flight["n"]
                for j in range(min(depth, n)):
                    sub["cmd"].put(j)
                prev = None
                for k in range(n):
                    tr0 = time.perf_counter()
                    kk, conf_k, t_sub, err = sub["done"].get()          # (submitted in order: kk == k)
                    if err is not None:
                        raise err
                    assert kk == k
                    wait(k)
                    tr1 = time.perf_counter()
                    prev, done = finish(k, conf_k, prev)
                    out = done or out
                    if k + depth < n:
                        sub["cmd"].put(k + depth)                       # the context of step k is free again
                    kt_ = (prev or done)[3 if prev else 4]
                    if step_trace is not None:
                        step_trace.append((tr1 - tr0, time.perf_counter() - tr1, t_sub, kt_["ms_total"]))
                    acc = dict(kt_) if acc is None else {x: acc[x] + kt_[x] for x in acc}
                if prev is not None:
                    out = finish_end(prev)
                return out, acc
            if in_flight["n"] >= 2:
                # one host thread (the default): batch k + depth is launched after batch k is finished; n submits and n
                # finishes, every batch complete inside the timed region.  Gate "end": a batch's count kernel starts when the
                # previous batch's last kernel is done (an event on the device, no host latency between the batches; one
                # batch's kernels at a time); "tail": when it is past its row-bound DP kernels (beside the folds and the
                # join); "none": as soon as the previous count kernel is done (beside all of that batch's DP kernels)
                depth = in_flight["n"]
                confs = {j: submit(j) for j in range(min(depth, n))}
                prev = None
                for k in range(n):
                    tr0 = time.perf_counter()
                    wait(k)
                    tr1 = time.perf_counter()
                    prev, done = finish(k, confs.pop(k), prev)
                    out = done or out
                    tr2 = time.perf_counter()
                    if k + depth < n:
                        confs[k + depth] = submit(k + depth)
                    kt_ = (prev or done)[3 if prev else 4]
                    if step_trace is not None:
                        step_trace.append((tr1 - tr0, tr2 - tr1, time.perf_counter() - tr2, kt_["ms_total"]))
                    acc = dict(kt_) if acc is None else {x: acc[x] + kt_[x] for x in acc}
                if prev is not None:
                    out = finish_end(prev)
                return out, acc
            pending = submit(0)
            prev = None
            for k in range(n):
                tr0 = time.perf_counter()
                wait(k)
                if not layer2:
                    callers[k % NCTX].synchronize()      # layer 1 has no separate wait: the kernels of step k are done here
                tr1 = time.perf_counter()
                nxt = submit(k + 1) if k + 1 < n else None
                tr2 = time.perf_counter()
                prev, done = finish(k, pending, prev)
                out = done or out
                kt_ = (prev or done)[3 if prev else 4]
                if step_trace is not None:
                    step_trace.append((tr1 - tr0, time.perf_counter() - tr2, tr2 - tr1, kt_["ms_total"]))
                acc = dict(kt_) if acc is None else {x: acc[x] + kt_[x] for x in acc}
                pending = nxt
            if prev is not None:
                out = finish_end(prev)
            return out, acc

        run_steps(max(args.warmup, 2))              # both contexts warm (workspace allocations)
        if not args.in_flight or (args.in_flight == 2 and args.gate == "auto"):
            # how many batches to keep in flight, and gated how, is the caller's choice and depends on the shape (overlap pays
            # where the DP tail is short latency-bound work next to a short count kernel: 1000x; the device-side gate at the
            # end of the previous batch takes the host's wake-up + launch latency out of every step): measured here,
            # outside the timed region
            # (first in the list = the default: batches one after another on the device, eight queued (four until round 6:
            # on some boxes one `finish` in six takes 7-16 ms instead of 0.4 -- the boxes grant 16 of 256 cores and have
            # neighbours -- which four queued batches, 11 ms of work, do not cover: profiles/r06_host_hiccups.md), so that
            # a host thread that loses the CPU for a few milliseconds does not leave the device idle; another mode has to beat the best one before it by 2 %.  "none": the count kernel of
            # batch k + 1 beside the DP kernels of batch k -- pays at 10 000x now that the count kernel runs 1024-thread
            # workgroups and the queue is deep enough to keep count kernels back to back; "tail" pays at 1000x)
            modes = [(2, "end"), (2, "tail"), (2, "none")] if args.in_flight == 2 else \
                    [(8, "end"), (8, "none"), (4, "none"), (3, "tail"), (1, "tail")]
            trial = {}
            for m in modes * 3:                     # (three rounds, the fastest of each mode: a box's first seconds are noisy)
                set_mode(*m)
                run_steps(2)
                torch.cuda.synchronize(dev)
                t0_ = time.perf_counter()
                _, acc_ = run_steps(16)
                torch.cuda.synchronize(dev)
                trial[m] = min(trial.get(m, 1e9), (time.perf_counter() - t0_) / 16)
                if m[1] == "end" or m[0] == 1:      # one batch's kernels at a time: the kernels' own durations
                    alone_kt.update({x: acc_[x] / 16 for x in acc_})
            if world > 1:                           # one choice for all ranks
                tt = torch.tensor([trial[m] for m in modes], dtype=torch.float64, device=xdev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                trial = {m: float(tt[i]) for i, m in enumerate(modes)}
            # (a mode has to beat the best one before it in the list by 2 % to be taken)
            best = modes[0]
            for m in modes[1:]:
                if trial[m] < 0.98 * trial[best]:
                    best = m
            set_mode(*best)
            in_flight["note"] = "chosen in the warm-up: " + ", ".join(
                "%.3f ms per step with %s" % (1e3 * trial[m], "one batch in flight" if m[0] == 1 else "%d, gate %s" % m)
                for m in modes)

    def timed_block():
        """EXACTLY --steps steps between two barrier + synchronize pairs -> (seconds: max over ranks, last step, kernel times)"""
        acc = None
        if os.environ.get("LFQ_BENCH_NO_GC"):
            import gc
            gc.collect()
            gc.disable()
        barrier()
        t0 = time.perf_counter()
        if pipelined:
            out, acc = run_steps(args.steps)
        else:
            for _ in range(args.steps):
                out = step()
                acc = out[4] if acc is None else {k: acc[k] + out[4][k] for k in acc}
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=xdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out, acc

    # the reported value is the FIRST block after the warm-up; the further blocks only show the run's own spread
    elapsed, (conf, st, recs, text, kt), kt_acc = timed_block()
    block_s = [elapsed]
    for _ in range(max(args.repeats, 1) - 1):
        block_s.append(timed_block()[0])
    work = (callers[(args.steps - 1) % NCTX] if pipelined else caller).dp_work()
    if pipelined and step_trace:
        tail = step_trace[-len(block_s) * args.steps:]
        for i, (w, f, sb, kms) in enumerate(tail):
            sys.stderr.write("[step %3d] wait %.3f  finish %.3f  submit %.3f  sum %.3f ms   kernels %.3f ms\n"
                             % (i, 1e3 * w, 1e3 * f, 1e3 * sb, 1e3 * (w + f + sb), kms))

    final_line = None
    if rank == 0:
        steps = max(args.steps, 1)
        ms_per_step = 1e3 * elapsed / steps
        total_cols = ncols * world if my_bins is None else ncols    # weak: a shard per GPU; strong: one genome
        value = total_cols * steps / elapsed
        kt = {k: v / steps for k, v in kt_acc.items()}
        n_launch = max(int(round(kt["n_segments"])), 1)       # count-kernel launches per step
        # Dominant kernel = the one with the largest duration per step: the count kernel (one launch, HBM-bound).
        # The DP kernels run concurrently on three streams; their span is the `dp` block below.
        # packed nt + lazy record counts: lfq_count_lean_kernel<one BQ threshold, columns per workgroup, chunks in flight per lane>;
        # the byte layout: lfq_count_fast_kernel<packed, strand planes, one BQ threshold, columns per workgroup>
        wpw = os.environ.get("LFQ_COUNT_WAVES_PER_WG", "16")
        wpw = wpw if wpw in ("4", "8") else "16"
        ahead = os.environ.get("LFQ_COUNT_AHEAD_DEEP", "2")
        ahead = ahead if ahead in ("3", "4") else "2"
        count_name = ("lfq_count_fast_kernel<false, false, true, %s>" % wpw) if args.nt_bytes else \
            ("lfq_count_lean_kernel<true, %s, %s>" % (wpw, ahead))
        if depth < 4096:
            lpg = 4 if depth <= 320 else 8 if depth <= 900 else 16     # lfq_launch_count's choice at the default knobs
            count_name = ("lfq_count_multi_kernel<false, false, %d>" if args.nt_bytes else "lfq_count_shallow_kernel<false, %d>") % lpg
        dom_ms = kt["ms_count"] / n_launch
        moved = (work["bytes_read_count"] + work["bytes_written_count"]) / n_launch     # layout bytes, this launch
        alg_bytes = my_cols * (4.0 * depth + 80.0) / n_launch
        achieved = moved / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        pmc = None
        if world == 1 and not args.no_pmc:
            child = ["--config", args.config, "--plant-period", str(args.plant_period)]
            if args.depth:
                child += ["--depth", str(args.depth)]
            if args.cols:
                child += ["--cols", str(args.cols)]
            if args.nt_bytes:
                child += ["--nt-bytes"]
            pmc = live_pmc(child, count_name)
        traffic = pmc["traffic"] if pmc else None
        if pmc and pmc.get("kernel"):
            count_name = pmc["kernel"]
        overlapped = bool(pipelined and in_flight["n"] >= 2 and in_flight["gate"] != "end")
        # kernels of consecutive batches beside each other: a kernel's duration is then what it takes while sharing the machine;
        # the DP span and the count kernel's own rate are those of the warm-up block that ran batch after batch
        dp_ms = alone_kt["ms_dp"] if (overlapped and alone_kt.get("ms_dp")) else kt["ms_dp"]
        valu_busy = None
        if pmc and pmc.get("valu_insts_dp") and dp_ms > 0:
            # wave-instructions issued by the DP kernels / what 1024 SIMDs can issue in the DP span (one VALU
            # wave-instruction occupies its SIMD for 4 cycles)
            valu_busy = pmc["valu_insts_dp"] / (N_SIMD * CLK_HZ / 4.0 * dp_ms * 1e-3)
        line = {
            "metric": "pileup columns/sec at depth %d" % depth,
            "value": value, "unit": "columns/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "repeats": {"blocks": len(block_s), "steps_per_block": args.steps,
                        "ms_per_step_first": 1e3 * block_s[0] / steps,
                        "ms_per_step_min": 1e3 * min(block_s) / steps,
                        "ms_per_step_median": 1e3 * float(np.median(block_s)) / steps,
                        "ms_per_step_max": 1e3 * max(block_s) / steps,
                        "note": "`value` / `ms_per_step` are the first block (the contract's K steps after W warm-up steps); "
                                "the other blocks are the same K steps timed again"},
            "config": {
                # a --depth / --cols override is a shape of its own, not the BASELINE config whose defaults it started from
                "workload": ("%s: synthetic %.0f Mb genome per GPU, uniform %dx depth, SNV-only, %s, dynamic Bonferroni "
                             "(BASELINE.json configs[%d])"
                             % (args.config, ncols / 1e6, depth,
                                "default filter applied" if cfg_filter else "--no-default-filter", cfg_idx))
                            if not (args.depth or args.cols) else
                            ("custom shape (no BASELINE config): synthetic %.2f Mb genome per GPU, uniform %dx depth, SNV-only, %s, "
                             "dynamic Bonferroni" % (ncols / 1e6, depth,
                                                     "default filter applied" if cfg_filter else "--no-default-filter")),
                "columns_per_gpu": my_cols, "bins_rank0": len(my_bins) if my_bins is not None else 1, "depth": depth, "planted_snv_period": args.plant_period,
                "sharding": "region shard per GPU; per step one test-count all-gather and one record gather to rank 0 (`exchange`)",
                "rccl_ranks": comm_ranks, "exchange_backend": (dist.get_backend() if (world > 1 or force_dist) else None),
                "exchange": exchange,
                "records_per_step": int(len(recs)), "tested_columns_rank0": int(st.n_tested),
                "nt_layout": "bytes" if args.nt_bytes else "packed nibbles (LFQ_TRACKS_NT_PACKED)",
                "kernel_ms": kt,
                "pipeline": ("a context per queued batch: host finish of step k under the kernels of the steps behind it; batches in flight: %d%s (%s)"
                             % (in_flight["n"], "" if in_flight["n"] == 1 else ", gate " + in_flight["gate"], in_flight["note"]))
                            if pipelined else "none",
                "batches_in_flight": in_flight["n"] if pipelined else 1,
                "batch_gate": (in_flight["gate"] if in_flight["n"] >= 2 else None) if pipelined else None,
                # with two batches in flight and a gate other than "end" the kernels of consecutive steps overlap: the
                # per-step kernel times then sum to more than the step, and ms_step - ms_kernels is not a host time
                "kernel_times_overlap": bool(pipelined and in_flight["n"] >= 2 and in_flight["gate"] != "end"),
                "ms_kernels": kt["ms_total"],
                "host_ms_per_step_not_hidden": (ms_per_step - kt["ms_total"])
                                               if not (pipelined and in_flight["n"] >= 2 and in_flight["gate"] != "end") else None,
            },
            "roofline": {
                "bound": "hbm", "kernel": count_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "bytes_per_launch": moved,
                "bytes_note": "bytes this instantiation moves by layout: observations x (0.5 nt + 1 bq) + 9 B header in, "
                              "1 B class flag out per column + 64 B record out per column (per TESTED column where the "
                              "shared-wavefront kernel runs on the context's own dense array: nothing reads the others); "
                              "reported by the library per batch",
                "avg_launch_ms": dom_ms,
                "traffic_frac": (traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and dom_ms > 0 else None,
                "traffic_over_layout_bytes": (traffic / moved) if traffic and moved else None,
                "frac_of_measured_copy_peak": achieved / 6290.0,
                # the count kernel of batch k + 1 ran beside the DP kernels of batch k in the timed region (gate "none" /
                # "tail"): `achieved` above is its rate while sharing the machine; this is its rate in the warm-up block
                # that ran the same batches one after another (gate "end")
                "kernel_alone": ({"avg_launch_ms": alone_kt["ms_count"] / n_launch,
                                  "achieved": moved / (alone_kt["ms_count"] / n_launch * 1e-3) / 1e9,
                                  "frac": moved / (alone_kt["ms_count"] / n_launch * 1e-3) / 1e9 / HBM_PEAK_GBS}
                                 if (pipelined and overlapped and alone_kt.get("ms_count")) else None),
                # the whole step against the same peak: the bytes the count kernel moves / the step's wall time
                "step": {"achieved": moved * n_launch / (ms_per_step * 1e-3) / 1e9,
                         "frac": moved * n_launch / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "pmc": {k: pmc[k] for k in ("fetch_kib", "write_kib", "source")} if pmc else None,
                "algorithmic_8d": {"bytes_per_launch": alg_bytes, "achieved": alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0,
                                   "note": "SURVEY 8(d): 4*depth+80 B per column; the default filter configuration needs only the "
                                           "nt and bq tracks for the counts, so this ratio to peak can exceed 1"},
            },
            "dp": {
                "cells": work["cells"], "rows": work["rows"], "span_ms": dp_ms,
                "span_ms_timed_region": kt["ms_dp"],
                "cells_per_s": work["cells"] / (dp_ms * 1e-3) if dp_ms > 0 else None,
                "valu_busy": valu_busy,
                "valu_insts_per_step": pmc.get("valu_insts_dp") if pmc else None,
                "valu_by_kernel": pmc.get("valu_by_kernel") if pmc else None,
                "columns": {"light": work["n_light"], "mid": work["n_mid"], "big": work["n_big"],
                            "light_finished_by_retry_kernel": work["n_light_retry"]},
                "note": "cells = sum over tested columns of sum_{n<=N*} min(n, K), N* = this implementation's pruning row "
                        "(track order); device counters of the last step.  span = last count kernel's end -> all DP kernels done",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = min(args.cpu_sample_cols or cfg_sample, my_cols)
            base, ores = cpu_baseline(seed, depth, args.plant_period, sample, cfg_filter)
            line["cpu_baseline"] = base
            # VCF concordance on the sample: same records, same QUAL, from the full GPU run
            exp = [(c, int(ores["qual"][c, a])) for c in range(sample) for a in range(3) if ores["emitted"][c, a]]
            got = [(int(r["col"]), int(r["qual"])) for r in recs if r["col"] < sample]
            line["config"]["vcf_concordance"] = {"sample_columns": sample, "reference_records": len(exp),
                                                 "gpu_records": len(got), "identical": exp == got}
            line["config"]["speedup_vs_cpu_1thread"] = value / base["value"]
        if world == 1 and not args.no_full_check and my_bins is None:
            # the whole timed batch against the oracle: every column's integer outputs, every record, the VCF text
            # (the reference's own full-size check compares whole VCFs, tests/parallel.sh:40-51)
            try:
                line["config"]["vcf_concordance_sample"] = line["config"].get("vcf_concordance")
                line["config"]["vcf_concordance"] = full_check(args, caller, la, batch, d_counts, d_pvals, pv_cap, seed,
                                                                depth, my_cols, cfg_filter, recs, text)
            except Exception as e:
                line["config"]["vcf_concordance"] = {"error": repr(e), "identical": False}
        # the concordance of the timed batch once more as scalars (a reader that keeps only scalar fields of `config` still
        # sees what was compared and whether it was identical)
        vc = line["config"].get("vcf_concordance") or {}
        line["config"]["vcf_identical"] = vc.get("identical")
        line["config"]["records_compared"] = vc.get("records_compared", vc.get("gpu_records"))
        line["config"]["columns_compared"] = vc.get("columns_compared", vc.get("sample_columns"))
        line["config"]["max_dlogp_upto_600"] = vc.get("max_dlogp_upto_600")
        line["config"]["max_dlogp_beyond_600"] = vc.get("max_dlogp_beyond_600")
        line["config"]["max_dlogp_beyond_600_over_bound"] = vc.get("max_dlogp_beyond_600_over_bound")
        line["config"]["max_dlogp_device_vs_80bit_truth"] = vc.get("max_dlogp_device_vs_80bit_truth")
        line["config"]["n_pvalues_vs_80bit_truth"] = vc.get("n_pvalues_vs_80bit_truth")
        if world == 1 and not args.no_secondary:
            sec = {}
            try:
                sec["host_abi"] = bench_host_abi(caller, la, seed_of(2), 1000, 200000, args.plant_period, 5)
            except Exception as e:      # secondary figures never take the headline down
                sec["host_abi"] = {"error": repr(e)}
            try:
                sec["chain"] = bench_chain(caller, la, 2000000, 1000000, 2)
            except Exception as e:
                sec["chain"] = {"error": repr(e)}
            try:                        # one worker that starts region k + 1 before it finishes region k (lofreq_amd_region.c)
                sec["chain_overlapped"] = bench_chain(caller, la, 2000000, 1000000, 4, overlap=True)
            except Exception as e:
                sec["chain_overlapped"] = {"error": repr(e)}
            try:                        # the same chain with two region workers (processes) sharing this GPU
                sec["chain_2_workers"] = chain_workers(2, 3)
            except BaseException as e:
                sec["chain_2_workers"] = {"error": repr(e)}
            line["config"]["secondary"] = sec
        final_line = line
    if world > 1 or force_dist:
        shard.shutdown()
        dist.destroy_process_group()
    if pipelined:
        if sub["thread"] is not None:
            sub["cmd"].put(None)
            sub["thread"].join()
        for c_ in callers[1:]:
            c_.close()
    caller.close()
    if final_line is not None:
        if world == 1 and args.config == "C3" and not (args.depth or args.cols or args.no_other_configs or args.no_secondary):
            # the other BASELINE configs on the same record: scalars only.  This process's HBM goes back
            # first (the contexts are closed, the tracks and the outputs dropped): a genome run sizes its BAQ scratch from the
            # free memory it finds
            batch = d_counts = d_pvals = out_bufs = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            final_line["config"].update(other_configs())
        print(json.dumps(final_line))


if __name__ == "__main__":
    main()
