"""-m gpu: the sharded path with REAL kernels -- two processes (gloo rendezvous on 127.0.0.1), each with its own
context on the one GPU of the box, running the bins shard.plan_regions deals them on a synthetic genome; rank 0's
merged records must be byte-identical to the single-process call of the whole genome (running Bonferroni factor,
QUAL, strand bias, order).  SURVEY 8e; the reference's own check is tests/parallel.sh:40-51."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]

SEED, DEPTH, NCOLS, PERIOD = 0x9E3779B97F4A7C15 ^ (5 << 32), 600, 24000, 53


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_bin(la, caller, dev, lo, hi):
    """layer 1 on one bin's columns (generated in HBM) -> (sparse records, tested columns)"""
    conf = la.VarcallConf()
    n = hi - lo
    batch = caller.synth_batch(SEED, DEPTH, n, plant_period=PERIOD, col_begin=lo)
    d_counts = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
    d_pvals = torch.zeros(n * 128, dtype=torch.uint8, device=dev)
    caller.snv_batch_device(batch, conf, d_counts, d_pvals, n)
    st = caller.batch_finish()
    pv = d_pvals[: st.n_pvals * 128].cpu().numpy().view(la.COL_PVALS_DTYPE).copy()
    return pv, int(st.n_tested)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lofreq_amd as la
    from lofreq_amd import shard
    dev = torch.device("cuda", 0)
    caller = la.SnvCaller(0)
    # the second half of the genome counts double: bins of unequal length, interleaved owners
    cost = lambda c, b, e: float((e - b) + max(0, e - max(b, NCOLS // 2)))
    bins, owner = shard.plan_regions([("synth", 0, NCOLS)], cost, world)
    mine = []
    for i, ((_, lo, hi), o) in enumerate(zip(bins, owner)):
        if o == rank:
            pv, n_tested = _run_bin(la, caller, dev, lo, hi)
            mine.append((i, lo, pv, n_tested))
    conf = la.VarcallConf()
    recs, total = shard.finish_bins(conf, mine, len(bins), dist, None)
    if rank == 0:
        np.save(out, recs.view(np.uint8))
        np.save(out + ".meta", np.array([total, conf.bonf_subst, conf.num_snv_tests, len(bins)]))
    caller.close()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_equal_single_process(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, ROOT)
    import lofreq_amd as la
    out = str(tmp_path / "recs.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out).view(la.SNV_RECORD_DTYPE)
    total, bonf, ntests, nbins = np.load(out + ".meta.npy")
    caller = la.SnvCaller(0)
    conf = la.VarcallConf()
    exp, _, st = caller.call_snvs(caller.synth_batch(SEED, DEPTH, NCOLS, plant_period=PERIOD), conf)
    caller.close()
    assert nbins >= 4
    assert total == st.n_tested and bonf == conf.bonf_subst and ntests == conf.num_snv_tests
    assert len(exp) > 100 and len(got) == len(exp)
    for k in la.SNV_RECORD_DTYPE.names:
        if k != "pad_":
            assert (got[k] == exp[k]).all(), k


@pytest.mark.timeout(600)
def test_bench_gpus_2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (torch.distributed.run, one process per
    rank) and prints ONE line with n_gpus = 2; on this one-GPU box both ranks share cuda:0 and exchange over gloo
    (LFQ_BENCH_ONE_GPU=1).  A world size that contradicts --gpus is refused."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LFQ_BENCH_ONE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--cols", "40000", "--repeats", "2"], env=env, capture_output=True, text=True, timeout=500)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["scaling"] == "weak"
    assert d["config"]["columns_per_gpu"] == 40000 and d["value"] > 0 and d["repeats"]["blocks"] == 2
    # the same shard alone: twice the columns per step with two ranks
    p1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                         "--cols", "40000", "--no-pmc", "--no-cpu-baseline", "--no-secondary", "--no-full-check"],
                        env=env, capture_output=True, text=True, timeout=500)
    assert p1.returncode == 0, p1.stderr[-2000:]
    d1 = json.loads([ln for ln in p1.stdout.splitlines() if ln.startswith("{")][0])
    assert d1["n_gpus"] == 1 and d1["config"]["records_per_step"] > 0
    # --gpus 2 under a launcher that started one rank: refused
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29513")
    p2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                        env=env2, capture_output=True, text=True, timeout=200)
    assert p2.returncode != 0 and "WORLD_SIZE" in (p2.stderr + p2.stdout)


@pytest.mark.timeout(900)
def test_bench_sharded_step_through_a_one_rank_rccl_communicator():
    """The N > 1 form of bench.py's step on one GPU: layer 1 + the shard exchange with a ONE-rank RCCL communicator
    (LFQ_BENCH_FORCE_DIST=1) -- test counts over the host group, the record gather asynchronous over RCCL and collected a
    step later -- gives the records of the layer-2 run, checked against the oracle by the run itself; so do the blocking
    forms kept for A/B runs (LFQ_BENCH_EXCHANGE=rccl, LFQ_BENCH_EXCHANGE_LAG=0)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LFQ_BENCH_ONE_GPU"):
        base.pop(k, None)

    def run(extra_env, *args):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--cols", "40000",
                            "--repeats", "1", "--no-pmc", "--no-cpu-baseline", "--no-secondary"] + list(args),
                           env=dict(base, **extra_env), capture_output=True, text=True, timeout=400)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
        assert len(lines) == 1, p.stdout[-2000:]
        return json.loads(lines[0])

    ref = run({})
    assert ref["config"]["vcf_identical"] is True and ref["config"]["records_per_step"] > 10
    for env in ({"LFQ_BENCH_FORCE_DIST": "1"},
                {"LFQ_BENCH_FORCE_DIST": "1", "LFQ_BENCH_EXCHANGE": "rccl"},
                {"LFQ_BENCH_FORCE_DIST": "1", "LFQ_BENCH_EXCHANGE_LAG": "0"}):
        d = run(env, "--shard-path")
        c = d["config"]
        assert c["exchange_backend"] == "nccl" and c["rccl_ranks"] == 1, c
        assert ("host transport" in c["exchange"]["counts"]) == (env.get("LFQ_BENCH_EXCHANGE") != "rccl"), c["exchange"]
        assert "lfq_shard_gather_start" in c["exchange"]["records"] and "ncclAllGather" in c["exchange"]["records"]
        assert c["vcf_identical"] is True and c["records_per_step"] == ref["config"]["records_per_step"], (env, c)
        assert c["records_compared"] == ref["config"]["records_compared"]


_PAR_WORKER = r'''
import ctypes as C, os, sys
import numpy as np
root, libpar, out = sys.argv[1:4]
sys.path.insert(0, root)
import lofreq_amd as la
from lofreq_amd import _lib
SEED, DEPTH, NCOLS, PERIOD = int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
L = _lib.load()
P = C.CDLL(libpar)
par = C.c_void_p()
assert P.lfq_par_init(C.byref(par), 1) == 0 and par.value
P.lfq_par_world.argtypes = P.lfq_par_rank.argtypes = [C.c_void_p]
P.lfq_par_ctx.argtypes = [C.c_void_p]; P.lfq_par_ctx.restype = C.c_void_p
world, rank = P.lfq_par_world(par), P.lfq_par_rank(par)
ctx = C.c_void_p(P.lfq_par_ctx(par))                     # the worker's context, created by lfq_par_init on its GPU
caller = la.SnvCaller.__new__(la.SnvCaller)
caller.L, caller.h, caller.device = L, ctx, 0
lo, hi = NCOLS * rank // world, NCOLS * (rank + 1) // world
# two flushes per worker, like a shim that fills two batches: the local running factor carries over
conf = la.VarcallConf()
pvs, tested = [], 0
mid = (lo + hi) // 2
for b, e in ((lo, mid), (mid, hi)):
    batch = caller.synth_batch(SEED, DEPTH, e - b, plant_period=PERIOD, col_begin=b)
    caller.call_snvs_submit(batch, conf)
    pv = np.zeros(e - b, _lib.COL_PVALS_DTYPE)
    n = C.c_int64(0); st = _lib.BatchStats()
    assert L.lfq_call_snvs_collect_pvals(caller.h, C.c_void_p(pv.ctypes.data), len(pv), C.byref(n), C.byref(st)) == 0
    pv = pv[: n.value].copy()
    pv["col"] += b                                       # the key: the global column
    pvs.append(pv)
    assert L.lfq_shard_advance_conf(C.byref(conf.c), st.n_tested) == 0      # this worker's own running factor
    tested += st.n_tested
pv = np.concatenate(pvs)
start = la.VarcallConf()                                 # what every worker started from
rec_p = C.c_void_p(); n_rec = C.c_int64(0)
P.lfq_par_merge_snvs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
rc = P.lfq_par_merge_snvs(par, C.byref(start.c), C.c_void_p(pv.ctypes.data), len(pv), tested, 0, C.byref(rec_p), C.byref(n_rec))
assert rc == 0, rc
if rank == 0:
    recs = np.frombuffer(C.string_at(rec_p.value, 64 * n_rec.value), la.SNV_RECORD_DTYPE).copy()
    np.save(out, recs.view(np.uint8))
    np.save(out + ".meta", np.array([start.bonf_subst, start.num_snv_tests]))
else:
    assert not rec_p.value
caller.h = None                                          # the context belongs to the lfq_par handle ...
P.lfq_par_destroy.argtypes = [C.c_void_p]
P.lfq_par_destroy(par)                                   # ... which destroys it
'''


@pytest.mark.timeout(600)
def test_c_workers_real_kernels_files_transport(tmp_path):
    """integration/lofreq_amd_parallel.c with REAL kernels: two worker processes (both on cuda:0 -- RCCL refuses two
    ranks on one GPU, so the all-gathers go through the files transport), each calling its half of a synthetic genome
    in two flushes through lfq_call_snvs_submit / lfq_call_snvs_collect_pvals, then lfq_par_merge_snvs: rank 0's merged
    records and the final conf equal the single-process call of the whole genome."""
    import subprocess
    import lofreq_amd as la
    libpar = str(tmp_path / "liblofreq_amd_parallel.so")
    subprocess.run(["gcc", "-std=gnu99", "-O1", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "integration", "lofreq_amd_parallel.c"), "-L" + os.path.join(ROOT, "lofreq_amd"),
                    "-llofreq_amd", "-Wl,-rpath," + os.path.join(ROOT, "lofreq_amd"), "-ldl", "-o", libpar],
                   check=True, capture_output=True, text=True)
    script = str(tmp_path / "worker.py")
    open(script, "w").write(_PAR_WORKER)
    out = str(tmp_path / "recs.npy")
    procs = []
    for r in range(2):
        env = dict(os.environ, LFQ_PAR_WORLD="2", LFQ_PAR_RANK=str(r), LFQ_PAR_RENDEZVOUS=str(tmp_path / "rdv"),
                   LFQ_PAR_TRANSPORT="files", LFQ_PAR_TIMEOUT_S="240", LFQ_DEVICE="0")
        procs.append(subprocess.Popen([sys.executable, script, ROOT, libpar, out, str(SEED), str(DEPTH), str(NCOLS), str(PERIOD)],
                                      env=env, stderr=subprocess.PIPE, text=True))
    for p in procs:
        _, err = p.communicate(timeout=500)
        assert p.returncode == 0, err[-3000:]
    got = np.load(out).view(la.SNV_RECORD_DTYPE)
    bonf, ntests = np.load(out + ".meta.npy")
    caller = la.SnvCaller(0)
    conf = la.VarcallConf()
    exp, _, st = caller.call_snvs(caller.synth_batch(SEED, DEPTH, NCOLS, plant_period=PERIOD), conf)
    caller.close()
    assert bonf == conf.bonf_subst and ntests == conf.num_snv_tests
    assert len(exp) > 100 and len(got) == len(exp)
    for k in la.SNV_RECORD_DTYPE.names:
        if k != "pad_":
            assert (got[k] == exp[k]).all(), k


_NCCL_WORKER = r'''
import os, sys
import numpy as np
root, out = sys.argv[1:3]
sys.path.insert(0, root)
os.environ["LFQ_SHARD_FORCE_COLLECTIVES"] = "1"          # the collectives run although the world has one rank
import torch
import torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", device_id=dev, rank=0, world_size=1)
import lofreq_amd as la
from lofreq_amd import shard
assert shard._FORCE_COLLECTIVES
caller = la.SnvCaller(0)
SEED, DEPTH, NCOLS, PERIOD = (int(x) for x in sys.argv[3:7])
batch = caller.synth_batch(SEED, DEPTH, NCOLS, plant_period=PERIOD)
conf = la.VarcallConf()
d_counts = torch.zeros(NCOLS * 64, dtype=torch.uint8, device=dev)
d_pvals = torch.zeros(NCOLS * 128, dtype=torch.uint8, device=dev)
caller.snv_batch_device(batch, conf, d_counts, d_pvals, NCOLS)
st = caller.batch_finish()
pv = d_pvals[: st.n_pvals * 128].cpu().numpy().view(la.COL_PVALS_DTYPE)
recs, total = shard.finish_shard(conf, pv, st.n_tested, None, 0, dist, dev)       # all-gather + gather on cuda tensors: RCCL
ones = torch.ones(1, dtype=torch.int64, device=dev); dist.all_reduce(ones)
np.save(out, recs.view(np.uint8))
np.save(out + ".meta", np.array([total, conf.bonf_subst, conf.num_snv_tests, int(ones.item()), dist.get_backend() == "nccl"]))
caller.close()
dist.destroy_process_group()
'''


def test_shard_exchange_through_rccl_one_rank(tmp_path):
    """the collectives of the N > 1 benchmark path (all_gather_into_tensor of int64 counts, gather of the uint8 record
    buffers, on CUDA tensors) through the real RCCL backend -- a process group of one rank, the collectives forced on
    (LFQ_SHARD_FORCE_COLLECTIVES): what an 8-GPU node runs per step, minus the peers"""
    import subprocess
    import lofreq_amd as la
    script = str(tmp_path / "w.py")
    open(script, "w").write(_NCCL_WORKER)
    out = str(tmp_path / "recs.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, script, ROOT, out, str(SEED), str(DEPTH), str(NCOLS), str(PERIOD)], env=env,
                       capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    got = np.load(out).view(la.SNV_RECORD_DTYPE)
    total, bonf, ntests, ranks, is_nccl = np.load(out + ".meta.npy")
    assert is_nccl == 1 and ranks == 1
    caller = la.SnvCaller(0)
    conf = la.VarcallConf()
    exp, _, st = caller.call_snvs(caller.synth_batch(SEED, DEPTH, NCOLS, plant_period=PERIOD), conf)
    caller.close()
    assert total == st.n_tested and bonf == conf.bonf_subst and ntests == conf.num_snv_tests
    assert len(exp) > 100 and got.tobytes() == exp.tobytes()
