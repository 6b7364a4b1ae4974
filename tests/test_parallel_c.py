"""integration/lofreq_amd_parallel.c -- N region workers that merge in memory (the C-side counterpart of the reference's
parallel wrapper epilogue, lofreq2_call_pparallel.py:131-185, 685-707) -- as separate PROCESSES over the files
transport (the all-gather through a shared directory: the stand-in for the RCCL communicator where there is no GPU),
against lofreq_amd/shard.py doing the same exchange in one process and against the unsharded result."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lofreq_amd import _lib          # noqa: E402
import lofreq_amd as la              # noqa: E402


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    d = tmp_path_factory.mktemp("par")
    exe = str(d / "parallel_harness")
    chk = subprocess.run(["gcc", "-std=gnu99", "-fsyntax-only", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
                          os.path.join(ROOT, "integration", "lofreq_amd_parallel.c")], capture_output=True, text=True)
    assert chk.returncode == 0 and not chk.stderr.strip(), chk.stderr
    _lib.load()
    subprocess.run(["gcc", "-std=gnu99", "-O1", "-g", "-Wall", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "integration"), os.path.join(ROOT, "integration", "lofreq_amd_parallel.c"),
                    os.path.join(ROOT, "tests", "parallel_harness.c"), "-L" + os.path.join(ROOT, "lofreq_amd"),
                    "-llofreq_amd", "-Wl,-rpath," + os.path.join(ROOT, "lofreq_amd"), "-Wl,-rpath-link,/opt/rocm/lib",
                    "-ldl", "-o", exe], check=True, capture_output=True, text=True)
    return exe


def _fake_shard(rng, n_cols, first_factor_cols):
    """sparse records of one shard as the device writes them: a few columns survive pruning, p-values around the
    emit threshold so that the exact Bonferroni factor decides, shard-local running factors"""
    n = int(rng.integers(20, 60))
    cols = np.sort(rng.choice(n_cols, n, replace=False))
    tested_before = np.sort(rng.choice(n_cols * 3, n, replace=False)) // 3 + np.arange(n)    # tested columns up to each record
    pv = np.zeros(n, _lib.COL_PVALS_DTYPE)
    pv["col"] = cols
    pv["bonf"] = 3 * (1 + tested_before)
    pv["ref_base"] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)]
    pv["counts"]["coverage"] = 1000
    for a in range(3):
        k = rng.integers(0, 40, n)
        pv["counts"]["alt_counts"][:, a] = k
        pv["counts"]["alt_raw_counts"][:, a] = k + rng.integers(0, 3, n)
        pv["counts"]["alt_fw"][:, a] = k // 2
        # around log(sig / bonf): some pass with the local factor and fail with the global one
        pv["logp"][:, a] = np.log(0.01 / (3.0 * (first_factor_cols + 1 + tested_before))) + rng.normal(0, 1.5, n)
        pv["status"][:, a] = np.where(k > 0, la.LFQ_PV_LOG, la.LFQ_PV_NONE)
    pv["counts"]["kmax"] = pv["counts"]["alt_counts"].max(axis=1)
    pv["counts"]["ref_fw"], pv["counts"]["ref_rv"] = 400, 380
    n_tested = int(tested_before[-1]) + 1 + int(rng.integers(0, 50))
    return pv, n_tested


@pytest.mark.parametrize("world,transport", [(2, "files"), (3, "files"), (2, "shm"), (8, "shm")])
def test_workers_merge_like_one_process(harness, tmp_path, world, transport):
    rng = np.random.default_rng(40 + world)
    n_cols = 50000
    shards, prefix = [], 0
    for r in range(world):
        pv, n_tested = _fake_shard(rng, n_cols, prefix)
        shards.append((pv, n_tested, int(rng.integers(0, 30))))
        prefix += n_tested
    # what ONE process computes: exact factors = local + 3 x tested columns of the earlier shards, global keys
    conf = la.VarcallConf()
    want, before = [], 0
    for r, (pv, n_tested, _) in enumerate(shards):
        g = pv.copy()
        g["col"] += r * n_cols
        g["bonf"] += 3 * before
        want.append(la.finalize_pvals(conf, g))
        before += n_tested
    want = np.concatenate(want)
    assert 10 < len(want) < sum(len(s[0]) * 3 for s in shards)
    # the same emit decisions with the shard-local factors would be different ones: the rebase matters
    local = np.concatenate([la.finalize_pvals(conf, s[0]) for s in shards])
    assert len(local) > len(want)

    rdv = str(tmp_path / "rdv")
    # what a crashed run under the same rendezvous path leaves behind: files of the right size for the first two
    # collectives (16 B of header + 16 / 8 B of payload) and a <rdv>.job older than any rank 0 would wait -- this run
    # must wait past all of them (they carry another run's nonce)
    for r in range(world):
        for k, n in ((0, 32), (1, 24)):
            with open("%s.ag%d.%d" % (rdv, k, r), "wb") as f:
                f.write(np.array([0x3152415051464c, 12345], np.uint64).tobytes() + b"\x07" * (n - 16))
    # ... and what a crashed attempt of the SAME launch leaves (a torchrun restart keeps run id and port, a second run inside
    # one srun step keeps job and step id -- ADVICE r05): a fresh <rdv>.job of the right size with that attempt's nonce and
    # tokens, its hello / ack files, an <rdv>.id and collective files under that nonce.  The late rank 0 below makes the
    # other ranks meet all of it first; the handshake of job_nonce() must not accept any of it
    stale = 0xDEADBEEF12345678
    hdr = lambda job: np.array([0x3152415051464c, job], np.uint64).tobytes()
    with open(rdv + ".job", "wb") as f:
        f.write(hdr(0) + np.array([stale] + [1000 + r for r in range(world)], np.uint64).tobytes())
    with open(rdv + ".id", "wb") as f:
        f.write(hdr(stale) + b"\x05" * 128)
    for r in range(1, world):
        open("%s.hello.%d" % (rdv, r), "wb").write(hdr(0) + np.array([1000 + r], np.uint64).tobytes())
        open("%s.ack.%d" % (rdv, r), "wb").write(hdr(stale) + np.array([1000 + r], np.uint64).tobytes())
    for k, n in ((2, 32), (3, 24)):
        for r in range(world):
            open("%s.ag%d.%d" % (rdv, k, r), "wb").write(hdr(stale) + b"\x09" * (n - 16))
    procs = []
    for r, (pv, n_tested, n_indel) in enumerate(shards):
        g = pv.copy()
        g["col"] += r * n_cols                       # the caller's job: a key that is unique over the whole run
        path = str(tmp_path / ("in%d" % r))
        with open(path, "wb") as f:
            f.write(np.array([n_tested, n_indel, len(g)], np.int64).tobytes())
            f.write(g.tobytes())
        env = dict(os.environ, LFQ_PAR_WORLD=str(world), LFQ_PAR_RANK=str(r), LFQ_PAR_RENDEZVOUS=rdv,
                   LFQ_PAR_TRANSPORT=transport, LFQ_PAR_TIMEOUT_S="60",
                   SLURM_JOB_ID="77", SLURM_STEP_ID="0", TORCHELASTIC_RUN_ID="abc", MASTER_ADDR="127.0.0.1", MASTER_PORT="29500")
        cmd = [harness, path, str(tmp_path / ("out%d" % r))]
        if r == 0:                                   # rank 0 comes last
            cmd = ["sh", "-c", "sleep 0.5; exec \"$@\"", "sh"] + cmd
        procs.append(subprocess.Popen(cmd, env=env, stderr=subprocess.PIPE))
    for p in procs:
        _, err = p.communicate(timeout=120)
        assert p.returncode == 0, err.decode()
    raw = open(tmp_path / "out0", "rb").read()
    assert not os.path.exists(tmp_path / "out1")
    assert not [f for f in os.listdir(tmp_path) if ".hello." in f or ".ack." in f or f in ("rdv.job", "rdv.id")]
    cs = C.sizeof(_lib.Conf)
    got_conf = _lib.Conf.from_buffer_copy(raw[:cs])
    n_rec = int(np.frombuffer(raw[cs:cs + 8], np.int64)[0])
    recs = np.frombuffer(raw[cs + 8:cs + 8 + 64 * n_rec], _lib.SNV_RECORD_DTYPE)
    assert n_rec == len(want) and recs.tobytes() == want.tobytes()
    total_tested = sum(s[1] for s in shards)
    total_indel = sum(s[2] for s in shards)
    assert got_conf.bonf_subst == 3 * total_tested and got_conf.num_snv_tests == 3 * total_tested
    assert got_conf.num_indel_tests == total_indel and got_conf.bonf_indel == (total_indel if total_indel else 1)
    off = cs + 8 + 64 * n_rec
    n_txt = int(np.frombuffer(raw[off:off + 8], np.int64)[0])
    assert raw[off + 8:off + 8 + n_txt].decode() == "".join("%d\tchr%d\n" % (r, r + 1) for r in range(world))
    left = [f for f in os.listdir(tmp_path) if f.startswith("rdv.ag")
            and open(tmp_path / f, "rb").read()[8:16] not in (hdr(stale)[8:], hdr(12345)[8:])]      # (not the planted ones)
    assert len(left) <= (world if transport == "files" else 0), left     # only the closing barrier's files stay behind (+ the stale run's)
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("lofreq_amd.")]      # the shared segment's name is gone


def test_fixed_bonferroni_is_not_rebased(harness, tmp_path):
    """`-b N` (bonf_dynamic = 0): every column carries the same factor N wherever it was called (lofreq_call.c:794 is
    the only place the factor moves), so the merge must leave the factors alone -- ranks > 0 used to add 3 x prefix"""
    world, n_cols, fixed = 2, 50000, 30000
    rng = np.random.default_rng(77)
    shards = []
    for r in range(world):
        pv, n_tested = _fake_shard(rng, n_cols, 0)
        pv["bonf"] = fixed
        for a in range(3):      # around log(sig / N): the factor decides
            pv["logp"][:, a] = np.log(0.01 / fixed) + rng.normal(0, 1.5, len(pv))
        pv["col"] += r * n_cols
        shards.append((pv, n_tested))
    conf = la.VarcallConf()
    conf.bonf_dynamic, conf.bonf_subst = 0, fixed
    want = np.concatenate([la.finalize_pvals(conf, s[0]) for s in shards])
    over = shards[1][0].copy()
    over["bonf"] += 3 * shards[0][1]
    assert len(la.finalize_pvals(conf, over)) < len(la.finalize_pvals(conf, shards[1][0]))   # the old behaviour loses calls
    rdv = str(tmp_path / "rdv")
    procs = []
    for r, (pv, n_tested) in enumerate(shards):
        path = str(tmp_path / ("in%d" % r))
        with open(path, "wb") as f:
            f.write(np.array([n_tested, 0, len(pv)], np.int64).tobytes())
            f.write(pv.tobytes())
        env = dict(os.environ, LFQ_PAR_WORLD=str(world), LFQ_PAR_RANK=str(r), LFQ_PAR_RENDEZVOUS=rdv,
                   LFQ_PAR_TRANSPORT="files", LFQ_PAR_TIMEOUT_S="60", LFQ_TEST_FIXED_BONF=str(fixed))
        procs.append(subprocess.Popen([harness, path, str(tmp_path / ("out%d" % r))], env=env, stderr=subprocess.PIPE))
    for p in procs:
        _, err = p.communicate(timeout=120)
        assert p.returncode == 0, err.decode()
    raw = open(tmp_path / "out0", "rb").read()
    cs = C.sizeof(_lib.Conf)
    got_conf = _lib.Conf.from_buffer_copy(raw[:cs])
    n_rec = int(np.frombuffer(raw[cs:cs + 8], np.int64)[0])
    recs = np.frombuffer(raw[cs + 8:cs + 8 + 64 * n_rec], _lib.SNV_RECORD_DTYPE)
    assert n_rec == len(want) and recs.tobytes() == want.tobytes()
    assert got_conf.bonf_subst == fixed and got_conf.num_snv_tests == 3 * sum(s[1] for s in shards)


def test_single_process_is_a_no_op(harness, tmp_path):
    """LFQ_PAR_WORLD unset: lfq_par_init hands back NULL (the harness treats that as an error: a plain run has no merge)"""
    path = str(tmp_path / "in")
    open(path, "wb").write(np.zeros(3, np.int64).tobytes())
    env = {k: v for k, v in os.environ.items() if not k.startswith("LFQ_PAR_")}
    p = subprocess.run([harness, path, str(tmp_path / "out")], env=env, capture_output=True)
    assert p.returncode == 4 and b"lfq_par_init: 0" in p.stderr


def test_pick_device_spreads_workers(tmp_path):
    """lfq_pick_device: LFQ_DEVICE > LOCAL_RANK > the first free worker slot of the node (lock files) -- eight worker
    processes alive at the same time on a (pretended) 4-GPU node take every GPU twice; a slot is reused after its
    worker has gone"""
    code = ("import sys, time, ctypes\n"
            "sys.path.insert(0, %r)\n"
            "from lofreq_amd import _lib\n"
            "L = _lib.load()\n"
            "s = ctypes.c_int(-9)\n"
            "d = L.lfq_pick_device(4, ctypes.byref(s))\n"
            "d2 = L.lfq_pick_device(4, None)\n"
            "print(d, s.value, d2, flush=True)\n"
            "sys.stdin.readline()\n" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("LFQ_DEVICE", "LOCAL_RANK")}
    env["LFQ_SLOT_DIR"] = str(tmp_path)

    def start():
        return subprocess.Popen([sys.executable, "-c", code], env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)

    ps = [start() for _ in range(8)]
    got = [tuple(int(x) for x in p.stdout.readline().split()) for p in ps]
    assert sorted(g[1] for g in got) == list(range(8))                  # eight different slots ...
    assert sorted(g[0] for g in got) == [0, 0, 1, 1, 2, 2, 3, 3]        # ... = every GPU twice
    assert all(g[0] == g[1] % 4 == g[2] for g in got)                   # asking again gives the same answer
    victim = [i for i, g in enumerate(got) if g[1] == 2][0]
    ps[victim].stdin.write("\n"); ps[victim].stdin.flush(); ps[victim].wait(timeout=30)
    late = start()
    assert tuple(int(x) for x in late.stdout.readline().split())[:2] == (2, 2)     # the freed slot, the same GPU
    for p in ps + [late]:
        if p.poll() is None:
            p.stdin.write("\n"); p.stdin.flush(); p.wait(timeout=30)
    # explicit settings win
    for var, val, want in (("LFQ_DEVICE", "3", 3), ("LOCAL_RANK", "6", 2)):
        e = dict(env, **{var: val})
        out = subprocess.run([sys.executable, "-c", code], env=e, input="\n", capture_output=True, text=True).stdout.split()
        assert int(out[0]) == want and int(out[1]) == -1
    e = dict(env, LFQ_DEVICE="7")
    out = subprocess.run([sys.executable, "-c", code], env=e, input="\n", capture_output=True, text=True).stdout.split()
    assert int(out[0]) == -1                                       # LFQ_ERR_INVALID: no such GPU
