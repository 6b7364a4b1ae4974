"""-m gpu: `bench.py --config C4` / `--config C5` (BASELINE.json configs[3] / [4]: a genome of reads through the
reads -> VCF chain, region- / BED-sharded) -- the sharded run (two ranks on this one GPU, exchange over gloo:
LFQ_BENCH_ONE_GPU=1) must write the same VCF, test counts and Bonferroni factors as the one-process run, and the line must
carry its own workload string, n_gpus = world, and the communicator's size.  A 1/16 genome of the same shape (32 bins)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


def _run(cfg, gpus, vcf, scale):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    if gpus > 1:
        env["LFQ_BENCH_ONE_GPU"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--gpus", str(gpus), "--steps", "1",
                        "--warmup", "1", "--genome-scale", str(scale), "--vcf-out", vcf, "--no-pmc"] + ([] if gpus == 1 else ["--no-cpu-baseline"]),
                       cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("cfg,scale", [("C4", 1 / 16), ("C5", 1 / 16)])
def test_sharded_genome_equals_one_process(tmp_path, cfg, scale):
    a = _run(cfg, 1, str(tmp_path / "one.vcf"), scale)
    b = _run(cfg, 2, str(tmp_path / "two.vcf"), scale)
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and b["config"]["rccl_ranks"] == 2 and b["scaling"] == "strong"
    assert a["config"]["workload"].startswith(cfg + ":") and ("configs[%d]" % (3 if cfg == "C4" else 4)) in a["config"]["workload"]
    assert a["config"]["bins"] == b["config"]["bins"] == 32 and b["config"]["bins_rank0"] == 16
    one, two = open(tmp_path / "one.vcf").read(), open(tmp_path / "two.vcf").read()
    assert one == two and one.count("\n") > 50
    for k in ("snv_tests", "indel_tests", "called_columns", "vcf_sha256", "vcf_lines"):
        assert a["config"][k] == b["config"][k], k
    assert a["config"]["snv_tests"] > 0 and a["value"] > 0
    # the bins cycle through four distinct sets of reads; the one-GPU line is a measured one: the dominant kernel's launch timed
    # with HIP events, the oracle chain timed on a sample of a bin
    assert a["config"]["distinct_bins"] == 4 and "4 distinct sets of reads" in a["data"]
    assert a["roofline"]["avg_launch_ms"] > 0 and 0 < a["roofline"]["frac"] < 1 and a["roofline"]["traffic"] is None
    assert a["cpu_baseline"]["value"] > 0 and a["cpu_baseline"]["cores"] == 1 and a["cpu_baseline"]["kind"] == "port"
    assert b["roofline"] is None and b["cpu_baseline"] is None
    if cfg == "C4":
        assert a["config"]["indel_tests"] > 100 and "INDEL" in one


def test_sharded_genome_at_eight_ranks(tmp_path):
    """what an 8-GPU node runs, on the one GPU of this box: eight ranks (four bins each), the test counts over the library's
    shared-memory host transport, records merged on rank 0 -- the one-process VCF, byte for byte (round 6; no node with eight
    GPUs has been available to any round)"""
    a = _run("C4", 1, str(tmp_path / "one.vcf"), 1 / 20)
    b = _run("C4", 8, str(tmp_path / "eight.vcf"), 1 / 20)
    assert b["n_gpus"] == 8 and b["config"]["rccl_ranks"] == 8 and b["config"]["bins"] == 32 and b["config"]["bins_rank0"] == 4
    one, eight = open(tmp_path / "one.vcf").read(), open(tmp_path / "eight.vcf").read()
    assert one == eight and one.count("\n") > 50 and "INDEL" in one
    for k in ("snv_tests", "indel_tests", "called_columns", "vcf_sha256"):
        assert a["config"][k] == b["config"][k], k
