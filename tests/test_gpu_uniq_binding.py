"""-m gpu: the packing core of the `lofreq uniq` binding (integration/lofreq_amd_uniqbatch.c -- what
integration/lofreq_amd_uniq.c runs inside `lofreq uniq` after the gates of uniq_snv) against the REAL liblofreq_amd.so:
tests/uniqbatch_harness.c feeds it the variants' columns of the golden fixtures the way mpileup hands over plp_col_t
(arrays gone right after the call); the UNIQ flags and UQ values it reports are the ones the reference's own 2.1.4 binary
wrote (tests/golden/uniq_*.json).  tests/test_uniq_binding.py (no GPU) proves the other half: a plp_col_t built with the
reference's own helpers reaches this core as the same arrays, and the values come back as the same INFO tags."""
import os
import struct
import subprocess

import numpy as np
import pytest

import golden_util as gu
from test_uniq_binding import _i32, _obs, _var, variant_blob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    d = tmp_path_factory.mktemp("uniqbatch")
    exe = str(d / "uniqbatch_harness")
    lib = os.path.join(ROOT, "lofreq_amd")
    subprocess.run(["gcc", "-std=gnu99", "-O1", "-g", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "integration"), os.path.join(ROOT, "integration", "lofreq_amd_uniqbatch.c"),
                    os.path.join(ROOT, "tests", "uniqbatch_harness.c"), "-L" + lib, "-llofreq_amd", "-Wl,-rpath," + lib,
                    "-Wl,-rpath-link,/opt/rocm/lib", "-lm", "-o", exe], check=True, capture_output=True, text=True)
    return exe


def _run(harness, tmp_path, use_det_lim, uni_freq, blobs):
    inp = str(tmp_path / "vars.bin")
    with open(inp, "wb") as f:
        f.write(_i32(use_det_lim) + struct.pack("<f", uni_freq) + _i32(len(blobs)) + b"".join(blobs))
    r = subprocess.run([harness, inp], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return [ln.split() for ln in r.stdout.splitlines()]


@pytest.mark.parametrize("path", gu.uniq_fixtures(), ids=lambda p: p.split("/")[-1])
def test_detlim_flags_of_the_reference_binary(harness, tmp_path, path):
    fx, _, _ = gu.load_uniq(path)
    got = _run(harness, tmp_path, 1, -1.0, [variant_blob(v, 0) for v in fx["variants"]])
    assert [(int(p), x) for p, x in got] == [(v["pos0"], "1" if v["uniq"] else "0") for v in fx["variants"]]


@pytest.mark.parametrize("path", gu.uniq_binom_fixtures(), ids=lambda p: p.split("/")[-1])
def test_binom_uq_values_of_the_reference_binary(harness, caller, tmp_path, path):
    import lofreq_amd as la
    fx, _, _ = gu.load_uniq(path)
    got = _run(harness, tmp_path, 0, -1.0, [variant_blob(v, 0) for v in fx["variants"]])
    assert [(int(p), x) for p, x in got] == [(v["pos0"], "-" if v["uq"] is None else str(v["uq"])) for v in fx["variants"]]
    # ... and the decision apply_uniq_filter_mtc takes on those tags (lofreq_uniq.c:140-206) is the binary's FILTER column
    uq = np.array([-1 if x == "-" else int(x) for _, x in got], np.int32)
    assert la.uniq_mtc(uq, fx["mtc"], fx["alpha"], 0).tolist() == [v["filter"] == "PASS" for v in fx["variants"]]


def test_gates_and_indel_counts(harness, oracle, tmp_path):
    """variants without a usable column get no tag; an indel variant in binomial mode is the scalar test on its event count"""
    snv = _var(10, "A", "G", "0.100000", {"A": _obs(20), "G": _obs(3)})
    wrong = _var(11, "C", "T", "0.200000", {"C": _obs(9)})
    nocov = _var(12, "C", "T", "0.200000", {"C": _obs(5)})
    dele = _var(15, "GAT", "G", "0.050000", {"G": _obs(30)})
    got = _run(harness, tmp_path, 0, -1.0, [variant_blob(snv, 0), variant_blob(wrong, 0, col_pos=12), variant_blob(nocov, 0, cov=0),
                                            variant_blob(dele, 0, cov=40, tails=2, dels=[("AT", 4), ("A", 1)])])
    uq_snv, _ = oracle.uniq_binom_batch(np.array([0] * 20 + [2] * 3, np.uint8), np.array([0, 23], np.uint64),
                                        np.array([0.1], np.float32), "G")
    pv, st = oracle.binom_cdf(38, 4, float(np.float32(0.05)))
    assert st == 0
    want_del = int(np.floor(-10.0 * np.log10(np.longdouble(pv))))
    assert [x for _, x in got] == [str(int(uq_snv[0])), "-", "-", str(want_del)]
