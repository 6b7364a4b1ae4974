"""-m gpu: ONE process from pileup columns to VCF text on the GPU through the packing core of the column binding
(integration/lofreq_amd_colbatch.c -- what integration/lofreq_amd_shim.c runs inside `lofreq call` after the gates of
call_vars) and the REAL liblofreq_amd.so: tests/colbatch_harness.c feeds it the columns of the golden fixtures the way
mpileup hands over plp_col_t (arrays freed right after the call), the lines it emits are compared with the VCFs the
reference's own 2.1.4 binary wrote (tests/golden/snv_*.json, indel_*.json) and, byte for byte, with what the ctypes path
of the other GPU tests produces from the same columns.  tests/test_shim.py (no GPU) proves the other half: a plp_col_t
built with the reference's own helpers reaches this core as the same arrays."""
import os
import subprocess
import sys

import numpy as np
import pytest

import golden_util as gu
import util
from test_shim import _i32, _indel_blob, _snv_columns_blob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    d = tmp_path_factory.mktemp("colbatch")
    exe = str(d / "colbatch_harness")
    lib = os.path.join(ROOT, "lofreq_amd")
    subprocess.run(["gcc", "-std=gnu99", "-O1", "-g", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "integration"), os.path.join(ROOT, "integration", "lofreq_amd_colbatch.c"),
                    os.path.join(ROOT, "tests", "colbatch_harness.c"), "-L" + lib, "-llofreq_amd", "-Wl,-rpath," + lib,
                    "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe], check=True, capture_output=True, text=True)
    return exe


def _run(harness, tmp_path, header, blob, ncols, args=(), env=None):
    inp = str(tmp_path / "cols.bin")
    with open(inp, "wb") as f:
        f.write(_i32(*header) + _i32(ncols) + blob)
    r = subprocess.run([harness, inp] + list(args), capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert lines and lines[-1].startswith("#conf ")
    return lines[:-1], [int(x) for x in lines[-1].split()[1:]]


def _no_af(line):
    f = line.split("\t")
    f[7] = ";".join(x for x in f[7].split(";") if not x.startswith("AF="))
    return "\t".join(f)


def _with_filter(line, filt):
    f = line.split("\t")
    f[6] = filt
    return "\t".join(f)


@pytest.mark.parametrize("path", gu.fixtures(), ids=lambda p: p.split("/")[-1])
@pytest.mark.parametrize("batch_cols", [0, 37])
def test_snv_columns_to_vcf_lines(harness, caller, tmp_path, path, batch_cols):
    import lofreq_amd as la
    fx, host = gu.load(path)
    kw, no_default_filter = gu.conf_kwargs(fx["call_args"])
    args = ["%s=%s" % (k, kw[k]) for k in ("min_bq", "min_alt_bq", "sig") if k in kw]
    header = (kw.get("bonf_dynamic", 1), kw.get("bonf_subst", 1), 1, 0, kw["flag"])
    env = {"LFQ_SHIM_BATCH_COLS": str(batch_cols)} if batch_cols else None       # several batches in flight, same lines
    got, (bonf_subst, n_tests, _, _, _) = _run(harness, tmp_path, header, _snv_columns_blob(fx), len(fx["columns"]), args, env)
    # the ctypes path on the same columns: byte-identical lines (FILTER '.', as report_var writes them) and counters
    conf = la.VarcallConf(**kw)
    recs, _, _ = caller.call_snvs(util.to_pileup_batch(la, host), conf, want_counts=True)
    pos0 = np.array([fx["columns"][int(r["col"])]["pos0"] for r in recs], np.int64)
    assert got == la.format_vcf(recs, "chr1", pos0=pos0).splitlines(), path
    assert (bonf_subst, n_tests) == (conf.bonf_subst, conf.num_snv_tests)
    if not fx.get("column_subset"):
        assert n_tests == fx["num_snv_tests"], path
    # the binary's VCF: `lofreq call` filters its own output (lofreq_call.c:1506-1538) unless --no-default-filter and -b
    dynamic = bool(conf.bonf_dynamic)
    if no_default_filter and not dynamic:
        kept = got
    else:
        thr = la.snvqual_thresh(conf.sig, conf.bonf_subst) if dynamic else 0
        keep = la.filter_records(recs, thr, apply_defaults=not no_default_filter)
        kept = [_with_filter(l, "PASS") for l, k in zip(got, keep) if k]
    assert len(kept) == len(fx["vcf"]) and len(kept) > 0, path
    for g, e in zip(kept, fx["vcf"]):       # modulo the two 2.1.4-vs-HEAD deltas: ;HQA=, raw counts -> AF
        assert _no_af(gu.strip_hqa(g)) == _no_af(e), (path, g, e)


@pytest.mark.parametrize("path", gu.indel_fixtures(), ids=lambda p: p.split("/")[-1])
def test_indel_columns_to_vcf_lines(harness, caller, tmp_path, path):
    import lofreq_amd as la
    fx, dicts = gu.load_indels(path)
    kw, no_default_filter = gu.conf_kwargs(fx["call_args"])
    header = (kw.get("bonf_dynamic", 1), kw.get("bonf_subst", 1), 0, 1, kw["flag"])
    pos = [c["pos0"] for c in fx["columns"]]
    got, (_, _, bonf_indel, n_tests, wo_idaq) = _run(harness, tmp_path, header, _indel_blob(dicts, pos), len(dicts))
    kw.pop("bonf_subst", None)
    conf = la.VarcallConf(**kw)
    cols = la.IndelColumns.from_columns(dicts)
    recs, ntests = la.call_indels(caller, cols, conf)
    want = [la.format_indel_record("chr1", pos[int(r["col"])], cols, r, None).rstrip("\n") for r in recs]
    assert got == want and len(got) > 0, path
    assert n_tests == ntests == fx["num_indel_tests"] and bonf_indel == conf.bonf_indel and wo_idaq == 0
    dynamic = bool(conf.bonf_dynamic)
    if no_default_filter and not dynamic:
        kept = got
    else:
        thr = la.snvqual_thresh(conf.sig, conf.bonf_indel) if dynamic else 0
        keep = la.filter_indel_records(recs, thr, apply_defaults=not no_default_filter)
        kept = [_with_filter(l, "PASS") for l, k in zip(got, keep) if k]
    assert kept == fx["vcf"], path
