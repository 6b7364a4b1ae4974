"""The reference-side binding under test (no GPU): integration/lofreq_amd_shim.c is compiled against the reference's
OWN plp.h / snpcaller.h / vcf.h / utils.h and linked with the reference's own utils.c + log.c, then driven by a mock
mpileup (tests/shim_harness.c) that rebuilds plp_col_t columns from the golden fixtures the way compile_plp_col
fills them (int_varray_add_value per nucleotide, add_ins_sequence / add_del_sequence = uthash insertion order) and
frees + poisons every column right after the callback, as plp.c:1440-1445 does.  A mock liblofreq_amd records the
packed batches the shim hands to lfq_call_snvs_batch / lfq_call_indels_batch; they must equal, byte for byte, what
tests/golden_util.py / lofreq_amd.indel.IndelColumns build from the same fixtures (the batches every GPU parity
test runs on).  Skipped where /root/reference is absent.

htslib is not in this image; plp.h and vcf.h include two of its headers only for the names `faidx_t` and `BGZF`
in prototypes the shim never calls, so the test puts two one-line forward declarations on the include path (in a
temp dir -- nothing of htslib is restated)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "lofreq")), reason="reference tree not mounted")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    d = tmp_path_factory.mktemp("shim")
    os.makedirs(d / "stub" / "htslib")
    (d / "stub" / "htslib" / "faidx.h").write_text("typedef struct faidx_t faidx_t;\n")
    (d / "stub" / "htslib" / "bgzf.h").write_text("#include <stdio.h>\ntypedef struct BGZF BGZF;\n")
    exe = str(d / "shim_harness")
    base = ["gcc", "-std=gnu99", "-O1", "-g", "-Wall", "-Wno-unused-function",
            "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "integration"), "-I" + os.path.join(REF, "lofreq"),
            "-I" + os.path.join(REF, "uthash"), "-I" + str(d / "stub"),
            os.path.join(ROOT, "integration", "lofreq_amd_shim.c"), os.path.join(ROOT, "integration", "lofreq_amd_colbatch.c"),
            os.path.join(ROOT, "tests", "shim_harness.c"),
            os.path.join(REF, "lofreq", "utils.c"), os.path.join(REF, "lofreq", "log.c"), "-lm", "-o", exe]
    # the shim itself must compile without a single warning against the reference headers
    chk = subprocess.run(["gcc", "-std=gnu99", "-fsyntax-only", "-Wall", "-Wextra", "-Wno-unused-parameter",
                          "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "integration"),
                          "-I" + os.path.join(REF, "lofreq"), "-I" + os.path.join(REF, "uthash"), "-I" + str(d / "stub"),
                          os.path.join(ROOT, "integration", "lofreq_amd_shim.c")], capture_output=True, text=True)
    assert chk.returncode == 0, chk.stderr
    assert "lofreq_amd_shim.c" not in chk.stderr, chk.stderr
    # ... and the packing core against include/lofreq_amd.h alone: no LoFreq header, no htslib
    chk = subprocess.run(["gcc", "-std=gnu99", "-fsyntax-only", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
                          os.path.join(ROOT, "integration", "lofreq_amd_colbatch.c")], capture_output=True, text=True)
    assert chk.returncode == 0 and not chk.stderr.strip(), chk.stderr
    asan = subprocess.run(base[:1] + ["-fsanitize=address", "-fno-omit-frame-pointer"] + base[1:], capture_output=True, text=True)
    if asan.returncode != 0:                        # no libasan in this image: the harness's poisoning still catches stale reads
        subprocess.run(base, check=True, capture_output=True, text=True)
    return exe


def _i32(*v):
    return struct.pack("<%di" % len(v), *[int(x) for x in v])


def _snv_columns_blob(fx, cons_indel_cols=()):
    """the per-nucleotide arrays of a golden SNV fixture -> the harness's column stream (no indel fields)"""
    out = []
    has_baq = "-B" not in fx["call_args"]
    for ci, col in enumerate(fx["columns"]):
        n_col = sum(len(o["bq"]) for o in col["obs"].values())
        cons = ord("+") if ci in cons_indel_cols else ord(col["ref"])
        out.append(_i32(col["pos0"], ord(col["ref"]), cons, n_col, n_col, 0, 0, 0, 0, 0, 0))
        for nt in "ACGTN":
            o = col["obs"].get(nt)
            if not o:
                out.append(_i32(0, 0, 0, 0))
                continue
            bq = gu.dec(o["bq"])
            n = len(bq)
            mq = gu.dec(o["mq"]) if isinstance(o["mq"], str) else np.asarray(o["mq"])
            baq = gu.dec(o["baq"]) if has_baq else np.full(n, -1)
            out.append(_i32(n, col["fwrv"][nt][0], 1 if has_baq else 0, 0))
            quad = np.stack([bq, baq, mq, np.full(n, -1)], axis=1).astype("<i4")
            out.append(quad.tobytes())
        for _side in range(2):
            out.append(_i32(0, 0, 0, 0))            # non_fw, non_rv, no non-event reads, no events
    return b"".join(out)


def _run(harness, tmp_path, header, blob, ncols, env=None):
    inp, outp = str(tmp_path / "cols.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        f.write(_i32(*header) + _i32(ncols) + blob)
    r = subprocess.run([harness, inp, outp], capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-2000:]
    return open(outp, "rb").read()


class _Reader:
    def __init__(self, b):
        self.b, self.o = b, 0

    def tag(self):
        t = self.b[self.o:self.o + 4]
        self.o += 4
        return t

    def i64(self):
        v = struct.unpack_from("<q", self.b, self.o)[0]
        self.o += 8
        return v

    def arr(self, dtype, n):
        a = np.frombuffer(self.b, dtype=dtype, count=n, offset=self.o)
        self.o += a.nbytes
        return a


@pytest.mark.parametrize("path", gu.fixtures(), ids=lambda p: p.split("/")[-1])
def test_shim_packs_snv_columns_like_golden_util(harness, tmp_path, path):
    fx, host = gu.load(path)
    ncols = len(fx["columns"])
    # header: bonf_dynamic, bonf_subst, no_indels, only_indels, flag
    out = _Reader(_run(harness, tmp_path, (1, 1, 1, 0, 3), _snv_columns_blob(fx), ncols))
    n_n_ref = sum(1 for c in fx["columns"] if c["ref"] == "N")       # call_vars returns at once (lofreq_call.c:892)
    assert out.tag() == b"SNVB"
    nc, n_obs, on_dev, has_baq, has_sq, max_obs, bonf = [out.i64() for _ in range(7)]
    keep = np.array([c["ref"] != "N" for c in fx["columns"]])
    assert nc == ncols - n_n_ref and on_dev == 0 and has_sq == 0 and bonf == 1
    col_off = out.arr("<u8", nc + 1)
    ref = out.arr("u1", nc)
    cov = out.arr("<i4", nc)
    nb = out.arr("<i4", nc)
    # the shim hands over the nt track nibble-packed (LFQ_TRACKS_NT_PACKED): observation o in byte (o >> 3) * 4 + (o & 3),
    # nibble (o & 7) >> 2
    assert out.i64() == 1
    raw = out.arr("u1", (n_obs + 7) // 8 * 4)
    o = np.arange(n_obs)
    tracks = {"nt": ((raw[(o >> 3) * 4 + (o & 3)] >> (4 * ((o & 7) >> 2))) & 15).astype(np.uint8)}
    tracks.update({k: out.arr("u1", n_obs) for k in (["bq", "mq"] + (["baq"] if has_baq else []))})
    depth = np.diff(host["col_off"].astype(np.int64))
    assert np.array_equal(np.diff(col_off.astype(np.int64)), depth[keep])
    assert np.array_equal(ref, host["ref_base"][keep])
    assert np.array_equal(cov, depth[keep]) and np.array_equal(nb, depth[keep])
    assert max_obs == depth[keep].max()
    sel = np.concatenate([np.arange(int(host["col_off"][c]), int(host["col_off"][c + 1])) for c in np.nonzero(keep)[0]]) \
        if keep.any() else np.zeros(0, np.int64)
    assert bool(has_baq) == (host["baq"] is not None)
    for k, a in tracks.items():
        assert np.array_equal(a, host[k][sel]), k
    assert out.tag() == b"DONE"
    assert out.i64() == 1 and out.i64() == 3 * nc      # bonf_subst / num_snv_tests written back from the library's conf


def test_shim_batches_in_flight(harness, tmp_path):
    """small batches (LFQ_SHIM_BATCH_COLS): a full batch is submitted and mpileup's thread goes on filling the other
    buffer set; every batch is complete when it is handed over, the batches cover the columns in order, the counters
    written back at the end are those of all of them"""
    path = gu.fixtures()[0]
    fx, host = gu.load(path)
    ncols = len(fx["columns"])
    out = _Reader(_run(harness, tmp_path, (1, 1, 1, 0, 3), _snv_columns_blob(fx), ncols, env={"LFQ_SHIM_BATCH_COLS": "7"}))
    keep = np.array([c["ref"] != "N" for c in fx["columns"]])
    depth = np.diff(host["col_off"].astype(np.int64))[keep]
    sel = np.concatenate([np.arange(int(host["col_off"][c]), int(host["col_off"][c + 1])) for c in np.nonzero(keep)[0]])
    got = {"nt": [], "bq": [], "mq": []}
    seen, n_batches = 0, 0
    while True:
        tag = out.tag()
        if tag != b"SNVB":
            break
        nc, n_obs, on_dev, has_baq, has_sq, max_obs, bonf = [out.i64() for _ in range(7)]
        assert nc <= 7 and bonf == 1 + 0 * n_batches            # (the mock advances no Bonferroni factor)
        col_off = out.arr("<u8", nc + 1)
        assert np.array_equal(np.diff(col_off.astype(np.int64)), depth[seen:seen + nc])
        out.arr("u1", nc); out.arr("<i4", nc); out.arr("<i4", nc)
        assert out.i64() == 1
        raw = out.arr("u1", (n_obs + 7) // 8 * 4)
        o = np.arange(n_obs)
        got["nt"].append(((raw[(o >> 3) * 4 + (o & 3)] >> (4 * ((o & 7) >> 2))) & 15).astype(np.uint8))
        got["bq"].append(out.arr("u1", n_obs)); got["mq"].append(out.arr("u1", n_obs))
        if has_baq:
            out.arr("u1", n_obs)
        seen += nc
        n_batches += 1
    assert tag == b"DONE" and n_batches == -(-int(keep.sum()) // 7) and seen == int(keep.sum())
    for k in got:
        assert np.array_equal(np.concatenate(got[k]), host[k][sel]), k
    assert out.i64() == 1 and out.i64() == 3 * seen


def test_shim_skips_snvs_at_consensus_indel_columns(harness, tmp_path):
    """call_vars: no SNV test where the consensus is an insertion / deletion (lofreq_call.c:929)"""
    path = gu.fixtures()[0]
    fx, host = gu.load(path)
    ncols = len(fx["columns"])
    skip = {3, 7}
    out = _Reader(_run(harness, tmp_path, (1, 1, 1, 0, 3), _snv_columns_blob(fx, skip), ncols))
    assert out.tag() == b"SNVB"
    nc = out.i64()
    n_ref_n = sum(1 for c in fx["columns"] if c["ref"] == "N")
    assert nc == ncols - n_ref_n - len([c for c in skip if fx["columns"][c]["ref"] != "N"])


def _indel_blob(cols, pos=None):
    out = []
    for i, c in enumerate(cols):
        out.append(_i32(pos[i] if pos is not None else 100 + i, ord(c["ref"]), ord(c["ref"]), c["coverage_plp"], c["coverage_plp"], c["num_tails"],
                        c["num_non_indels"], c["num_ins"], c["num_dels"], c["hrun"], 1))
        for _nt in range(5):
            out.append(_i32(0, 0, 0, 0))
        for sn in ("ins", "dels"):
            s = c[sn]
            out.append(_i32(s["non_fw"], s["non_rv"], len(s["ne_q"])))
            out.append(np.stack([np.asarray(s["ne_q"]), np.asarray(s["ne_mq"])], axis=1).astype("<i4").tobytes()
                       if len(s["ne_q"]) else b"")
            out.append(_i32(len(s["events"])))
            for e in s["events"]:
                key = e["key"].encode()
                n = len(e["q"])
                out.append(_i32(len(key)) + key + _i32(e["fw"], n))
                out.append(np.stack([np.asarray(e["q"]), np.asarray(e["aq"]), np.asarray(e["mq"]), np.asarray(e["sq"])],
                                    axis=1).astype("<i4").tobytes())
    return b"".join(out)


@pytest.mark.parametrize("path", gu.indel_fixtures(), ids=lambda p: p.split("/")[-1])
def test_shim_flattens_indel_columns_like_indel_columns(harness, tmp_path, path):
    import sys
    sys.path.insert(0, ROOT)
    from lofreq_amd.indel import IndelColumns
    fx, cols = gu.load_indels(path)
    out = _Reader(_run(harness, tmp_path, (1, 1, 0, 1, 11), _indel_blob(cols), len(cols)))
    # the shim flattens the columns that carry an event (lofreq_call.c:684, :706: no event, no test) with ref != N
    want = IndelColumns.from_columns([c for c in cols if c["ref"] != "N" and (c["num_ins"] or c["num_dels"])])
    assert out.tag() == b"INDB"
    nc = out.i64()
    assert nc == want.ncols
    assert np.array_equal(out.arr("u1", nc), want.ref_base)
    for name in ("coverage_plp", "num_tails", "num_non_indels", "num_ins", "num_dels", "hrun"):
        assert np.array_equal(out.arr("<i4", nc), getattr(want, name)), name
    for sd in range(2):
        w = want.sides[sd]
        n_ne, n_ev, n_rd, n_key = [out.i64() for _ in range(4)]
        assert (n_ne, n_ev, n_rd) == (len(w["ne_q"]), len(w["ev_fw"]), len(w["rd_q"]))
        for name, dt, n in (("non_fw", "<i4", nc), ("non_rv", "<i4", nc), ("ne_off", "<i8", nc + 1), ("ne_q", "<i2", n_ne),
                            ("ne_mq", "<i2", n_ne), ("ev_off", "<i8", nc + 1), ("key_off", "<i8", n_ev + 1)):
            assert np.array_equal(out.arr(dt, n), w[name]), (sd, name)
        assert out.arr("u1", n_key).tobytes() == w["key_chars"].tobytes()[:n_key]
        for name, dt, n in (("ev_fw", "<i4", n_ev), ("ev_rv", "<i4", n_ev), ("rd_off", "<i8", n_ev + 1), ("rd_q", "<i2", n_rd),
                            ("rd_aq", "<i2", n_rd), ("rd_mq", "<i2", n_rd), ("rd_sq", "<i2", n_rd)):
            assert np.array_equal(out.arr(dt, n), w[name]), (sd, name)
    assert out.tag() == b"DONE"
