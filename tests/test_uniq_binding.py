"""The reference-side binding of `lofreq uniq` under test (no GPU): integration/lofreq_amd_uniq.c is compiled against the
reference's OWN plp.h / vcf.h / utils.h / log.h and linked with the reference's own utils.c + log.c, then driven by a mock
mpileup (tests/uniq_harness.c) that plays main_uniq's loop (lofreq_uniq.c:690-730): one callback per variant with a
plp_col_t rebuilt from the golden fixtures the way compile_plp_col fills it, freed + poisoned right after the callback
(plp.c:1440-1445).  A mock liblofreq_amd records the ONE batch the binding hands to lfq_uniq_detlim_batch /
lfq_uniq_binom_batch; it must equal, byte for byte, what tests/golden_util.py builds from the same fixtures (the batches
the GPU tests of tests/test_gpu_uniq.py run on), and the INFO tags written back must be the 2.1.4 binary's.
Skipped where /root/reference is absent.  (htslib: as in tests/test_shim.py, two one-line forward declarations.)"""
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
    d = tmp_path_factory.mktemp("uniq")
    os.makedirs(d / "stub" / "htslib")
    (d / "stub" / "htslib" / "faidx.h").write_text("typedef struct faidx_t faidx_t;\n")
    (d / "stub" / "htslib" / "bgzf.h").write_text("#include <stdio.h>\ntypedef struct BGZF BGZF;\n")
    exe = str(d / "uniq_harness")
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "integration"), "-I" + os.path.join(REF, "lofreq"),
           "-I" + os.path.join(REF, "uthash"), "-I" + str(d / "stub")]
    base = ["gcc", "-std=gnu99", "-O1", "-g", "-Wall", "-Wno-unused-function"] + inc + [
        os.path.join(ROOT, "integration", "lofreq_amd_uniq.c"), os.path.join(ROOT, "integration", "lofreq_amd_uniqbatch.c"),
        os.path.join(ROOT, "tests", "uniq_harness.c"),
        os.path.join(REF, "lofreq", "utils.c"), os.path.join(REF, "lofreq", "log.c"), "-lm", "-o", exe]
    # the binding itself must compile without a single warning against the reference headers
    chk = subprocess.run(["gcc", "-std=gnu99", "-fsyntax-only", "-Wall", "-Wextra", "-Wno-unused-parameter"] + inc +
                         [os.path.join(ROOT, "integration", "lofreq_amd_uniq.c")], capture_output=True, text=True)
    assert chk.returncode == 0, chk.stderr
    assert "lofreq_amd_uniq.c" not in chk.stderr, chk.stderr
    # ... and the packing core against include/lofreq_amd.h alone: no LoFreq header, no htslib
    chk = subprocess.run(["gcc", "-std=gnu99", "-fsyntax-only", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
                          "-I" + os.path.join(ROOT, "integration"),
                          os.path.join(ROOT, "integration", "lofreq_amd_uniqbatch.c")], capture_output=True, text=True)
    assert chk.returncode == 0 and not chk.stderr.strip(), chk.stderr
    asan = subprocess.run(base[:1] + ["-fsanitize=address", "-fno-omit-frame-pointer"] + base[1:], capture_output=True, text=True)
    if asan.returncode != 0:                        # no libasan in this image: the harness's poisoning still catches stale reads
        subprocess.run(base, check=True, capture_output=True, text=True)
    return exe


def _i32(*v):
    return struct.pack("<%di" % len(v), *[int(x) for x in v])


def _s(x):
    b = x.encode()
    return _i32(len(b)) + b


def variant_blob(v, canned, chrom="chr1", info=None, has_col=1, col_pos=None, cov=None, tails=0, ins=(), dels=()):
    """one variant of a uniq fixture (or a hand-made one) -> the harness's stream"""
    n_col = sum(len(o["bq"]) for o in v["obs"].values())
    out = [_s(chrom), _s(v["ref"]), _s(v["alt"]), _s("AF=%s" % v["af"] if info is None else info),
           _i32(v["pos0"], has_col, v["pos0"] if col_pos is None else col_pos, ord(v["ref"][0]),
                n_col if cov is None else cov, tails, canned)]
    for nt in "ACGTN":
        o = v["obs"].get(nt)
        if not o:
            out.append(_i32(0, 0))
            continue
        bq = gu.dec(o["bq"])
        mq = np.array([int(o["mq"][2 * i:2 * i + 2], 16) for i in range(len(bq))])
        out.append(_i32(len(bq), v["fwrv"][nt][0]))
        out.append(np.stack([bq, mq], axis=1).astype("<i4").tobytes())
    for evs in (ins, dels):
        out.append(_i32(len(evs)))
        for key, cnt in evs:
            out.append(_s(key) + _i32(cnt))
    return b"".join(out)


def run(harness, tmp_path, use_det_lim, uni_freq, blobs):
    inp, outp = str(tmp_path / "vars.bin"), str(tmp_path / "dump.bin")
    with open(inp, "wb") as f:
        f.write(_i32(use_det_lim) + struct.pack("<f", uni_freq) + _i32(len(blobs)) + b"".join(blobs))
    r = subprocess.run([harness, inp, outp], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    infos = [ln.split("\t", 1) for ln in r.stdout.splitlines()]
    return [(int(p), i) for p, i in infos], open(outp, "rb").read(), r.stderr


class Dump:
    def __init__(self, b):
        self.b, self.o = b, 0

    def take(self, n):
        v = self.b[self.o:self.o + n]
        assert len(v) == n
        self.o += n
        return v

    def i64(self):
        return struct.unpack("<q", self.take(8))[0]

    def tracks(self, tag, with_alt):
        assert self.take(4) == tag
        ncols, n_obs, on_dev, has_baq, has_sq, max_obs, has_cov, flags = [self.i64() for _ in range(8)]
        d = dict(ncols=ncols, n_obs=n_obs, on_dev=on_dev, has_baq=has_baq, has_sq=has_sq, max_obs=max_obs, flags=flags)
        d["col_off"] = np.frombuffer(self.take((ncols + 1) * 8), "<u8")
        d["ref_base"] = np.frombuffer(self.take(ncols), np.uint8)
        d["cov"] = np.frombuffer(self.take(ncols * 4), "<i4") if has_cov else None
        d["af"] = np.frombuffer(self.take(ncols * 4), "<f4")
        d["nt"] = np.frombuffer(self.take((n_obs + 7) // 8 * 4), np.uint8)
        d["bq"] = np.frombuffer(self.take(n_obs), np.uint8)
        d["mq"] = np.frombuffer(self.take(n_obs), np.uint8)
        d["alt"] = self.take(ncols).decode() if with_alt else None
        return d


def unpack_nt(packed, n_obs):
    o = np.arange(n_obs)
    b = packed[(o >> 3) * 4 + (o & 3)]
    return np.where(o & 4, b >> 4, b & 15).astype(np.uint8)


def check_tracks(d, host, af):
    assert d["ncols"] == len(af) and d["on_dev"] == 0 and not d["has_baq"] and not d["has_sq"]
    assert d["flags"] == 1                                                  # LFQ_TRACKS_NT_PACKED
    assert d["col_off"].tolist() == host["col_off"].tolist()
    assert d["max_obs"] == int(np.diff(host["col_off"].astype(np.int64)).max())
    assert d["ref_base"].tolist() == host["ref_base"].tolist()
    assert d["af"].tobytes() == np.asarray(af, "<f4").tobytes()
    assert unpack_nt(d["nt"], d["n_obs"]).tolist() == host["nt"].tolist()
    assert d["bq"].tolist() == host["bq"].tolist() and d["mq"].tolist() == host["mq"].tolist()


@pytest.mark.parametrize("path", gu.uniq_fixtures(), ids=lambda p: p.split("/")[-1])
def test_detlim_fixture_one_batch_and_uniq_flags(harness, tmp_path, path):
    fx, host, af = gu.load_uniq(path)
    infos, dump, _ = run(harness, tmp_path, 1, -1.0, [variant_blob(v, int(v["uniq"])) for v in fx["variants"]])
    d = Dump(dump)
    t = d.tracks(b"UDET", False)
    assert d.o == len(dump)                                                 # ONE batch call, nothing else
    check_tracks(t, host, af)
    assert t["cov"] is None
    for (pos, info), v in zip(infos, fx["variants"]):
        assert pos == v["pos0"]
        assert info == ("AF=%s;UNIQ" % v["af"] if v["uniq"] else "AF=%s" % v["af"])
    assert any(v["uniq"] for v in fx["variants"]) and not all(v["uniq"] for v in fx["variants"])


@pytest.mark.parametrize("path", gu.uniq_binom_fixtures(), ids=lambda p: p.split("/")[-1])
def test_binom_fixture_one_batch_and_uq_tags(harness, tmp_path, path):
    fx, host, af = gu.load_uniq(path)
    uq = [(-1 if v["uq"] is None else v["uq"]) for v in fx["variants"]]
    infos, dump, _ = run(harness, tmp_path, 0, -1.0, [variant_blob(v, q) for v, q in zip(fx["variants"], uq)])
    d = Dump(dump)
    t = d.tracks(b"UBIN", True)
    assert d.o == len(dump)
    check_tracks(t, host, af)
    assert t["alt"] == "".join(v["alt"] for v in fx["variants"])
    assert t["cov"].tolist() == np.diff(host["col_off"].astype(np.int64)).tolist()     # coverage_plp of these columns
    for (pos, info), v, q in zip(infos, fx["variants"], uq):
        assert info == ("AF=%s;UQ=%d" % (v["af"], q) if q >= 0 else "AF=%s" % v["af"])


def _var(pos0, ref, alt, af, obs):
    return dict(pos0=pos0, ref=ref, alt=alt, af=af, obs=obs,
                fwrv={nt: [len(o["bq"]) // 2, len(o["bq"]) - len(o["bq"]) // 2] for nt, o in obs.items()})


def _obs(n, q="I", mq="3c"):
    return dict(bq=q * n, mq=mq * n)


def test_gates_indels_and_af_handling(harness, tmp_path):
    """what uniq_snv does around the test (lofreq_uniq.c:233-277, 342-370): wrong pileup -> error line, no tag; coverage
    < 1 after the tails of an indel variant -> nothing; an AF out of bounds is logged and reset; -f replaces every AF; an
    indel variant in binomial mode takes its count from the event table and never reaches the batch"""
    snv = _var(10, "A", "G", "0.100000", {"A": _obs(20), "G": _obs(3)})
    wrong = _var(11, "C", "T", "0.200000", {"C": _obs(9)})
    nocov = _var(12, "C", "T", "0.200000", {"C": _obs(5)})
    oob = _var(13, "T", "C", "1.500000", {"T": _obs(7), "C": _obs(1)})
    neg = _var(14, "T", "C", "-0.200000", {"T": _obs(6)})
    dele = _var(15, "GAT", "G", "0.050000", {"G": _obs(30)})
    insn = _var(16, "G", "GCC", "0.050000", {"G": _obs(12)})
    insx = _var(17, "G", "GTT", "0.050000", {"G": _obs(12)})
    blobs = [variant_blob(snv, 17),
             variant_blob(wrong, 99, col_pos=12),
             variant_blob(nocov, 99, cov=0),
             variant_blob(oob, 5),
             variant_blob(neg, 3),
             variant_blob(dele, 99, cov=40, tails=2, dels=[("AT", 4), ("A", 1)]),
             variant_blob(insn, 99, cov=25, ins=[("C", 2), ("CC", 3)]),
             variant_blob(insx, 99, cov=2, tails=2, ins=[("TT", 9)])]
    infos, dump, err = run(harness, tmp_path, 0, -1.0, blobs)
    d = Dump(dump)
    t = d.tracks(b"UBIN", True)
    assert t["ncols"] == 3 and t["alt"] == "GCC"
    assert t["af"].tolist() == [np.float32(0.1), np.float32(1.0), np.float32(0.01)]       # reset, lofreq_uniq.c:262-268
    assert t["cov"].tolist() == [23, 8, 6]
    # the two indel variants with coverage: scalar tests with the event counts, coverage minus tails
    calls = []
    while d.o < len(dump):
        assert d.take(4) == b"BINO"
        n, k = d.i64(), d.i64()
        calls.append((n, k, struct.unpack("<d", d.take(8))[0]))
    assert calls == [(38, 4, float(np.float32(0.05))), (25, 3, float(np.float32(0.05)))]
    assert [i for _, i in infos] == ["AF=0.100000;UQ=17", "AF=0.200000", "AF=0.200000", "AF=1.500000;UQ=5",
                                     "AF=-0.200000;UQ=3", "AF=0.050000;UQ=6", "AF=0.050000;UQ=6", "AF=0.050000"]
    assert "wrong pileup for var. pileup for chr1 13. var for chr1 12" in err
    assert err.count("Invalid (value out of bound) AF") == 2
    # det-lim mode: every variant with coverage goes through the column (indels too, :291-301); -f 0.25 replaces the AFs
    infos, dump, err = run(harness, tmp_path, 1, 0.25, [variant_blob(snv, 1), variant_blob(dele, 0, cov=40, tails=2),
                                                        variant_blob(insn, 1, cov=25)])
    d = Dump(dump)
    t = d.tracks(b"UDET", False)
    assert d.o == len(dump) and t["ncols"] == 3 and t["af"].tolist() == [0.25, 0.25, 0.25]
    assert [i for _, i in infos] == ["AF=0.100000;UNIQ", "AF=0.050000", "AF=0.050000;UNIQ"]
