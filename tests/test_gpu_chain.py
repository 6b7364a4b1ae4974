"""-m gpu: raw reads -> BAQ (lfq_baq_batch) -> pileup columns -> SNV calls (lfq_call_snvs_batch) -> VCF text, against
the VCF the reference's own 2.1.4 binary wrote from the same SAM with its on-the-fly BAQ (tests/golden/chain_*.json).
The pileup itself (compile_plp_col, plp.c) is out of scope and done here in numpy for all-M reads."""
import numpy as np
import pytest

import golden_util as gu

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180)]


def _pileup(fx, lb):
    """columns of the all-M reads of a fixture: what compile_plp_col hands to the callback (plp.c:797-1017):
    per read base nt4 code + strand, BQ, BAQ (lb byte - 33), MQ; bases below min_plp_bq = 3 are dropped"""
    import json
    glen = len(fx["genome"])
    per_col = [[] for _ in range(glen)]
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    for (pos0, flag, mapq, cigar, seq, qual), tag in zip(fx["reads"], lb):
        assert cigar == "%dM" % len(seq)
        strand = 1 if flag & 16 else 0
        for j, (b, q) in enumerate(zip(seq, qual)):
            bq = ord(q) - 33
            if bq < 3:
                continue
            per_col[pos0 + j].append((code.get(b, 4) | (strand << 3), bq, int(tag[j]) - 33, mapq))
    cols = [c for c in range(glen) if per_col[c]]
    nt = np.array([o[0] for c in cols for o in per_col[c]], np.uint8)
    bq = np.array([o[1] for c in cols for o in per_col[c]], np.uint8)
    baq = np.array([o[2] for c in cols for o in per_col[c]], np.uint8)
    mq = np.array([o[3] for c in cols for o in per_col[c]], np.uint8)
    off = np.zeros(len(cols) + 1, np.uint64)
    off[1:] = np.cumsum([len(per_col[c]) for c in cols])
    ref = np.frombuffer("".join(fx["genome"][c] for c in cols).encode(), np.uint8).copy()
    return cols, dict(nt=nt, bq=bq, baq=baq, mq=mq, sq=None, col_off=off, ref_base=ref)


@pytest.mark.parametrize("path", gu.chain_fixtures(), ids=lambda p: p.split("/")[-1])
def test_reads_to_vcf_matches_reference_binary(caller, path):
    import json
    import lofreq_amd as la
    import util
    fx = json.load(open(path))
    reads = [{"pos0": r[0], "cigar": gu.parse_cigar(r[3]), "seq": la.encode_seq(r[4]),
              "qual": np.array([ord(c) - 33 for c in r[5]], np.uint8)} for r in fx["reads"]]
    lb = la.baq_batch(caller, reads, fx["genome"].encode(), extended=True)          # lofreq call: extended BAQ
    cols, host = _pileup(fx, lb)
    kw, no_default_filter = gu.conf_kwargs(fx["call_args"])
    conf = la.VarcallConf(**kw)
    recs, _, st = caller.call_snvs(util.to_pileup_batch(la, host), conf)
    assert conf.num_snv_tests == fx["num_snv_tests"]
    thr = la.snvqual_thresh(conf.sig, conf.bonf_subst)
    keep = la.filter_records(recs, thr, apply_defaults=not no_default_filter)
    pos0 = np.array([cols[int(r["col"])] for r in recs], np.int64)
    text = la.format_vcf(recs, "chr1", pos0=pos0, keep=keep, filter_str="PASS")
    got = [gu.strip_hqa(l) for l in text.splitlines()]
    assert got == fx["vcf"]


# ---- the read-level binding (integration/lofreq_amd_region.c): BAM records in, VCF lines out --------------------------------

def _build_region_lib(tmp_path):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = str(tmp_path / "liblofreq_amd_region.so")
    subprocess.run(["gcc", "-std=gnu99", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC",
                    "-I" + os.path.join(root, "include"), os.path.join(root, "integration", "lofreq_amd_region.c"),
                    "-L" + os.path.join(root, "lofreq_amd"), "-llofreq_amd", "-Wl,-rpath," + os.path.join(root, "lofreq_amd"),
                    "-o", lib], check=True, capture_output=True, text=True)
    return lib


class _RegionOpts(__import__("ctypes").Structure):
    _fields_ = [(k, __import__("ctypes").c_int) for k in
                ("use_baq", "baq_extended", "use_idaq", "use_sq", "def_nm_q", "call_indels", "only_indels", "min_mq", "max_mq",
                 "min_plp_bq", "min_plp_idq", "no_orphan")]


def _bam_fields(r):
    """a read of the fixtures as the fields of its BAM record (bam1_t): 4-bit packed bases, cigar words, Z strings"""
    nt16 = np.array([1, 2, 4, 8, 15], np.uint8)[np.asarray(r["seq"], np.uint8)]
    if len(nt16) & 1:
        nt16 = np.append(nt16, 0)
    seq4 = ((nt16[0::2] << 4) | nt16[1::2]).astype(np.uint8)
    cig = np.asarray([(l << 4) | "MIDNSHP=X".index(o) for o, l in r["cigar"]], np.uint32)
    z = lambda t: None if t is None else bytes(np.asarray(t, np.uint8)) + b"\0"
    return seq4, cig, z(r.get("bi")), z(r.get("bd"))


def _run_regions(caller, lib, reads, ref, regions, conf, call_indels=True, only_indels=False, extra_reads=()):
    """drive lfq_region_* like the region loop of mpileup() would: per region the reads that overlap it, in file order"""
    import ctypes as C
    P = C.CDLL(lib)
    lines = []
    EMIT = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)
    cb = EMIT(lambda user, s: lines.append(s.decode().rstrip("\n")))
    o = _RegionOpts()
    P.lfq_region_opts_init(C.byref(o))
    o.use_idaq, o.call_indels, o.only_indels = 1 if call_indels else 0, 1 if call_indels else 0, 1 if only_indels else 0
    h = C.c_void_p()
    P.lfq_region_open.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, EMIT, C.c_void_p]
    assert P.lfq_region_open(C.byref(h), caller.h, C.byref(conf.c), C.byref(o), cb, None) == 0
    P.lfq_region_begin.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int64]
    P.lfq_region_add_read.argtypes = [C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_char_p, C.c_char_p]
    P.lfq_region_end.argtypes = [C.c_void_p]
    P.lfq_region_close.argtypes = [C.c_void_p, C.c_void_p]
    taken = 0
    for beg, end in regions:
        assert P.lfq_region_begin(h, b"chr1", ref, len(ref), beg, end) == 0
        for r in reads:
            rlen = sum(l for op, l in r["cigar"] if op in "MDN=X")
            if r["pos0"] >= end or r["pos0"] + rlen <= beg:
                continue                                    # (what sam_itr_querys leaves out)
            seq4, cig, bi, bd = _bam_fields(r)
            q = np.asarray(r["qual"], np.uint8)
            rc = P.lfq_region_add_read(h, r["pos0"], 16 if r["reverse"] else 0, r["mapq"], len(cig), cig.ctypes.data, len(q),
                                       seq4.ctypes.data, q.ctypes.data, bi, bd)
            assert rc in (0, 1)
            taken += rc
        assert P.lfq_region_end(h) == 0
    wo = C.c_int64(-1)
    assert P.lfq_region_close(h, C.byref(wo)) == 0
    return lines, taken, wo.value


def _epilogue(la, lines, conf):
    """main_call's epilogue on the emitted lines (lofreq_call.c:1519-1538): QUAL thresholds from the final factors, PASS"""
    thr_s, thr_i = la.snvqual_thresh(conf.sig, conf.bonf_subst), la.snvqual_thresh(conf.sig, conf.bonf_indel)
    out = []
    for ln in lines:
        f = ln.split("\t")
        if int(f[5]) < (thr_i if "INDEL" in f[7] else thr_s):
            continue
        assert f[6] == "."
        f[6] = "PASS"
        out.append(gu.strip_hqa("\t".join(f)))
    return out


@pytest.mark.parametrize("path", gu.plpindel_fixtures(), ids=lambda p: p.split("/")[-1])
def test_region_binding_reproduces_the_binary_vcf(caller, tmp_path, path):
    """integration/lofreq_amd_region.c fed with BAM records (4-bit bases, cigar words, BI / BD strings -- a mock bam1_t):
    one region, and the same genome cut into three regions whose reads overlap the cuts; the VCF of `lofreq call
    --call-indels` of the reference's 2.1.4 binary, test counts included, either way"""
    import lofreq_amd as la
    lib = _build_region_lib(tmp_path)
    fx, reads = gu.load_plpindel(path, with_alnqual_tags=False)
    ref = fx["genome"].encode()
    kw, _ = gu.conf_kwargs(fx["call_args"])
    n = len(ref)
    for regions in ([(0, n)], [(0, n // 3), (n // 3, n // 3 + 37), (n // 3 + 37, n)]):
        conf = la.VarcallConf(**kw)
        lines, taken, wo = _run_regions(caller, lib, reads, ref, regions, conf)
        assert taken >= len(reads)
        assert conf.num_snv_tests == fx["all"]["num_tests"]["snv"] and conf.num_indel_tests == fx["all"]["num_tests"]["indel"]
        assert _epilogue(la, lines, conf) == fx["all"]["vcf"], regions
        assert wo == 0                                      # every indel call had its alignment qualities (ai / ad by the device)
    conf = la.VarcallConf(**kw)
    lines, _, _ = _run_regions(caller, lib, reads, ref, [(0, n)], conf, only_indels=True)
    assert conf.num_indel_tests == fx["only_indels"]["num_tests"]["indel"] and conf.num_snv_tests == 0
    assert _epilogue(la, lines, conf) == fx["only_indels"]["vcf"]


@pytest.mark.parametrize("path", gu.chain_fixtures(), ids=lambda p: p.split("/")[-1])
def test_region_binding_snv_fixtures(caller, tmp_path, path):
    """the SNV-only fixtures (no BI / BD: no consensus indel can arise, the indel pileup is not even run) through the
    same binding, incl. a mapping-quality floor"""
    import json
    import lofreq_amd as la
    lib = _build_region_lib(tmp_path)
    fx = json.load(open(path))
    reads = [{"pos0": r[0], "cigar": gu.parse_cigar(r[3]), "seq": la.encode_seq(r[4]),
              "qual": np.array([ord(c) - 33 for c in r[5]], np.uint8), "mapq": r[2], "reverse": bool(r[1] & 16)}
             for r in fx["reads"]]
    ref = fx["genome"].encode()
    kw, ndf = gu.conf_kwargs(fx["call_args"])
    conf = la.VarcallConf(**kw)
    n = len(ref)
    lines, taken, _ = _run_regions(caller, lib, reads, ref, [(0, n // 2), (n // 2, n)], conf, call_indels=False)
    assert conf.num_snv_tests == fx["num_snv_tests"]
    got = _epilogue(la, lines, conf)
    if not ndf:                                             # the default filter is `lofreq filter`'s business (lfq_filter_records)
        got = [l for l in got if l in set(fx["vcf"])]
    assert got == fx["vcf"]
