"""CPU tests: the C-ABI library loads, exports every symbol include/lofreq_amd.h declares, agrees on
struct layouts, and fails loudly (no CPU fallback) when no GPU is present.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    from lofreq_amd import _lib
    L = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "lofreq_amd.h")).read()
    declared = set(re.findall(r"\b(lfq_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), "symbol %s declared in include/lofreq_amd.h but not exported" % name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert L.lfq_abi_version() == _lib.LFQ_ABI_VERSION == int(re.search(r"#define LFQ_ABI_VERSION (\d+)", hdr).group(1))


def test_struct_layouts():
    from lofreq_amd import _lib
    assert _lib.COL_COUNTS_DTYPE.itemsize == 64
    assert _lib.COL_PVALS_DTYPE.itemsize == 128
    assert _lib.SNV_RECORD_DTYPE.itemsize == 64
    assert C.sizeof(_lib.Conf) == 80        # + approx_threshold_n, pad (snpcaller.h:62)
    assert _lib.INDEL_CALL_DTYPE.itemsize == 48 and _lib.INDEL_RECORD_DTYPE.itemsize == 80
    assert C.sizeof(_lib.IndelColumnsC) == 8 * 8 + 2 * 15 * 8 + 8
    assert C.sizeof(_lib.Tracks) == 12 * 8


def test_conf_defaults_match_reference():
    """init_varcall_conf (snpcaller.c:627-651) + defaults.h"""
    import lofreq_amd as la
    c = la.VarcallConf()
    assert (c.min_bq, c.min_alt_bq, c.def_alt_bq) == (6, 6, 0)
    assert (c.min_jq, c.min_alt_jq, c.def_alt_jq) == (0, 0, 0)
    assert c.min_cov == 1 and c.bonf_dynamic == 1 and c.bonf_subst == 1
    assert c.sig == np.float32(0.01) and c.flag == (la.LFQ_USE_MQ | la.LFQ_USE_BAQ | la.LFQ_USE_IDAQ)
    assert c.bonf_indel == 1 and c.num_indel_tests == 0
    assert c.approx_threshold_n == -1           # snpcaller.c:650


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import lofreq_amd as la
    with pytest.raises(RuntimeError, match="no usable HIP device"):
        la.SnvCaller(0)


def test_host_side_functions_match_oracle(oracle):
    """Host pieces of the product (no GPU needed) against the oracle: Fisher/SB, thresholds, FDR,
    VCF text, final filter, 80-bit p-value conversion."""
    import lofreq_amd as la
    from lofreq_amd import _lib
    L, O = _lib.load(), oracle.lib()
    rng = np.random.default_rng(4)
    for _ in range(500):
        a, b, c, d = [int(x) for x in rng.integers(0, 3000, 4)]
        assert L.lfq_sb_phred(a, b, c, d) == O.orc_sb_phred(a, b, c, d)
    for bonf in (1, 3, 29727, 3000000, 10 ** 9):
        assert la.snvqual_thresh(0.01, bonf) == O.orc_snvqual_thresh(0.01, bonf)
    # records -> text and filter
    n = 200
    rec = np.zeros(n, la.SNV_RECORD_DTYPE)
    rec["col"] = np.arange(n) * 3
    rec["qual"] = rng.integers(0, 3000, n)
    rec["dp"] = rng.integers(5, 5000, n)
    rec["alt_fw"] = rng.integers(0, 100, n)
    rec["alt_rv"] = rng.integers(0, 100, n)
    rec["alt_rv"][(rec["alt_fw"] + rec["alt_rv"]) == 0] = 1
    rec["ref_fw"] = rng.integers(0, 2000, n)
    rec["ref_rv"] = rng.integers(0, 2000, n)
    rec["alt_raw_count"] = rec["alt_fw"] + rec["alt_rv"]
    rec["hqa"] = rec["alt_raw_count"]
    rec["ref"], rec["alt"] = b"A", b"G"
    for i in range(n):
        rec["sb"][i] = L.lfq_sb_phred(int(rec["ref_fw"][i]), int(rec["ref_rv"][i]), int(rec["alt_fw"][i]),
                                      int(rec["alt_rv"][i]))
    for thr, defaults in [(0, True), (84, False), (500, True)]:
        keep = la.filter_records(rec, thr, apply_defaults=defaults)
        arr = lambda k: (C.c_int * n)(*[int(x) for x in rec[k]])
        k2 = (C.c_int * n)()
        O.orc_default_filter(arr("qual"), arr("dp"), arr("sb"), arr("alt_fw"), arr("alt_rv"), n, thr,
                             1 if defaults else 0, k2)
        assert [bool(x) for x in keep] == [bool(k2[i]) for i in range(n)]
    buf = C.create_string_buffer(512)
    for i in range(0, n, 17):
        r = rec[i]
        m = O.orc_format_snv(buf, 512, b"chrX", int(r["col"]), b"A", b"G", int(r["qual"]), int(r["dp"]),
                             int(r["alt_raw_count"]), int(r["sb"]), int(r["ref_fw"]), int(r["ref_rv"]),
                             int(r["alt_fw"]), int(r["alt_rv"]), int(r["hqa"]), 1, b"PASS")
        assert la.format_vcf_record(r, "chrX", int(r["col"]), "PASS") == buf.raw[:m].decode()
    text = la.format_vcf(rec, "chrX", filter_str="PASS")
    assert text.count("\n") == n and text.startswith("chrX\t1\t.\tA\tG\t")
    # p-value conversion incl. the clamp (snpcaller.c:1047-1059)
    LDBL_MAX, LDBL_MIN = np.finfo(np.longdouble).max, np.finfo(np.longdouble).tiny
    assert la.pvalue_from_log(-20000.0, la.LFQ_PV_LOG) == LDBL_MIN
    assert la.pvalue_from_log(-3670.0, la.LFQ_PV_LOG_FECLAMP) == LDBL_MIN
    assert la.pvalue_from_log(-20.0, la.LFQ_PV_LOG_FECLAMP) == LDBL_MAX
    assert la.pvalue_from_log(-1e5, la.LFQ_PV_UNDERFLOW) == LDBL_MIN
    p = la.pvalue_from_log(-3670.2412275820329, la.LFQ_PV_LOG)
    assert abs(float(np.log(p)) + 3670.2412275820329) < 1e-12


def test_release_library_reads_ten_environment_variables():
    """lfq_knobs() of the release library reads ten documented variables, none of which changes a result;
    device selection (lfq_pick_device) reads three more.  LFQ_DEBUG_SKIP and the launch-shape / fallback selectors are strings
    of the tuning build only (liblofreq_amd_tune.so, lfq_host.cpp under -DLFQ_TUNE)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def env_names(lib):
        data = open(os.path.join(root, "lofreq_amd", lib), "rb").read()
        return {m.decode() for m in re.findall(rb"(?<![A-Z0-9_])((?:LFQ|LOCAL)_[A-Z0-9_]{3,})\x00", data)}
    knobs = {"LFQ_TIMING", "LFQ_SINGLE_STREAM", "LFQ_DEBUG_SYNC", "LFQ_PRIVATE_STREAM", "LFQ_SYNC_UPLOAD", "LFQ_BAQ_SCRATCH_MB",
             "LFQ_HOST_THREADS", "LFQ_HOST_LOOP_THREADS", "LFQ_HOST_SPIN_US", "LOCAL_WORLD_SIZE"}
    device = {"LFQ_DEVICE", "LFQ_SLOT_DIR", "LOCAL_RANK"}
    rel = env_names("liblofreq_amd.so")
    assert rel == knobs | device, sorted(rel ^ (knobs | device))
    tune = env_names("liblofreq_amd_tune.so")
    assert rel < tune and {"LFQ_DEBUG_SKIP", "LFQ_LIGHT_KERNEL", "LFQ_PILEUP_ATOMIC", "LFQ_SPLIT_POOL_CELLS"} <= tune
