"""lfq_filter_vars & friends -- `lofreq filter` in every mode (lofreq_filter.c:376-677, 861-1331) -- against the FILTER
columns, the ##FILTER header lines and the set of written variants of the reference's own 2.1.4 binary
(tests/golden/filter_modes.json, oracle/make_golden.py --filter-only: 90 variants x 18 option sets), and the two
special-case entry points `lofreq call` uses (lfq_filter_records / lfq_filter_indel_records) against the general one."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lofreq_amd import _lib          # noqa: E402

MTC = {"bonf": 1, "bonferroni": 1, "holm": 2, "holmbonf": 2, "holm-bonf": 2, "fdr": 3}


def conf_from_args(L, args):
    """main_filter's option loop (lofreq_filter.c:1099-1171) + the defaults step"""
    c = _lib.FilterConf()
    L.lfq_filter_conf_init(C.byref(c))
    no_defaults, it = False, iter(args)
    for a in it:
        if a == "--no-defaults": no_defaults = True
        elif a == "--only-snvs": c.only_snvs = 1
        elif a == "--only-indels": c.only_indels = 1
        elif a == "--sb-no-compound": c.sb_no_compound = 1
        elif a == "--sb-incl-indels": c.sb_incl_indels = 1
        elif a == "-v": c.dp_min = int(next(it))
        elif a == "-V": c.dp_max = int(next(it))
        elif a == "-a": c.af_min = float(next(it))
        elif a == "-A": c.af_max = float(next(it))
        elif a == "-B": c.sb_thresh = int(next(it))
        elif a == "-b": c.sb_mtc_type = MTC[next(it)]
        elif a == "-c": c.sb_alpha = float(next(it))
        elif a == "-Q": c.snvqual_thresh = int(next(it))
        elif a == "-q": c.snvqual_mtc_type = MTC[next(it)]
        elif a == "-r": c.snvqual_alpha = float(next(it))
        elif a == "-s": c.snvqual_ntests = int(next(it))
        elif a == "-K": c.indelqual_thresh = int(next(it))
        elif a == "-k": c.indelqual_mtc_type = MTC[next(it)]
        elif a == "-l": c.indelqual_alpha = float(next(it))
        elif a == "-m": c.indelqual_ntests = int(next(it))
        else: raise ValueError(a)
    if not no_defaults:
        L.lfq_filter_conf_defaults(C.byref(c))
    return c


def vars_from_vcf(lines):
    v = np.zeros(len(lines), _lib.FILTER_VAR_DTYPE)
    pos = []
    for i, l in enumerate(lines):
        f = l.split("\t")
        info = dict((kv.split("=") + [""])[:2] for kv in f[7].split(";"))
        pos.append(int(f[1]))
        v["is_indel"][i] = int(len(f[3]) > 1 or len(f[4]) > 1 or "INDEL" in info)       # vcf_var_is_indel, vcf.c:328-337
        v["qual"][i] = -1 if f[5] == "." else int(f[5])
        v["dp"][i], v["sb"][i] = int(info["DP"]), int(info["SB"])
        v["af"][i] = np.float32(float(info["AF"]))                                     # strtof of the text
        dp4 = [int(x) for x in info["DP4"].split(",")]
        v["alt_fw"][i], v["alt_rv"][i] = dp4[2], dp4[3]
    return v, pos


import pytest       # noqa: E402


@pytest.mark.parametrize("fixture", ["filter_modes.json", "filter_modes_small.json"])
def test_every_mode_against_the_binary(fixture):
    L = _lib.load()
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", fixture)))
    v, pos = vars_from_vcf(fx["vcf"])
    assert v["is_indel"].sum() > 5 and (v["qual"] == -1).sum() == 1
    if fixture == "filter_modes_small.json":        # the AF filter is live next to a correction when nothing underflows
        assert any("af_" in f for case in fx["cases"] for f in case["filter"])
    else:                                           # ... and silently off when a QUAL of 49314 is in the file
        assert not any("af_" in f for f in fx["cases"][15]["filter"]) and "-a" in fx["cases"][15]["args"]
    buf = C.create_string_buffer(4096)
    for case in fx["cases"]:
        c = conf_from_args(L, case["args"])
        fail = np.zeros(len(v), np.uint32)
        rc = L.lfq_filter_vars(C.byref(c), v.ctypes.data, len(v), fail.ctypes.data)
        assert rc == 0, case["args"]
        kept = [i for i in range(len(v)) if fail[i] != 128]                # --only-snvs / --only-indels drop the rest
        assert [pos[i] for i in kept] == case["pos"], case["args"]
        got = []
        for i in kept:
            L.lfq_filter_string(C.byref(c), int(fail[i]), buf, 4096)
            got.append(buf.value.decode())
        assert got == case["filter"], (case["args"], [(a, b) for a, b in zip(got, case["filter"]) if a != b][:5])
        assert [pos[i] for i in kept if fail[i] == 0] == case["passed_pos"], case["args"]
        n = L.lfq_filter_header_lines(C.byref(c), buf, 4096)
        assert n < 4096 and buf.value.decode().splitlines() == case["filter_lines"], case["args"]
    # the conflicts main_filter rejects (lofreq_filter.c:1177-1227)
    for bad in (["-B", "30", "-b", "fdr"], ["-Q", "50", "-q", "bonf"], ["-K", "50", "-k", "bonf"], ["--only-snvs", "--only-indels"],
                ["-v", "50", "-V", "20"], ["-A", "1.5"]):
        c = conf_from_args(L, ["--no-defaults"] + bad)
        fail = np.zeros(len(v), np.uint32)
        assert L.lfq_filter_vars(C.byref(c), v.ctypes.data, len(v), fail.ctypes.data) == -1, bad


def test_call_time_entry_points_are_the_special_case():
    """lfq_filter_records(snvqual_thresh, apply_defaults) = lfq_filter_vars with -Q thresh [+ defaults] on the same records"""
    L = _lib.load()
    rng = np.random.default_rng(3)
    n = 400
    rec = np.zeros(n, _lib.SNV_RECORD_DTYPE)
    rec["qual"] = rng.choice([30, 55, 56, 57, 90, 400], n)
    rec["dp"] = rng.choice([4, 9, 10, 11, 200, 5000], n)
    rec["alt_fw"] = rng.integers(0, 60, n)
    rec["alt_rv"] = np.where(rng.random(n) < 0.5, rng.integers(0, 60, n), rng.integers(0, 3, n))
    rec["alt_raw_count"] = np.maximum(1, rec["alt_fw"] + rec["alt_rv"])
    rec["dp"] = np.maximum(rec["dp"], rec["alt_raw_count"])
    rec["sb"] = rng.choice([0, 2, 15, 40, 90, 300], n)
    for thresh, defaults in ((56, 1), (56, 0), (0, 1)):
        keep = np.zeros(n, np.uint8)
        assert L.lfq_filter_records(rec.ctypes.data, n, thresh, defaults, keep.ctypes.data) == 0
        v = np.zeros(n, _lib.FILTER_VAR_DTYPE)
        for i in range(n):
            L.lfq_filter_var_from_snv(rec[i:i + 1].ctypes.data, v[i:i + 1].ctypes.data)
        c = _lib.FilterConf()
        L.lfq_filter_conf_init(C.byref(c))
        c.snvqual_thresh = thresh
        if defaults:
            L.lfq_filter_conf_defaults(C.byref(c))
        fail = np.zeros(n, np.uint32)
        assert L.lfq_filter_vars(C.byref(c), v.ctypes.data, n, fail.ctypes.data) == 0
        assert np.array_equal(keep != 0, fail == 0), (thresh, defaults)
        assert 0 < keep.sum() < n
