"""-m gpu: the code paths that only run under an environment knob (fallbacks for inputs the default route cannot take:
no pool space, unsorted reads, wide bands, shallow / deep launch shapes; the measured-and-rejected A/B paths of rounds
1-3 were deleted in round 4, profiles/NOTES.md) get the parity tests of their area, each in a process of its own with the
knob set -- so that nothing that can be selected at run time is untested.  The knobs are read once per process
(lfq_knobs(), lfq_internal.h); DESIGN.md lists them.

Round 6: the release library reads ten variables, none of which changes a result; every other knob exists only in the tuning
build lofreq_amd/liblofreq_amd_tune.so (the same objects, lfq_host.cpp compiled under -DLFQ_TUNE), which the child processes
here load through LFQ_AMD_LIB.  test_release_library_ignores_the_tuning_knobs holds the release library to that."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

TUNE_LIB = os.path.join(ROOT, "lofreq_amd", "liblofreq_amd_tune.so")

DP = ["tests/test_gpu_parity.py", "-k", "default_conf_random or row_split or ragged_deep or underflow or fe_clamp or cells_below or "
      "golden_reference or dynamic_bonferroni or edge_cases"]
BAQ = ["tests/test_gpu_baq.py"]
PLP = ["tests/test_gpu_pileup.py", "tests/test_gpu_plpindel.py", "-k", "not host_loops and not window_search"]
CHAIN = ["tests/test_gpu_plpindel.py", "-k", "chain or device_only or device_packing", "tests/test_gpu_indel.py"]

CASES = [
    ("LFQ_SPLIT_POOL_CELLS=0", DP),          # no row split: every long column on the unsplit kernels
    ("LFQ_SPLIT_POOL_CELLS=60000", DP),      # a pool that runs out: split and unsplit columns in one batch
    ("LFQ_LIGHT_KERNEL=wave", DP),           # one light column per wavefront (what K >= 32 gets anyway)
    ("LFQ_SCREEN_ROUNDS=2", DP),             # nearly every light column through the retry kernel
    ("LFQ_SINGLE_STREAM=1", DP),             # every kernel on one stream (what the counter passes run)
    ("LFQ_NO_SB_PRECOMPUTE=1", DP),          # strand bias computed at collect time only
    ("LFQ_COUNT_MULTI_BELOW=0", DP),         # shallow batches on the one-column-per-wavefront count kernel
    ("LFQ_COUNT_LPG4_BELOW=100000", DP),     # shared-wavefront count kernel: four lanes per column whatever the depth
    ("LFQ_COUNT_LPG8_BELOW=100000", DP),     # ... eight (and four for the shallowest batches)
    ("LFQ_COUNT_LPG4_BELOW=0", DP),          # ... never four
    ("LFQ_COUNT_WAVES_PER_WG=4 LFQ_COUNT_MULTI_BELOW=0", DP),    # the lean count kernel with 4 columns per workgroup (default 16) ...
    ("LFQ_COUNT_WAVES_PER_WG=8 LFQ_COUNT_MULTI_BELOW=0", DP),    # ... and 8, on every batch
    ("LFQ_SEG_MAX=2", DP),                   # a big column in two row segments, a mid-class column in one piece after its first stretch (what a context with LFQ_GATE_NONE runs)
    ("LFQ_SEG_MAX_MID=3 LFQ_SEG_MAX_BIG=4", DP),   # ... other segment counts per class
    ("LFQ_BIG_ON_SIDE=1", DP),               # the unsplit big columns behind the big chain (what a context with LFQ_GATE_NONE runs)
    ("LFQ_JOIN_ON_SIDE=0 LFQ_TAIL_LIGHT=0 LFQ_HEAVY_AFTER_SCREEN=0", DP),   # the stream plan of round 4: join on the light chain's stream, tail event behind the retry kernel
    ("LFQ_TAIL_LIGHT=2", DP),                # tail event of the light chain behind the scan
    ("LFQ_COUNT_SHALLOW_LDS_PAD=44000", DP), # shared-wavefront count kernel with two workgroups per CU (what a context with LFQ_GATE_NONE launches)
    ("LFQ_COUNT_LEAN_LDS_PAD=54000 LFQ_COUNT_WAVES_PER_WG=8 LFQ_COUNT_AHEAD_DEEP=4 LFQ_COUNT_MULTI_BELOW=0", DP),    # lean count kernel capped to three workgroups per CU
    ("LFQ_PRIVATE_STREAM=1", DP),            # every context with a launch stream of its own (lfq_set_private_stream) ...
    ("LFQ_PRIVATE_STREAM=1", CHAIN),         # ... on the read-set chain as well
    ("LFQ_PILEUP_TILES=0", PLP),             # a wavefront per position instead of tiles of 64 positions
    ("LFQ_BAQ_ONE_VARIANT=1", BAQ),          # every wavefront through the register kernel's instantiation with the N case
    ("LFQ_BAQ_LDS=0", BAQ),                  # every read through the all-HBM BAQ kernel (what wide bands get)
    ("LFQ_PILEUP_ATOMIC=1", PLP),            # read-major pileup kernels (what unsorted reads get)
    ("LFQ_INDEL_HOST_PACK=1", CHAIN),        # indel pseudo-columns packed on the host
    ("LFQ_SYNC_UPLOAD=1", CHAIN),            # lfq_readset_create waits for its copies itself
    ("LFQ_HOST_SPIN_US=-1", CHAIN),          # host loops on threads created per loop
]


@pytest.mark.parametrize("knob,sel", CASES, ids=[c[0] for c in CASES])
def test_knob_selected_paths(knob, sel):
    env = dict(os.environ, LFQ_AMD_LIB=TUNE_LIB, **dict(kv.split("=") for kv in knob.split()))
    p = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"] + sel, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=800)
    tail = (p.stdout + p.stderr)[-1500:]
    assert p.returncode == 0, tail
    assert " passed" in p.stdout and "failed" not in p.stdout.splitlines()[-1], tail


_SKIP_PROBE = r"""
import sys
sys.path.insert(0, "tests")
import numpy as np
import lofreq_amd as la
import util
caller = la.SnvCaller(0)
host = util.random_batch(np.random.default_rng(5), 64, 800, 1500, planted={c: 0.3 for c in range(0, 64, 2)})
recs, _, st = caller.call_snvs(util.to_pileup_batch(la, host), la.VarcallConf())
print("RECS", len(recs))
"""


def test_release_library_ignores_the_tuning_knobs():
    """LFQ_DEBUG_SKIP drops DP classes -- i.e. calls -- in the tuning build and must do nothing in the release library"""
    out = {}
    for name, lib in (("release", None), ("tune", TUNE_LIB)):
        env = dict(os.environ, LFQ_DEBUG_SKIP="light,mid,big")
        env.pop("LFQ_AMD_LIB", None)
        if lib:
            env["LFQ_AMD_LIB"] = lib
        p = subprocess.run([sys.executable, "-c", _SKIP_PROBE], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-1500:]
        out[name] = int(p.stdout.split("RECS")[1])
    env = dict(os.environ)
    env.pop("LFQ_AMD_LIB", None)
    p = subprocess.run([sys.executable, "-c", _SKIP_PROBE], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    plain = int(p.stdout.split("RECS")[1])
    assert plain >= 20 and out["release"] == plain and out["tune"] < plain, (plain, out)
