"""The pin is reproducible as committed: where the reference tree is mounted, `make -C oracle ref` + oracle/make_golden.py
regenerate fixtures from the reference's own 2.1.4 binary that are byte-identical to the committed ones -- incl. the
default-filter fixtures, for which `lofreq call` shells out to `lofreq filter` (lofreq_call.c:1506-1551; the binary is
unpacked as oracle/_ref/bin/lofreq for that).  Skipped where /root/reference is absent."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isfile("/root/reference/dist/lofreq_star-2.1.4_linux-x86-64.tgz"),
                                reason="reference dist not mounted")


def test_regenerated_snv_fixtures_are_byte_identical(tmp_path):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    env = {k: v for k, v in os.environ.items()}
    env["LFQ_GOLDEN_OUT"] = str(tmp_path)
    env["PATH"] = "/usr/bin:/bin"                       # nothing named `lofreq` on PATH but what the generator puts there
    p = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_golden.py"), "--snv-only"], env=env,
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    made = sorted(os.listdir(tmp_path))
    assert made == ["snv_deep.json", "snv_default.json", "snv_minbq_sig.json", "snv_nobaq_dynamic_nofilter.json",
                    "snv_nofilter_fixedbonf.json"]
    for f in made:
        a = open(tmp_path / f, "rb").read()
        b = open(os.path.join(ROOT, "tests", "golden", f), "rb").read()
        assert a == b, f
