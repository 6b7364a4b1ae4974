"""run bench.py's C3 step with an alternative build of the library: python bench_lib.py <lib.so> [bench args]"""
import sys, os, runpy
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, root)
from lofreq_amd import _lib
_lib.LIB_PATH = sys.argv[1]
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
