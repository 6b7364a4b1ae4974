import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
from lofreq_amd import _lib
_lib.LIB_PATH = os.path.join(os.getcwd(), "lofreq_amd", "liblofreq_amd_prof.so")
import lofreq_amd as la, numpy as np
os.environ.setdefault("LFQ_DEBUG_SKIP", "light,mid")
c = la.SnvCaller(0)
b = c.synth_batch(0x9E3779B97F4A7C15 ^ (3 << 32), 10000, 1000000)
for it in range(3):
    conf = la.VarcallConf(); recs, _, st = c.call_snvs(b, conf, records_capacity=1 << 16)
out = (C.c_int32 * 64)()
_lib.load().lfq_debug_counters(c.h, out)
v = list(out)
n = max(v[60], 1)
print("folds", v[60], "per fold us (wall_clock 100MHz ticks*0.01):", [round(x / n * 0.01, 1) for x in v[52:59]], "kernel ms", c.kernel_times())
