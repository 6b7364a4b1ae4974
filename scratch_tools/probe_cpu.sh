cd $GRAFT_REPO_ROOT
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>&1)"; echo "cfs: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>&1)"; nproc; cat /proc/loadavg
grep -c processor /proc/cpuinfo; cat /sys/fs/cgroup/cpuset.cpus.effective 2>&1 | head -2
python - <<'PY'
import sys, time, os
sys.path.insert(0,'oracle')
import full_check as fc, pyoracle as orc
orc.build()
print("budget", fc.cpu_budget())
seed=0x9E3779B97F4A7C15 ^ (3 << 32)
for procs in (16, 64, 128, 254):
    n=procs*200
    ranges=[(i*200+1, 200, 0) for i in range(procs)]
    t0=time.time()
    fc.run_ranges(seed, 10000, 997, ranges, procs=procs, chunk_cols=200)
    dt=time.time()-t0
    print(procs, "procs:", n, "cols in %.2f s = %.0f cols/s" % (dt, n/dt))
PY
python -m pytest tests -m gpu -x -q -k "deep_tail" -s 2>&1 | grep -v "^$" | tail -5
