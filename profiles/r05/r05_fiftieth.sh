# Round 5: the lean count kernel's column header as ONE scalar round trip (plain scalar arguments, the first fourteen dwords
# preloaded into scalar registers at wave start; the four header loads requested back to back) against three behind one another
# (kernel arguments -> null tests -> reference base -> column offsets).  Libraries: old = before; nopre = new code without
# -amdgpu-kernarg-preload-count; default = new code with it (every kernel of the library compiled with the flag).
# (the three libraries were built from profiles/experiments/r05_count_header_one_round_trip.patch; the code was reverted -- profiles/NOTES.md)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
one() {     # $1 = lib, $2 = shape args, $3 = mode args
  env LFQ_AMD_LIB=$GRAFT_REPO_ROOT/lofreq_amd/$1 python bench.py $2 $3 --steps 60 --warmup 10 --repeats 3 --no-cpu-baseline --no-pmc --no-secondary --no-full-check 2>gpurun_out/r05_x.err | grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['config']['kernel_ms']; r = d['repeats']
print('%-26s %-12s %-26s step %.3f (min %.3f max %.3f)  count %.3f  dp %.3f (l %.3f m %.3f b %.3f)  records %d' % (
    '$1', '$2', '$3', r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'], k['ms_count'], k['ms_dp'],
    k['ms_dp_light'], k['ms_dp_mid'], k['ms_dp_big'], d['config']['records_per_step']))" || tail -3 gpurun_out/r05_x.err
}
for i in 1 2 3; do
for lib in liblofreq_amd_old.so liblofreq_amd_nopre.so liblofreq_amd.so; do
one $lib "--config C3" "--in-flight 4 --gate none"
done
done
for lib in liblofreq_amd_old.so liblofreq_amd_nopre.so liblofreq_amd.so; do
one $lib "--config C3" "--in-flight 4 --gate end"
one $lib "--config C2" "--in-flight 4 --gate none"
one $lib "--depth 200 --cols 3750000" "--in-flight 4 --gate end"
done
