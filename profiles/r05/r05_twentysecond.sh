# Round 5: lifetimes of the lean count kernel's wavefronts, alone and beside another batch's DP kernels (a -DLFQ_COUNT_STAMP build on the box)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f lofreq_amd/csrc/build/lfq_kernels.o
make -C lofreq_amd/csrc EXTRA=-DLFQ_COUNT_STAMP 2>&1 | grep -i "error"
for u in 2 4; do
  echo "#### LFQ_COUNT_AHEAD_DEEP=$u"
  LFQ_COUNT_AHEAD_DEEP=$u python profiles/wave_stamps.py
done > gpurun_out/r05_wave_stamps.txt 2>&1
cat gpurun_out/r05_wave_stamps.txt
