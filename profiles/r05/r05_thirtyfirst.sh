# Round 5: the shader clock the lean count kernel's wavefronts see, alone and beside the DP kernels (-DLFQ_COUNT_STAMP build on the box)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f lofreq_amd/csrc/build/lfq_kernels.o
make -C lofreq_amd/csrc EXTRA=-DLFQ_COUNT_STAMP 2>&1 | grep -i "error"
python profiles/wave_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_wave_stamps2.txt
cat gpurun_out/r05_wave_stamps2.txt
