cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03g
python -m pytest tests -m gpu -q > gpurun_out/r03g/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r03g/pytest.log
tail -25 gpurun_out/r03g/pytest.log | cut -c1-400
