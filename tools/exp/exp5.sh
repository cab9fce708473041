cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/e5_tests.txt 2>&1
tail -15 gpurun_out/e5_tests.txt
