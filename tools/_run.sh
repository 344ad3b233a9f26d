cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_wavelets.py -m gpu -x -q > gpurun_out/w1_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/w1_tests.log
