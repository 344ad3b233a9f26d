cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s3_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/s3_tests.log
