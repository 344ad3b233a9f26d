cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_rowfinal" > gpurun_out/x2_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/x2_tests.log
