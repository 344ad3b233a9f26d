cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/m3_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/m3_tests.log
bash tools/ab.sh m3 m2 2>&1 | tail -12
