cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_wavelets.py -m gpu -x -q > gpurun_out/r1_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r1_tests.log
bash tools/ab.sh r1 r0 2>&1 | tail -12
