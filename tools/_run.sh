cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_rowfinal" > gpurun_out/x1_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/x1_tests.log
bash tools/ab.sh x1 rf1 2>&1 | tail -12
bash tools/traffic.sh x1 > /dev/null 2>&1; grep "mb_per_plane\|rowfinal" gpurun_out/x1_traffic.json
