cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_rowfinal or multistream or seed_sweep or width_sweep or otsu" > gpurun_out/rf5_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/rf5_tests.log
bash tools/ab.sh rf5 rf0 2>&1 | tail -12
L=$PWD/aind_smartspim_destripe_amd/_lib
bash tools/ab_kernels.sh $L/libdsx_rf0.so $L/libdsx_hip.so 1
