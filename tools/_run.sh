cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_zarr_chunk_map.py -m gpu -x -q > gpurun_out/h1_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/h1_tests.log
L=$PWD/aind_smartspim_destripe_amd/_lib
bash tools/ab_kernels.sh $L/libdsx_base.so $L/libdsx_hip.so 2 | tee gpurun_out/h1_abk.txt
bash tools/ab.sh h1 base 2>&1 | tail -12
