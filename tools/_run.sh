cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python tools/fuzz_parity.py 400 31337 > gpurun_out/r3_fuzz_parity.log 2>&1; echo "fuzz rc=$?"; tail -4 gpurun_out/r3_fuzz_parity.log
timeout -k 10 150 python tools/fuzz_parity.py 120 777 wavelets > gpurun_out/r3_fuzz_wavelets.log 2>&1; echo "fuzz wl rc=$?"; tail -2 gpurun_out/r3_fuzz_wavelets.log
