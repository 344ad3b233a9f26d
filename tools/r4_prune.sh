#!/bin/bash
# pruned forward FFT (product) against a variant built with  tools/build_variant.sh noprune -DDSX_FWD_PRUNE=0
# (the session recorded in profiles/r4_fft_prune_and_coarse_stream_ab.txt also ran "diag DSX_PIPE=0 / 3": a schedule experiment whose code was not kept)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r4c_prune_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -5 gpurun_out/r4c_prune_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
L=$PWD/aind_smartspim_destripe_amd/_lib
OUT=gpurun_out/r4c_prune_ab.txt; : > $OUT
for round in 1 2 3; do
  for spec in "hip" "noprune" "hip DSX_STREAMS=1" "noprune DSX_STREAMS=1"; do
    lib=${spec%% *}; envs=""; [ "$spec" != "$lib" ] && envs=${spec#* }
    r=$(env DSX_LIB=$L/libdsx_$lib.so $envs timeout -k 10 200 python bench.py --steps 100 --warmup 20 --cpu-planes 0 --settle 0.5 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "[$spec] $r" | tee -a $OUT
  done
done
