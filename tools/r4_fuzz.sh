#!/bin/bash
# long randomised parity run with the launch-geometry switches drawn per case: bash tools/r4_fuzz.sh [cases] [seed] [out tag]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python tools/fuzz_parity.py ${1:-500} ${2:-20261005} geometry > gpurun_out/r4_fuzz_geometry${3:-}.log 2>&1; echo "geometry fuzz rc=$?"; tail -2 gpurun_out/r4_fuzz_geometry${3:-}.log | cut -c1-400
