#!/bin/bash
# long randomised parity runs on the current build: launch-geometry switches drawn per case, then the plain run
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python tools/fuzz_parity.py ${1:-500} ${2:-20261005} geometry > gpurun_out/r4_fuzz_geometry.log 2>&1; echo "geometry fuzz rc=$?"; tail -3 gpurun_out/r4_fuzz_geometry.log | cut -c1-400
