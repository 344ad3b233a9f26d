#!/bin/bash
# the whole -m gpu suite on the current build (one process), per-test durations
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r4_gpu_suite.log 2>&1; rc=$?
echo "gpu tests rc=$rc"; tail -22 gpurun_out/r4_gpu_suite.log | cut -c1-250
exit $rc
