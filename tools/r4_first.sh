#!/bin/bash
# round 4, first GPU call: the whole -m gpu suite on the new build, then the bench (driver style) as the round's baseline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r4a_gpu_tests.log 2>&1; rc=$?
echo "gpu tests rc=$rc"; tail -25 gpurun_out/r4a_gpu_tests.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 100 --warmup 20 > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r4a_bench.json')); print(d['value'], d['ms_per_step'], d['verified'], d['roofline']['frac'], d['cpu_baseline'])"
