#!/bin/bash
# round-3 baseline: GPU tests, headline bench, single-stream per-kernel events
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s > gpurun_out/r3_tests0.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r3_tests0.log
tail -8 gpurun_out/r3_tests0.log
python bench.py --steps 100 --warmup 10 > gpurun_out/r3_bench0.json 2> gpurun_out/r3_bench0.err; echo "bench rc=$?"
cat gpurun_out/r3_bench0.json
DSX_STREAMS=1 python bench.py --steps 20 --warmup 5 --cpu-planes 0 --kernel-breakdown > gpurun_out/r3_bench0_s1.json 2>/dev/null
cat gpurun_out/r3_bench0_s1.json
