#!/bin/bash
# round-2 baseline: GPU tests, headline bench (+ per-kernel events), single-stream chain trace, 1-rank RCCL rehearsal
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s > gpurun_out/r2_tests0.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2_tests0.log
tail -5 gpurun_out/r2_tests0.log
python bench.py --steps 100 --warmup 10 > gpurun_out/r2_bench0.json 2> gpurun_out/r2_bench0.err; echo "bench rc=$?"
cat gpurun_out/r2_bench0.json
DSX_STREAMS=1 python bench.py --steps 20 --warmup 5 --cpu-planes 0 --kernel-breakdown > gpurun_out/r2_bench0_s1.json 2>/dev/null
cat gpurun_out/r2_bench0_s1.json
bash tools/chain_trace.sh > gpurun_out/r2_chain0.log 2>&1
DSX_FORCE_COMM=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-planes 0 > gpurun_out/r2_dist1.json 2> gpurun_out/r2_dist1.err; echo "dist rc=$?"
cat gpurun_out/r2_dist1.json; tail -3 gpurun_out/r2_dist1.err
