#!/bin/bash
# The fused histogram / row-filter kernel (DSX_FUSE_HIST, default 1 here; FH=0 for the separate kernels) over streams x
# batch sizes, 3 steps each.  Extra environment is passed on, e.g. DSX_ABLATE=4096 (nobody waits at the plane barrier)
# or 8192 (no Otsu arithmetic) -- wrong results, timing only.  Appends to gpurun_out/fh_probe.txt.
cd $GRAFT_REPO_ROOT
for spec in "1 72" "1 256" "4 72" "4 128" "4 256" "2 256"; do
  set -- $spec
  r=$(DSX_FUSE_HIST=${FH:-1} DSX_STREAMS=$1 timeout -k 5 120 python bench.py --batch $2 --steps 3 --warmup 1 --cpu-planes 0 --settle 0 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "ablate=${DSX_ABLATE:-0} fuse=${FH:-1} streams=$1 batch=$2 -> $r" | tee -a gpurun_out/fh_probe.txt
done
