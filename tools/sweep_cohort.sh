#!/bin/bash
# planes per step (batch = cohort) x sub-cohort streams sweep of the headline bench -> gpurun_out/<tag>_cohort.txt
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2}_cohort.txt; : > $OUT
for round in 1 2; do
for B in 64 96 128 192 256 320 384 512; do
  for S in 2 4; do
    r=$(DSX_STREAMS=$S python bench.py --batch $B --cohort $B --steps $((51200 / B)) --warmup 20 --cpu-planes 0 --settle 0 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "batch=$B streams=$S $r" | tee -a $OUT
  done
done
done
