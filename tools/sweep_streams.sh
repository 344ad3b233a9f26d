#!/bin/bash
# headline bench against the number of sub-cohort streams and the batch size
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-sweep}_streams.txt; : > $OUT
for b in 256 512; do
  for s in 2 3 4 6 8; do
    r=$(DSX_STREAMS=$s python bench.py --batch $b --cohort $b --steps $((25600 / b)) --warmup 10 --cpu-planes 0 --settle 0 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "batch=$b streams=$s $r" | tee -a $OUT
  done
done
