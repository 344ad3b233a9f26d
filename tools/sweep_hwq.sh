#!/bin/bash
# headline bench against the number of hardware queues the HIP runtime spreads its streams over
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2}_hwq.txt; : > $OUT
for round in 1 2; do
for q in default 2 4 6 8 12 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  r=$(python bench.py --steps 100 --warmup 20 --cpu-planes 0 --settle 0 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "GPU_MAX_HW_QUEUES=$q $r" | tee -a $OUT
done
done
