#!/bin/bash
# A/B two builds inside one GPU session: tools/ab.sh <libA.so> <libB.so> [rounds]
# (timings from different gpurun boxes differ by a few percent; only same-session numbers compare)
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do
    DSX_LIB=$L python bench.py --cpu-planes 0 --steps 40 --warmup 5 --settle 0.3 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'])"
  done
done
