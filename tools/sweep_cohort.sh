#!/bin/bash
# cohort (planes per launch chain) x sub-cohort streams sweep of the headline bench
for C in 256 128 64 32; do
  for S in 4 8; do
    DSX_STREAMS=$S python bench.py --cpu-planes 0 --steps 6 --cohort $C 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cohort', $C, 'streams', $S, 'value', d['value'], 'ms', d['ms_per_step'])"
  done
done
