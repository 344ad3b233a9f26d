#!/bin/bash
# sub-cohort streams sweep of the headline bench
for S in 1 2 3 4 5 6 8; do
  DSX_STREAMS=$S python bench.py --cpu-planes 0 --steps 6 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('streams', $S, 'value', d['value'], 'ms', d['ms_per_step'])"
done
