#!/bin/bash
# target number of waves per march launch (row segmentation) sweep, 4 streams
for Wv in 1536 2304 3072 4096 6144 9216; do
  DSX_MARCH_WAVES=$Wv python bench.py --cpu-planes 0 --steps 6 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('march waves', $Wv, 'value', d['value'], 'ms', d['ms_per_step'])"
done
