#!/bin/bash
# per-kernel HIP-event times of two builds in one GPU session (single stream): tools/ab_kernels.sh libA libB [rounds]
A=$1; B=$2; R=${3:-2}
for i in $(seq $R); do
  for L in $A $B; do
    DSX_LIB=$L DSX_STREAMS=1 python bench.py --cpu-planes 0 --steps 2 --warmup 1 --settle 0.2 --no-verify --kernel-breakdown 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms']
print('$L', 'value', d['value'], ' '.join('%s=%.3f' % (n[2:].replace('_march','').replace('(',':').rstrip(')'), v['ms']) for n, v in k.items()))"
  done
done
