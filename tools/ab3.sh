#!/bin/bash
# per-kernel times (single stream) and 4-stream throughput of N builds in one GPU session: tools/ab3.sh lib1 lib2 ...
for i in 1 2; do
  for L in "$@"; do
    DSX_LIB=$L DSX_STREAMS=1 python bench.py --cpu-planes 0 --steps 2 --warmup 1 --settle 0.2 --no-verify --kernel-breakdown 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms']
print('$L', '1-stream', d['value'], ' '.join('%s=%.3f' % (n[2:].replace('_march','').replace('(',':').rstrip(')'), v['ms']) for n, v in k.items()))"
    DSX_LIB=$L python bench.py --cpu-planes 0 --steps 40 --warmup 5 --settle 0.3 --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$L', '4-stream', d['value'], d['ms_per_step'])"
  done
done
