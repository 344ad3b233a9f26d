#!/bin/bash
# Row-filter ablations (diagnosis only; results are wrong with DSX_ABLATE != 0): per-kernel HIP-event times
# needs a -DDSX_DIAG build:  tools/build_variant.sh diag -DDSX_DIAG  (the product library ignores DSX_ABLATE)
export DSX_LIB=${DSX_LIB:-$(dirname "$0")/../aind_smartspim_destripe_amd/_lib/libdsx_diag.so}
for A in 0 1 2 4 8 3 7; do
  DSX_ABLATE=$A DSX_STREAMS=1 python bench.py --cpu-planes 0 --steps 2 --warmup 1 --settle 0.2 --no-verify --kernel-breakdown 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms']
print('ablate', $A, 'value', d['value'], ' '.join('%s=%.3f' % (n.split('(')[0][2:6] + n[-6:-1], v['ms']) for n, v in k.items()))"
done
